"""Scalar logging with TensorBoard-compatible event files, no TensorBoard needed.

Same surface as the reference's ``Logger`` (``utils/logger.py:7-32``):
``Logger(log_dir).scalar_summary(tag, value, step)``, ``close()`` and the
``info`` / ``debug`` / ``warning`` aliases.  The reference delegates to
``tensorboardX.SummaryWriter``; neither tensorboardX nor tensorboard is a
dependency here, so the event-file format is written directly:

  record  = uint64 len | uint32 masked_crc32c(len) | bytes data | uint32 masked_crc32c(data)
  data    = Event{ double wall_time = 1; int64 step = 2; string file_version = 3 |
                   Summary summary = 5 { repeated Value value = 1 { string tag = 1; float simple_value = 2 } } }

``tensorboard --logdir`` reads these files.  :func:`read_events` parses them back
(used by the tests and by ``utils/reward_plot.py``).  A JSON-lines mirror
(``scalars.jsonl``) can be written alongside (``log_backend: jsonl | both``).
"""
from __future__ import annotations

import json
import logging
import os
import socket
import struct
import time
from typing import Iterator, List, Optional, Tuple

_log = logging.getLogger("d4pg_pytorch_amd")

# ---------------------------------------------------------------------------
# CRC-32C (Castagnoli), table driven
# ---------------------------------------------------------------------------
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    crc = 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        crc = tbl[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------
# minimal protobuf encoding
# ---------------------------------------------------------------------------

def _varint(n: int) -> bytes:
    n &= 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_bytes(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag: str, value: float, step: int, wall_time: float) -> bytes:
    val = _field_bytes(1, tag.encode("utf-8")) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
    summary = _field_bytes(1, val)
    return (_varint((1 << 3) | 1) + struct.pack("<d", wall_time) +
            _varint((2 << 3) | 0) + _varint(int(step)) + _field_bytes(5, summary))


def encode_version_event(wall_time: float) -> bytes:
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _field_bytes(3, b"brain.Event:2")


def _record(data: bytes) -> bytes:
    hdr = struct.pack("<Q", len(data))
    return hdr + struct.pack("<I", _masked_crc(hdr)) + data + struct.pack("<I", _masked_crc(data))


class EventFileWriter:
    def __init__(self, log_dir: str, flush_secs: float = 5.0):
        os.makedirs(log_dir, exist_ok=True)
        fn = f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}"
        self.path = os.path.join(log_dir, fn)
        self._fh = open(self.path, "wb")
        self._fh.write(_record(encode_version_event(time.time())))
        self._last_flush = time.time()
        self._flush_secs = flush_secs

    def add_scalar(self, tag: str, value: float, step: int, wall_time: Optional[float] = None) -> None:
        if self._fh is None:
            return
        self._fh.write(_record(encode_scalar_event(tag, value, step, wall_time or time.time())))
        now = time.time()
        if now - self._last_flush > self._flush_secs:
            self._fh.flush()
            self._last_flush = now

    def flush(self) -> None:
        if self._fh is not None:
            self._fh.flush()

    def close(self) -> None:
        if self._fh is not None:
            self._fh.flush()
            self._fh.close()
            self._fh = None


# ---------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------

def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _parse_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield num, wt, v


def read_events(path: str, check_crc: bool = True) -> List[Tuple[float, int, str, float]]:
    """Parse an event file (or every event file in a directory) into
    ``(wall_time, step, tag, value)`` tuples."""
    if os.path.isdir(path):
        out: List[Tuple[float, int, str, float]] = []
        for fn in sorted(os.listdir(path)):
            if "tfevents" in fn:
                out += read_events(os.path.join(path, fn), check_crc)
        return out
    out = []
    with open(path, "rb") as fh:
        blob = fh.read()
    pos = 0
    while pos + 12 <= len(blob):
        hdr = blob[pos:pos + 8]
        (ln,) = struct.unpack("<Q", hdr)
        (hcrc,) = struct.unpack("<I", blob[pos + 8:pos + 12])
        data = blob[pos + 12:pos + 12 + ln]
        if len(data) < ln or pos + 16 + ln > len(blob):
            break   # truncated tail (writer still running)
        (dcrc,) = struct.unpack("<I", blob[pos + 12 + ln:pos + 16 + ln])
        if check_crc and (hcrc != _masked_crc(hdr) or dcrc != _masked_crc(data)):
            raise ValueError(f"{path}: CRC mismatch at offset {pos}")
        pos += 16 + ln
        wall, step, summary = 0.0, 0, None
        for num, wt, v in _parse_fields(data):
            if num == 1 and wt == 1:
                (wall,) = struct.unpack("<d", v)
            elif num == 2 and wt == 0:
                step = v
            elif num == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for num, wt, v in _parse_fields(summary):
            if num != 1:
                continue
            tag, val = "", None
            for n2, w2, v2 in _parse_fields(v):
                if n2 == 1 and w2 == 2:
                    tag = v2.decode("utf-8")
                elif n2 == 2 and w2 == 5:
                    (val,) = struct.unpack("<f", v2)
            if val is not None:
                out.append((wall, int(step), tag, float(val)))
    return out


# ---------------------------------------------------------------------------
# Logger
# ---------------------------------------------------------------------------

class Logger:
    """General scalar logger (reference ``utils/logger.py``)."""

    def __init__(self, log_dir: str, backend: str = "tb"):
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        self.writer = EventFileWriter(log_dir) if backend in ("tb", "both") else None
        self._jsonl = open(os.path.join(log_dir, "scalars.jsonl"), "a") if backend in ("jsonl", "both") else None
        self.info = _log.info
        self.debug = _log.debug
        self.warning = _log.warning

    def scalar_summary(self, tag: str, value: float, step: int) -> None:
        """Log one scalar at ``step`` (the learner update step in every caller)."""
        if self.writer is not None:
            self.writer.add_scalar(tag, float(value), int(step))
        if self._jsonl is not None:
            self._jsonl.write(json.dumps({"wall": time.time(), "step": int(step), "tag": tag,
                                          "value": float(value)}) + "\n")

    def flush(self) -> None:
        if self.writer is not None:
            self.writer.flush()
        if self._jsonl is not None:
            self._jsonl.flush()

    def close(self) -> None:
        if self.writer is not None:
            self.writer.close()
        if self._jsonl is not None:
            self._jsonl.close()
            self._jsonl = None
