"""Transition rows from rollout workers on other GPUs (or other processes on the learner's GPU) into the learner's
HBM, without a host hop.

The reference's explorers push one pickled transition per environment step through a pipe to the sampler
(``models/d4pg/engine.py:44-47``, ``models/agent.py:101-105``).  A rollout worker (``actors/rollout_worker.py``)
produces a whole segment of ``T x E`` rows per graph replay in ITS device memory; the learner owns a staging ring in
ITS device memory, exported through a HIP IPC handle:

    worker k                                             learner
    --------                                             -------
    wait until head[k] - tail[k] < slots
    copy rows -> slot head[k] % slots   (device-to-device: same GPU, or a peer write over xGMI)
    synchronise its stream
    head[k] += 1                        (host atomics)
                                                         sees head[k] > taken[k]
                                                         replay_insert(slot) on its stream, event behind it
                                                         ... event complete: tail[k] += 1   (polled, never waited for)

The counters live in a ``ControlBlock`` of their own (shared-memory atomics); nothing here blocks the learner.
A worker that is restarted picks up the counters where its predecessor left them.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional, Tuple

import torch

from .hip_ipc import SharedDeviceBuffer
from .ipc import ControlBlock


def mailbox_path(ctrl_name: str, learner_rank: int = 0) -> str:
    """Handle file of the mailbox of learner rank ``learner_rank`` (every learner rank that is dealt rollout workers
    owns one, in its own GPU's memory)."""
    suffix = ".rows.hipipc" if learner_rank == 0 else f".rows{int(learner_rank)}.hipipc"
    return os.path.join("/dev/shm", ctrl_name.strip("/") + suffix)


class RowMailbox:
    HEAD, TAIL = 0, 1      # per worker k: counter slots 2 k, 2 k + 1

    def __init__(self, shared: SharedDeviceBuffer, counters: ControlBlock, n_workers: int, slots: int,
                 rows_per_slot: int, row_ld: int, owner: bool):
        self.shared, self.counters = shared, counters
        self.n_workers, self.slots, self.rows_per_slot, self.row_ld = n_workers, slots, rows_per_slot, row_ld
        self.owner = owner
        self.view = shared.tensor.view(n_workers, slots, rows_per_slot, row_ld)
        self.taken = [counters.load(2 * k + self.HEAD) for k in range(n_workers)] if owner else None
        self.inflight: List[Tuple[torch.cuda.Event, int]] = []
        self.rows_received = 0

    # ------------------------------------------------------------------ learner side
    @staticmethod
    def create(ctrl_name: str, n_workers: int, rows_per_slot: int, row_ld: int, device, slots: int = 3,
               learner_rank: int = 0) -> "RowMailbox":
        dev = torch.device(device)
        shared = SharedDeviceBuffer.create(n_workers * slots * rows_per_slot * row_ld, dev)
        counters = ControlBlock.create(n_workers=max(1, n_workers))
        meta = (shared.handle + int(shared.numel).to_bytes(8, "little") + int(dev.index or 0).to_bytes(4, "little")
                + int(n_workers).to_bytes(4, "little") + int(slots).to_bytes(4, "little")
                + int(rows_per_slot).to_bytes(4, "little") + int(row_ld).to_bytes(4, "little") + counters.name.encode())
        path = mailbox_path(ctrl_name, learner_rank)
        with open(path + ".tmp", "wb") as fh:
            fh.write(meta)
        os.replace(path + ".tmp", path)
        mb = RowMailbox(shared, counters, n_workers, slots, rows_per_slot, row_ld, True)
        mb._path = path
        return mb

    def poll(self, replay) -> int:
        """Insert every filled slot into ``replay`` (enqueue only) and hand finished slots back.  Returns the rows
        inserted by this call."""
        self.release()
        n = 0
        for k in range(self.n_workers):
            head = self.counters.load(2 * k + self.HEAD)
            while self.taken[k] < head:
                replay.insert(self.view[k, self.taken[k] % self.slots])
                ev = torch.cuda.Event()
                ev.record()
                self.inflight.append((ev, k))
                self.taken[k] += 1
                n += self.rows_per_slot
        self.rows_received += n
        return n

    def release(self, block: bool = False) -> None:
        while self.inflight:
            ev, k = self.inflight[0]
            if block:
                ev.synchronize()
            elif not ev.query():
                return
            self.counters.add(2 * k + self.TAIL, 1)
            self.inflight.pop(0)

    # ------------------------------------------------------------------ worker side
    @staticmethod
    def open(ctrl_name: str, device, timeout_s: float = 60.0, should_stop=None, learner_rank: int = 0) -> Optional["RowMailbox"]:
        """Map a learner rank's staging ring into this process (peer memory when ``device`` is another GPU)."""
        path, deadline = mailbox_path(ctrl_name, learner_rank), time.time() + timeout_s
        while not os.path.exists(path):
            if time.time() > deadline or (should_stop is not None and should_stop()):
                return None
            time.sleep(0.01)
        raw = open(path, "rb").read()
        handle, numel, dev_index = raw[:64], int.from_bytes(raw[64:72], "little"), int.from_bytes(raw[72:76], "little")
        n_workers, slots, rows, row_ld = (int.from_bytes(raw[76 + 4 * i: 80 + 4 * i], "little") for i in range(4))
        dev = torch.device(device)
        mine = dev.index or 0
        if mine != dev_index and not torch.cuda.can_device_access_peer(mine, dev_index):
            raise RuntimeError(f"cuda:{mine} has no peer access to the learner's cuda:{dev_index}")
        shared = SharedDeviceBuffer.open(handle, numel, dev)
        counters = ControlBlock.attach(raw[92:].decode())
        return RowMailbox(shared, counters, n_workers, slots, rows, row_ld, False)

    def push(self, k: int, rows: torch.Tensor, should_stop=None) -> bool:
        """Worker ``k``: deliver one segment (``rows_per_slot x row_ld``, on this process's device).  Blocks while the
        ring is full; returns False if told to stop first."""
        assert rows.shape == (self.rows_per_slot, self.row_ld)
        head = self.counters.load(2 * k + self.HEAD)
        while head - self.counters.load(2 * k + self.TAIL) >= self.slots:
            if should_stop is not None and should_stop():
                return False
            time.sleep(0.0002)
        self.view[k, head % self.slots].copy_(rows, non_blocking=True)
        torch.cuda.current_stream(rows.device).synchronize()      # the rows are in the learner's memory ...
        self.counters.store(2 * k + self.HEAD, head + 1)          # ... before the learner is told
        return True

    def close(self) -> None:
        if self.owner:
            self.release(block=True)
            try:
                os.unlink(self._path)
            except OSError:
                pass
        self.view = None
        self.shared.close()
        self.counters.close()
