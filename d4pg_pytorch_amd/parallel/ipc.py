"""Python front-ends of the native IPC primitives in ``csrc/host_runtime.cpp``.

=================  ==========================================================
 primitive          replaces (reference)
=================  ==========================================================
 :class:`ShmRing`    ``replay_queue`` — ``mp.Queue(64)`` of pickled transitions
                     with silent drops (``models/d4pg/engine.py:114``,
                     ``models/agent.py:101-105``)
 :class:`WeightSlab` ``learner_w_queue`` — a queue where each weight snapshot
                     reaches one explorer only (``models/d4pg/d4pg.py:141-146``)
 :class:`ControlBlock` ``training_on`` / ``update_step`` / ``global_episode``
                     ``mp.Value`` ints updated without locks
                     (``models/d4pg/engine.py:115-117``, ``models/agent.py:64``)
=================  ==========================================================

All three live in POSIX shared memory and are addressed by name, so a worker
process only needs the (picklable) name to attach.
"""
from __future__ import annotations

import ctypes as C
import os
import uuid
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libd4pg_host.so")
_lib: Optional[C.CDLL] = None

vp, u64, i64 = C.c_void_p, C.c_uint64, C.c_int64
fp = C.POINTER(C.c_float)


def host_lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            from .. import build
            build.build_host()
        L = C.CDLL(HOST_LIB_PATH)
        sig = {
            "ring_create": (vp, [C.c_char_p, u64, u64]), "ring_attach": (vp, [C.c_char_p]),
            "ring_detach": (None, [vp]), "shm_remove": (C.c_int, [C.c_char_p]),
            "ring_capacity": (u64, [vp]), "ring_row_floats": (u64, [vp]), "ring_data_ptr": (vp, [vp]),
            "ring_data_bytes": (u64, [vp]), "ring_size": (u64, [vp]), "ring_dropped": (u64, [vp]),
            "ring_pushed_total": (u64, [vp]), "ring_close": (None, [vp]), "ring_is_closed": (C.c_int, [vp]),
            "ring_push": (i64, [vp, vp, i64, i64]), "ring_pop": (i64, [vp, vp, i64]),
            "ring_peek": (i64, [vp, C.POINTER(u64), i64]), "ring_advance": (None, [vp, i64]),
            "ring_peek_from": (i64, [vp, u64, C.POINTER(u64), i64]),
            "ring_set_check": (None, [vp, C.c_int]), "ring_seq_errors": (u64, [vp]),
            "slab_create": (vp, [C.c_char_p, u64]), "slab_attach": (vp, [C.c_char_p]),
            "slab_detach": (None, [vp]), "slab_data_ptr": (vp, [vp]), "slab_n_floats": (u64, [vp]),
            "slab_version": (u64, [vp]), "slab_step": (i64, [vp]), "slab_begin_write": (None, [vp]),
            "slab_end_write": (None, [vp, i64]), "slab_publish": (None, [vp, vp, i64]),
            "slab_read": (C.c_int, [vp, vp, C.POINTER(u64)]),
            "ctrl_create": (vp, [C.c_char_p, u64]), "ctrl_attach": (vp, [C.c_char_p]),
            "ctrl_detach": (None, [vp]), "ctrl_slots": (u64, [vp]), "ctrl_load": (i64, [vp, u64]),
            "ctrl_store": (None, [vp, u64, i64]), "ctrl_add": (i64, [vp, u64, i64]),
            "ctrl_max": (i64, [vp, u64, i64]),
            "ou_create": (vp, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, i64, u64]),
            "ou_destroy": (None, [vp]), "ou_reset": (None, [vp, C.c_int]), "ou_state": (vp, [vp]),
            "ou_sample": (None, [vp, vp, vp, C.c_float, C.c_float, vp]),
            "nstep_create": (vp, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
            "nstep_destroy": (None, [vp]), "nstep_row_ld": (C.c_int, [vp]), "nstep_reset": (None, [vp, C.c_int]),
            "nstep_push": (i64, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64]),
            "actor_forward": (None, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_float]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def unique_name(prefix: str) -> str:
    return f"/{prefix}_{os.getpid()}_{uuid.uuid4().hex[:10]}"


def np_ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class _Shm:
    """Common create/attach/cleanup handling (owner unlinks the name on close)."""
    _create = _attach = _detach = ""

    def __init__(self, name: str, handle: int, owner: bool):
        if not handle:
            raise OSError(f"{type(self).__name__}: cannot {'create' if owner else 'attach'} shared segment {name}")
        self.name, self._h, self._owner = name, handle, owner

    def close(self) -> None:
        if getattr(self, "_h", None):
            getattr(host_lib(), self._detach)(self._h)
            self._h = None
            if self._owner:
                host_lib().shm_remove(self.name.encode())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ShmRing(_Shm):
    """SPSC ring of fixed-width float32 rows."""
    _detach = "ring_detach"

    @classmethod
    def create(cls, capacity: int, row_floats: int, name: Optional[str] = None) -> "ShmRing":
        name = name or unique_name("d4pg_ring")
        return cls(name, host_lib().ring_create(name.encode(), capacity, row_floats), True)

    @classmethod
    def attach(cls, name: str) -> "ShmRing":
        return cls(name, host_lib().ring_attach(name.encode()), False)

    @property
    def capacity(self) -> int:
        return host_lib().ring_capacity(self._h)

    @property
    def row_floats(self) -> int:
        return host_lib().ring_row_floats(self._h)

    def __len__(self) -> int:
        return host_lib().ring_size(self._h)

    @property
    def dropped(self) -> int:
        return host_lib().ring_dropped(self._h)

    @property
    def pushed_total(self) -> int:
        return host_lib().ring_pushed_total(self._h)

    def set_check(self, on: bool = True) -> None:
        """Race detection: the producer stamps every slot with the running index of its row and the
        consumer verifies it on consumption (``seq_errors``).  Set by the creator before any push."""
        host_lib().ring_set_check(self._h, 1 if on else 0)

    @property
    def seq_errors(self) -> int:
        return host_lib().ring_seq_errors(self._h)

    def mark_closed(self) -> None:
        host_lib().ring_close(self._h)

    @property
    def closed(self) -> bool:
        return bool(host_lib().ring_is_closed(self._h))

    def push(self, rows: np.ndarray, timeout_ms: int = 10_000) -> int:
        """Producer side.  Returns the number of rows accepted; the rest is
        counted in :attr:`dropped`.  ``timeout_ms=0`` never blocks."""
        rows = _f32c(rows)
        if rows.ndim == 1:
            rows = rows.reshape(1, -1)
        assert rows.shape[1] == self.row_floats, (rows.shape, self.row_floats)
        return host_lib().ring_push(self._h, np_ptr(rows), rows.shape[0], int(timeout_ms))

    def pop(self, max_rows: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.empty((max_rows, self.row_floats), dtype=np.float32)
        n = host_lib().ring_pop(self._h, np_ptr(out), min(max_rows, out.shape[0]))
        return out[:n]

    def peek(self, max_rows: int, ahead: int = 0) -> Tuple[int, int]:
        """(row offset into the data area, contiguous row count) without consuming.  ``ahead``: rows past the
        read cursor that were peeked before and are still waiting for :meth:`advance` (a DMA in flight)."""
        off = u64(0)
        if ahead:
            n = host_lib().ring_peek_from(self._h, int(ahead), C.byref(off), max_rows)
        else:
            n = host_lib().ring_peek(self._h, C.byref(off), max_rows)
        return off.value, n

    def advance(self, rows: int) -> None:
        host_lib().ring_advance(self._h, rows)

    def data_view(self) -> np.ndarray:
        """The whole data area as ``[capacity, row_floats]`` (for zero-copy staging)."""
        nbytes = host_lib().ring_data_bytes(self._h)
        buf = (C.c_char * nbytes).from_address(host_lib().ring_data_ptr(self._h))
        return np.frombuffer(buf, dtype=np.float32).reshape(self.capacity, self.row_floats)

    @property
    def data_ptr(self) -> int:
        return host_lib().ring_data_ptr(self._h)

    @property
    def data_bytes(self) -> int:
        return host_lib().ring_data_bytes(self._h)


class WeightSlab(_Shm):
    """Seq-locked broadcast buffer for a flat float32 parameter vector."""
    _detach = "slab_detach"

    @classmethod
    def create(cls, n_floats: int, name: Optional[str] = None) -> "WeightSlab":
        name = name or unique_name("d4pg_slab")
        return cls(name, host_lib().slab_create(name.encode(), n_floats), True)

    @classmethod
    def attach(cls, name: str) -> "WeightSlab":
        return cls(name, host_lib().slab_attach(name.encode()), False)

    @property
    def n_floats(self) -> int:
        return host_lib().slab_n_floats(self._h)

    @property
    def version(self) -> int:
        return host_lib().slab_version(self._h)

    @property
    def update_step(self) -> int:
        return host_lib().slab_step(self._h)

    def publish(self, flat: np.ndarray, update_step: int = 0) -> None:
        flat = _f32c(flat).reshape(-1)
        assert flat.size == self.n_floats
        host_lib().slab_publish(self._h, np_ptr(flat), int(update_step))

    def begin_write(self) -> None:
        host_lib().slab_begin_write(self._h)

    def end_write(self, update_step: int = 0) -> None:
        host_lib().slab_end_write(self._h, int(update_step))

    def data_view(self) -> np.ndarray:
        buf = (C.c_char * (self.n_floats * 4)).from_address(host_lib().slab_data_ptr(self._h))
        return np.frombuffer(buf, dtype=np.float32)

    @property
    def data_ptr(self) -> int:
        return host_lib().slab_data_ptr(self._h)


class SlabReader:
    """Per-consumer cursor over a :class:`WeightSlab`."""

    def __init__(self, slab: WeightSlab):
        self.slab = slab
        self._last = u64(0)
        self.buf = np.zeros(slab.n_floats, dtype=np.float32)

    def poll(self) -> bool:
        """Copy the newest snapshot into :attr:`buf` if there is one not yet seen."""
        rc = host_lib().slab_read(self.slab._h, np_ptr(self.buf), C.byref(self._last))
        if rc < 0:
            raise TimeoutError("weight slab writer appears stuck mid-publication")
        return rc == 1

    @property
    def version(self) -> int:
        return self._last.value


class ControlBlock(_Shm):
    """Shared 64-bit atomic counters / flags."""
    _detach = "ctrl_detach"
    # fixed slots
    STOP, UPDATE_STEP, GLOBAL_EPISODE, ENV_STEPS, LEARNER_ALIVE, FAILED = range(6)
    #: device-resident weight hand-off (parallel/hip_ipc.py): completed publications / learner step of the newest
    GPU_WEIGHT_VERSION, GPU_WEIGHT_STEP = 6, 7
    #: publications whose device-to-device copies have been ENQUEUED (>= completed).  Publication p (1-based) writes
    #: slot (p - 1) % 2, so a reader of publication v was overtaken iff this reads >= v + 2 when its pack has finished.
    GPU_WEIGHT_ISSUED = 8
    #: slots 9 .. 15 are free for tools that share a control block (``bench.py`` counts finished rollout workers in 10)
    TOOL_SLOT0 = 10
    N_FIXED = 16

    @classmethod
    def create(cls, n_workers: int = 64, name: Optional[str] = None) -> "ControlBlock":
        name = name or unique_name("d4pg_ctrl")
        return cls(name, host_lib().ctrl_create(name.encode(), cls.N_FIXED + 2 * n_workers), True)

    @classmethod
    def attach(cls, name: str) -> "ControlBlock":
        return cls(name, host_lib().ctrl_attach(name.encode()), False)

    def load(self, slot: int) -> int:
        return host_lib().ctrl_load(self._h, slot)

    def store(self, slot: int, value: int) -> None:
        host_lib().ctrl_store(self._h, slot, int(value))

    def add(self, slot: int, value: int = 1) -> int:
        """Atomic fetch-add; returns the NEW value."""
        return host_lib().ctrl_add(self._h, slot, int(value))

    def max(self, slot: int, value: int) -> int:
        return host_lib().ctrl_max(self._h, slot, int(value))

    # --- named accessors -------------------------------------------------
    @property
    def n_workers(self) -> int:
        return (host_lib().ctrl_slots(self._h) - self.N_FIXED) // 2

    @property
    def stop(self) -> bool:
        return self.load(self.STOP) != 0

    def request_stop(self) -> None:
        self.store(self.STOP, 1)

    @property
    def update_step(self) -> int:
        return self.load(self.UPDATE_STEP)

    @property
    def global_episode(self) -> int:
        return self.load(self.GLOBAL_EPISODE)

    @property
    def env_steps(self) -> int:
        return self.load(self.ENV_STEPS)

    def heartbeat(self, worker: int, now_ms: int) -> None:
        self.store(self.N_FIXED + worker, now_ms)

    def last_heartbeat(self, worker: int) -> int:
        return self.load(self.N_FIXED + worker)

    def worker_env_steps_slot(self, worker: int) -> int:
        return self.N_FIXED + self.n_workers + worker


def cleanup_names(names: List[str]) -> None:
    """Best-effort unlink of shared segments (supervisor teardown after a crash)."""
    for n in names:
        try:
            host_lib().shm_remove(n.encode())
        except Exception:
            pass
