"""Device buffers shared between processes through HIP IPC memory handles.

The reference's exploiter reads the learner's target policy in place: ``target_policy_net.share_memory()``
makes torch pass CUDA IPC handles across the spawn boundary (``models/d4pg/engine.py:129-144``), no copies.
Here the learner owns a :class:`SharedDeviceBuffer` holding ``[online actor | target actor]`` in the flat fp32
master layout and refreshes it with two device-to-device copies on its own stream (no host synchronisation);
GPU agents on the same machine map it with :meth:`SharedDeviceBuffer.open` and build the packed copies their
inference kernel streams straight from it (``actors/gpu_actor.py``).  Reader and writer are not synchronised,
like the reference's (torn reads are possible and harmless for a slowly moving Polyak average).

The 64-byte handle travels through the ``ControlBlock``-adjacent shared-memory file created by the engine
(:func:`write_handle` / :func:`read_handle`).  ``HSA_ENABLE_IPC_MODE_LEGACY=0`` must be in the environment of
both processes before the HIP runtime starts (dmabuf-only hosts).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


class _RawCuda:
    """Minimal ``__cuda_array_interface__`` carrier so that torch can wrap a raw device pointer."""

    def __init__(self, ptr: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class SharedDeviceBuffer:
    def __init__(self, ptr: int, numel: int, handle: bytes, owner: bool, device):
        self.ptr, self.numel, self.handle, self.owner = ptr, numel, handle, owner
        self.device = torch.device(device)
        self.tensor = torch.as_tensor(_RawCuda(ptr, numel), device=self.device)

    @staticmethod
    def create(numel: int, device="cuda") -> "SharedDeviceBuffer":
        from ..ops import hip as H
        dev = torch.device(device)
        with torch.cuda.device(dev):
            p, h = C.c_void_p(), C.create_string_buffer(64)
            H._check(H.lib().d4pg_ipc_alloc(numel * 4, C.byref(p), h), "d4pg_ipc_alloc")
        return SharedDeviceBuffer(p.value, numel, h.raw, True, dev)

    @staticmethod
    def open(handle: bytes, numel: int, device="cuda") -> "SharedDeviceBuffer":
        from ..ops import hip as H
        dev = torch.device(device)
        with torch.cuda.device(dev):
            p = C.c_void_p()
            H._check(H.lib().d4pg_ipc_open(C.c_char_p(handle), C.byref(p)), "d4pg_ipc_open")
        return SharedDeviceBuffer(p.value, numel, handle, False, dev)

    def close(self) -> None:
        from ..ops import hip as H
        if self.ptr:
            self.tensor = None
            (H.lib().d4pg_ipc_free if self.owner else H.lib().d4pg_ipc_close)(C.c_void_p(self.ptr))
            self.ptr = 0


def handle_path(ctrl_name: str) -> str:
    return os.path.join("/dev/shm", ctrl_name.strip("/") + ".hipipc")


def write_handle(ctrl_name: str, handle: bytes, numel: int, device_index: int) -> str:
    path = handle_path(ctrl_name)
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(handle + int(numel).to_bytes(8, "little") + int(device_index).to_bytes(4, "little"))
    os.replace(tmp, path)
    return path


def read_handle(ctrl_name: str) -> Optional[tuple]:
    try:
        with open(handle_path(ctrl_name), "rb") as fh:
            raw = fh.read()
    except FileNotFoundError:
        return None
    if len(raw) != 76:
        return None
    return raw[:64], int.from_bytes(raw[64:72], "little"), int.from_bytes(raw[72:76], "little")
