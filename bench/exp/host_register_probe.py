#!/usr/bin/env python
"""Which host memory can the HIP runtime pin for DMA on this box?  (zero-copy ring ingest, models/learner.py)"""
import ctypes as C
import mmap
import os
import resource
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from d4pg_pytorch_amd.parallel.ipc import ShmRing  # noqa: E402

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipGetErrorString.restype = C.c_char_p
torch.cuda.init()
torch.zeros(1, device="cuda")
print("RLIMIT_MEMLOCK", resource.getrlimit(resource.RLIMIT_MEMLOCK), flush=True)


def probe(name, ptr, nbytes):
    for flags in (0, 1, 2, 3):
        rc = hip.hipHostRegister(ptr, nbytes, flags)
        print(f"{name}: {nbytes} bytes flags={flags} -> {rc} {hip.hipGetErrorString(rc).decode()}", flush=True)
        if rc == 0:
            hip.hipHostUnregister(ptr)


ring = ShmRing.create(4096, 56)
probe("shm ring data (MAP_SHARED, /dev/shm)", ring.data_ptr, ring.data_bytes)
probe("shm ring data, 64 KB only", ring.data_ptr, 65536)
anon = mmap.mmap(-1, 1 << 20)
buf = (C.c_char * (1 << 20)).from_buffer(anon)
probe("anonymous private mmap", C.addressof(buf), 1 << 20)
arr = torch.zeros(1 << 18)
probe("malloc (torch cpu tensor)", arr.data_ptr(), arr.numel() * 4)
fd = os.open("/tmp/probe_file", os.O_RDWR | os.O_CREAT)
os.ftruncate(fd, 1 << 20)
fm = mmap.mmap(fd, 1 << 20)
fbuf = (C.c_char * (1 << 20)).from_buffer(fm)
probe("MAP_SHARED file in /tmp", C.addressof(fbuf), 1 << 20)
