"""Builds the native code in-tree.

* ``csrc/libd4pg_hip.so``  — HIP kernels, ``hipcc --offload-arch=gfx950`` (cross-compiles
  without a GPU).  Objects are linked against the HIP runtime that PyTorch itself
  loads (``torch/lib/libamdhip64.so``), never a second copy: stream handles and
  device pointers are shared with torch.
* ``csrc/libd4pg_host.so`` — C++ host runtime (shared-memory transition rings,
  seq-locked weight slab, n-step assembler, OU noise, CPU actor inference), ``g++``.

Run ``python -m d4pg_pytorch_amd.build`` or call :func:`build_all`.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIP_SOURCES = ["gemm.hip", "loss.hip", "optim.hip", "replay.hip", "nstep.hip", "fused.hip", "fused32.hip", "rollout.hip", "planar_env.hip", "analytic_env.hip", "ant_env.hip"]
HOST_SOURCES = ["host_runtime.cpp", "planar_host.cpp"]
HIP_LIB = os.path.join(CSRC, "libd4pg_hip.so")
HOST_LIB = os.path.join(CSRC, "libd4pg_host.so")
ARCH = "gfx950"


def _run(cmd: List[str]) -> None:
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"command failed: {' '.join(cmd)}\n{proc.stdout}")


def _stale(target: str, sources: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def torch_lib_dir() -> str:
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


#: translation units of csrc/fused.hip / csrc/fused32.hip: -1 = C API + weight gradients, the others one
#: hidden width each (0 = the run-time-width variant, bf16 pipeline only)
FUSED_PARTS = {"fused.hip": (-1, 13, 4, 8, 0), "fused32.hip": (-1, 13, 4, 8)}


#: optimisation level per source where it is not -O3.  ant_env.hip: the fully unrolled 14 x 14 system in leg order
#: is miscompiled at -O2 / -O3 by hipcc 7.2 (wrong accelerations on the device; the -O1 build and the host build of
#: the same source agree with the numpy simulator) and -O1 is also the fastest build of it (290 registers, 84 us per
#: step of 4096 ants against 125 us for the -O3 build in link order); tests/test_spatial_env.py checks the result.
#: (planar_env.hip at -O1: 10 - 20 % faster launched stand-alone; inside the captured rollout Walker2d and the
#: walker are unchanged (68.8, 111 us per step of 4096 environments) and HalfCheetah gains 12 % (85 -> 75 us): left at -O3.)
OPT_LEVEL = {"ant_env.hip": "-O1"}


def build_hip(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    headers = [os.path.join(CSRC, h) for h in ("common.h", "loss_dev.h", "replay_dev.h", "phase_common.h", "mlp32_dev.h", "planar_sim.h", "ant_sim.h")]
    deps = srcs + headers
    if not force and not _stale(HIP_LIB, deps):
        return HIP_LIB
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    hipcc = _hipcc()
    jobs = []
    stamps = bool(os.environ.get("D4PG_STAMPS"))    # diagnostic build: per-layer clock stamps (bench/stamps.py)
    for s in srcs:
        base = os.path.basename(s)
        # fused.hip is built as five objects, one per hidden width of the phase kernels plus the main one
        # (see the head of the file); the stamp build keeps it in one piece
        parts = FUSED_PARTS[base] if (base in FUSED_PARTS and not (stamps and base == "fused.hip")) else (None,)
        for part in parts:
            o = os.path.join(objdir, base + ("" if part is None else f".part{part}") + ".o")
            if force or _stale(o, [s] + headers):
                cmd = [hipcc, OPT_LEVEL.get(base, "-O3"), f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-c", s, "-o", o]
                if part is not None:
                    cmd.insert(1, f"-DD4PG_KSH_PART={part}")
                if stamps:
                    cmd.insert(1, "-DD4PG_STAMPS")
                if verbose:
                    print(" ".join(cmd), flush=True)
                jobs.append(cmd)
            objs.append(o)
    if jobs:        # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "0")) or (os.cpu_count() or 1), 8))
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(_run, jobs))
    tl = torch_lib_dir()
    tmp = HIP_LIB + ".tmp"
    _run(["g++", "-shared", "-o", tmp] + objs + [f"-L{tl}", "-l:libamdhip64.so"])
    os.replace(tmp, HIP_LIB)
    return HIP_LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if not force and not _stale(HOST_LIB, srcs + [os.path.join(CSRC, "planar_sim.h"), os.path.join(CSRC, "ant_sim.h")]):
        return HOST_LIB
    tmp = HOST_LIB + ".tmp"
    cmd = ["g++", "-O3", "-march=x86-64-v3", "-fPIC", "-shared", "-std=c++17", "-pthread", "-fopenmp-simd"] + srcs + \
          ["-o", tmp, "-lrt"]
    if verbose:
        print(" ".join(cmd), flush=True)
    _run(cmd)
    os.replace(tmp, HOST_LIB)
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_host(force, verbose)
    build_hip(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built:", HIP_LIB, HOST_LIB)
