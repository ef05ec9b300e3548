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

import hashlib
import json
import os
import re
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


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


_INCLUDE = re.compile(rb'^\s*#\s*include\s*"([^"]+)"', re.M)


def _closure(src: str) -> List[str]:
    """`src` and every header of csrc/ it includes, directly or not (quoted includes only)."""
    seen, todo = [], [src]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        for m in _INCLUDE.finditer(_read(f)):
            todo.append(os.path.join(os.path.dirname(f), m.group(1).decode()))
    return sorted(seen)


def _digest(cmd: List[str], files: List[str]) -> str:
    """What an output depends on: the command line (flags, optimisation level, defines) and the CONTENT of
    its inputs.  Modification times say nothing after a fresh checkout or a copy to another machine."""
    h = hashlib.sha256()
    h.update("\0".join(os.path.basename(c) if os.sep in c else c for c in cmd).encode())
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + hashlib.sha256(_read(f)).digest())
    return h.hexdigest()


def _manifest_path(target: str) -> str:
    return target + ".manifest.json"


def _recorded(target: str) -> str:
    """Digest stored next to `target` when it was built, '' if either is missing."""
    try:
        with open(_manifest_path(target)) as f:
            return json.load(f)["digest"] if os.path.exists(target) else ""
    except (OSError, ValueError, KeyError):
        return ""


def _record(target: str, digest: str, parts: dict) -> None:
    with open(_manifest_path(target), "w") as f:
        json.dump({"digest": digest, "parts": parts}, f, indent=1, sort_keys=True)


#: what the last build_all() did, for the caller to print ("reused" or "compiled N objects" per library)
LAST_BUILD: dict = {}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def torch_lib_dir() -> str:
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


#: translation units of csrc/fused.hip / csrc/fused32.hip: -1 = C API + weight gradients, the others one
#: hidden width each (0 = the run-time-width variants: bf16 pipeline, and the 16x16x4 path of the fp32 pipeline)
FUSED_PARTS = {"fused.hip": (-1, 13, 4, 8, 0), "fused32.hip": (-1, 13, 4, 8, 2, 7, 10, 16, 0)}


#: optimisation level per source where it is not -O3.  ant_env.hip: -O1 is the FASTEST build of the fully unrolled
#: 14 x 14 system (85.6 us per step of 4096 ants against 98.8 at -O2 and 98.9 at -O3: 290 registers against 450), and
#: since the parameter pointer passes through an empty asm in every sub-step all three levels are correct (csrc/ant_sim.h
#: has the cause of the wrong -O2 / -O3 builds of earlier revisions; tests/test_spatial_env.py holds the device build
#: to the host simulator step by step at whatever level is built).
OPT_LEVEL: dict = {"ant_env.hip": "-O1"}


def _hip_units(stamps: bool):
    """(source, part, object path, command) of every translation unit of the HIP library."""
    objdir = os.path.join(CSRC, "build")
    hipcc = _hipcc()
    units = []
    for name in HIP_SOURCES:
        s = os.path.join(CSRC, name)
        # fused.hip / fused32.hip are built as one object per hidden width of the phase kernels plus the main one
        # (see the head of the files); the stamp build keeps fused.hip in one piece
        parts = FUSED_PARTS[name] if (name in FUSED_PARTS and not (stamps and name == "fused.hip")) else (None,)
        for part in parts:
            o = os.path.join(objdir, name + ("" if part is None else f".part{part}") + ".o")
            # --offload-compress: the gfx950 code objects are stored compressed in the fat binary (the HIP runtime inflates
            # them at load): the library shrinks from 15.9 to about 4 MB, which every copy of the tree to a GPU box pays for
            cmd = [hipcc, OPT_LEVEL.get(name, "-O3"), f"--offload-arch={ARCH}", "--offload-compress", "-fPIC", "-std=c++17",
                   "-c", s, "-o", o]
            if part is not None:
                cmd.insert(1, f"-DD4PG_KSH_PART={part}")
            if stamps:
                cmd.insert(1, "-DD4PG_STAMPS")
            units.append((s, part, o, cmd))
    return units


def build_hip(force: bool = False, verbose: bool = False) -> str:
    stamps = bool(os.environ.get("D4PG_STAMPS"))    # diagnostic build: per-layer clock stamps (bench/stamps.py)
    units = _hip_units(stamps)
    want = {os.path.basename(o): _digest(cmd, _closure(s)) for s, _, o, cmd in units}
    lib_digest = hashlib.sha256(json.dumps(want, sort_keys=True).encode()).hexdigest()
    if not force and _recorded(HIP_LIB) == lib_digest:
        LAST_BUILD["hip"] = "reused (digest of sources + flags matches the library's manifest)"
        return HIP_LIB
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    jobs = []
    for s, _, o, cmd in units:
        if force or _recorded(o) != want[os.path.basename(o)]:
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((o, cmd))
    if jobs:        # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "0")) or (os.cpu_count() or 1), 8))
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(_run, [cmd for _, cmd in jobs]))
        for o, _ in jobs:
            _record(o, want[os.path.basename(o)], {})
    tl = torch_lib_dir()
    tmp = HIP_LIB + ".tmp"
    _run(["g++", "-shared", "-o", tmp] + [o for _, _, o, _ in units] + [f"-L{tl}", "-l:libamdhip64.so"])
    os.replace(tmp, HIP_LIB)
    _record(HIP_LIB, lib_digest, want)
    LAST_BUILD["hip"] = f"compiled {len(jobs)} of {len(units)} objects, linked"
    return HIP_LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    tmp = HOST_LIB + ".tmp"
    cmd = ["g++", "-O3", "-march=x86-64-v3", "-fPIC", "-shared", "-std=c++17", "-pthread", "-fopenmp-simd"] + srcs + \
          ["-o", tmp, "-lrt"]
    digest = _digest(cmd, sorted({f for s in srcs for f in _closure(s)}))
    if not force and _recorded(HOST_LIB) == digest:
        LAST_BUILD["host"] = "reused (digest of sources + flags matches the library's manifest)"
        return HOST_LIB
    if verbose:
        print(" ".join(cmd), flush=True)
    _run(cmd)
    os.replace(tmp, HOST_LIB)
    _record(HOST_LIB, digest, {})
    LAST_BUILD["host"] = f"compiled {len(srcs)} sources, linked"
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_host(force, verbose)
    build_hip(force, verbose)
    if verbose:
        print(f"build: host runtime {LAST_BUILD['host']}; HIP kernels {LAST_BUILD['hip']}", flush=True)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built:", HIP_LIB, HOST_LIB)
