"""Build a variant of the HIP library for an A/B run inside ONE gpurun call (timings differ by a few
per cent from box to box, so variants are compared on the same box):

    python bench/exp/build_variant.py nw8 -DD4PG_NW=8      ->  bench/exp/variants/libd4pg_hip_nw8.so
    D4PG_HIP_LIB=bench/exp/variants/libd4pg_hip_nw8.so python bench.py ...
    python bench/exp/build_variant.py stamps32 --src=fused32.hip -DD4PG_STAMPS
    python bench/exp/build_variant.py ant_o2 --src=ant_env.hip --opt=-O2

Only ``fused.hip`` (or the file named by ``--src=``) is recompiled with the extra flags, as ONE translation
unit; the other objects come from the regular build.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from d4pg_pytorch_amd import build as B   # noqa: E402

tag, flags = sys.argv[1], sys.argv[2:]
src, opt = "fused.hip", None
for f in list(flags):
    if f.startswith("--src="):
        src = f.split("=", 1)[1]
        flags.remove(f)
    elif f.startswith("--opt="):       # optimisation level of the variant (default: the regular build's)
        opt = f.split("=", 1)[1]
        flags.remove(f)
B.build_hip()
out_dir = os.path.join(ROOT, "bench", "exp", "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"fused_{tag}.o")
subprocess.check_call([B._hipcc(), opt or B.OPT_LEVEL.get(src, "-O3"), f"--offload-arch={B.ARCH}", "-fPIC", "-std=c++17", *flags, "-c",
                       os.path.join(B.CSRC, src), "-o", obj])
objs = [obj]
for s in B.HIP_SOURCES:
    if s == src:
        continue
    parts = B.FUSED_PARTS.get(s, (None,))
    objs += [os.path.join(B.CSRC, "build", s + ("" if p is None else f".part{p}") + ".o") for p in parts]
lib = os.path.join(out_dir, f"libd4pg_hip_{tag}.so")
subprocess.check_call(["g++", "-shared", "-o", lib] + objs + [f"-L{B.torch_lib_dir()}", "-l:libamdhip64.so"])
os.remove(obj)
print(lib)
