#!/bin/bash
# Where do kernel arguments live?  The fp32 phase kernels read their argument struct in place, layer by layer (scalar
# loads from the kernel-argument segment): if that segment is host memory every first touch of a 64-byte line is a trip
# over the host link.  A/B of the runtime's placement switch.
set -e
O=gpurun_out/kernarg_ab
mkdir -p $O
for v in default 0 1; do
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout -k 10 120 python bench.py --no-bf16 > $O/fp32_$v.json 2> $O/fp32_$v.err
  timeout -k 10 120 python bench.py --dtype bf16 > $O/bf16_$v.json 2> $O/bf16_$v.err
  python - <<PY
import json
for d in ("fp32","bf16"):
    j=json.loads(open("$O/%s_$v.json"%d).read().strip().splitlines()[-1])
    print("HIP_FORCE_DEV_KERNARG=$v", d, j["value"], round(j["ms_per_step"]*1000,2), "us")
PY
done
