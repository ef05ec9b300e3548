#!/bin/bash
# 4096 rows at the other hidden widths: 4x4x1 kernels (16-row tiles) against the 16x16x4 kernels (what `auto` picks there)
set -o pipefail
O=gpurun_out/r3_width_big
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us', d['implementation'][-22:])"; }
for h in 64 128 200 256 300 512; do
b walker4096_h${h}_4x4 --env Walker2d-v2 --batch 4096 --hidden $h --steps 800 --warmup 100 --mfma 4x4 || exit 1
b walker4096_h${h}_auto --env Walker2d-v2 --batch 4096 --hidden $h --steps 800 --warmup 100 || exit 1
done
