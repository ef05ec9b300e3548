#!/bin/bash
# 1024 / 2048 rows at the other hidden widths: 4x4x1 kernels (what `auto` picks there) against the 16x16x4 kernels
set -o pipefail
O=gpurun_out/r3_width_big
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us', d['implementation'][-22:])"; }
for B in 1024 2048; do
for h in 128 256 512; do
b walker${B}_h${h}_auto --env Walker2d-v2 --batch $B --hidden $h --steps 1000 --warmup 100 || exit 1
b walker${B}_h${h}_16x16 --env Walker2d-v2 --batch $B --hidden $h --steps 1000 --warmup 100 --mfma 16x16 || exit 1
done
done
