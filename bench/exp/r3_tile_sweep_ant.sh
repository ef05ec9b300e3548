#!/bin/bash
# 4x4x1 kernels, Ant shape: rows per workgroup (4 / 8) by batch size
set -o pipefail
O=gpurun_out/r3_tile_sweep
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name failed"; return 0; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for B in 512 640 768 896 1024; do
b ant${B}_t4 --env Ant-v2 --batch $B --steps 1500 --warmup 200 --tile-rows 4
b ant${B}_t8 --env Ant-v2 --batch $B --steps 1500 --warmup 200 --tile-rows 8
done
