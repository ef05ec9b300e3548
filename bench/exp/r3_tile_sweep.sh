#!/bin/bash
# 4x4x1 kernels, Walker2d shape: rows per workgroup (4 / 8 / 16) by batch size, against what `auto` picks
set -o pipefail
O=gpurun_out/r3_tile_sweep
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name failed"; return 0; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for B in 384 512 640 768 896 1024; do
b walker${B}_t4 --env Walker2d-v2 --batch $B --steps 1500 --warmup 200 --tile-rows 4
b walker${B}_t8 --env Walker2d-v2 --batch $B --steps 1500 --warmup 200 --tile-rows 8
done
for B in 1280 1536 1792 2048; do
b walker${B}_t8 --env Walker2d-v2 --batch $B --steps 1000 --warmup 100 --tile-rows 8
b walker${B}_t16 --env Walker2d-v2 --batch $B --steps 1000 --warmup 100 --tile-rows 16 --mfma 4x4
b walker${B}_w16 --env Walker2d-v2 --batch $B --steps 1000 --warmup 100 --mfma 16x16
done
