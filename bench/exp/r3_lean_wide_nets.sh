#!/bin/bash
# Networks too wide for four 16-row LDS tiles (512, 768, 1024): the one-workgroup build on 8-row tiles against the lean
# build on 16-row tiles (two shared tiles: fits up to ~1000 columns, one workgroup per CU)
set -o pipefail
O=gpurun_out/r3_lean_wide_nets
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 200 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name failed"; tail -2 $O/bench_$name.err; return 0; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for h in 512 768 1024; do
for B in 256 1024 2048 4096; do
b walker${B}_h${h}_occ1 --env Walker2d-v2 --batch $B --hidden $h --mfma 16x16 --occupancy 1 --steps 400 --warmup 50
b walker${B}_h${h}_occ2 --env Walker2d-v2 --batch $B --hidden $h --mfma 16x16 --occupancy 2 --steps 400 --warmup 50
done
done
