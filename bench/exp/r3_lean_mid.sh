#!/bin/bash
# Between 2048 and 4096 rows: one against two workgroups per CU (129 .. 255 row tiles of 16: more workgroups than CUs in a
# launch, fewer than two per CU)
set -o pipefail
O=gpurun_out/r3_lean_mid
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us', d['implementation'])"; }
for B in 2560 3072 3584; do
b walker${B}_4x4 --env Walker2d-v2 --batch $B --steps 800 --warmup 100 --mfma 4x4 || exit 1
b walker${B}_occ1 --env Walker2d-v2 --batch $B --steps 800 --warmup 100 --mfma 16x16 --occupancy 1 || exit 1
b walker${B}_occ2 --env Walker2d-v2 --batch $B --steps 800 --warmup 100 --mfma 16x16 --occupancy 2 || exit 1
done
