#!/bin/bash
# Where the 16x16x4 kernels start to win over the 4x4x1 ones (just above 2048 rows) and where two workgroups per CU do
set -o pipefail
O=gpurun_out/r3_lean_mid
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for B in 2064 2176 2304; do
b walker${B}_4x4 --env Walker2d-v2 --batch $B --steps 800 --warmup 100 --mfma 4x4 || exit 1
b walker${B}_occ1 --env Walker2d-v2 --batch $B --steps 800 --warmup 100 --mfma 16x16 --occupancy 1 || exit 1
done
for B in 3200 3328 3456; do
b walker${B}_occ1 --env Walker2d-v2 --batch $B --steps 800 --warmup 100 --mfma 16x16 --occupancy 1 || exit 1
b walker${B}_occ2 --env Walker2d-v2 --batch $B --steps 800 --warmup 100 --mfma 16x16 --occupancy 2 || exit 1
done
b ant3072_occ1 --env Ant-v2 --batch 3072 --steps 800 --warmup 100 --mfma 16x16 --occupancy 1 || exit 1
b ant3072_occ2 --env Ant-v2 --batch 3072 --steps 800 --warmup 100 --mfma 16x16 --occupancy 2 || exit 1
b ant3072_4x4 --env Ant-v2 --batch 3072 --steps 800 --warmup 100 --mfma 4x4 || exit 1
