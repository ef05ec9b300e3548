#!/bin/bash
# Ant shape (119 inputs) at 512 .. 2048 rows: 4x4x1 against 16x16x4 kernels
set -o pipefail
O=gpurun_out/r3_ant_mid
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for B in 512 1024 2048; do
b ant${B}_4x4 --env Ant-v2 --batch $B --steps 1500 --warmup 200 --mfma 4x4 || exit 1
b ant${B}_16x16 --env Ant-v2 --batch $B --steps 1500 --warmup 200 --mfma 16x16 || exit 1
done
b ant256_4x4 --env Ant-v2 --batch 256 --mfma 4x4 || exit 1
b ant256_16x16 --env Ant-v2 --batch 256 --mfma 16x16 || exit 1
