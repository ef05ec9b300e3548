#!/bin/bash
# Weight-gradient kernel at 4096 rows: 16 waves with 4 operand pairs in flight against 8 waves with 8 / 12 / 16
# (D4PG_WGRAD_DEPTH selected instances of a WG_DEPTH template parameter that existed for this run only: 220 - 222 us
# against 216.6, not kept; as the kernel is now the variable is ignored)
set -o pipefail
O=gpurun_out/r3_wgrad_depth
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
b ant4096_w16 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
export D4PG_WGRAD_WAVES=8
b ant4096_w8_d4 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
for dp in 8 12 16; do
export D4PG_WGRAD_DEPTH=$dp
b ant4096_w8_d$dp --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
done
b walker2048_w8_d16 --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 || exit 1
