#!/bin/bash
# Lean launches at 4096 rows: the short role of a launch starts some microseconds late, so that its hidden layer meets
# the long role's loss rows instead of its hidden layers.  (D4PG_ROLE_DELAY_US / D4PG_AHEAD_DELAY_US drove a wait that was built
# for this run only: every delay cost time, bench/exp/README.md; as the kernels are now the variables are ignored.)
set -o pipefail
O=gpurun_out/r3_delay
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for c in 0 25 35 45 55; do
export D4PG_ROLE_DELAY_US=$c D4PG_AHEAD_DELAY_US=0
b ant4096_c${c}_a0 --env Ant-v2 --batch 4096 --steps 600 --warmup 100 || exit 1
done
export D4PG_ROLE_DELAY_US=0
for a in 15 25 35 45; do
export D4PG_AHEAD_DELAY_US=$a
b ant4096_c0_a$a --env Ant-v2 --batch 4096 --steps 600 --warmup 100 || exit 1
done
