#!/bin/bash
# Lean kernels at 8192 rows: the critic's forward pass for the next update made in the actor-phase launch (1024 workgroups) or not
# (D4PG_AHEAD_TILES overrode HipCoreF32.ahead_tiles for this run only: no difference, bench/exp/README.md; it is ignored now)
set -o pipefail
O=gpurun_out/r3_ahead8192
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for t in 256 512; do
export D4PG_AHEAD_TILES=$t
b walker8192_ahead$t --env Walker2d-v2 --batch 8192 --steps 400 --warmup 50 || exit 1
b ant8192_ahead$t --env Ant-v2 --batch 8192 --steps 400 --warmup 50 || exit 1
done
export D4PG_AHEAD_TILES=1024
b walker16384_ahead1024 --env Walker2d-v2 --batch 16384 --steps 200 --warmup 30 || exit 1
export D4PG_AHEAD_TILES=256
b walker16384_ahead256 --env Walker2d-v2 --batch 16384 --steps 200 --warmup 30 || exit 1
