#!/bin/bash
# Lean build beyond 4096 rows: the two roles of a launch in halves of the grid (0) or in alternating runs of 256 workgroups (1)
set -o pipefail
O=gpurun_out/r3_lean
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for m in 0 1; do
export D4PG_ROLE_MAP=$m
b walker8192_occ2_map$m --env Walker2d-v2 --batch 8192 --steps 400 --warmup 50 --occupancy 2 || exit 1
b ant8192_occ2_map$m --env Ant-v2 --batch 8192 --steps 400 --warmup 50 --occupancy 2 || exit 1
done
b ant8192_occ1 --env Ant-v2 --batch 8192 --steps 400 --warmup 50 --occupancy 1 || exit 1
export D4PG_ROLE_MAP=1
timeout -k 10 300 python -u -m pytest tests/test_hip_f32.py -x -q --timeout 300 --timeout-method thread -k "drawn_ahead or kw26" > $O/pytest_map1.log 2>&1 || { tail -40 $O/pytest_map1.log; exit 1; }
tail -3 $O/pytest_map1.log
