#!/bin/bash
# Prioritised replay at 4096 rows on the lean kernels: the next batch drawn inside the actor-phase launch (room for 512 workgroups)
set -o pipefail
O=gpurun_out/r3_lean_per
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_hip_f32.py tests/test_replicated_dp.py tests/test_hbm_replay.py -x -q --timeout 300 --timeout-method thread -k "per or prior or PER or replay" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us', d.get('losses_finite'))"; }
b ant4096_per --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --per || exit 1
D4PG_PER_AHEAD=0 b ant4096_per_noahead --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --per || exit 1
b walker4096_per --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 --per || exit 1
