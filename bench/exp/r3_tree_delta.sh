#!/bin/bash
# priority write-back by differences: tests, then the prioritised headline with and without it
O=gpurun_out/r3_tree_delta
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_hbm_replay.py tests/test_hip_f32.py tests/test_replicated_dp.py tests/test_hip_bf16.py -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for d in 1 0; do
  D4PG_TREE_DELTA=$d timeout -k 10 150 python bench.py --per --no-extras > $O/bench_per_delta$d.json 2> $O/bench_per_delta$d.err || exit 1
  D4PG_TREE_DELTA=$d timeout -k 10 150 python bench.py --per --no-extras --no-bf16 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 > $O/bench_ant_per_delta$d.json 2> $O/bench_ant_per_delta$d.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_tree_delta/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('bf16_ms_per_step'))
PY
