#!/bin/bash
# the rollout-worker learning test, several times in one process, with the worker's own account of its policy refreshes
O=gpurun_out/r3_worker_flaky
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6; do
  timeout -k 10 200 python -u -m pytest tests/test_rollout_worker.py -k engine_trains -x -q -s --timeout 170 --timeout-method thread > $O/run$i.log 2>&1
  echo "run $i exit $?"; grep -a "Exit rollout\|AssertionError: (\|passed\|failed" $O/run$i.log | cut -c1-250
done
