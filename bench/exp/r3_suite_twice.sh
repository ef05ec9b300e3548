#!/bin/bash
# the GPU suite twice in a row on one box: anything that only fails sometimes should show here
O=gpurun_out/r3_suite_twice
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  timeout -k 10 500 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider > $O/run$i.log 2>&1; echo "run $i exit $?"; tail -4 $O/run$i.log | cut -c1-200
done
