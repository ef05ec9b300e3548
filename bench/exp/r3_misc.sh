#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3_misc
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_replicated_dp.py tests/test_gpu_engine.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_misc/pytest.log 2>&1 || { tail -40 gpurun_out/r3_misc/pytest.log; exit 1; }
tail -3 gpurun_out/r3_misc/pytest.log
O=$GRAFT_REPO_ROOT/gpurun_out/r3_misc
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o walker --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench/learn_planar.py" --env Walker2d-v2 --updates 20000 --envs 256 --segment 32 --overlap --out $O/prof_run > $O/prof.log 2>&1 &&
cd "$GRAFT_REPO_ROOT" && python tools/rocprof_summary.py stats $O/prof $O/kernel_stats_walker2d_on_device_overlap.csv && rm -rf $O/prof && head -12 $O/kernel_stats_walker2d_on_device_overlap.csv | cut -c1-160
