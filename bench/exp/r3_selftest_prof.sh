#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_selftest
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --dp-selftest --dp-mode grad_allreduce --no-extras --no-bf16 --steps 300 --warmup 50 > $O/bench.json 2> $O/bench.err &&
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py stats $O/prof $O/kernel_stats_dp_selftest_fp32.csv && rm -rf $O/prof && head -8 $O/kernel_stats_dp_selftest_fp32.csv | cut -c1-150
timeout -k 10 600 python -u -m pytest tests/test_dp_hip.py tests/test_gpu_engine.py tests/test_hip_kernels.py -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 200 python bench.py --dp-selftest --no-extras --no-bf16 > $O/bench_selftest.json 2>> $O/bench.err && grep -o '"allreduce_probe_us[^,]*,\|"grad_allreduce_ms_per_step[^,]*,\|"replicated_ms_per_step[^,]*,' $O/bench_selftest.json
timeout -k 10 200 python bench.py --dp-selftest --no-extras --no-bf16 --env Ant-v2 --batch 4096 > $O/bench_selftest_ant.json 2>> $O/bench.err && grep -o '"allreduce_probe_us[^,]*,\|"grad_allreduce_ms_per_step[^,]*,\|"replicated_ms_per_step[^,]*,' $O/bench_selftest_ant.json
timeout -k 10 200 python bench.py --dp-selftest --no-extras --dtype bf16 > $O/bench_selftest_bf16.json 2>> $O/bench.err && grep -o '"allreduce_probe_us[^,]*,\|"grad_allreduce_ms_per_step[^,]*,\|"replicated_ms_per_step[^,]*,' $O/bench_selftest_bf16.json
