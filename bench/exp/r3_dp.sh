#!/bin/bash
# chi-square test of the ride-along draw, thread-rank data-parallel cases, optimiser kernel, exchange-mode self-test
set -o pipefail
mkdir -p gpurun_out/r3_dp
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_hip_f32.py tests/test_dp_hip.py tests/test_hip_kernels.py tests/test_replicated_dp.py -x -q --timeout 170 --timeout-method thread > gpurun_out/r3_dp/pytest.log 2>&1 || { tail -30 gpurun_out/r3_dp/pytest.log; exit 1; }
tail -3 gpurun_out/r3_dp/pytest.log
timeout -k 10 120 python bench.py --no-extras > gpurun_out/r3_dp/bench_plain.json 2> gpurun_out/r3_dp/bench_plain.err && cat gpurun_out/r3_dp/bench_plain.json &&
timeout -k 10 120 python bench.py --dp-selftest --no-extras > gpurun_out/r3_dp/bench_selftest.json 2> gpurun_out/r3_dp/bench_selftest.err && cat gpurun_out/r3_dp/bench_selftest.json &&
timeout -k 10 120 python bench.py --dp-selftest --no-extras --env Ant-v2 --batch 4096 > gpurun_out/r3_dp/bench_selftest_ant.json 2> gpurun_out/r3_dp/bench_selftest_ant.err && cat gpurun_out/r3_dp/bench_selftest_ant.json
