#!/bin/bash
# two-launch rollout step: tests, then the on-device learning runs it is meant to speed up
set -o pipefail
mkdir -p gpurun_out/r3_rollout2
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_gpu_rollout.py tests/test_spatial_env.py tests/test_planar_envs.py tests/test_analytic_envs.py tests/test_rollout_worker.py -x -q --timeout 170 --timeout-method thread > gpurun_out/r3_rollout2/pytest.log 2>&1 || { tail -40 gpurun_out/r3_rollout2/pytest.log; exit 1; }
tail -3 gpurun_out/r3_rollout2/pytest.log
# on-device learning: Walker2d 200 000 updates, 256 environments x 32-step segments (29.4 s with four launches per step)
timeout -k 10 400 python -u bench/learn_planar.py --env Walker2d-v2 --updates 200000 --envs 256 --segment 32 --out gpurun_out/r3_rollout2 > gpurun_out/r3_rollout2/walker_200k.log 2>&1 && tail -3 gpurun_out/r3_rollout2/walker_200k.log
