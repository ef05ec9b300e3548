#!/bin/bash
# on-device learning with the two-launch rollout step, segments collected beside the updates
set -o pipefail
mkdir -p gpurun_out/r3_rollout3
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -u bench/learn_planar.py --env Walker2d-v2 --updates 200000 --envs 256 --segment 32 --seeds 3 --overlap --out gpurun_out/r3_rollout3 > gpurun_out/r3_rollout3/walker_200k_overlap.log 2>&1 && grep -v curve gpurun_out/r3_rollout3/walker_200k_overlap.log &&
timeout -k 10 300 python -u bench/learn_planar.py --env BipedalWalker-v2 HalfCheetah-v2 Hopper-v2 --updates 200000 --envs 256 --segment 32 --overlap --out gpurun_out/r3_rollout3 > gpurun_out/r3_rollout3/others_200k_overlap.log 2>&1 && grep -v curve gpurun_out/r3_rollout3/others_200k_overlap.log &&
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3_rollout3/prof" -o walker -- python3 "$GRAFT_REPO_ROOT/bench/learn_planar.py" --env Walker2d-v2 --updates 20000 --envs 256 --segment 32 --overlap --out "$GRAFT_REPO_ROOT/gpurun_out/r3_rollout3/prof_run" > "$GRAFT_REPO_ROOT/gpurun_out/r3_rollout3/prof.log" 2>&1 &&
cd "$GRAFT_REPO_ROOT" && ls gpurun_out/r3_rollout3/prof && rm -f gpurun_out/r3_rollout3/prof/*kernel_trace.csv
