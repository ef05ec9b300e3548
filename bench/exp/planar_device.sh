#!/bin/bash
# Device-resident planar tasks: rollout throughput beside the learner and on-device learning runs.
#   bash bench/exp/planar_device.sh [updates]
O=gpurun_out/planar_device
mkdir -p $O
U=${1:-200000}
for env in Hopper-v2 Walker2d-v2 HalfCheetah-v2 BipedalWalker-v2; do
  timeout -k 10 200 python bench/topology.py --gpu-rollout --rollout-env $env --envs 4096 --segments 100 > $O/rollout_$env.json 2> $O/rollout_$env.err || exit 1
  tail -1 $O/rollout_$env.json | cut -c1-400
done
timeout -k 10 900 python bench/learn_planar.py --updates $U --envs 256 --out $O > $O/learn.log 2>&1 || { tail -20 $O/learn.log; exit 1; }
grep -v curve $O/learn.log
