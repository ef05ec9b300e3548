#!/bin/bash
# on-device learning with the two-launch rollout step, segments beside the updates: three seeds of every articulated-body task
O=gpurun_out/r3_learn_seeds
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -u bench/learn_planar.py --env BipedalWalker-v2 HalfCheetah-v2 Hopper-v2 --updates 200000 --seeds 3 --overlap --out $O > $O/planar.log 2>&1 || { tail -5 $O/planar.log; exit 1; }
grep -v curve $O/planar.log
timeout -k 10 500 python -u bench/learn_planar.py --env Ant-v2 --updates 400000 --seeds 3 --overlap --out $O > $O/ant.log 2>&1 || { tail -5 $O/ant.log; exit 1; }
grep -v curve $O/ant.log
