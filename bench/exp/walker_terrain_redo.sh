#!/bin/bash
# Everything measured on BipedalWalker, again (after the ground became the gym task's grass instead of a flat track).
O=gpurun_out/walker_redo
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_planar_envs.py "tests/test_learning_gpu.py::test_d4pg_learns_the_device_bipedal_walker_and_half_cheetah" -m gpu -x -q --timeout 200 --timeout-method thread > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench/topology.py --gpu-rollout --rollout-env BipedalWalker-v2 --envs 4096 --segments 100 2>/dev/null | tail -1 > $O/rollout.json; cut -c150-420 $O/rollout.json
timeout -k 10 300 python bench/learn_planar.py --env BipedalWalker-v2 --updates 200000 --envs 256 --eval-every 20000 --out $O > $O/learn.log 2>&1 || { tail $O/learn.log; exit 1; }
tail -2 $O/learn.log | cut -c1-500
bash bench/exp/topology_real_env.sh 200000 | cut -c1-420
bash bench/exp/e2e_planar.sh bipedal_d4pg configs/bipedal_d4pg.yml 400000 max_ep_length=1600 | cut -c1-500
