#!/bin/bash
# Everything measured on the lunar lander, again (after the leg-travel rule): parity + learning tests, in-process
# learning runs, train.py runs, the two lander panels of the reference's figure, rollout throughput.
O=gpurun_out/lander_redo
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_lunar_lander.py "tests/test_learning_gpu.py::test_d4pg_learns_to_land_the_lunar_lander" -m gpu -x -q --timeout 200 --timeout-method thread > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 400 python bench/learn_lander.py --updates 60000 --seeds 2 --out $O > $O/learn.log 2>&1 || { tail -20 $O/learn.log; exit 1; }
grep -v curve $O/learn.log
bash bench/exp/e2e_lander.sh > $O/e2e.log 2>&1 || { tail -20 $O/e2e.log; exit 1; }
cut -c1-500 $O/e2e.log
timeout -k 10 200 python bench/topology.py --gpu-rollout --rollout-env LunarLanderContinuous-v2 --envs 4096 --segments 100 > $O/rollout.json 2> $O/rollout.err || exit 1
tail -1 $O/rollout.json | cut -c1-400
