#!/bin/bash
# Device-resident Ant, Reacher and inverted double pendulum: rollout throughput and on-device learning runs.
O=gpurun_out/ant_device
mkdir -p $O
for env in Ant-v2 Reacher-v2 InvertedDoublePendulum-v2; do
  timeout -k 10 200 python bench/topology.py --gpu-rollout --rollout-env $env --envs 4096 --segments 100 > $O/rollout_$env.json 2> $O/rollout_$env.err || { tail -5 $O/rollout_$env.err; exit 1; }
  tail -1 $O/rollout_$env.json | cut -c1-420
done
timeout -k 10 300 python bench/learn_planar.py --env Reacher-v2 InvertedDoublePendulum-v2 --updates 100000 --envs 64 --eval-every 5000 --out $O > $O/learn_small.log 2>&1 || { tail -20 $O/learn_small.log; exit 1; }
grep -v curve $O/learn_small.log
timeout -k 10 600 python bench/learn_planar.py --env Ant-v2 --updates ${1:-400000} --envs 256 --eval-every 20000 --out $O > $O/learn_ant.log 2>&1 || { tail -20 $O/learn_ant.log; exit 1; }
cat $O/learn_ant.log | cut -c1-900
