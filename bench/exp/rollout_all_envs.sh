#!/bin/bash
# Device-rollout throughput of every environment that exists as a kernel (4096 environments, policy in the loop).
O=gpurun_out/rollout_all_r3
mkdir -p $O
for env in Pendulum-v0 InvertedPendulum-v2 LunarLanderContinuous-v2 Reacher-v2 InvertedDoublePendulum-v2 Hopper-v2 BipedalWalker-v2 Walker2d-v2 HalfCheetah-v2 Ant-v2; do
  timeout -k 10 200 python bench/topology.py --gpu-rollout --rollout-env $env --envs 4096 --segments 100 2> $O/$env.err | tail -1 > $O/$env.json || { tail -5 $O/$env.err; exit 1; }
  python - <<PY
import json
r = json.load(open("$O/$env.json"))
print("$env", r["env_steps_per_s_rollout_alone"] / 1e6, r["us_per_env_step_all_envs"], r["env_steps_per_s_beside_learner"] / 1e6, r["updates_per_s_beside_rollout"], r["updates_per_s_learner_alone"])
PY
done
