#!/bin/bash
# The reference's README figure on the native tasks: exploiter episode reward against the learner's update step,
# D3PG vs D4PG on Pendulum-v0, LunarLanderContinuous-v2 and BipedalWalker-v2, the shipped configurations
# (100 000 updates, batch 256, 4 agents = 1 exploiter + 3 explorers), 8 environments per explorer and one update per
# 8 environment steps.  Leaves the exploiters' scalars under gpurun_out/figure/; plot them with
#   python -m d4pg_pytorch_amd.utils.reward_plot --run ENV:LABEL:gpurun_out/figure/<name>.json ... --out plot.png
O=gpurun_out/figure
mkdir -p $O
one() {
  name=$1; cfg=$2; shift; shift
  timeout -k 10 600 python train.py --config $cfg --set device=cuda results_path=$O/$name envs_per_agent=8 updates_per_env_step=0.125 num_episode_save=100000 save_reward_threshold=1000000 max_env_steps_per_agent=1000000000 "$@" > $O/$name.log 2>&1 || { tail -20 $O/$name.log; return 1; }
  run=$(ls -d $O/$name/* | tail -1)
  python - <<PY
import json, sys
sys.path.insert(0, ".")
from d4pg_pytorch_amd.utils.reward_plot import load_xy
x, y = load_xy("$run/agent-0", "agent/reward")
json.dump([[0.0, float(a), float(b)] for a, b in zip(x, y)], open("$O/$name.json", "w"))      # [wall, step, value] rows, the
print("$name", len(y), "episodes; first 5", [round(v, 1) for v in y[:5]], "last 10 mean", round(sum(y[-10:]) / max(1, len(y[-10:])), 1))      # layout of a TensorBoard export
PY
  rm -rf $O/$name
}
one pendulum_d4pg configs/pendulum_d4pg.yml & one pendulum_d3pg configs/pendulum_d3pg.yml & wait
one lander_d4pg configs/lunar_lander_continuous_d4pg.yml & one lander_d3pg configs/lunar_lander_continuous_d3pg.yml & wait
one bipedal_d4pg configs/bipedal_d4pg.yml max_ep_length=1600 & one bipedal_d3pg configs/bipedal_d3pg.yml max_ep_length=1600 & wait
