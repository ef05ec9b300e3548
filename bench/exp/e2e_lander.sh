#!/bin/bash
# The whole process topology (train.py: learner + HBM replay on the GPU, CPU explorer processes on the native lunar
# lander, GPU exploiter playing the target policy) on the reference's LunarLander D4PG configuration, with the update rate
# tied to the environment steps so that the run is comparable with the reference's learning curve (updates on the x axis).
set -e
O=gpurun_out/e2e_lander
rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python train.py --config configs/mi355x/lunar_lander_d4pg_1gpu.yml --set results_path=$O/$name num_steps_train=40000 envs_per_agent=16 updates_per_env_step=0.25 num_episode_save=100000 "$@" > $O/$name.log 2>&1; python - <<PY
import glob, json, os, sys
sys.path.insert(0, ".")
from d4pg_pytorch_amd.utils.reward_plot import load_xy
run = sorted(glob.glob("$O/$name/*"))[-1]
out = {"run": os.path.basename(run)}
for agent in sorted(glob.glob(run + "/agent-*")):
    x, y = load_xy(agent, "agent/reward")
    if len(y):
        n = len(y)
        q = [round(float(sum(y[i * n // 4:(i + 1) * n // 4]) / max(1, len(y[i * n // 4:(i + 1) * n // 4]))), 1) for i in range(4)]
        out[os.path.basename(agent)] = {"episodes": n, "mean_return_by_quarter_of_the_run": q, "last20_mean": round(float(sum(y[-20:]) / len(y[-20:])), 1), "last_update": int(x[-1])}
for tag in ("perf/updates_per_s", "perf/env_steps_per_s"):
    try:
        x, y = load_xy(run + "/learner", tag)
        out[tag] = round(float(sum(y[len(y)//2:]) / max(1, len(y[len(y)//2:]))), 1)
    except Exception as e:
        out[tag] = repr(e)
json.dump(out, open("$O/$name.json", "w"), indent=1)
print("$name", json.dumps(out))
PY
}
run d4pg_reference_support
run d4pg_support_pm5 v_min=-5 v_max=5
run d3pg model=d3pg
