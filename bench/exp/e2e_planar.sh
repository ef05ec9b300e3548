#!/bin/bash
# Do the planar articulated-body tasks (envs/planar.py) learn?  train.py: GPU learner + HBM replay, CPU explorer
# processes on the host simulator, GPU exploiter; the update rate is tied to the environment steps.
#   bash bench/exp/e2e_planar.sh <name> <config> <updates> [key=value ...]
O=gpurun_out/e2e_planar
mkdir -p $O
name=$1; cfg=$2; upd=$3; shift; shift; shift
timeout -k 10 ${E2E_TIMEOUT:-900} python train.py --config $cfg --set device=cuda results_path=$O/$name num_steps_train=$upd num_agents=13 envs_per_agent=8 updates_per_env_step=0.125 num_episode_save=100000 save_reward_threshold=1000000 max_env_steps_per_agent=1000000000 "$@" > $O/$name.log 2>&1
python - <<PY
import glob, json, os, sys
sys.path.insert(0, ".")
from d4pg_pytorch_amd.utils.reward_plot import load_xy
run = sorted(glob.glob("$O/$name/*"))[-1]
out = {"run": os.path.basename(run)}
x, y = load_xy(run + "/agent-0", "agent/reward")
n = len(y)
out["exploiter"] = {"episodes": n, "mean_return_by_eighth_of_the_run": [round(float(sum(y[i*n//8:(i+1)*n//8]) / max(1, len(y[i*n//8:(i+1)*n//8]))), 1) for i in range(8)],
                    "last20_mean": round(float(sum(y[-20:]) / max(1, len(y[-20:]))), 1), "best": round(float(max(y)), 1) if n else None, "last_update": int(x[-1]) if n else None}
for tag in ("perf/updates_per_s", "perf/env_steps_per_s"):
    try:
        xx, yy = load_xy(run + "/learner", tag)
        out[tag] = round(float(sum(yy[len(yy)//2:]) / max(1, len(yy[len(yy)//2:]))), 1)
    except Exception as e:
        out[tag] = repr(e)
json.dump(out, open("$O/$name.json", "w"), indent=1)
print("$name", json.dumps(out))
PY
rm -rf $O/$name
