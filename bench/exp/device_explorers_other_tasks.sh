#!/bin/bash
# The device-explorer configuration with other tasks substituted (--set env=...), as its header says one can.
O=gpurun_out/device_explorers
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python train.py --config configs/mi355x/walker2d_d4pg_device_explorers.yml --set results_path=$O/$name num_episode_save=100000 save_reward_threshold=1000000 "$@" > $O/$name.log 2>&1 || { tail -20 $O/$name.log; return 1; }
python - <<PY
import glob, json, sys
sys.path.insert(0, ".")
from d4pg_pytorch_amd.utils.reward_plot import load_xy
run = sorted(glob.glob("$O/$name/*"))[-1]
x, y = load_xy(run + "/agent-0", "agent/reward")
n = len(y)
out = {"run": "$name", "episodes": n, "exploiter_return_by_quarter": [round(float(sum(y[i*n//4:(i+1)*n//4]) / max(1, len(y[i*n//4:(i+1)*n//4]))), 1) for i in range(4)], "last10": round(float(sum(y[-10:]) / max(1, len(y[-10:]))), 1)}
for tag in ("perf/updates_per_s", "perf/env_steps_per_s", "agent/reward_device_explorers"):
    try:
        xx, yy = load_xy(run + "/learner", tag)
        out[tag] = round(float(sum(yy[len(yy)//2:]) / max(1, len(yy[len(yy)//2:]))), 1)
    except Exception as e:
        out[tag] = repr(e)
json.dump(out, open("$O/$name.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/$name; }
run bipedal env=BipedalWalker-v2 state_dim=24 action_dim=4 v_min=-100 v_max=300 max_ep_length=1600 num_steps_train=150000 && \
run ant env=Ant-v2 state_dim=111 action_dim=8 v_min=0 v_max=300 num_steps_train=300000 && \
run half_cheetah env=HalfCheetah-v2 state_dim=17 action_dim=6 v_min=-50 v_max=600 num_steps_train=150000
