#!/bin/bash
# train.py with device-resident Walker2d explorers (GPU exploiter on the host simulator), and a kernel trace of the
# device rollout of Walker2d beside the learner.
O=gpurun_out/planar_device
mkdir -p $O
timeout -k 10 400 python train.py --config configs/mi355x/walker2d_d4pg_device_explorers.yml --set results_path=$O/train num_steps_train=${1:-150000} num_episode_save=100000 save_reward_threshold=1000000 > $O/train_walker2d.log 2>&1 || { tail -30 $O/train_walker2d.log; exit 1; }
python - <<PY
import glob, json, sys
sys.path.insert(0, ".")
from d4pg_pytorch_amd.utils.reward_plot import load_xy
run = sorted(glob.glob("$O/train/*"))[-1]
x, y = load_xy(run + "/agent-0", "agent/reward")
n = len(y)
out = {"episodes": n, "exploiter_return_by_quarter": [round(float(sum(y[i*n//4:(i+1)*n//4]) / max(1, len(y[i*n//4:(i+1)*n//4]))), 1) for i in range(4)], "last10": round(float(sum(y[-10:]) / max(1, len(y[-10:]))), 1)}
for tag in ("perf/updates_per_s", "perf/env_steps_per_s", "agent/reward_device_explorers"):
    try:
        xx, yy = load_xy(run + "/learner", tag)
        out[tag] = round(float(sum(yy[len(yy)//2:]) / max(1, len(yy[len(yy)//2:]))), 1)
    except Exception as e:
        out[tag] = repr(e)
json.dump(out, open("$O/train_walker2d.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/train
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench/topology.py --gpu-rollout --rollout-env Walker2d-v2 --envs 4096 --segments 50 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err || exit 1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py stats $O/prof $O/kernel_stats_walker2d_rollout.csv && rm -rf $O/prof && head -12 $O/kernel_stats_walker2d_rollout.csv | cut -c1-200
