#!/bin/bash
# Every shipped reference configuration through train.py on the GPU topology AS SHIPPED (fused fp32 learner, HBM replay,
# CPU explorer processes on the native simulators, GPU exploiter), only shortened: 3000 updates, 4 environments per
# explorer, one update per 8 environment steps.  One JSON line per configuration in <out>/summary.jsonl; exit 1 if any failed.
#   tools/smoke_all_configs_gpu.sh [out_dir]
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/smoke_gpu}
rm -rf "$out"; mkdir -p "$out"
fail=0
for c in configs/*.yml; do
  n=$(basename "$c" .yml)
  t0=$(date +%s.%N)
  timeout -k 10 150 python train.py --config "$c" --set device=cuda num_steps_train=3000 envs_per_agent=4 updates_per_env_step=0.125 \
      num_episode_save=100000 results_path="$out/$n" > "$out/$n.log" 2>&1
  rc=$?
  [ $rc -ne 0 ] && fail=1
  python - "$n" "$rc" "$out" "$t0" <<'PY' >> "$out/summary.jsonl"
import glob, json, sys, time
sys.path.insert(0, ".")
from d4pg_pytorch_amd.utils.reward_plot import load_xy
n, rc, out, t0 = sys.argv[1], int(sys.argv[2]), sys.argv[3], float(sys.argv[4])
row = {"config": n, "rc": rc, "wall_s": round(time.time() - t0, 1)}
runs = sorted(glob.glob(f"{out}/{n}/*"))
if runs:
    for tag in ("perf/updates_per_s", "perf/env_steps_per_s", "learner/value_loss"):
        try:
            x, y = load_xy(runs[-1] + "/learner", tag)
            row[tag] = round(float(y[-1]), 4) if len(y) else None
        except Exception as e:
            row[tag] = repr(e)
    x, y = load_xy(runs[-1] + "/agent-0", "agent/reward")
    row["exploiter_episodes"] = len(y)
print(json.dumps(row))
PY
  rm -rf "$out/$n"
  tail -1 "$out/summary.jsonl"
  [ $rc -eq 124 ] && break       # a timeout: start nothing more on the GPU
done
exit $fail
