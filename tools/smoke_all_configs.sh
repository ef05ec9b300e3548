#!/bin/bash
# Every shipped reference configuration through train.py on the CPU topology (sampler + learner + agents), shrunk to
# seconds: 40 updates, 32-wide nets, batch 32.  Prints one line per configuration; exit status 1 if any failed.
#   tools/smoke_all_configs.sh [out_dir]
cd "$(dirname "$0")/.."
out=${1:-/tmp/d4pg_smoke_all}
rm -rf "$out"; mkdir -p "$out"
fail=0
for c in configs/*.yml; do
  n=$(basename "$c" .yml)
  timeout 120 python train.py --config "$c" --set device=cpu agent_device=cpu num_steps_train=40 dense_size=32 batch_size=32 \
      replay_mem_size=5000 max_ep_length=60 results_path="$out/$n" > "$out/$n.log" 2>&1
  rc=$?
  [ $rc -ne 0 ] && fail=1
  echo "$n rc=$rc"
done
exit $fail
