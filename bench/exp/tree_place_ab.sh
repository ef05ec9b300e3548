#!/bin/bash
# Where the priority write-back rides: critic's weight-gradient launch (1) or beside the actor phase (0), by pipeline and batch
set -e
O=gpurun_out/tree_place
mkdir -p $O
for v in 1 0; do
  export D4PG_TREE_IN_WGRAD=$v
  for cfg in "bf16_256 --dtype bf16 --per" "fp32_256 --no-bf16 --per" "bf16_w2048 --dtype bf16 --per --env Walker2d-v2 --batch 2048" "fp32_w2048 --no-bf16 --per --env Walker2d-v2 --batch 2048" "bf16_ant4096 --dtype bf16 --per --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --steps 1000 --warmup 100" "fp32_ant4096 --no-bf16 --per --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --steps 1000 --warmup 100"; do
    set -- $cfg; name=$1; shift
    timeout -k 10 200 python bench.py "$@" > $O/${name}_$v.json 2> $O/${name}_$v.err
    python -c "import json;d=json.loads(open('$O/${name}_$v.json').read().strip().splitlines()[-1]);print('tree_in_wgrad=$v','$name',d['value'],round(d['ms_per_step']*1000,2),'us')"
  done
done
