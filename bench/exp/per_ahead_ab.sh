#!/bin/bash
# A/B of the prioritised-replay schedule (fp32 pipeline): next batch drawn and critic forward made ahead inside
# the actor-phase launch (default) vs the sampling role in the update's last launch (D4PG_PER_AHEAD=0).
set -e
out=gpurun_out/per_ahead
mkdir -p $out
timeout -k 10 120 python bench.py --per --no-bf16 > $out/per_new.json 2> $out/per_new.err
D4PG_PER_AHEAD=0 timeout -k 10 120 python bench.py --per --no-bf16 > $out/per_old.json 2> $out/per_old.err
timeout -k 10 120 python bench.py --no-bf16 > $out/uniform.json 2> $out/uniform.err
timeout -k 10 200 python bench.py --per --no-bf16 --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --steps 1000 --warmup 100 > $out/ant_per_new.json 2> $out/ant_per_new.err
D4PG_PER_AHEAD=0 timeout -k 10 200 python bench.py --per --no-bf16 --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --steps 1000 --warmup 100 > $out/ant_per_old.json 2> $out/ant_per_old.err
for f in per_new per_old uniform ant_per_new ant_per_old; do python - <<PY
import json
d=json.loads(open("$out/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"]*1000, "us", d["losses_finite"])
PY
done
