#!/bin/bash
# Weight-gradient kernel: 16 waves (default from 128 rows) against 8, by batch size
set -o pipefail
O=gpurun_out/r3_wgrad_waves
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { echo "$name failed"; return 0; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for B in 512 1024 2048 4096; do
unset D4PG_WGRAD_WAVES
b walker${B}_w16 --env Walker2d-v2 --batch $B --steps 1000 --warmup 100
export D4PG_WGRAD_WAVES=8
b walker${B}_w8 --env Walker2d-v2 --batch $B --steps 1000 --warmup 100
done
