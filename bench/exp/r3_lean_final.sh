#!/bin/bash
# Records of the lean build (two workgroups per CU) at 4096 and 8192 rows: kernel times, update times beside the
# one-workgroup build on the same box
set -o pipefail
O=gpurun_out/r3_lean_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ks() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --no-bf16 --no-extras --steps 300 --warmup 50 "$@" > $O/${name}_bench.json 2> $O/${name}.err || return 1; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; head -5 $O/kernel_stats_$name.csv | cut -c1-140; }
ks ant4096_lean --env Ant-v2 --batch 4096 || exit 1
ks walker4096_lean --env Walker2d-v2 --batch 4096 || exit 1
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for occ in 1 2; do
b walker4096_occ$occ --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 --occupancy $occ || exit 1
b ant4096_occ$occ --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --occupancy $occ || exit 1
b ant4096_per_occ$occ --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --per --occupancy $occ || exit 1
b ant4096_d3pg_occ$occ --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --model d3pg --occupancy $occ || exit 1
b walker8192_occ$occ --env Walker2d-v2 --batch 8192 --steps 400 --warmup 50 --occupancy $occ || exit 1
b ant8192_occ$occ --env Ant-v2 --batch 8192 --steps 400 --warmup 50 --occupancy $occ || exit 1
done
b ant4096_dp_selftest --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --dp-selftest || exit 1
