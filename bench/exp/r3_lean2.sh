#!/bin/bash
# Lean build: the whole fp32 test file, then kernel times at batch 4096 with two workgroups per CU
set -o pipefail
O=gpurun_out/r3_lean
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_hip_f32.py -x -q --timeout 300 --timeout-method thread > $O/pytest_f32.log 2>&1 || { tail -40 $O/pytest_f32.log; exit 1; }
tail -3 $O/pytest_f32.log
ks() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --no-bf16 --no-extras --steps 300 --warmup 50 "$@" > $O/${name}_bench.json 2> $O/${name}.err || return 1; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; head -5 $O/kernel_stats_$name.csv | cut -c1-140; }
ks ant4096_occ2 --env Ant-v2 --batch 4096 --occupancy 2 || exit 1
ks walker4096_occ2 --env Walker2d-v2 --batch 4096 --occupancy 2 || exit 1
