#!/bin/bash
# Lean build: weight ring of 3 (default) against 4 k-groups (bench/exp/variants/libd4pg_hip_ring4.so: part 0 of fused32.hip
# with -DD4PG_RING16_LEAN=4) on one box
set -o pipefail
O=gpurun_out/r3_lean_ring
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for v in ring3 ring4 ring3 ring4; do
  if [ $v = ring3 ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$v.so; fi
  b ant4096_$v --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
  b walker4096_$v --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
done
export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_ring4.so
timeout -k 10 300 python -u -m pytest tests/test_hip_f32.py -x -q --timeout 300 --timeout-method thread -k "gradients" > $O/pytest_ring4.log 2>&1 || { tail -20 $O/pytest_ring4.log; exit 1; }
tail -1 $O/pytest_ring4.log
