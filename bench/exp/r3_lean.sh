#!/bin/bash
# Lean build of the 16x16x4 phase kernels (two workgroups per CU): tests, then batch 4096 with one and two workgroups per CU
set -o pipefail
O=gpurun_out/r3_lean
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests/test_hip_f32.py -x -q --timeout 300 --timeout-method thread -k "occupancy or 4096 or drawn_ahead" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
b walker4096_occ1 --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 --occupancy 1 || exit 1
b walker4096_occ2 --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 --occupancy 2 || exit 1
b ant4096_occ1 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --occupancy 1 || exit 1
b ant4096_occ2 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --occupancy 2 || exit 1
b walker2048_occ2 --env Walker2d-v2 --batch 2048 --steps 1000 --warmup 100 --mfma 16x16 --occupancy 2 || exit 1
b walker8192_occ1 --env Walker2d-v2 --batch 8192 --steps 400 --warmup 50 --occupancy 1 || exit 1
b walker8192_occ2 --env Walker2d-v2 --batch 8192 --steps 400 --warmup 50 --occupancy 2 || exit 1
