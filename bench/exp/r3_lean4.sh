#!/bin/bash
# Inputs of a phase in one round trip (issue / commit apart, tile copies with four loads in flight) and the kernel bodies
# behind two __global__ wrappers: the fp32 tests, then the headline and the large batches.  (Both were measured with this
# script and not kept, bench/exp/README.md; what is left of them is copy16x4 in the lean build's mid-kernel tile copies.)
set -o pipefail
O=gpurun_out/r3_lean5
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_hip_f32.py tests/test_replicated_dp.py -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
b lunar256 || exit 1
b lunar256_per --per || exit 1
b walker1024 --env Walker2d-v2 --batch 1024 --steps 2000 --warmup 200 || exit 1
b walker2048 --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 || exit 1
b walker2048_16x16 --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 --mfma 16x16 || exit 1
b walker4096 --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
b ant4096 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
b ant4096_occ1 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --occupancy 1 || exit 1
b ant4096_per --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --per || exit 1
b lunar256_h300_16x16 --hidden 300 --mfma 16x16 || exit 1
