#!/bin/bash
# fp32 learner tests and the update time of the main shapes (a check after a kernel edit)
set -o pipefail
O=gpurun_out/r3_check
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_hip_f32.py tests/test_replicated_dp.py tests/test_gpu_rollout.py -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
b lunar256 || exit 1
b walker2048 --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 || exit 1
b walker4096 --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
b ant4096 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
b lunar256_h32 --hidden 32 || exit 1
