#!/bin/bash
# what `auto` picks between 2048 and 4096 rows after the thresholds moved, and the fp32 tests
set -o pipefail
O=gpurun_out/r3_check2
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_hip_f32.py tests/test_replicated_dp.py tests/test_data_parallel.py tests/test_dp_hip.py -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us', d['implementation'])"; }
for B in 2048 2064 3072 3584 4096; do
b walker${B}_auto --env Walker2d-v2 --batch $B --steps 800 --warmup 100 || exit 1
done
b walker3072_per --env Walker2d-v2 --batch 3072 --steps 800 --warmup 100 --per || exit 1
b walker3584_per --env Walker2d-v2 --batch 3584 --steps 800 --warmup 100 --per || exit 1
b walker2056_auto --env Walker2d-v2 --batch 2056 --steps 800 --warmup 100 || exit 1
