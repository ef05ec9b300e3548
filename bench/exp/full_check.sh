# whole GPU suite + the headline numbers (one box, so the numbers are comparable with each other)
set -e
O=gpurun_out/${1:-check}
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python -c "import json;d=json.load(open('$O/$name.json'));print('$name',d['value'],round(d['ms_per_step']*1000,2))"; }
run bf16
run bf16_per --per
run bf16_d3pg --model d3pg
run fp32 --dtype fp32
run walker_b512 --env Walker2d-v2 --batch 512
run walker_b1024 --env Walker2d-v2 --batch 1024
run walker_b2048 --env Walker2d-v2 --batch 2048
run walker_b4096 --env Walker2d-v2 --batch 4096
run ant4096_d4pg --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000
run ant4096_d3pg --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --model d3pg
run ant4096_d4pg_per --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --per
run dp_selftest --dp-selftest
