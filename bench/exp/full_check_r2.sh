# every headline / stress number on ONE box, so that they are comparable with each other
# (fp32 is the headline; every fp32 run on one GPU also times bf16 and reports it as bf16_value)
set -e
O=gpurun_out/${1:-final_r2}
mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python -c "import json;d=json.load(open('$O/$name.json'));print('$name',d['value'],round(d['ms_per_step']*1000,2),d.get('bf16_ms_per_step'))"; }
run fp32
run fp32_driver_style --steps 20 --warmup 5
run fp32_per --per
run fp32_d3pg --model d3pg
run fp32_layers --layers
run torch_baseline --baseline --steps 300 --warmup 30
run walker2d_b256 --env Walker2d-v2 --batch 256
run walker2d_b512 --env Walker2d-v2 --batch 512
run walker2d_b1024 --env Walker2d-v2 --batch 1024
run walker2d_b2048 --env Walker2d-v2 --batch 2048
run walker2d_b4096 --env Walker2d-v2 --batch 4096 --steps 1000 --warmup 100
run ant4096_10M_d4pg --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --steps 1000 --warmup 100
run ant4096_10M_d3pg --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --model d3pg --steps 1000 --warmup 100
run ant4096_10M_d4pg_per --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 --per --steps 1000 --warmup 100
