# large batches in fp32: fused row-local pipeline vs one launch per layer (weight-stationary grouped GEMMs)
run() { timeout -k 10 200 python bench.py --steps 320 --warmup 64 --no-bf16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], round(d['ms_per_step']*1000,1))" "$@"; }
for b in 1024 2048 4096 8192; do run --env Walker2d-v2 --batch $b; run --env Walker2d-v2 --batch $b --layers; done
run --env Ant-v2 --batch 4096; run --env Ant-v2 --batch 4096 --layers
