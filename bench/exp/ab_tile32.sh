# fp32 pipeline: update time by rows per phase workgroup (4 / 8 / 16) at large batches
run() { timeout -k 10 200 python bench.py --steps 320 --warmup 64 --no-bf16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], round(d['ms_per_step']*1000,1))" "$@"; }
for t in 4 8 16; do run --env Ant-v2 --batch 4096 --tile-rows $t; done
for t in 4 8 16; do run --env Walker2d-v2 --batch 2048 --tile-rows $t; done
for t in 4 8; do run --env Walker2d-v2 --batch 1024 --tile-rows $t; done
for t in 4 8; do run --env Walker2d-v2 --batch 512 --tile-rows $t; done
