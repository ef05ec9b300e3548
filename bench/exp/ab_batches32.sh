# fp32 update time at the Walker2d shape by batch (default tile policy)
run() { timeout -k 10 200 python bench.py --steps 640 --warmup 64 --no-bf16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], round(d['ms_per_step']*1000,1))" "$@"; }
for b in 256 512 1024 2048 4096; do run --env Walker2d-v2 --batch $b; done
