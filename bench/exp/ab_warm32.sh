set -e
for w in 0 32 64 128; do
  echo "warmers=$w"
  D4PG_L2_WARMERS=$w timeout -k 10 120 python bench.py --dtype fp32 --no-bf16 --steps 2000 --warmup 200 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
