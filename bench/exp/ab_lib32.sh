# update time of the fp32 pipeline per library variant (two rounds, one box): bash bench/exp/ab_lib32.sh default pf1 ...
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$v.so; fi
  for shape in "--batch 256" "--env Walker2d-v2 --batch 2048"; do
    r=$(timeout -k 10 120 python bench.py --dtype fp32 --no-bf16 --steps 2000 --warmup 200 $shape 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v [$shape] $r"
  done
done
done
