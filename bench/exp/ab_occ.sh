# large-batch A/B of library variants (two workgroups per CU need <= 128 registers and <= 80 KB of LDS each)
set -e
O=gpurun_out/$1; shift
mkdir -p $O
for rep in 1 2; do for tag in "$@"; do
  if [ $tag = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$tag.so; fi
  run() { n=$1; shift; timeout -k 10 200 python bench.py --steps 1000 --warmup 128 "$@" > $O/${tag}_${n}_$rep.json 2> $O/${tag}_${n}_$rep.err; python -c "import json;d=json.load(open('$O/${tag}_${n}_$rep.json'));print('$tag','$n',round(d['ms_per_step']*1000,2))"; }
  run walker4096 --env Walker2d-v2 --batch 4096; run ant4096 --env Ant-v2 --batch 4096 --prefill 400000; run walker8192_t16 --env Walker2d-v2 --batch 8192 --tile-rows 16; run d4pg256
done; done
