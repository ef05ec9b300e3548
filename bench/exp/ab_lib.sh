# usage: ab_lib.sh <outdir> <tag> ...: bf16 tests on the in-tree library, then update time (LunarLander batch 256,
# D3PG, Walker2d batch 2048, PER) per library variant ("default" = in-tree), two rounds
set -e
O=gpurun_out/$1; shift
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "bf16 or fused or replicated" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do for tag in "$@"; do
  if [ $tag = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$tag.so; fi
  run() { n=$1; shift; timeout -k 10 120 python bench.py --steps 3200 --warmup 320 "$@" > $O/${tag}_${n}_$rep.json 2> $O/${tag}_${n}_$rep.err; python -c "import json;d=json.load(open('$O/${tag}_${n}_$rep.json'));print('$tag','$n',round(d['ms_per_step']*1000,2))"; }
  run d4pg256; run d3pg256 --model d3pg; run walker2048 --env Walker2d-v2 --batch 2048; run per256 --per
done; done
