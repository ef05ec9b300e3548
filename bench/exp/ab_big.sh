# large-batch A/B: library variants x L2 warmers at batch 4096 / 8192 (Walker2d shape)
set -e
O=gpurun_out/$1; shift
mkdir -p $O
for tag in "$@"; do for W in 0 32; do for B in 4096 8192; do for T in 16 32; do
  if [ $tag = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$tag.so; fi
  D4PG_L2_WARMERS=$W timeout -k 10 120 python bench.py --env Walker2d-v2 --batch $B --tile-rows $T --steps 1500 --warmup 200 > $O/${tag}_w${W}_b${B}_t$T.json 2> $O/${tag}_w${W}_b${B}_t$T.err
  python -c "import json;d=json.load(open('$O/${tag}_w${W}_b${B}_t$T.json'));print('$tag warmers',$W,'batch',$B,'tile',$T,round(d['ms_per_step']*1000,2))"
done; done; done; done
