# usage: ab_variants.sh <outdir> <tag> [<tag> ...]   ("default" = the in-tree library)
set -e
O=gpurun_out/$1; shift
mkdir -p $O
for rep in 1 2; do for tag in "$@"; do for B in 256 2048; do
  if [ $tag = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$tag.so; fi
  timeout -k 10 120 python bench.py --env Walker2d-v2 --batch $B --steps 3000 --warmup 300 > $O/${tag}_b${B}_$rep.json 2> $O/${tag}_b${B}_$rep.err
  python -c "import json;d=json.load(open('$O/${tag}_b${B}_$rep.json'));print('$tag',$B,round(d['ms_per_step']*1000,2))"
done; done; done
