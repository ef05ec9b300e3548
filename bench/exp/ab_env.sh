# usage: ab_env.sh <outdir> <VAR> <value> [<value> ...]: update time at batch 256 / 2048 (+ PER at 256) per value of VAR
set -e
O=gpurun_out/$1; VAR=$2; shift; shift
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "bf16 or fused or replicated" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do for v in "$@"; do
  export $VAR=$v
  for B in 256 2048; do
    timeout -k 10 120 python bench.py --env Walker2d-v2 --batch $B --steps 3000 --warmup 300 > $O/v${v}_b${B}_$rep.json 2> $O/v${v}_b${B}_$rep.err
    python -c "import json;d=json.load(open('$O/v${v}_b${B}_$rep.json'));print('$VAR=$v',$B,round(d['ms_per_step']*1000,2))"
  done
  timeout -k 10 120 python bench.py --per --steps 3000 --warmup 300 > $O/v${v}_per_$rep.json 2> $O/v${v}_per_$rep.err
  python -c "import json;d=json.load(open('$O/v${v}_per_$rep.json'));print('$VAR=$v','per256',round(d['ms_per_step']*1000,2))"
done; done
