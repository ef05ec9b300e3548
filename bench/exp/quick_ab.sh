# bf16 pipeline tests + update time at three batch sizes (A/B runs of kernel variants go through this)
set -e
O=gpurun_out/${1:-ab}
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "bf16 or wgrad or fused or replicated" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for B in 256 1024 2048; do
  timeout -k 10 120 python bench.py --env Walker2d-v2 --batch $B --steps 3000 --warmup 300 > $O/b$B.json 2> $O/b$B.err
  python -c "import json;d=json.load(open('$O/b$B.json'));print($B,round(d['ms_per_step']*1000,2))"
done
timeout -k 10 120 python bench.py --per --steps 3000 --warmup 300 > $O/per.json 2> $O/per.err
python -c "import json;d=json.load(open('$O/per.json'));print('per',round(d['ms_per_step']*1000,2))"
