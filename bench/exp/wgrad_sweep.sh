# bf16 pipeline tests, then update time by global batch and wgrad waves per workgroup
set -e
mkdir -p gpurun_out/ws
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "bf16 or wgrad or fused or replicated" > gpurun_out/ws/tests.log 2>&1 || { tail -30 gpurun_out/ws/tests.log; exit 1; }
tail -2 gpurun_out/ws/tests.log
for B in 256 1024 2048; do for W in 4 8 16; do
  D4PG_WGRAD_WAVES=$W timeout -k 10 120 python bench.py --env Walker2d-v2 --batch $B --steps 2000 --warmup 200 > gpurun_out/ws/b${B}_w$W.json 2> gpurun_out/ws/b${B}_w$W.err
  python -c "import json;d=json.load(open('gpurun_out/ws/b${B}_w$W.json'));print($B,$W,round(d['ms_per_step']*1000,2))"
done; done
