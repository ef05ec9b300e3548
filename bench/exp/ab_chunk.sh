set -e
O=gpurun_out/chunk; mkdir -p $O
for rep in 1 2; do for K in 8 16 32 64; do
  timeout -k 10 120 python bench.py --chunk $K --steps 3072 --warmup 320 > $O/k$K.json 2> $O/k$K.err
  python -c "import json;d=json.load(open('$O/k$K.json'));print('chunk',$K,round(d['ms_per_step']*1000,2))"
done; done
timeout -k 10 120 python bench.py --chunk 32 --dp-selftest --steps 3072 --warmup 320 > $O/self32.json 2> $O/self32.err
python -c "import json;d=json.loads(open('$O/self32.json').read().strip().splitlines()[-1]);print('selftest chunk 32',round(d['ms_per_step']*1000,2))"
