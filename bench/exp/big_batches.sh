set -e
O=gpurun_out/bigb; mkdir -p $O
for B in 8192 16384 32768; do
  timeout -k 10 200 python bench.py --env Ant-v2 --batch $B --steps 300 --warmup 40 --prefill 400000 > $O/b$B.json 2> $O/b$B.err
  python -c "import json;d=json.load(open('$O/b$B.json'));print('Ant batch',$B,round(d['ms_per_step']*1000,1),'us')"
done
