# the headline benchmark with step counts that do not line up with the 32-update chunks
set -e
O=gpurun_out/odd; mkdir -p $O
for sw in "50 10" "20 5" "100 20" "1000 100" "3000 300"; do set -- $sw
  timeout -k 10 120 python bench.py --steps $1 --warmup $2 > $O/s$1.json 2> $O/s$1.err
  python -c "import json;d=json.load(open('$O/s$1.json'));print('steps',$1,'warmup',$2,round(d['ms_per_step']*1000,2),'us')"
done
