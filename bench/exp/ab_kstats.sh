# usage: ab_kstats.sh <outdir> <tag> ...: per-kernel mean durations (rocprofv3) per library variant
set -e
O=gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  if [ $tag = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$tag.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/s_$tag -o run --output-format csv -- python3 bench.py --steps 800 --warmup 100 > $O/$tag.json 2> $O/$tag.err
  python tools/rocprof_summary.py stats $O/s_$tag $O/ks_$tag.csv > /dev/null
  rm -rf $O/s_$tag
  echo "== $tag $(python -c "import json;print(round(json.load(open('$O/$tag.json'))['ms_per_step']*1000,2))")"
  sed -n 2,4p $O/ks_$tag.csv | cut -d, -f1,4 | cut -c1-90
done
