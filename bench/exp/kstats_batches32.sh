# per-kernel durations of the fp32 update at global batches 512 / 1024 / 2048 (what a replicated data-parallel step costs)
set -e
O=gpurun_out/ks_batches32
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 512 1024 2048; do
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$b -o run --output-format csv -- python3 bench.py --steps 640 --warmup 64 --no-bf16 --env Walker2d-v2 --batch $b > $O/b$b.json 2> $O/b$b.err
  python tools/rocprof_summary.py stats $O/s_$b $O/kernel_stats_b$b.csv; rm -rf $O/s_$b
  echo "batch $b"; head -4 $O/kernel_stats_b$b.csv | cut -d, -f1-4 | cut -c1-120
done
