# per-kernel times of one update at global batches 512 / 1024 / 2048 (the step of replicated DP at 2 / 4 / 8 GPUs)
set -e
mkdir -p gpurun_out/batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 512 1024 2048; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/batch/s$B -o run --output-format csv -- python3 bench.py --env Walker2d-v2 --batch $B --steps 800 --warmup 100 > gpurun_out/batch/b$B.json 2> gpurun_out/batch/b$B.err
  python tools/rocprof_summary.py stats gpurun_out/batch/s$B gpurun_out/batch/kernel_stats_$B.csv
  rm -rf gpurun_out/batch/s$B
done
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/batch/s2048t32 -o run --output-format csv -- python3 bench.py --env Walker2d-v2 --batch 2048 --tile-rows 32 --steps 800 --warmup 100 > gpurun_out/batch/b2048t32.json 2> gpurun_out/batch/b2048t32.err
python tools/rocprof_summary.py stats gpurun_out/batch/s2048t32 gpurun_out/batch/kernel_stats_2048t32.csv
rm -rf gpurun_out/batch/s2048t32
