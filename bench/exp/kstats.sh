set -e
mkdir -p gpurun_out/ks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/ks/s -o run --output-format csv -- python3 bench.py --steps 800 --warmup 100 $@ > gpurun_out/ks/b.json 2> gpurun_out/ks/b.err
python tools/rocprof_summary.py stats gpurun_out/ks/s gpurun_out/ks/kernel_stats.csv
rm -rf gpurun_out/ks/s
head -5 gpurun_out/ks/kernel_stats.csv | cut -c1-120
python -c "import json;print(json.load(open('gpurun_out/ks/b.json'))['ms_per_step'])"
