set -e
mkdir -p gpurun_out/v5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/v5/gpu_tests.log 2>&1
tail -3 gpurun_out/v5/gpu_tests.log
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/v5/stats -o run --output-format csv -- python3 bench.py --steps 800 --warmup 100 > gpurun_out/v5/stats_bench.json 2> gpurun_out/v5/stats.err
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/v5/stats_per -o run --output-format csv -- python3 bench.py --steps 800 --warmup 100 --per > gpurun_out/v5/stats_per_bench.json 2> gpurun_out/v5/stats_per.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/v5/pmc1 -o run --output-format csv -- python3 bench.py --steps 100 --warmup 20 --no-graph > gpurun_out/v5/pmc1_bench.json 2> gpurun_out/v5/pmc1.err
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum -d gpurun_out/v5/pmc2 -o run --output-format csv -- python3 bench.py --steps 100 --warmup 20 --no-graph > gpurun_out/v5/pmc2_bench.json 2> gpurun_out/v5/pmc2.err
python tools/rocprof_summary.py pmc gpurun_out/v5/pmc1 gpurun_out/v5/pmc1.json --note "bench.py --steps 100 --no-graph (LunarLander D4PG bf16, B=256)"
python tools/rocprof_summary.py pmc gpurun_out/v5/pmc2 gpurun_out/v5/pmc2.json --note "bench.py --steps 100 --no-graph (LunarLander D4PG bf16, B=256)"
python tools/rocprof_summary.py stats gpurun_out/v5/stats gpurun_out/v5/kernel_stats.csv
python tools/rocprof_summary.py stats gpurun_out/v5/stats_per gpurun_out/v5/kernel_stats_per.csv
rm -rf gpurun_out/v5/stats gpurun_out/v5/stats_per
ls gpurun_out/v5
