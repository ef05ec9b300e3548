#!/bin/bash
# HalfCheetah seed 300 took 30.8 s where seeds 301 / 302 took 17.8 s: again, with a kernel trace of a shorter run
O=$GRAFT_REPO_ROOT/gpurun_out/r3_cheetah_slow
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python -u bench/learn_planar.py --env HalfCheetah-v2 --updates 100000 --seeds 2 --overlap --out $O > $O/run.log 2>&1; grep -v curve $O/run.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o c --output-format csv -- python3 $GRAFT_REPO_ROOT/bench/learn_planar.py --env HalfCheetah-v2 --updates 60000 --seeds 1 --overlap --out $O/p > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py stats $O/prof $O/kernel_stats_cheetah_seed300.csv && rm -rf $O/prof && head -7 $O/kernel_stats_cheetah_seed300.csv | cut -c1-150
