#!/bin/bash
# The Ant-shape stress configuration (batch 4096, prioritised replay, fp32: the lean kernels) through train.py on one GPU
set -o pipefail
O=gpurun_out/r3_lean_e2e
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python train.py --config configs/mi355x/ant_stress_d4pg.yml --set num_learners=1 num_steps_train=3000 replay_mem_size=2000000 results_path=$O/results > $O/train.log 2>&1 || { tail -30 $O/train.log; exit 1; }
tail -12 $O/train.log
python tools/summarize_run.py $O/results/* > $O/summary.txt 2>&1; tail -15 $O/summary.txt
rm -rf $O/results
