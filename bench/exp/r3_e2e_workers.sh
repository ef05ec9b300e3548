#!/bin/bash
# train.py end to end with a rollout worker process (here beside the learner on the one GPU of the box): Walker2d, 150 000 updates
O=gpurun_out/r3_e2e_workers
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u train.py --config configs/mi355x/walker2d_d4pg_device_explorers.yml --set rollout_devices=[0] num_steps_train=150000 results_path=$O/results > $O/train.log 2>&1; rc=$?
echo "train exit $rc"; grep -a "Exit\|results in\|restart" $O/train.log | tail -8
python tools/summarize_run.py $(ls -d $O/results/*/ | tail -1) > $O/summary.json 2>/dev/null; cut -c1-1500 $O/summary.json
rm -rf $O/results/*/learner/*.pt $O/results/*/agent-0/*.pt
exit $rc
