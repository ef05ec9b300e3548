#!/bin/bash
# Mixed topology soak: CPU explorer processes (rings) AND a rollout worker process (row mailbox) feed one prioritised HBM
# replay; GPU exploiter through HIP IPC; 100 000 updates.  Everything must exit with code 0 and nothing may be dropped.
O=gpurun_out/r3_soak
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u train.py --config configs/mi355x/walker2d_d4pg_device_explorers.yml --set rollout_devices=[0] num_agents=3 envs_per_agent=8 replay_memory_prioritized=1 num_steps_train=100000 results_path=$O/results > $O/train.log 2>&1; rc=$?
echo "train exit $rc"; grep -a "Exit\|done\|results in\|restart\|Traceback\|Error" $O/train.log | tail -12
rm -rf $O/results
exit $rc
