#!/bin/bash
# Two data-parallel learner ranks (sharing the one GPU of the box, collectives through gloo) with two rollout worker
# processes dealt to them round-robin: every rank's replay shard is fed through its own row mailbox.
O=gpurun_out/r3_dp_workers
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export D4PG_DIST_BACKEND=gloo
timeout -k 10 400 python -u train.py --config configs/mi355x/pendulum_d4pg_device_explorers.yml --set num_learners=2 learner_devices=[0,0] rollout_devices=[0,0] num_steps_train=4000 results_path=$O/results > $O/train.log 2>&1; rc=$?
echo "train exit $rc"; grep -a "Exit\|done\|results in\|restart\|Traceback\|Error\|all-reduce\|dp_mode\|replicated" $O/train.log | tail -14 | cut -c1-300
rm -rf $O/results
exit $rc
