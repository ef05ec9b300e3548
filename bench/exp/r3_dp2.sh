#!/bin/bash
# two-rank rehearsal of bench.py on the one-GPU box (gloo: RCCL refuses two ranks on one device)
O=gpurun_out/r3_dp2
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export D4PG_DIST_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 200 --warmup 32 > $O/dp2.json 2> $O/dp2.err; rc=$?; echo "dp2 exit $rc"; cut -c1-3000 $O/dp2.json; tail -5 $O/dp2.err | cut -c1-300
exit $rc
