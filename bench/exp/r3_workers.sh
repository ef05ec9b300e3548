#!/bin/bash
# Rollout workers, any-width inference, bench extras: tests first, then the driver-style bench and a two-rank rehearsal.
O=gpurun_out/r3_workers
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_rollout_worker.py tests/test_gpu_rollout.py tests/test_gpu_engine.py -m gpu -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1; rc=$?; echo "tests exit $rc"; tail -25 $O/tests.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench exit $?"; cut -c1-1800 $O/bench_driver_style.json
export D4PG_DIST_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 200 --warmup 32 > $O/dp2.json 2> $O/dp2.err; echo "dp2 exit $?"; cut -c1-2500 $O/dp2.json; tail -5 $O/dp2.err | cut -c1-300
