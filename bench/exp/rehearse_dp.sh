# Rehearsal of the multi-rank benchmark on a box with ONE GPU: both ranks share the card and
# the collectives go through gloo (RCCL refuses two ranks on one device).
set -e
mkdir -p gpurun_out/rehearse
export D4PG_DIST_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 400 --warmup 48 > gpurun_out/rehearse/dp2.json 2> gpurun_out/rehearse/dp2.err
cat gpurun_out/rehearse/dp2.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 4 --steps 203 --warmup 21 > gpurun_out/rehearse/dp4.json 2> gpurun_out/rehearse/dp4.err
cat gpurun_out/rehearse/dp4.json
unset D4PG_DIST_BACKEND
timeout -k 10 300 python bench.py --dp-selftest --steps 1000 --warmup 100 > gpurun_out/rehearse/selftest.json 2> gpurun_out/rehearse/selftest.err
cat gpurun_out/rehearse/selftest.json
timeout -k 10 300 python bench.py --dp-selftest --steps 1000 --warmup 100 --dtype bf16 > gpurun_out/rehearse/selftest_bf16.json 2> gpurun_out/rehearse/selftest_bf16.err
cat gpurun_out/rehearse/selftest_bf16.json
timeout -k 10 300 python bench.py --dp-selftest --steps 640 --warmup 64 --env Ant-v2 --batch 4096 > gpurun_out/rehearse/selftest_ant4096.json 2> gpurun_out/rehearse/selftest_ant4096.err
cat gpurun_out/rehearse/selftest_ant4096.json
timeout -k 10 300 python bench.py --dp-selftest --steps 640 --warmup 64 --env Ant-v2 --batch 4096 --dtype bf16 > gpurun_out/rehearse/selftest_ant4096_bf16.json 2> gpurun_out/rehearse/selftest_ant4096_bf16.err
cat gpurun_out/rehearse/selftest_ant4096_bf16.json
