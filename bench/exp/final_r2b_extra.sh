#!/bin/bash
# the remaining evidence on the final kernels: GPU test log, per-layer stamps, whole topology (uniform / PER),
# device explorers (pendulum, lunar lander), rehearsal of the multi-rank paths
O=gpurun_out/final_r2b_extra
mkdir -p $O
timeout -k 10 700 python -u -m pytest tests -m gpu -q --timeout 150 --timeout-method thread > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
bash bench/exp/stamps32_run.sh 256 > /dev/null 2>&1; cp gpurun_out/stamps/f32_b256.json $O/stamps_fp32.json
timeout -k 10 280 python bench/topology.py --config configs/mi355x/bipedal_d4pg_per_8explorers.yml --updates 60000 --set replay_memory_prioritized=0 envs_per_agent=8 synthetic_env=true --out $O/topo_uniform > $O/topo_uniform.json 2> $O/topo_uniform.err; tail -c 300 $O/topo_uniform.json; echo
timeout -k 10 280 python bench/topology.py --config configs/mi355x/bipedal_d4pg_per_8explorers.yml --updates 60000 --set envs_per_agent=8 synthetic_env=true --out $O/topo_per > $O/topo_per.json 2> $O/topo_per.err; tail -c 300 $O/topo_per.json; echo
timeout -k 10 200 python bench/topology.py --gpu-rollout > $O/rollout_pendulum.json 2> $O/rollout_pendulum.err; tail -c 400 $O/rollout_pendulum.json; echo
timeout -k 10 200 python bench/topology.py --gpu-rollout --rollout-env LunarLanderContinuous-v2 > $O/rollout_lander.json 2> $O/rollout_lander.err; tail -c 400 $O/rollout_lander.json; echo
bash bench/exp/rehearse_dp.sh > $O/rehearse.log 2>&1; cp gpurun_out/rehearse/*.json $O/ 2>/dev/null; tail -3 $O/rehearse.log | cut -c1-300
