#!/bin/bash
# Simulator kernels without restrict-qualified parameters: correctness against the host simulators and the time of a
# step at each optimisation level (ant), and the planar kernels against their former restrict-qualified build.
O=gpurun_out/r3_sim_levels
mkdir -p $O
bash bench/exp/ant_rootcause/run_step.sh > $O/run_step.log 2>&1 || { tail -5 $O/run_step.log; exit 1; }
tail -3 $O/run_step.log | cut -c1-300
for v in default ant_O1 ant_O2; do
  if [ $v = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$v.so; fi
  timeout -k 10 120 python bench/exp/ant_variant_check.py > $O/ant_check_$v.txt 2>&1 || { tail -5 $O/ant_check_$v.txt; exit 1; }
  echo "ant $v: $(tail -2 $O/ant_check_$v.txt | tr '\n' ' ')"
done
unset D4PG_HIP_LIB
timeout -k 10 300 python -u -m pytest tests/test_spatial_env.py tests/test_planar_envs.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1; echo "tests exit $?"; tail -3 $O/tests.log
for v in default; do
  if [ $v = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$v.so; fi
  for env in Hopper-v2 Walker2d-v2 HalfCheetah-v2 BipedalWalker-v2 Ant-v2; do
    timeout -k 10 120 python bench/topology.py --gpu-rollout --rollout-env $env --envs 4096 --segments 60 > $O/rollout_${v}_$env.json 2> $O/rollout_${v}_$env.err || { tail -5 $O/rollout_${v}_$env.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/rollout_${v}_$env.json").read().strip().splitlines()[-1])
print("$v $env", {k: d[k] for k in d if "steps_per" in k or "us_per" in k or "updates_per" in k})
PY
  done
done
