#!/bin/bash
# The 8-explorer BipedalWalker topology on the REAL task (C++ host simulator in the explorer processes) instead of
# shape-only dynamics: prioritised and uniform replay, 200 000 updates each.
O=gpurun_out/topology_real
mkdir -p $O
for mode in per uniform; do
  extra=""; [ $mode = uniform ] && extra="replay_memory_prioritized=0"
  timeout -k 10 400 python bench/topology.py --config configs/mi355x/bipedal_d4pg_per_8explorers.yml --updates ${1:-200000} --set envs_per_agent=8 max_ep_length=1600 max_env_steps_per_agent=1000000000 $extra --out $O/run_$mode > $O/topo_$mode.json 2> $O/topo_$mode.err || { tail -5 $O/topo_$mode.err; exit 1; }
  tail -1 $O/topo_$mode.json | cut -c1-600
  grep "agent 0\]" $O/topo_$mode.json | tail -3
  rm -rf $O/run_$mode
done
