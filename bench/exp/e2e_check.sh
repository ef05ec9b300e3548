# end-to-end runs on the current kernels: multi-rank rehearsal, whole topology, learning on Pendulum
set -e
O=gpurun_out/${1:-e2e}
mkdir -p $O
bash bench/exp/rehearse_dp.sh > $O/rehearse.log 2>&1 || { tail -30 $O/rehearse.log; exit 1; }
cp gpurun_out/rehearse/dp2.json $O/dp2.json; cp gpurun_out/rehearse/dp4.json $O/dp4.json
echo rehearsal ok
timeout -k 10 280 python bench/topology.py --config configs/mi355x/bipedal_d4pg_per_8explorers.yml --updates 80000 --set replay_memory_prioritized=0 envs_per_agent=8 synthetic_env=true --out $O/topo_uniform > $O/topo_uniform.json 2> $O/topo_uniform.err
tail -c 600 $O/topo_uniform.json; echo
timeout -k 10 280 python bench/topology.py --config configs/mi355x/bipedal_d4pg_per_8explorers.yml --updates 80000 --set envs_per_agent=8 synthetic_env=true --out $O/topo_per > $O/topo_per.json 2> $O/topo_per.err
tail -c 600 $O/topo_per.json; echo
for m in d4pg d3pg; do
  timeout -k 10 200 python bench/learn_pendulum.py --model $m --dtype bf16 --out $O/learn_$m > $O/learn_$m.json 2> $O/learn_$m.err
  tail -c 400 $O/learn_$m.json; echo
done
timeout -k 10 200 python bench/learn_pendulum.py --model d4pg --dtype bf16 --per 1 --out $O/learn_d4pg_per > $O/learn_d4pg_per.json 2> $O/learn_d4pg_per.err
tail -c 400 $O/learn_d4pg_per.json; echo
