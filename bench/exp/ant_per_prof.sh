#!/bin/bash
# rocprofv3 kernel trace of the Ant-shape batch-4096 update with a 10M-row tree: prioritised vs uniform, fp32 and bf16
set -e
O=gpurun_out/${1:-ant_per_prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ks() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --steps 300 --warmup 64 --env Ant-v2 --batch 4096 --replay-size 10000000 --prefill 1000000 "$@" > $O/${name}_bench.json 2> $O/${name}.err; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; }
ks fp32_per --no-bf16 --per
ks fp32_uni --no-bf16
ks bf16_per --dtype bf16 --per
ks bf16_uni --dtype bf16
for v in fp32_per fp32_uni bf16_per bf16_uni; do echo "== $v"; head -6 $O/kernel_stats_$v.csv | cut -c1-130; done
