#!/bin/bash
# rocprofv3 kernel trace of the prioritised fp32 update: per-kernel times of the current schedule (next batch drawn
# and critic forward made ahead inside the actor-phase launch) and of the older one (D4PG_PER_AHEAD=0)
set -e
O=gpurun_out/${1:-per_prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ks() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --steps 800 --warmup 100 "$@" > $O/${name}_bench.json 2> $O/${name}.err; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; }
ks per_new --no-bf16 --per
export D4PG_PER_AHEAD=0
ks per_old --no-bf16 --per
for v in per_new per_old; do echo "== $v"; head -5 $O/kernel_stats_$v.csv | cut -c1-140; done
