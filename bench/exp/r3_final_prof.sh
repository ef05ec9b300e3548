#!/bin/bash
# Profile of the headline (fp32, batch 256) and of its prioritised-replay and exchange-mode variants: kernel
# times and one PMC pass, then the headline numbers of the other shapes on the same box.
O=gpurun_out/r3_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ks() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --no-bf16 --no-extras --steps 300 --warmup 50 "$@" > $O/${name}_bench.json 2> $O/${name}.err || return 1; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; head -5 $O/kernel_stats_$name.csv | cut -c1-140; }
ks lunar256 || exit 1
ks lunar256_per --per || exit 1
ks lunar256_h300 --hidden 300 || exit 1
pmc() { name=$1; note=$2; shift; shift; ctrs=$1; shift; timeout -s KILL 150 rocprofv3 --pmc $ctrs -d $O/p_$name -o run --output-format csv -- python3 bench.py --steps 40 --warmup 8 --no-graph --no-bf16 --no-extras "$@" > $O/${name}_bench.json 2> $O/${name}.err; rc=$?; python tools/rocprof_summary.py pmc $O/p_$name $O/$name.json --note "$note"; rm -rf $O/p_$name; return $rc; }
pmc pmcA_lunar256 "fp32 headline, batch 256" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" || exit 1
pmc pmcB_lunar256 "fp32 headline, batch 256" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" || exit 1
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 --no-extras "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
b driver_style --steps 20 --warmup 5 || exit 1
b walker512 --env Walker2d-v2 --batch 512 || exit 1
b walker1024 --env Walker2d-v2 --batch 1024 --steps 2000 --warmup 200 || exit 1
b walker2048 --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 || exit 1
b walker4096 --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
b ant4096 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
b ant4096_per --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --per || exit 1
b lunar256_d3pg --model d3pg || exit 1
ls $O
