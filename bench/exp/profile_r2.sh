# rocprofv3 evidence for the fp32 (headline) and bf16 pipelines: per-kernel durations (uniform and PER) and PMC passes
set -e
O=gpurun_out/${1:-prof_r2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ks() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --steps 800 --warmup 100 "$@" > $O/${name}_bench.json 2> $O/${name}.err; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; }
ks fp32 --no-bf16
ks fp32_per --no-bf16 --per
ks bf16 --dtype bf16
ks fp32_ant4096 --no-bf16 --env Ant-v2 --batch 4096
pmc() { name=$1; note=$2; shift; shift; ctrs=$1; shift; timeout -s KILL 150 rocprofv3 --pmc $ctrs -d $O/p_$name -o run --output-format csv -- python3 bench.py --steps 100 --warmup 20 --no-graph "$@" > $O/${name}_bench.json 2> $O/${name}.err; python tools/rocprof_summary.py pmc $O/p_$name $O/$name.json --note "$note"; rm -rf $O/p_$name; }
pmc pmc1_fp32 "bench.py --steps 100 --no-graph --no-bf16 (LunarLander D4PG fp32, B=256)" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --no-bf16
pmc pmc2_fp32 "bench.py --steps 100 --no-graph --no-bf16 (LunarLander D4PG fp32, B=256)" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum" --no-bf16
pmc pmc3_fp32 "bench.py --steps 100 --no-graph --no-bf16 (LunarLander D4PG fp32, B=256)" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" --no-bf16
pmc pmc1_bf16 "bench.py --steps 100 --no-graph --dtype bf16 (LunarLander D4PG bf16, B=256)" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --dtype bf16
ls $O
