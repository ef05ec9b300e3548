#!/bin/bash
# Opening measurements on one box: the ant optimisation-level bisect, and PMC / kernel-time profiles of the
# fp32 pipeline at large batches (which of LDS reads, MFMA issue and L2 -> L1 weight traffic binds the 8 / 16-row tiles).
O=gpurun_out/r3_first
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash bench/exp/ant_rootcause/run.sh > $O/ant_rootcause.log 2>&1; echo "ant rootcause exit $?"; tail -12 $O/ant_rootcause.log
timeout -k 10 120 python bench/exp/ant_variant_check.py > $O/ant_check_default_O1.txt 2>&1; tail -2 $O/ant_check_default_O1.txt
D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_ant_o3.so timeout -k 10 120 python bench/exp/ant_variant_check.py > $O/ant_check_O3.txt 2>&1; tail -2 $O/ant_check_O3.txt
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
b lunar256 --steps 3000 --warmup 300 || exit 1
b walker1024 --env Walker2d-v2 --batch 1024 --steps 2000 --warmup 200 || exit 1
b walker2048 --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 || exit 1
b ant4096 --env Ant-v2 --batch 4096 --steps 800 --warmup 100 || exit 1
ks() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --no-bf16 --steps 300 --warmup 50 "$@" > $O/${name}_bench.json 2> $O/${name}.err; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; }
ks walker2048 --env Walker2d-v2 --batch 2048 || exit 1
pmc() { name=$1; note=$2; shift; shift; ctrs=$1; shift; timeout -s KILL 150 rocprofv3 --pmc $ctrs -d $O/p_$name -o run --output-format csv -- python3 bench.py --steps 40 --warmup 8 --no-graph --no-bf16 "$@" > $O/${name}_bench.json 2> $O/${name}.err; rc=$?; python tools/rocprof_summary.py pmc $O/p_$name $O/$name.json --note "$note"; rm -rf $O/p_$name; return $rc; }
for shape in "walker2048 --env Walker2d-v2 --batch 2048" "ant4096 --env Ant-v2 --batch 4096"; do
  set -- $shape; tag=$1; shift
  pmc pmcA_$tag "fp32 $tag" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "$@" || exit 1
  pmc pmcB_$tag "fp32 $tag" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" "$@" || exit 1
  pmc pmcC_$tag "fp32 $tag" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "$@" || exit 1
done
ls $O
