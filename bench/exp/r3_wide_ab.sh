#!/bin/bash
# 16x16x4 phase kernels: ring depth / explicit LDS prefetch variants on one box, per-layer stamps, PMC.
O=gpurun_out/r3_wide_ab
mkdir -p $O
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us')"; }
for v in default ring4 ring5 ldsasm; do
  if [ $v = default ]; then unset D4PG_HIP_LIB; else export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_$v.so; fi
  if [ $v = ldsasm ]; then timeout -k 10 300 python -u -m pytest tests/test_hip_f32.py -m gpu -q -x --timeout 120 --timeout-method thread -k "gradients" > $O/tests_$v.log 2>&1 || { tail -5 $O/tests_$v.log; exit 1; }; tail -1 $O/tests_$v.log; fi
  b ant4096_$v --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --mfma 16x16 || exit 1
  b walker4096_$v --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 --mfma 16x16 || exit 1
  b walker2048_$v --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 --mfma 16x16 || exit 1
done
export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_stamps32.so
D4PG_STAMP_ENV=Ant-v2 D4PG_STAMP_MFMA=16x16 timeout -k 10 150 python bench/stamps.py 4096 fp32 > $O/stamps_ant4096_16x16.json 2> $O/stamps_ant4096.err || tail -3 $O/stamps_ant4096.err
D4PG_STAMP_ENV=Walker2d-v2 D4PG_STAMP_MFMA=16x16 timeout -k 10 150 python bench/stamps.py 2048 fp32 > $O/stamps_walker2048_16x16.json 2> $O/stamps_w2048.err || tail -3 $O/stamps_w2048.err
unset D4PG_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() { name=$1; note=$2; shift; shift; ctrs=$1; shift; timeout -s KILL 150 rocprofv3 --pmc $ctrs -d $O/p_$name -o run --output-format csv -- python3 bench.py --steps 40 --warmup 8 --no-graph --no-bf16 "$@" > $O/${name}_bench.json 2> $O/${name}.err; rc=$?; python tools/rocprof_summary.py pmc $O/p_$name $O/$name.json --note "$note"; rm -rf $O/p_$name; return $rc; }
pmc pmcA_ant4096_16x16 "fp32 16x16x4 ant4096" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" --env Ant-v2 --batch 4096 --mfma 16x16
pmc pmcB_ant4096_16x16 "fp32 16x16x4 ant4096" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" --env Ant-v2 --batch 4096 --mfma 16x16
