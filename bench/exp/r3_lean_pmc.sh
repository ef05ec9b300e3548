#!/bin/bash
# PMC passes of the lean 16x16x4 phase kernels (two workgroups per CU), Ant shape, 4096 rows
O=gpurun_out/r3_lean_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() { name=$1; note=$2; shift; shift; ctrs=$1; shift; timeout -s KILL 150 rocprofv3 --pmc $ctrs -d $O/p_$name -o run --output-format csv -- python3 bench.py --steps 40 --warmup 8 --no-graph --no-bf16 --no-extras "$@" > $O/${name}_bench.json 2> $O/${name}.err; rc=$?; python tools/rocprof_summary.py pmc $O/p_$name $O/$name.json --note "$note"; rm -rf $O/p_$name; return $rc; }
pmc pmcA_ant4096_lean "fp32 16x16x4 lean build (two workgroups per CU), Ant shape, 4096 rows" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" --env Ant-v2 --batch 4096 --occupancy 2 || exit 1
pmc pmcB_ant4096_lean "fp32 16x16x4 lean build (two workgroups per CU), Ant shape, 4096 rows" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" --env Ant-v2 --batch 4096 --occupancy 2 || exit 1
pmc pmcA_ant4096_occ1 "fp32 16x16x4, one workgroup per CU, Ant shape, 4096 rows (same box)" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" --env Ant-v2 --batch 4096 --occupancy 1 || exit 1
python - <<'PY'
import json
for n in ("pmcA_ant4096_lean", "pmcA_ant4096_occ1", "pmcB_ant4096_lean"):
    d = json.load(open(f"gpurun_out/r3_lean_pmc/{n}.json"))
    for k, v in d.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                if "phase32" in kk: print(n, kk[:60], {a: b for a, b in vv.items() if a.startswith(("SQ_", "TCC_"))})
PY
