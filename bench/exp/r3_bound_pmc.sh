#!/bin/bash
# What the 16x16x4 phase kernels are bound by at 4096 rows: address path (TA), vector cache (TCP) and issue counters
O=gpurun_out/r3_bound_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 60 rocprofv3 -L > $O/avail.txt 2>&1
grep -c . $O/avail.txt
pmc() { name=$1; note=$2; shift; shift; ctrs=$1; shift; timeout -s KILL 150 rocprofv3 --pmc $ctrs -d $O/p_$name -o run --output-format csv -- python3 bench.py --steps 40 --warmup 8 --no-graph --no-bf16 --no-extras "$@" > $O/${name}_bench.json 2> $O/${name}.err; rc=$?; [ $rc = 0 ] && python tools/rocprof_summary.py pmc $O/p_$name $O/$name.json --note "$note"; rm -rf $O/p_$name; return $rc; }
pmc pmcD_ant4096_occ1 "fp32 16x16x4, one workgroup per CU, Ant 4096" "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --env Ant-v2 --batch 4096 --occupancy 1 || { tail -5 $O/pmcD_ant4096_occ1.err; exit 1; }
pmc pmcD_ant4096_lean "fp32 16x16x4 lean, Ant 4096" "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --env Ant-v2 --batch 4096 --occupancy 2 || { tail -5 $O/pmcD_ant4096_lean.err; exit 1; }
pmc pmcE_ant4096_occ1 "fp32 16x16x4, one workgroup per CU, Ant 4096" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES" --env Ant-v2 --batch 4096 --occupancy 1 || { tail -5 $O/pmcE_ant4096_occ1.err; exit 1; }
pmc pmcE_ant4096_lean "fp32 16x16x4 lean, Ant 4096" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES" --env Ant-v2 --batch 4096 --occupancy 2 || { tail -5 $O/pmcE_ant4096_lean.err; exit 1; }
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_bound_pmc/pmc*.json")):
    if f.endswith("_bench.json"): continue
    d = json.load(open(f))
    for k, v in d.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                if "phase32" in kk or "wgrad32" in kk: print(f.split("/")[-1], kk[:48], {a: round(b) for a, b in vv.items() if a.startswith(("SQ_", "TCC_", "TA_", "TCP_"))})
PY
