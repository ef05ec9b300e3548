#!/bin/bash
# Instruction-cache counters of the phase kernels (their code is 85 - 240 KB; the instruction cache holds 64 KB):
# headline shape on the 4x4x1 kernels, Ant 4096 on the 16x16x4 kernels (one and two workgroups per CU)
O=gpurun_out/r3_icache_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() { name=$1; note=$2; shift; shift; ctrs=$1; shift; timeout -s KILL 150 rocprofv3 --pmc $ctrs -d $O/p_$name -o run --output-format csv -- python3 bench.py --steps 40 --warmup 8 --no-graph --no-bf16 --no-extras "$@" > $O/${name}_bench.json 2> $O/${name}.err; rc=$?; [ $rc = 0 ] && python tools/rocprof_summary.py pmc $O/p_$name $O/$name.json --note "$note"; rm -rf $O/p_$name; return $rc; }
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
pmc pmcI_lunar256 "fp32 headline, batch 256" "$C" || { tail -5 $O/pmcI_lunar256.err; exit 1; }
pmc pmcI_ant4096_occ1 "fp32 16x16x4, one workgroup per CU, Ant 4096" "$C" --env Ant-v2 --batch 4096 --occupancy 1 || { tail -5 $O/pmcI_ant4096_occ1.err; exit 1; }
pmc pmcI_ant4096_lean "fp32 16x16x4 lean, Ant 4096" "$C" --env Ant-v2 --batch 4096 --occupancy 2 || { tail -5 $O/pmcI_ant4096_lean.err; exit 1; }
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_icache_pmc/pmcI*.json")):
    if f.endswith("_bench.json"): continue
    d = json.load(open(f))
    for k, v in d.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                if "phase32" in kk or "wgrad32" in kk: print(f.split("/")[-1], kk[:44], {a: round(b) for a, b in vv.items() if a.startswith(("SQ_", "SQC_"))})
PY
