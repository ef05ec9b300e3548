#!/bin/bash
# Per-layer clock stamps of the 16x16x4 phase kernels at 4096 rows, one and two workgroups per CU
O=gpurun_out/r3_lean
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_stamps32.so
for occ in 1 2; do
D4PG_STAMP_OCC=$occ D4PG_STAMP_ENV=Ant-v2 timeout -k 10 150 python bench/stamps.py 4096 fp32 > $O/stamps_ant4096_occ$occ.json 2> $O/stamps_ant4096_occ$occ.err || { tail -3 $O/stamps_ant4096_occ$occ.err; exit 1; }
python - <<PY
import json
d=json.load(open('$O/stamps_ant4096_occ$occ.json'))
for k in ('critic_launch_spans_last_iter_us_rel_wg0_entry','critic_phase','actor_phase'): print($occ, k, json.dumps(d[k]))
PY
done
