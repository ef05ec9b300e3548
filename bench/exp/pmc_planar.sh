#!/bin/bash
# Hardware counters of the environment kernels (one pass): instructions per wave, wave cycles, busy cycles.
O=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/p -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench/exp/planar_variant_check.py > $O/out.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/p/*counter_collection.csv $O/p/*/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] || { tail -5 $O/out.log; exit 1; }
python - "$f" <<'PY' | tee $O/pmc_planar_env_kernels.txt
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "planar_step" not in k:
        continue
    k = k[k.index("<") + 1:k.index(">")]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
for k, d in acc.items():
    print(k, "dispatches", n[k], {c: round(v / max(1, n[k]), 1) for c, v in d.items()})
PY
rm -rf $O/p
