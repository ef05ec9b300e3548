"""Condense rocprofv3 output into the small files kept under ``profiles/``.

    python tools/rocprof_summary.py pmc   <dir with *counter_collection.csv> out.json  [--note "..."]
    python tools/rocprof_summary.py stats <dir with *kernel_stats.csv>       out.csv

``pmc``: per kernel name, the mean over dispatches of every collected counter (rocprofv3
reports one row per dispatch and counter, already summed over the chip's shader engines),
plus register / LDS / scratch footprints of the kernel.
``stats``: copies the kernel-stats table (first CSV found, searched recursively).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def _find(root: str, suffix: str):
    hits = sorted(glob.glob(os.path.join(root, "**", f"*{suffix}"), recursive=True))
    if not hits:
        sys.exit(f"no *{suffix} under {root}")
    return hits


def summarize_pmc(root: str, out: str, note: str) -> None:
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    meta = {}
    for path in _find(root, "counter_collection.csv"):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k].add((path, row["Dispatch_Id"]))
                meta.setdefault(k, {f: row.get(f) for f in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                                              "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                              "SGPR_Count") if f in row})
    kernels = {}
    for k, counters in acc.items():
        n = max(len(disp[k]), 1)
        kernels[k] = {c: round(v / n, 1) for c, v in sorted(counters.items())}
        kernels[k]["dispatches"] = n
        kernels[k].update({f.lower(): int(float(v)) for f, v in meta[k].items() if v not in (None, "")})
    with open(out, "w") as fh:
        json.dump({"tool": "rocprofv3 --pmc (per-dispatch means, summed over the chip)", "config": note,
                   "kernels": kernels}, fh, indent=1)
    print(f"{out}: {len(kernels)} kernels")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["pmc", "stats"])
    ap.add_argument("root")
    ap.add_argument("out")
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    if a.what == "pmc":
        summarize_pmc(a.root, a.out, a.note)
    else:
        shutil.copyfile(_find(a.root, "kernel_stats.csv")[0], a.out)
        print(a.out)


if __name__ == "__main__":
    main()
