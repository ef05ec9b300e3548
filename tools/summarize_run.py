#!/usr/bin/env python
"""Render a run directory into one PNG report following misc/summary_config.json.

    python tools/summarize_run.py results/<run> [--out report.png]
"""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from d4pg_pytorch_amd.utils.reward_plot import load_xy, smooth   # noqa: E402


def expand(run, pattern):
    if pattern.startswith("{"):
        return [d for p in pattern.strip("{}").split(",") for d in glob.glob(os.path.join(run, p))]
    return sorted(glob.glob(os.path.join(run, pattern)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("run")
    ap.add_argument("--out", default=None)
    ap.add_argument("--layout", default=os.path.join(ROOT, "misc", "summary_config.json"))
    args = ap.parse_args()
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    layout = json.load(open(args.layout))
    charts = [(sec, ch) for sec in layout["sections"] for ch in sec["charts"]]
    cols = 3
    rows = (len(charts) + cols - 1) // cols
    fig, axes = plt.subplots(rows, cols, figsize=(6 * cols, 3.5 * rows), squeeze=False)
    for ax, (sec, ch) in zip(axes.ravel(), charts):
        drawn = False
        for d in expand(args.run, sec["dir_glob"]):
            x, y = load_xy(d, ch["tag"])
            if x.size:
                ax.plot(x, smooth(y, ch.get("smooth", 1)), label=os.path.basename(d))
                drawn = True
        ax.set_title(ch["title"], fontsize=9)
        ax.set_xlabel(layout["x_axis"], fontsize=8)
        if drawn:
            ax.legend(fontsize=7)
        ax.grid(alpha=0.3)
    fig.suptitle(f"{layout['title']}: {os.path.basename(os.path.normpath(args.run))}")
    fig.tight_layout()
    out = args.out or os.path.join(args.run, "report.png")
    fig.savefig(out, dpi=110)
    print(out)


if __name__ == "__main__":
    main()
