#!/usr/bin/env python
"""Does the compute path change what the agent learns?  Seeds x {eager PyTorch fp32, fused HIP fp32, fused HIP
bf16} x {D4PG, D3PG} on the pendulum, trained in-process with device-resident explorers
(``d4pg_pytorch_amd/utils/learning.py``).

    python bench/learn_parity.py --seeds 5 --updates 20000 --out gpurun_out/learning

Writes one JSON line per run to ``<out>/runs.jsonl`` and a Markdown table to ``<out>/table.md``.
A random policy scores about -1200 .. -1500 per 200-step episode; a good one -150 .. -250.
"""
import argparse
import json
import os
import statistics as st
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--updates", type=int, default=20000)
    ap.add_argument("--models", nargs="*", default=["d4pg", "d3pg"])
    ap.add_argument("--cores", nargs="*", default=["torch", "fp32", "bf16"])
    ap.add_argument("--out", default="gpurun_out/learning")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    runs = []
    with open(os.path.join(args.out, "runs.jsonl"), "a") as fh:
        for model in args.models:
            for core in args.cores:
                for seed in range(args.seeds):
                    r = train_pendulum_on_device(model, core, args.updates, seed=100 + seed)
                    runs.append(r)
                    fh.write(json.dumps(r) + "\n")
                    fh.flush()
                    print(f"{model} {core} seed {seed}: last5 {r['last5_mean_return']} best {r['best_return']} "
                          f"first>-300 {r['first_eval_step_above_-300']} wall {r['wall_s']}s", flush=True)
    lines = ["| model | core | seeds | initial return | last-5-evals mean return (mean ± sd over seeds) | min / max over seeds "
             "| best eval (mean) | first eval > -300 (median update) | seeds reaching -300 | s per run |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for model in args.models:
        for core in args.cores:
            rs = [r for r in runs if r["model"] == model and r["core"] == core]
            if not rs:
                continue
            last = [r["last5_mean_return"] for r in rs]
            firsts = [r["first_eval_step_above_-300"] for r in rs if r["first_eval_step_above_-300"] is not None]
            lines.append(f"| {model} | {rs[0]['core_class']} | {len(rs)} | {st.mean(r['initial_return'] for r in rs):.0f} "
                         f"| {st.mean(last):.1f} ± {st.pstdev(last):.1f} | {min(last):.1f} / {max(last):.1f} "
                         f"| {st.mean(r['best_return'] for r in rs):.1f} | {int(st.median(firsts)) if firsts else '-'} "
                         f"| {len(firsts)}/{len(rs)} | {st.mean(r['wall_s'] for r in rs):.1f} |")
    with open(os.path.join(args.out, "table.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines), flush=True)


if __name__ == "__main__":
    main()
