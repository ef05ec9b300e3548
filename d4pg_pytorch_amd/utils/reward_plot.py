"""Offline learning-curve plots (reference ``utils/reward_plot.py:12-56``).

The reference loads scalars exported from TensorBoard as JSON rows
``[wall_time, step, value]``, smooths them with a trailing window of 5 and draws
one line per algorithm and one panel per environment into ``plot.png``.  This
version does the same and can ALSO read the event files / ``scalars.jsonl``
written by :mod:`utils.logger` directly, so no TensorBoard export step is needed:

    python -m d4pg_pytorch_amd.utils.reward_plot \\
        --run Pendulum-v0:D4PG:results/Pendulum-v0-d4pg-2026-.../agent-0 \\
        --run Pendulum-v0:D3PG:results/Pendulum-v0-d3pg-2026-.../agent-0 --out plot.png
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, List, Tuple

import numpy as np

from .logger import read_events


def load_xy(path: str, tag: str = "agent/reward") -> Tuple[np.ndarray, np.ndarray]:
    """``(steps, values)`` from a TensorBoard JSON export, a ``scalars.jsonl`` file, an event
    file or a directory of event files."""
    if os.path.isfile(path) and path.endswith(".json"):
        rows = np.asarray(json.load(open(path)), dtype=np.float64)     # [wall, step, value]
        return rows[:, 1], rows[:, 2]
    if os.path.isfile(path) and path.endswith(".jsonl"):
        rows = [json.loads(l) for l in open(path) if l.strip()]
        rows = [r for r in rows if r["tag"] == tag]
        return np.asarray([r["step"] for r in rows], float), np.asarray([r["value"] for r in rows], float)
    ev = [e for e in read_events(path) if e[2] == tag]
    return np.asarray([e[1] for e in ev], float), np.asarray([e[3] for e in ev], float)


def smooth(y: np.ndarray, window: int = 5) -> np.ndarray:
    """Trailing moving average; the first ``window-1`` points average what exists so far."""
    y = np.asarray(y, dtype=np.float64)
    if y.size == 0 or window <= 1:
        return y
    c = np.cumsum(np.insert(y, 0, 0.0))
    out = np.empty_like(y)
    for i in range(y.size):
        lo = max(0, i + 1 - window)
        out[i] = (c[i + 1] - c[lo]) / (i + 1 - lo)
    return out


def plot_data(runs: Dict[str, List[Tuple[str, str]]], out: str = "plot.png", tag: str = "agent/reward",
              window: int = 5) -> str:
    """``runs``: ``{env: [(label, path), ...]}`` -> one panel per environment."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    envs = list(runs)
    fig, axes = plt.subplots(1, len(envs), figsize=(6 * len(envs), 4), squeeze=False)
    for ax, env in zip(axes[0], envs):
        for label, path in runs[env]:
            x, y = load_xy(path, tag)
            ax.plot(x, smooth(y, window), label=label)
        ax.set_title(env)
        ax.set_xlabel("learner update step")
        ax.set_ylabel("episode reward")
        ax.legend()
        ax.grid(alpha=0.3)
    fig.tight_layout()
    fig.savefig(out, dpi=120)
    plt.close(fig)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="append", required=True, metavar="ENV:LABEL:PATH")
    ap.add_argument("--out", default="plot.png")
    ap.add_argument("--tag", default="agent/reward")
    ap.add_argument("--window", type=int, default=5)
    args = ap.parse_args()
    runs: Dict[str, List[Tuple[str, str]]] = {}
    for spec in args.run:
        env, label, path = spec.split(":", 2)
        runs.setdefault(env, []).append((label, path))
    print(plot_data(runs, args.out, args.tag, args.window))


if __name__ == "__main__":
    main()
