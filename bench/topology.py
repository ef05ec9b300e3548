#!/usr/bin/env python
"""Throughput of the whole process topology: learner updates/s and environment steps/s.

    python bench/topology.py --config configs/mi355x/bipedal_d4pg_per_8explorers.yml \
        --updates 20000 [--set key=value ...]

Runs ``Engine.train()`` (learner on the GPU with the HBM replay + priority tree, CPU
explorer processes feeding it through shared-memory rings, GPU exploiter) for the given
number of learner updates and prints one JSON line.  Environments without a simulator in
this image run as shape-only synthetic environments (``synthetic_env: 1``), which is what
the throughput configurations prescribe.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d4pg_pytorch_amd.config import read_config          # noqa: E402
from d4pg_pytorch_amd.engine import load_engine          # noqa: E402
from d4pg_pytorch_amd.utils.logger import read_events    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--updates", type=int, default=20000)
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--out", default="/tmp/topology_bench")
    args = ap.parse_args()
    overrides = [f"num_steps_train={args.updates}", f"results_path={args.out}", "num_episode_save=1000000",
                 "save_reward_threshold=1e9"] + list(args.set)
    cfg = read_config(args.config, overrides)
    t0 = time.time()
    engine = load_engine(cfg)
    exp = engine.train()
    wall = time.time() - t0
    ev = read_events(os.path.join(exp, "learner"))
    ups = [e[3] for e in ev if e[2] == "perf/updates_per_s"]
    eps = [e[3] for e in ev if e[2] == "perf/env_steps_per_s"]
    steady = slice(len(ups) // 3, None)       # skip start-up (replay fill, graph capture)
    out = {
        "config": os.path.basename(args.config), "env": cfg["env"], "model": cfg["model"],
        "num_agents": cfg["num_agents"], "envs_per_agent": cfg["envs_per_agent"],
        "replay": ("per-" if cfg["replay_memory_prioritized"] else "uniform-") + cfg["replay_device"],
        "compute_dtype": cfg.get("compute_dtype"), "batch_size": cfg["batch_size"],
        "updates": engine.final_update_step, "env_steps": engine.final_env_steps, "episodes": engine.final_episodes,
        "wall_s_including_spawn": round(wall, 2),
        "steady_updates_per_s": round(sum(ups[steady]) / max(1, len(ups[steady])), 1) if ups else None,
        "steady_env_steps_per_s": round(sum(eps[steady]) / max(1, len(eps[steady])), 1) if eps else None,
        "ring_dropped": max([e[3] for e in ev if e[2] == "data_struct/ring_dropped"] or [0]),
        "exit_codes_ok": all(c == 0 for c in engine.exit_codes.values()),
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
