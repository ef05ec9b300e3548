#!/usr/bin/env python
"""Throughput of the whole process topology: learner updates/s and environment steps/s.

    python bench/topology.py --config configs/mi355x/bipedal_d4pg_per_8explorers.yml \
        --updates 20000 [--set key=value ...]

Runs ``Engine.train()`` (learner on the GPU with the HBM replay + priority tree, CPU
explorer processes feeding it through shared-memory rings, GPU exploiter) for the given
number of learner updates and prints one JSON line.  Environments without a simulator in
this image run as shape-only synthetic environments (``synthetic_env: 1``), which is what
the throughput configurations prescribe.

    python bench/topology.py --gpu-rollout [--envs 4096] [--segment 32] [--segments 200]

measures the device-resident explorers instead (``actors/gpu_rollout.py``: policy, OU noise, pendulum
dynamics, n-step rows and replay insert on the GPU): environment steps/s of the rollout alone, and of the
rollout running beside the learner (one segment per 32 updates, on its own stream) with the updates/s the
learner keeps; plus the latency of one ``GpuActor.act`` (one launch, one synchronisation).
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


#: ids with dynamics of their own among the device kernels (everything else: shape-only dynamics)
NATIVE_DEVICE_ENVS = ("Pendulum", "InvertedPendulum-v2", "LunarLanderContinuous", "Hopper-", "Walker2d-", "HalfCheetah-", "BipedalWalker-", "Reacher-",
                      "InvertedDoublePendulum-", "Ant-")


def gpu_rollout_bench(args):
    import numpy as np
    import torch
    from d4pg_pytorch_amd.actors.gpu_actor import GpuActor
    from d4pg_pytorch_amd.actors.gpu_rollout import DeviceRollout
    from d4pg_pytorch_amd.config import default_config
    from d4pg_pytorch_amd.envs.device_envs import make_device_env
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    E, T = args.envs, args.segment
    pend = args.rollout_env.startswith("Pendulum")
    cfg = default_config(env=args.rollout_env, model="d4pg", batch_size=256, dense_size=400, device="cuda",
                         n_step_returns=5, replay_mem_size=1_000_000, v_min=-20.0 if pend else -100.0,
                         v_max=0.0 if pend else 100.0, replay_memory_prioritized=args.per,
                         synthetic_env=not args.rollout_env.startswith(NATIVE_DEVICE_ENVS))
    core = build_core(cfg, "cuda")
    rp = HbmReplay(cfg, "cuda", core.dev_state)
    core.attach_replay(rp)
    ro = DeviceRollout(cfg, make_device_env(cfg, E, "cuda", seed=1), rp, "cuda", segment=T, seed=1)
    S_, A_ = cfg["state_dim"], cfg["action_dim"]
    flat, base = core.online.flat.data_ptr(), core.layout.actor.base

    def sync_time(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(3):                                     # capture + warm-up
        ro.collect_overlapped(flat, base)
    # ---- the rollout alone (refresh, segment, insert) ----
    alone = sync_time(lambda: ro.collect_overlapped(flat, base), args.segments)
    # ---- beside the learner: 32 updates per segment ----
    def burst():
        done = 0
        while done < 32:
            done += core.update_from_replay_many(32 - done)
    for _ in range(4):
        burst()
    learner_only = sync_time(burst, args.segments)

    def both():
        ro.collect_overlapped(flat, base)
        burst()
    both()
    together = sync_time(both, args.segments)
    # ---- one act() of a GPU agent ----
    act_us = {}
    for n in (1, 64):
        ga = GpuActor(S_, A_, 400, max_batch=n)
        ga.load_flat(core.online.actor_flat().detach().cpu().numpy())
        x = np.zeros((n, S_), np.float32)
        for _ in range(20):
            ga.act(x)
        t0 = time.perf_counter()
        for _ in range(500):
            ga.act(x)
        act_us[str(n)] = round((time.perf_counter() - t0) / 500 * 1e6, 1)
        launches = ga.launches_per_act
    mean_ret, n_ep = ro.episode_stats()
    steps = args.segments * T * E
    out = {"mode": "gpu_rollout", "env": "%s (device kernels: %s)" % (args.rollout_env, type(ro.env).__name__), "envs": E, "segment_steps": T,
           "segments": args.segments, "replay": ("per-" if args.per else "uniform-") + "hbm", "dense_size": 400,
           "env_steps_per_s_rollout_alone": round(steps / alone, 0),
           "us_per_env_step_all_envs": round(alone / (args.segments * T) * 1e6, 2),
           "updates_per_s_learner_alone": round(args.segments * 32 / learner_only, 1),
           "updates_per_s_beside_rollout": round(args.segments * 32 / together, 1),
           "env_steps_per_s_beside_learner": round(steps / together, 0),
           "env_steps_per_update": T * E / 32,
           "gpu_actor_act_us_by_batch": act_us, "gpu_actor_launches_per_act": launches,
           "episodes": n_ep, "mean_return": mean_ret, "replay_rows": len(rp)}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config")
    ap.add_argument("--gpu-rollout", action="store_true")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--rollout-env", default="Pendulum-v0",
                    help="--gpu-rollout: Pendulum-v0, InvertedPendulum-v2 (cart-pole) or any known id (shape-only dynamics)")
    ap.add_argument("--segment", type=int, default=32)
    ap.add_argument("--segments", type=int, default=200)
    ap.add_argument("--per", type=int, default=0)
    ap.add_argument("--updates", type=int, default=20000)
    ap.add_argument("--set", nargs="*", default=[])
    ap.add_argument("--out", default="/tmp/topology_bench")
    args = ap.parse_args()
    if args.gpu_rollout:
        return gpu_rollout_bench(args)
    if not args.config:
        ap.error("--config is required (or --gpu-rollout)")
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
