#!/usr/bin/env python
"""Train D4PG on the device BipedalWalker, then let the trained TARGET policy walk the HOST simulator (C++ build of
the same source, a fresh random terrain) and save what `Agent.save_replay_gif` would record (models/agent.py:147-163).

    python bench/exp/walker_gif.py --updates 120000 --out gpurun_out/walker_gif
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d4pg_pytorch_amd.actors.native import CpuActor                      # noqa: E402
from d4pg_pytorch_amd.config import default_config                       # noqa: E402
from d4pg_pytorch_amd.envs import create_env_wrapper                     # noqa: E402
from d4pg_pytorch_amd.utils.learning import train_on_device              # noqa: E402
from d4pg_pytorch_amd.utils.misc import frames_to_gif                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="BipedalWalker-v2")
    ap.add_argument("--updates", type=int, default=120000)
    ap.add_argument("--out", default="gpurun_out/walker_gif")
    ap.add_argument("--v-min", type=float, default=-100.0)
    ap.add_argument("--v-max", type=float, default=300.0)
    ap.add_argument("--limit", type=int, default=1600)
    ap.add_argument("--every", type=int, default=6, help="record every n-th frame")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    r = train_on_device("d4pg", "fp32", args.updates, seed=300, envs=256, eval_every=20000, env=args.env,
                        max_ep_length=args.limit, v_min=args.v_min, v_max=args.v_max, return_actor=True)
    flat = r.pop("actor_flat")
    cfg = default_config(env=args.env, model="d4pg", device="cpu", max_ep_length=args.limit)
    env = create_env_wrapper(cfg, 1, seed=12345)
    actor = CpuActor(env.state_dim, env.action_dim, int(cfg["dense_size"]))
    actor.load_flat(flat)
    obs, total, frames = env.reset(), 0.0, []
    for t in range(args.limit):
        if t % args.every == 0:
            frames.append(env.render(0))
        obs, rew, done = env.step(actor.act(obs))
        total += float(rew[0])
        if done[0]:
            break
    frames.append(env.render(0))
    name = args.env.split("-")[0].lower()
    gif = frames_to_gif(frames, os.path.join(args.out, f"{name}_trained.gif"), duration_ms=20 * args.every)
    out = {"env": args.env, "updates": r["updates"], "device_eval_curve": r["curve"], "host_episode_return": round(total, 1),
           "host_episode_steps": t + 1, "frames": len(frames), "gif": os.path.basename(gif), "host_env": type(env).__name__}
    json.dump(out, open(os.path.join(args.out, f"{name}_trained.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
