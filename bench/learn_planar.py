#!/usr/bin/env python
"""Learning runs on the articulated-body and closed-form tasks with device-resident explorers (``csrc/planar_env.hip``,
``csrc/ant_env.hip``, ``csrc/analytic_env.hip``),
the fused fp32 learner and the reference's hyper-parameters (batch 256, 400-wide nets, lr 5e-4, tau 0.001,
gamma 0.99, 5-step returns, 51 atoms): nothing but the evaluation result leaves the GPU.

    python bench/learn_planar.py --env Walker2d-v2 --updates 200000 --envs 256 --out gpurun_out/planar

The greedy TARGET policy is evaluated on 64 fresh episodes.  One JSON line per run in ``<out>/runs.jsonl``.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d4pg_pytorch_amd.utils.learning import train_on_device   # noqa: E402

#: value support = the range discounted returns can take (gamma 0.99), episode limit of the task
TASKS = {"Hopper-v2": (0.0, 400.0, 1000), "Walker2d-v2": (0.0, 400.0, 1000), "HalfCheetah-v2": (-50.0, 600.0, 1000),
         "BipedalWalker-v2": (-100.0, 300.0, 1600), "Ant-v2": (0.0, 300.0, 1000), "Reacher-v2": (-12.0, 0.0, 50),
         "InvertedDoublePendulum-v2": (0.0, 1000.0, 1000)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", nargs="*", default=["Hopper-v2", "Walker2d-v2", "HalfCheetah-v2", "BipedalWalker-v2"])
    ap.add_argument("--seeds", type=int, default=1)
    ap.add_argument("--updates", type=int, default=200000)
    ap.add_argument("--model", default="d4pg")
    ap.add_argument("--core", default="fp32")
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--segment", type=int, default=32)
    ap.add_argument("--eval-every", type=int, default=10000)
    ap.add_argument("--tau", type=float, default=0.001)
    ap.add_argument("--v-min", type=float, default=None)
    ap.add_argument("--v-max", type=float, default=None)
    ap.add_argument("--out", default="gpurun_out/planar")
    ap.add_argument("--overlap", action="store_true", help="collect the next segment beside the updates (the learner's own loop)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "runs.jsonl"), "a") as fh:
        for env in args.env:
            v_min, v_max, limit = TASKS[env]
            v_min = args.v_min if args.v_min is not None else v_min
            v_max = args.v_max if args.v_max is not None else v_max
            for seed in range(args.seeds):
                r = train_on_device(args.model, args.core, args.updates, seed=300 + seed, envs=args.envs,
                                             segment=args.segment, eval_every=args.eval_every, env=env, max_ep_length=limit,
                                             tau=args.tau, v_min=v_min, v_max=v_max, overlap=args.overlap)
                r["overlap"] = bool(args.overlap)
                r["envs"], r["env_steps_per_update"] = args.envs, args.envs * args.segment / 32
                r["env_steps_per_s"] = round(r["env_steps"] / r["wall_s"], 0)
                r["updates_per_s"] = round(r["updates"] / r["wall_s"], 1)
                fh.write(json.dumps(r) + "\n")
                fh.flush()
                print(f"{env} {args.model} seed {seed}: initial {r['initial_return']} last5 {r['last5_mean_return']} best "
                      f"{r['best_return']} env steps {r['env_steps']} wall {r['wall_s']}s ({r['updates_per_s']} updates/s, "
                      f"{r['env_steps_per_s']} env steps/s)", flush=True)
                print("   curve:", r["curve"], flush=True)


if __name__ == "__main__":
    main()
