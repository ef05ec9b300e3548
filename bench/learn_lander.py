#!/usr/bin/env python
"""Learning runs on the native lunar lander (``envs/lunar_lander.py``, the task of ``LunarLanderContinuous-v2``)
with device-resident explorers, the fused fp32 learner and the reference's hyper-parameters
(``configs/lunar_lander_continuous_d4pg.yml``: batch 256, 400-wide nets, lr 5e-4, tau 0.001, gamma 0.99, 5-step
returns, 51 atoms; reward / 100).  The greedy TARGET policy is evaluated on 64 fresh episodes.

    python bench/learn_lander.py --updates 60000 --seeds 3 --out gpurun_out/lander

A random policy scores about -280, crashing costs -100, a landing pays +100 and a good landing run totals
>= 200 (the conventional "solved" level of the gym task; this is the same task on its own simulator, so the
level is comparable in kind only).  One JSON line per run in ``<out>/runs.jsonl``.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--updates", type=int, default=60000)
    ap.add_argument("--models", nargs="*", default=["d4pg", "d3pg"])
    ap.add_argument("--cores", nargs="*", default=["fp32"])
    ap.add_argument("--envs", type=int, default=16)
    ap.add_argument("--eval-every", type=int, default=2000)
    ap.add_argument("--tau", type=float, default=0.001)
    ap.add_argument("--out", default="gpurun_out/lander")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "runs.jsonl"), "a") as fh:
        for model in args.models:
            for core in args.cores:
                for seed in range(args.seeds):
                    r = train_pendulum_on_device(model, core, args.updates, seed=200 + seed, envs=args.envs,
                                                 eval_every=args.eval_every, env="LunarLanderContinuous-v2",
                                                 max_ep_length=1000, tau=args.tau)
                    r["tau"] = args.tau
                    r["first_eval_step_above_200"] = next((s for s, v in r["curve"] if v > 200.0), None)
                    r["first_eval_step_above_0"] = next((s for s, v in r["curve"] if v > 0.0), None)
                    fh.write(json.dumps(r) + "\n")
                    fh.flush()
                    print(f"{model} {core} tau {args.tau} seed {seed}: initial {r['initial_return']} last5 {r['last5_mean_return']} "
                          f"best {r['best_return']} first>0 {r['first_eval_step_above_0']} first>200 "
                          f"{r['first_eval_step_above_200']} env steps {r['env_steps']} wall {r['wall_s']}s", flush=True)
                    print("   curve:", r["curve"], flush=True)


if __name__ == "__main__":
    main()
