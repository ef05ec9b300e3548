#!/usr/bin/env python
"""Learning-curve check on the built-in Pendulum: trains with the full multi-process
topology and prints the exploiter's (greedy, target policy) episode rewards.

    python bench/learn_pendulum.py --model d4pg --dtype fp32 --updates 30000 [--device cuda]

Prints one JSON line: reward quartiles over the run and the mean of the last 20 episodes.
A random policy scores about -1200 .. -1500 per 200-step episode; a good one -150 .. -250.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d4pg_pytorch_amd.engine import load_engine            # noqa: E402
from d4pg_pytorch_amd.utils.logger import read_events      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="d4pg")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--updates", type=int, default=30000)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--ratio", type=float, default=1.0, help="max learner updates per environment step")
    ap.add_argument("--per", type=int, default=0)
    ap.add_argument("--out", default="/tmp/learn_pendulum")
    ap.add_argument("--seed", type=int, default=2019)
    args = ap.parse_args()
    cfg = dict(env="Pendulum-v0", model=args.model, state_dim=3, action_dim=1, action_low=-2, action_high=2,
               num_agents=args.agents, batch_size=256, num_steps_train=args.updates, max_ep_length=1000,
               replay_mem_size=1_000_000, n_step_returns=5 if args.model == "d4pg" else 1,
               discount_rate=0.99, device=args.device, agent_device="cpu", dense_size=400,
               v_min=-20.0, v_max=0.0,          # support in the units the learner sees (reward / 100, gamma 0.99)
               tau=0.001, critic_learning_rate=5e-4, actor_learning_rate=5e-4,
               replay_memory_prioritized=args.per, compute_dtype=args.dtype, updates_per_env_step=args.ratio,
               min_replay_size=1000, results_path=args.out, random_seed=args.seed, num_episode_save=10_000,
               save_reward_threshold=1e9)
    t0 = time.time()
    engine = load_engine(cfg)
    exp = engine.train()
    wall = time.time() - t0
    ev = [e for e in read_events(os.path.join(exp, "agent-0")) if e[2] == "agent/reward"]
    rewards = np.asarray([e[3] for e in ev])
    steps = np.asarray([e[1] for e in ev])
    q = len(rewards) // 4
    out = {"model": args.model, "dtype": args.dtype, "device": args.device, "updates": engine.final_update_step,
           "env_steps": engine.final_env_steps, "episodes_exploiter": int(len(rewards)), "wall_s": round(wall, 1),
           "reward_quarters_mean": [round(float(rewards[i * q:(i + 1) * q].mean()), 1) for i in range(4)] if q else [],
           "reward_last20_mean": round(float(rewards[-20:].mean()), 1) if len(rewards) else None,
           "reward_best": round(float(rewards.max()), 1) if len(rewards) else None,
           "first_step_above_-300": int(steps[np.argmax(rewards > -300)]) if (rewards > -300).any() else None}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
