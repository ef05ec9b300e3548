#!/usr/bin/env python
"""Which knobs make D4PG learn the pendulum quickly?  Fused fp32 core, one seed per variant."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d4pg_pytorch_amd.utils.learning import train_pendulum_on_device   # noqa: E402

VARIANTS = {
    "base_40k": dict(updates=40000),
    "ce_40k": dict(updates=40000, critic_loss="ce"),
    "tau005_40k": dict(updates=40000, tau=0.005),
    "ce_tau005_40k": dict(updates=40000, critic_loss="ce", tau=0.005),
    "nstep1_40k": dict(updates=40000, n_step_returns=1),
    "envs32_40k": dict(updates=40000, envs=32),
    "vmin5_40k": dict(updates=40000, v_min=-5.0),
    "fixed_gamma_40k": dict(updates=40000, use_per_sample_gamma=False),
}
only = sys.argv[1:]
for name, kw in VARIANTS.items():
    if only and name not in only:
        continue
    kw = dict(kw)
    r = train_pendulum_on_device("d4pg", "fp32", seed=100, eval_every=2000, **kw)
    print(name, json.dumps({k: r[k] for k in ("last5_mean_return", "best_return", "first_eval_step_above_-300", "wall_s",
                                               "value_loss", "policy_loss")}), [c[1] for c in r["curve"]], flush=True)
