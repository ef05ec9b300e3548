"""Does the pendulum run collapse to a constant-torque policy when the explorers deliver many more environment steps per
update than the reference's ratio?  In-process device explorers (deterministic), D4PG fp32, greedy target-policy return
after the budget, by environments (x 32 steps per 32 updates = environment steps per update), tau and seed."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d4pg_pytorch_amd.utils.learning import train_on_device

os.makedirs("gpurun_out/r3_ratio", exist_ok=True)
log = open("gpurun_out/r3_ratio/runs.jsonl", "w")
for envs in (8, 40, 64):
    for tau in (0.001, 0.005):
        for seed in (103, 104, 105):
            r = train_on_device("d4pg", "fp32", updates=30_000, seed=seed, eval_every=5000, envs=envs, tau=tau)
            rec = {"envs": envs, "tau": tau, "seed": seed, "final": r["final_return"], "best": r["best_return"], "curve": r["curve"]}
            print(json.dumps(rec), flush=True)
            log.write(json.dumps(rec) + "\n")
            log.flush()
