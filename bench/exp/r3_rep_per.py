"""How robustly does replicated data parallelism with prioritised replay learn the pendulum (two thread-ranks, one GPU)?
Greedy return of the target policy by tau, update budget and replay kind, several runs each (the interleaving of the two
threads differs from run to run)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_replicated_dp import _replicated_per_pendulum_run

out = []
os.makedirs("gpurun_out/r3_rep_per", exist_ok=True)
log = open("gpurun_out/r3_rep_per/runs.jsonl", "w")
for tau, updates, per in ((0.001, 20_000, 1), (0.001, 30_000, 1), (0.005, 20_000, 1), (0.001, 20_000, 0), (0.005, 20_000, 0)):
    for run in range(3):
        _, _, ret = _replicated_per_pendulum_run(tau, updates, per=per)
        out.append({"tau": tau, "updates": updates, "per": per, "run": run, "return": round(ret, 1)})
        print(json.dumps(out[-1]), flush=True)
        log.write(json.dumps(out[-1]) + "\n")
        log.flush()
