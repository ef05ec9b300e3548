"""On-device pendulum learning with 4096-row batches (the 16x16x4 kernels, two workgroups per CU, critic's pass made
ahead) beside the same run at the usual 256 rows: does the large-batch path learn like the other one."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d4pg_pytorch_amd.utils.learning import train_on_device

for batch, occ, envs in ((256, "auto", 8), (4096, "auto", 128), (4096, "1", 128)):
    for seed in (0, 1):
        r = train_on_device("d4pg", "fp32", updates=30_000, seed=seed, eval_every=5000, envs=envs, batch_size=batch,
                            phase_occupancy=occ, tau=0.005, replay_mem_size=2_000_000)
        print(json.dumps({"batch": batch, "phase_occupancy": occ, "seed": seed,
                          "curve": [(s, round(x, 1)) for s, x in r["curve"]], "wall_s": round(r["wall_s"], 1)}), flush=True)
