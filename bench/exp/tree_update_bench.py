#!/usr/bin/env python
"""Stand-alone cost of a priority write-back (d4pg_tree_update) by batch size and number of workgroups, on a tree of
10M (default) rows with 1M filled: what the role that rides in the learner's launches costs by itself."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from d4pg_pytorch_amd.config import default_config          # noqa: E402
from d4pg_pytorch_amd.models.core import build_core          # noqa: E402
from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay     # noqa: E402


def main():
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    cfg = default_config(env="Ant-v2", model="d4pg", batch_size=256, device="cuda", replay_mem_size=cap,
                         replay_memory_prioritized=1, v_min=0.0, v_max=4000.0)
    core = build_core(cfg, "cuda:0")
    rp = HbmReplay(cfg, "cuda:0", core.dev_state)
    fill = min(cap, 1_000_000)
    for off in range(0, fill, 50_000):
        rp.insert(torch.randn(50_000, rp.row_ld, device="cuda"))
    torch.cuda.synchronize()
    out = {"capacity": cap, "filled": fill, "levels": rp.L, "us": {}}
    g = torch.Generator(device="cuda").manual_seed(1)
    for n in (256, 1024, 4096):
        idx = torch.randint(0, fill, (n,), device="cuda", generator=g)
        prio = torch.rand(n, device="cuda", generator=g) + 0.01
        for parts in (1, 2, 4, 8, 16, 32):
            if parts > 1 and (rp.L < 2 or n < 512):
                continue
            rp.tree_parts = lambda n_, p=parts: p
            for _ in range(5):
                rp.enqueue_priority_update(idx, prio, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                rp.enqueue_priority_update(idx, prio, n)
            e1.record()
            torch.cuda.synchronize()
            out["us"][f"n{n}_parts{parts}"] = round(e0.elapsed_time(e1) * 10.0, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
