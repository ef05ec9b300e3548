#!/usr/bin/env python
"""Per-layer timing inside the fused phase kernels (diagnostic build only).

    D4PG_STAMPS=1 python -m d4pg_pytorch_amd.build --force
    python bench/stamps.py [batch] [bf16|fp32]

Workgroup 0 of ``critic_phase_kernel`` / ``actor_phase_kernel`` stamps the 100 MHz wall
clock at every layer seam; this script runs a few updates and prints the deltas in µs.
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import synthetic_rows  # noqa: E402
from d4pg_pytorch_amd.config import default_config  # noqa: E402
from d4pg_pytorch_amd.models.core import build_core  # noqa: E402
from d4pg_pytorch_amd.ops import hip as H  # noqa: E402
from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay  # noqa: E402

CRITIC = ["zero", "load", "ta1", "ta2", "ta3", "tc1", "tc2", "tc3+cf1", "cf2", "cf3", "loss", "cb3", "cb2"]
ACTOR = ["prologue", "cf1", "cf2", "cf3", "loss", "cb3", "cb2", "cb1", "ab3", "ab2"]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    # D4PG_STAMP_ENV / D4PG_STAMP_MFMA / D4PG_STAMP_OCC: another shape / the other set of fp32 phase kernels / workgroups per CU
    cfg = default_config(env=os.environ.get("D4PG_STAMP_ENV", "LunarLanderContinuous-v2"), model="d4pg", batch_size=batch,
                         device="cuda", v_min=-200.0, v_max=200.0, compute_dtype=dtype, replay_mem_size=100_000,
                         phase_mfma=os.environ.get("D4PG_STAMP_MFMA", "auto"),
                         phase_occupancy=os.environ.get("D4PG_STAMP_OCC", "auto"))
    core = build_core(cfg, "cuda")
    rp = HbmReplay(cfg, "cuda", core.dev_state)
    core.attach_replay(rp)
    rp.insert(synthetic_rows(50_000, rp.S, rp.A, rp.row_ld, 0.95, "cuda", 1))
    fn = H.lib().d4pg_read_stamps if dtype == "bf16" else H.lib().d4pg_read_stamps32
    acc = None
    n = 0
    for it in range(60):
        # steady state: a burst of back-to-back updates, the stamps are those of the last one (a launch
        # after an idle gap runs at a lower clock with cold instruction / scalar caches and reads 2x)
        done = 0
        while done < 64:
            done += core.update_from_replay_many(64 - done)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        assert fn(buf) == 0
        if dtype != "bf16":
            # the span stamps are running minima / maxima: re-arm them before the next burst
            H.lib().d4pg_reset_stamps32()
        if it < 10:
            continue
        v = list(buf)
        rows = []
        for k, names in ((0, CRITIC), (1, ACTOR)):
            t = [v[k * 32 + 31]] + [v[k * 32 + i] for i in range(len(names))]
            if k == 0 and t[8] < t[7]:     # the online critic's layers were made ahead: stamps 7 and 8 not taken
                t[8] = t[9] = t[7]         # "tc3+cf1" and "cf2" read 0, "cf3" is the target critic's last layer
            rows.append([(t[i + 1] - t[i]) / 100.0 for i in range(len(names))])
        acc = rows if acc is None else [[a + b for a, b in zip(ra, rb)] for ra, rb in zip(acc, rows)]
        n += 1
    out = {}
    if dtype != "bf16":
        v = v + [0] * 64
    out["loss_detail_last_iter"] = {"cf3_barrier_to_row0_start": (v[16] - v[9]) / 100.0, "row0_math": (v[17] - v[16]) / 100.0,
                                    "row0_stores": (v[18] - v[17]) / 100.0, "row1_total": (v[19] - v[18]) / 100.0,
                                    "row1_end_to_barrier": (v[10] - v[19]) / 100.0}
    # weight-gradient kernel (the actor's: last launch of the update), workgroup 0 / wave 0
    out["wgrad_last_iter"] = {"entry_to_state_prefetch_issued": (v[32 + 21] - v[32 + 20]) / 100.0,
                              "stash_loads_and_mfma": (v[32 + 22] - v[32 + 21]) / 100.0,
                              "lds_write_and_barrier": (v[32 + 23] - v[32 + 22]) / 100.0,
                              "reduce_adam_stores_issued": (v[32 + 24] - v[32 + 23]) / 100.0,
                              "actor_phase_end_to_wgrad_entry": (v[32 + 20] - v[32 + 9]) / 100.0}
    if dtype != "bf16":
        # inside two layers of the critic phase (workgroup 0, wave 0): entry -> own MFMAs ahead of the prefetch of
        # the next layer -> prefetch issued -> tail MFMAs + epilogue -> barrier passed
        for name, b, after, before in (("ta1_narrow", 13, 2, 1), ("ta2_hidden", 26, 3, 2)):
            out["inside_" + name + "_last_iter"] = {
                "barrier_to_entry": (v[b] - v[before]) / 100.0, "mfma_ahead_of_pre": (v[b + 1] - v[b]) / 100.0,
                "pre_issue": (v[b + 2] - v[b + 1]) / 100.0, "tail_and_epilogue": (v[b + 3] - v[b + 2]) / 100.0,
                "to_barrier_passed": (v[after] - v[b + 3]) / 100.0}
        base = v[31]      # workgroup 0 entry
        out["critic_launch_spans_last_iter_us_rel_wg0_entry"] = {
            "critic_role_first_start": (v[20] - base) / 100.0, "critic_role_last_end": (v[21] - base) / 100.0,
            "actor_fwd_role_first_start": (v[22] - base) / 100.0, "actor_fwd_role_last_end": (v[23] - base) / 100.0,
            "warmers_last_end": (v[25] - base) / 100.0}
    for names, r, key in ((CRITIC, acc[0], "critic_phase"), (ACTOR, acc[1], "actor_phase")):
        out[key] = {nm: round(x / n, 2) for nm, x in zip(names, r)}
        out[key]["total_us"] = round(sum(r) / n, 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
