"""Per-layer relative error (units of 1e-4) of the fused fp32 gradients against autograd, Ant shape, three seeds:
critic W1, b1, W2, b2, W3, b3, then the actor's.  Shows the signature of a unit on the other side of its ReLU than in
torch's run (seeds 7 and 8: 2 - 3e-4 on the actor's first layer, less on the layers after it; seed 9: none).

    python bench/exp/r3_remk_err.py [batch] [phase_occupancy]

With the patch of notes/rem_tile_in_pass_experiment.patch applied, D4PG_NO_REMK=1 switches its K-split off (A/B)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_hip_f32 import _cfg, _batch
from d4pg_pytorch_amd.models.core import TorchCore, build_core

kw = {"env": "Ant-v2", "batch_size": int(sys.argv[1]) if len(sys.argv) > 1 else 4096}
if len(sys.argv) > 2:
    kw["phase_occupancy"] = sys.argv[2]
cfg = _cfg("d4pg", use_hip_graph=False, **kw)
for seed in (7, 8, 9):
    hipc, torc = build_core(cfg, "cuda"), TorchCore(cfg, "cuda")
    b = _batch(cfg, seed)
    tg, hg = {}, {}
    torc.grad_sync = lambda g: tg.setdefault(len(tg), g.detach().clone())
    hipc.grad_sync = lambda g: hg.setdefault(len(hg), g.detach().clone())
    torc.update(b); hipc.update(b)
    lay = hipc.layout
    out = []
    for net, k in ((lay.critic, 0), (lay.actor, 1)):
        for s in net.layers:
            for off, n in ((s.w_off, s.w_numel), (s.b_off, s.out_dim)):
                a_, b_ = hg[k][off - net.base: off - net.base + n], tg[k][off - net.base: off - net.base + n]
                out.append(round(((a_ - b_).norm() / b_.norm()).item() * 1e4, 2))
    print("lean" if hipc.lean else "fat", "remk" if not os.environ.get("D4PG_NO_REMK") else "whole", seed, out)
