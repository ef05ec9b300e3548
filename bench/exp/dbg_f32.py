import sys, torch
sys.path.insert(0, '.')
from tests.test_hip_f32 import _cfg, _batch
from d4pg_pytorch_amd.models.core import TorchCore, build_core
kw = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
model = sys.argv[2] if len(sys.argv) > 2 else "d4pg"
cfg = _cfg(model, use_hip_graph=False, **kw)
hipc, torc = build_core(cfg, "cuda"), TorchCore(cfg, "cuda")
print(type(hipc).__name__, "mt", hipc.mt)
b = _batch(cfg, 7)
tg, hg = {}, {}
torc.grad_sync = lambda g: tg.setdefault(len(tg), g.detach().clone())
hipc.grad_sync = lambda g: hg.setdefault(len(hg), g.detach().clone())
from d4pg_pytorch_amd.models.nets import ParamPack
old = ParamPack(hipc.layout, flat=hipc.online.flat.detach().clone())
p_t = torc.update(b); p_h = hipc.update(b).clone()
print("prio rel", ((p_h - p_t).norm() / p_t.norm()).item())
lay = hipc.layout
for net, k, nm in ((lay.critic, 0, "critic"), (lay.actor, 1, "actor")):
    for i, s in enumerate(net.layers):
        for off, n, what in ((s.w_off, s.w_numel, "W"), (s.b_off, s.out_dim, "b")):
            a_, b_ = hg[k][off - net.base: off - net.base + n], tg[k][off - net.base: off - net.base + n]
            print(nm, i, what, "rel", ((a_ - b_).norm() / b_.norm().clamp_min(1e-30)).item(), "norms", a_.norm().item(), b_.norm().item())
torch.cuda.synchronize()
for n in ("g_xin", "g_h1", "g_h2", "a_daT", "a_dh2T", "a_dh1T", "q_val", "sT", "a_h1T", "a_h2T", "c_dlT", "gc_lg"):
    t = getattr(hipc, n)
    print(n, tuple(t.shape), "abs sum", t.abs().sum().item(), "nan", torch.isnan(t).any().item())
print("q torch", torc.read_metrics(), "hip", hipc.read_metrics())

new = hipc.online
pi_old = old.actor.forward(b.state, "tanh")
z = torch.linspace(cfg["v_min"], cfg["v_max"], cfg["num_atoms"], device="cuda")
for nm, crit in (("critic_new", new.critic), ("critic_old", old.critic)):
    lg = crit.forward(torch.cat([b.state, pi_old], 1), "none")
    q = (torch.softmax(lg, 1) * z).sum(1)
    print(nm, "q ref sum", q.abs().sum().item(), "max diff to hip q_val", (q - hipc.q_val).abs().max().item())
print("pi in g_xin vs ref", (hipc.g_xin[:, hipc.S:hipc.S + hipc.A] - pi_old).abs().max().item())
from d4pg_pytorch_amd.models.nets import refresh_shadows32, unpack32
want = torch.zeros_like(hipc.shadow_online)
refresh_shadows32(hipc.online.flat.detach(), want, hipc.layout, hipc.sl)
for nm, slots in (("actor", hipc.sl.actor), ("critic", hipc.sl.critic)):
    for i, s in enumerate(slots):
        for tag, p in (("f", s.f), ("b", s.b)):
            if p is None: continue
            a_, b_ = hipc.shadow_online[p.off:p.off + p.numel], want[p.off:p.off + p.numel]
            print(nm, i, tag, "shadow max diff", (a_ - b_).abs().max().item(), "nonzero", (a_ != 0).sum().item(), (b_ != 0).sum().item())
