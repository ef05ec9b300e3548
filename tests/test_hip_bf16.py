"""Fused row-local bf16 pipeline (csrc/fused.hip) against the fp32 torch oracle (GPU only)."""
import numpy as np
import pytest
import torch

from d4pg_pytorch_amd.config import default_config

pytestmark = pytest.mark.gpu


def _cfg(model, **kw):
    base = dict(env="LunarLanderContinuous-v2", model=model, batch_size=256, dense_size=400,
                v_min=-20.0, v_max=20.0, device="cuda", num_steps_train=1000, random_seed=11,
                compute_dtype="bf16")
    base.update(kw)
    return default_config(**base)


def _batch(cfg, seed, device="cuda"):
    from d4pg_pytorch_amd.models.core import Batch
    g = torch.Generator().manual_seed(seed)
    B, S, A = cfg["batch_size"], cfg["state_dim"], cfg["action_dim"]
    return Batch(state=torch.randn(B, S, generator=g), action=torch.rand(B, A, generator=g) * 2 - 1,
                 reward=torch.randn(B, generator=g), next_state=torch.randn(B, S, generator=g),
                 done=(torch.rand(B, generator=g) < 0.1).float(),
                 gamma=torch.full((B,), cfg["discount_rate"] ** cfg["n_step_returns"]),
                 weights=torch.rand(B, generator=g) + 0.5, indices=torch.arange(B)).to(device)


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


CASES = [("d4pg", {}), ("d4pg", {"critic_loss": "ce", "dense_size": 128}), ("d3pg", {}),
         ("d4pg", {"env": "Ant-v2", "batch_size": 128, "v_min": 0.0, "v_max": 40.0}),
         ("ddpg", {"env": "Pendulum-v0", "dense_size": 100, "batch_size": 64}),
         ("d4pg", {"env": "Walker2d-v2", "batch_size": 2048, "dense_size": 256}),      # 32-row tiles
         ("d4pg", {"env": "BipedalWalker-v2", "dense_size": 72, "num_atoms": 101})]   # runtime-K path, 2 atom slots


@pytest.mark.parametrize("model,kw", CASES)
def test_bf16_gradients_match_fp32_autograd(model, kw):
    from d4pg_pytorch_amd.models.core import TorchCore, build_core
    from d4pg_pytorch_amd.models.hip_core_bf16 import HipCoreBf16
    cfg = _cfg(model, use_hip_graph=False, **kw)
    hipc, torc = build_core(cfg, "cuda"), TorchCore(cfg, "cuda")
    assert isinstance(hipc, HipCoreBf16)
    b = _batch(cfg, 7)
    tg, hg = {}, {}
    torc.grad_sync = lambda g: tg.setdefault(len(tg), g.detach().clone())
    hipc.grad_sync = lambda g: hg.setdefault(len(hg), g.detach().clone())
    p_t = torc.update(b)
    p_h = hipc.update(b).clone()
    torch.testing.assert_close(p_h, p_t, rtol=3e-2, atol=1e-4)
    for k, name in ((0, "critic"), (1, "actor")):
        cos = torch.nn.functional.cosine_similarity(hg[k], tg[k], dim=0).item()
        assert cos > 0.999, (name, cos)
        assert _rel(hg[k], tg[k]) < 6e-2, (name, _rel(hg[k], tg[k]))
    # per-layer check so that a wrong small tensor cannot hide behind a large one
    lay = hipc.layout
    for net, k in ((lay.critic, 0), (lay.actor, 1)):
        for s in net.layers:
            for off, n in ((s.w_off, s.w_numel), (s.b_off, s.out_dim)):
                a_, b_ = hg[k][off - net.base: off - net.base + n], tg[k][off - net.base: off - net.base + n]
                if b_.norm() > 1e-8 and n >= 8:   # a 1-element sum of cancelling terms is bf16 noise
                    assert _rel(a_, b_) < 0.12, (net.base, off, _rel(a_, b_))   # deepest layers see 5 bf16 roundings


@pytest.mark.parametrize("graph", [False, True])
def test_bf16_core_trains_like_fp32_and_keeps_shadows_in_sync(graph):
    from d4pg_pytorch_amd.models.core import TorchCore, build_core
    from d4pg_pytorch_amd.models.nets import refresh_shadows
    cfg = _cfg("d4pg", use_hip_graph=graph)
    hipc, torc = build_core(cfg, "cuda"), TorchCore(cfg, "cuda")
    for it in range(10):
        b = _batch(cfg, 100 + it)
        hipc.update(b)
        torc.update(b)
    torch.cuda.synchronize()
    # Adam normalises the step, so the distance to the eager fp32 trajectory is measured in Adam steps
    # (lr = 5e-4): a parameter whose gradient sign bf16 rounding flips drifts by up to 2 lr per update.
    # Ten updates move a parameter by at most 10 lr; the bf16 pipeline must stay well inside that, on
    # average and for all but 1 % of the parameters.
    lr = cfg["critic_learning_rate"]
    diff = (hipc.online.flat - torc.online.flat.detach()).abs() / lr
    moved = (torc.online.flat.detach() - build_core(cfg, "cuda").online.flat).abs() / lr
    print(f"bf16 vs eager after 10 updates, in Adam steps: mean {diff.mean().item():.3f}, "
          f"p99 {diff.quantile(0.99).item():.3f}, max {diff.max().item():.3f}; eager moved mean {moved.mean().item():.3f}")
    assert diff.mean().item() < 0.15 * moved.mean().item()
    assert diff.quantile(0.99).item() < 2.0
    # and no parameter anywhere near the worst case (opposite directions on every one of the ten updates: 20 steps)
    assert diff.max().item() < 10.0, diff.max().item()
    mh, mt = hipc.read_metrics(), torc.read_metrics()
    assert mh["count"] == 10
    assert abs(mh["value_loss"] - mt["value_loss"]) < 2e-2 * max(1.0, abs(mt["value_loss"]))
    assert abs(mh["policy_loss"] - mt["policy_loss"]) < 2e-2 * max(1.0, abs(mt["policy_loss"])) + 1e-2
    # shadows written by the optimiser kernel == a rebuild from the fp32 masters
    for flat, sh in ((hipc.online.flat, hipc.shadow_online), (hipc.target.flat, hipc.shadow_target)):
        want = torch.zeros_like(sh)
        refresh_shadows(flat.detach(), want, hipc.layout, hipc.sl)
        if sh is hipc.shadow_target:
            # target keeps forward shadows only; compare those ranges
            for slots in (hipc.sl.actor, hipc.sl.critic):
                for s in slots:
                    assert torch.equal(sh[s.f_off: s.f_off + s.f_rows * s.f_ld], want[s.f_off: s.f_off + s.f_rows * s.f_ld])
        else:
            assert torch.equal(sh, want)


@pytest.mark.parametrize("per", [False, True])
def test_bf16_update_from_replay_and_resume(per, tmp_path):
    from d4pg_pytorch_amd.models.core import build_core
    from d4pg_pytorch_amd.replay.hbm_replay import HbmReplay
    cfg = _cfg("d4pg", env="BipedalWalker-v2", replay_mem_size=50_000, replay_memory_prioritized=int(per))
    core = build_core(cfg, "cuda")
    rp = HbmReplay(cfg, "cuda", core.dev_state)
    core.attach_replay(rp)
    g = torch.Generator().manual_seed(1)
    rows = torch.randn(20_000, rp.row_ld, generator=g)
    S, A = rp.S, rp.A
    rows[:, S + A + 1] = (torch.rand(20_000, generator=g) < 0.05).float()
    rows[:, S + A + 2] = 0.99 ** 5
    rp.insert(rows.cuda())
    for _ in range(30):
        prio = core.update_from_replay()
    torch.cuda.synchronize()
    m = core.read_metrics()
    assert m["count"] == 30 and np.isfinite(m["value_loss"]) and np.isfinite(m["policy_loss"])
    assert torch.isfinite(core.online.flat).all() and torch.all(prio > 0)
    bundle = core.state_bundle()
    other = build_core(_cfg("d4pg", env="BipedalWalker-v2", random_seed=3), "cuda")
    other.load_state_bundle(bundle)
    assert torch.equal(other.shadow_online, core.shadow_online)
    b = _batch(cfg, 5)
    core.update(b)
    other.update(b)
    torch.testing.assert_close(other.online.flat, core.online.flat, rtol=0, atol=0)
