"""In-process learning runs on one GPU: device-resident explorers feed the HBM replay, any learner core
trains from it, the greedy target policy is evaluated on a held-out batch of device environments.

One process, no host hops in the loop, so a 20 000-update run takes seconds with the fused HIP cores: cheap
enough to compare cores (eager PyTorch fp32 / fused fp32 / fused bf16) over several seeds, which is how a
compute path is shown not to change what the agent learns (``bench/learn_parity.py``, ``tests/test_learning_gpu.py``).

The loop is the reference's data flow (``models/agent.py:51-154`` explorers with OU noise and n-step rows,
``models/d4pg/d4pg.py:36-171`` learner, the exploiter playing the TARGET policy greedily) at a fixed ratio of
environment steps to updates.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional

import torch

from ..config import default_config


#: value support in the units the learner sees, per environment (pendulum: reward / 100 with gamma 0.99;
#: cart-pole stand-in: +1 per step; lunar lander: reward / 100, episode returns between about -4 and +3)
SUPPORT = {"Pendulum-v0": (-20.0, 0.0), "InvertedPendulum-v2": (0.0, 100.0), "LunarLanderContinuous-v2": (-5.0, 5.0)}


def pendulum_config(model: str, core: str, seed: int, env: str = "Pendulum-v0", **overrides) -> Dict:
    v_min, v_max = SUPPORT.get(env, (-100.0, 100.0))
    cfg = dict(env=env, model=model, batch_size=256, dense_size=400, device="cuda", n_step_returns=5,
               discount_rate=0.99, tau=0.001, critic_learning_rate=5e-4, actor_learning_rate=5e-4,
               replay_mem_size=1_000_000, replay_memory_prioritized=0, random_seed=seed,
               v_min=v_min, v_max=v_max,
               compute_dtype="bf16" if core == "bf16" else "fp32",
               fused_learner="torch" if core == "torch" else "auto")
    cfg.update(overrides)
    return default_config(**cfg)


class _TorchSampler:
    """Uniform batches for the eager core out of the same HBM rows the HIP cores sample with their kernels."""

    def __init__(self, replay, batch: int, seed: int):
        self.rp, self.B = replay, batch
        self.gen = torch.Generator(device=replay.device).manual_seed(seed)

    def __call__(self):
        from ..models.core import Batch
        rp = self.rp
        idx = torch.randint(0, rp.size, (self.B,), generator=self.gen, device=rp.device)
        r = rp.rows[idx]
        S, A = rp.S, rp.A
        return Batch(state=r[:, :S], action=r[:, S:S + A], reward=r[:, S + A], next_state=r[:, S + A + 3: 2 * S + A + 3],
                     done=r[:, S + A + 1], gamma=r[:, S + A + 2], weights=None, indices=idx)


class GreedyEvaluator:
    """Mean undiscounted return of the greedy policy over the first episode of each of ``envs`` fresh device
    environments (pendulum: 200 steps each)."""

    def __init__(self, cfg: Dict, envs: int, device, seed: int):
        from ..actors.gpu_actor import PackedActor32
        from ..envs.device_envs import make_device_env
        self.device = torch.device(device)
        self.env = make_device_env(dict(cfg, env_backend="native"), envs, device, seed)      # (never gym: evaluation on the device)
        S, A = self.env.state_dim, self.env.action_dim
        self.actor = PackedActor32(S, A, int(cfg["dense_size"]), device)
        self.scale = float(cfg["action_high"]) if cfg.get("scale_action") else 1.0
        z = lambda *s: torch.zeros(*s, device=self.device)      # noqa: E731
        self.obs, self.nobs, self.act = z(envs, S), z(envs, S), z(envs, 4 * ((A + 3) // 4))
        self.rew, self.done, self.trunc = z(envs), z(envs), z(envs)
        self.first = z(envs)         # return of the first episode of every environment
        self.alive = z(envs)

    def __call__(self, flat_ptr: int, actor_base: int) -> float:
        from ..actors.gpu_actor import PackedActor32
        from ..ops import hip as H
        self.actor.pack_from(flat_ptr, actor_base)
        self.env.reset(self.obs)
        self.first.zero_()
        self.alive.fill_(1.0)
        scale = float(getattr(self.env, "reward_scale", 1.0))
        for _ in range(self.env.time_limit):
            self.actor.infer(self.actor.infer_args(self.obs, self.act, self.env.E, self.scale), self.env.E)
            self.env.step(self.act, self.obs, self.nobs, self.rew, self.done, self.trunc)
            self.first.add_(self.rew * self.alive, alpha=scale)          # rewards of the first episode only
            self.alive.mul_(1.0 - torch.clamp(self.done + self.trunc, max=1.0))
        return float(self.first.mean().item())


def train_on_device(model: str = "d4pg", core: str = "fp32", updates: int = 20_000, seed: int = 0,
                    eval_every: int = 1000, envs: int = 8, segment: int = 32, eval_envs: int = 64,
                    warmup_segments: int = 8, device="cuda", env: str = "Pendulum-v0", return_actor: bool = False,
                    overlap: bool = False, **overrides) -> Dict:
    """Train on a device-resident environment (the pendulum by default; any id ``envs/device_envs.py::make_device_env``
    knows: the lunar lander, the articulated-body tasks, ...); returns the evaluation curve and summary numbers.

    ``core``: ``"torch"`` (eager PyTorch fp32, the oracle), ``"fp32"`` (fused HIP, default compute path) or
    ``"bf16"`` (fused HIP, bf16 MFMA operands).  ``envs * segment`` environment steps are collected per 32
    updates (8 per update by default).  ``overlap``: the next segment runs on the rollout's stream BESIDE the 32 updates
    (its policy is then 32 updates old, as an explorer process's is between two weight publications) instead of ahead
    of them; the learner's own loop (``models/learner.py``) works this way."""
    from ..actors.gpu_rollout import DeviceRollout
    from ..envs.device_envs import make_device_env
    from ..models.core import build_core
    from ..replay.hbm_replay import HbmReplay
    cfg = pendulum_config(model, core, seed, env, **overrides)
    torch.manual_seed(seed)
    c = build_core(cfg, device, seed=seed)
    rp = HbmReplay(cfg, device, c.dev_state) if hasattr(c, "dev_state") else None
    if rp is None:          # the eager core keeps no device state block: give the replay its own
        from ..ops import hip as H
        st = torch.zeros(H.C.sizeof(H.DevStateMirror) // 4, dtype=torch.float32, device=device)
        rp = HbmReplay(cfg, device, st)
    fused = hasattr(c, "attach_replay")
    if fused:
        c.attach_replay(rp)
    sampler = None if fused else _TorchSampler(rp, int(cfg["batch_size"]), seed)
    ro = DeviceRollout(cfg, make_device_env(dict(cfg, env_backend="native"), envs, device, seed + 1), rp, device, segment=segment, seed=seed + 2)
    ev = GreedyEvaluator(cfg, eval_envs, device, seed + 3)
    base_a = c.layout.actor.base
    for _ in range(warmup_segments):            # random-ish data before the first update (min_replay_size)
        ro.refresh_weights(c.online.flat.data_ptr(), base_a)
        ro.collect()
    curve: List = []
    t0 = time.time()
    step = 0
    next_eval = 0
    while step < updates:
        if step >= next_eval:
            curve.append((step, ev(c.target.flat.data_ptr(), base_a)))
            next_eval += eval_every
        if overlap:
            ro.collect_overlapped(c.online.flat.data_ptr(), base_a)      # rows of the last segment in, the next one off
        else:
            ro.refresh_weights(c.online.flat.data_ptr(), base_a)
            ro.collect()
        n = min(32, updates - step)
        if fused:
            done = 0
            while done < n:
                done += c.update_from_replay_many(n - done)
        else:
            for _ in range(n):
                c.update(sampler())
        step += n
    ro.drain()
    curve.append((step, ev(c.target.flat.data_ptr(), base_a)))
    torch.cuda.synchronize()
    wall = time.time() - t0
    rets = [r for _, r in curve]
    first: Optional[int] = next((s for s, r in curve if r > -300.0), None)
    m = c.read_metrics()
    extra = {"actor_flat": c.target.actor_flat().detach().cpu().numpy()} if return_actor else {}      # the TARGET policy's weights
    return {**extra, "env": env, "model": model, "core": core, "core_class": type(c).__name__, "seed": seed, "updates": step,
            "env_steps": ro.env_steps, "wall_s": round(wall, 1), "curve": [(s, round(r, 1)) for s, r in curve],
            "initial_return": round(rets[0], 1), "final_return": round(rets[-1], 1),
            "last5_mean_return": round(sum(rets[-5:]) / len(rets[-5:]), 1), "best_return": round(max(rets), 1),
            "first_eval_step_above_-300": first, "value_loss": m.get("value_loss"), "policy_loss": m.get("policy_loss")}


#: the name this function had while the pendulum was the only device environment
train_pendulum_on_device = train_on_device
