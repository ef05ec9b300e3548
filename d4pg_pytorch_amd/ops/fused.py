"""Tensor-level front-ends of the HIP kernels (one function per op).

Each function accepts ordinary torch tensors.  On a CUDA (ROCm) tensor it
launches the hand-written kernel; on a CPU tensor it evaluates the reference
formulation in ``ops/ref.py`` (the CPU-only configuration has no GPU to run on).
The learner does not go through these wrappers — it enqueues the kernels on
preallocated buffers (``models/hip_core*.py``) — they exist for users who want a
single op, and for tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ref
from ..models.nets import pad4


def _H():
    from . import hip as H
    H.lib()
    return H


def _padded(x: torch.Tensor, ld: int) -> torch.Tensor:
    if x.shape[1] == ld and x.is_contiguous():
        return x
    out = torch.zeros(x.shape[0], ld, dtype=torch.float32, device=x.device)
    out[:, : x.shape[1]] = x
    return out


def d4pg_critic_loss(logits_t: torch.Tensor, logits: torch.Tensor, reward: torch.Tensor, done: torch.Tensor,
                     gamma: torch.Tensor, v_min: float, v_max: float, weights: Optional[torch.Tensor] = None,
                     kind: str = "bce") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Fused softmax + categorical projection + loss.

    Returns ``(per_sample_loss[B], priority[B], dlogits[B, N], projected_target[B, N])`` where
    ``dlogits`` is the gradient of ``mean_b(w_b L_b)`` w.r.t. the online logits."""
    B, N = logits.shape
    if not logits.is_cuda:
        m = ref.l2_project(torch.softmax(logits_t, 1), reward, done, gamma, v_min, v_max)
        _, per, prio = ref.d4pg_critic_loss(logits, m, weights, kind)
        return per, prio, ref.d4pg_dlogits(logits, m, weights, kind), m
    H = _H()
    ld = pad4(N)
    lt, lo = _padded(logits_t.float(), ld), _padded(logits.float(), ld)
    dl, proj = torch.zeros_like(lo), torch.zeros_like(lo)
    per = torch.empty(B, device=logits.device)
    prio = torch.empty(B, device=logits.device)
    g = torch.as_tensor(gamma, dtype=torch.float32, device=logits.device).expand(B).contiguous()
    a = H.D4pgLossArgs()
    a.logits_t, a.logits = lt.data_ptr(), lo.data_ptr()
    r_, d_ = reward.float().contiguous(), done.float().contiguous()
    a.reward, a.done, a.gamma = r_.data_ptr(), d_.data_ptr(), g.data_ptr()
    w_ = None if weights is None else weights.float().contiguous()
    a.weights = None if w_ is None else w_.data_ptr()
    a.dlogits, a.per_loss, a.priority, a.proj_out, a.metrics = dl.data_ptr(), per.data_ptr(), prio.data_ptr(), proj.data_ptr(), None
    a.B, a.N, a.ld, a.use_ce = B, N, ld, int(kind == "ce")
    a.v_min, a.v_max, a.fixed_gamma, a.inv_B = float(v_min), float(v_max), -1.0, 1.0 / B
    H.launch("d4pg_critic_loss", a)
    return per, prio, dl[:, :N], proj[:, :N]


def l2_project(probs_logits: torch.Tensor, reward, done, gamma, v_min: float, v_max: float,
               from_logits: bool = True) -> torch.Tensor:
    """Categorical projection of ``r + g z`` onto the support.  ``probs_logits`` are the target
    critic's logits (``from_logits=True``) or probabilities."""
    if not probs_logits.is_cuda:
        p = torch.softmax(probs_logits, 1) if from_logits else probs_logits
        return ref.l2_project(p, reward, done, gamma, v_min, v_max)
    lg = probs_logits if from_logits else torch.log(probs_logits.clamp_min(1e-45))
    return d4pg_critic_loss(lg, torch.zeros_like(lg), reward, done, gamma, v_min, v_max)[3]


def expected_q(logits: torch.Tensor, v_min: float, v_max: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(q[B], dlogits[B, N])`` with ``dlogits`` the gradient of ``-mean(q)``."""
    B, N = logits.shape
    if not logits.is_cuda:
        lg = logits.detach().clone().requires_grad_(True)
        q = ref.expected_q(lg, v_min, v_max)
        (-q.mean()).backward()
        return q.detach(), lg.grad
    H = _H()
    ld = pad4(N)
    lo = _padded(logits.float(), ld)
    dl, q = torch.zeros_like(lo), torch.empty(B, device=logits.device)
    a = H.D4pgActorLossArgs()
    a.logits, a.dlogits, a.q_out, a.metrics = lo.data_ptr(), dl.data_ptr(), q.data_ptr(), None
    a.B, a.N, a.ld, a.v_min, a.v_max, a.inv_B = B, N, ld, float(v_min), float(v_max), 1.0 / B
    H.launch("d4pg_actor_loss", a)
    return q, dl[:, :N]


def polyak_(target: torch.Tensor, online: torch.Tensor, tau: float) -> torch.Tensor:
    """In-place multi-tensor soft update of one flat buffer (numel % 4 == 0 on the GPU)."""
    if not target.is_cuda:
        ref.polyak_(target, online, tau)
        return target
    H = _H()
    H._check(H.lib().d4pg_polyak(target.data_ptr(), online.data_ptr(), target.numel(), float(tau), H.stream_ptr()),
             "d4pg_polyak")
    return target


def nstep_assemble(reward: torch.Tensor, done: torch.Tensor, n: int, gamma: float,
                   truncated: Optional[torch.Tensor] = None):
    """N-step targets of a ``[T, E]`` rollout segment: ``(R, gamma**k, next_index, done_flag)``."""
    if not reward.is_cuda:
        return ref.nstep_assemble(reward, done, n, gamma, truncated)
    H = _H()
    T, E = reward.shape
    r, d = reward.float().contiguous(), done.float().contiguous()
    tr = None if truncated is None else truncated.float().contiguous()
    R, gk, dd = torch.empty_like(r), torch.empty_like(r), torch.empty_like(r)
    nxt = torch.empty(T, E, dtype=torch.int32, device=r.device)
    a = H.NstepArgs()
    a.reward, a.done, a.trunc = r.data_ptr(), d.data_ptr(), None if tr is None else tr.data_ptr()
    a.R, a.gk, a.nxt, a.d = R.data_ptr(), gk.data_ptr(), nxt.data_ptr(), dd.data_ptr()
    a.T, a.E, a.n, a.gamma = T, E, int(n), float(gamma)
    H.launch("d4pg_nstep", a)
    return R, gk, nxt.long(), dd


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           activation: str = "none") -> torch.Tensor:
    """``act(x @ weight.T + bias)`` on the fp32 MFMA kernel (``csrc/gemm.hip``)."""
    if not x.is_cuda:
        y = torch.nn.functional.linear(x, weight, bias)
        return {"none": y, "relu": torch.relu(y), "tanh": torch.tanh(y)}[activation]
    H = _H()
    M, K = x.shape
    N = weight.shape[0]
    xp, wp = _padded(x.float(), pad4(K)), _padded(weight.float(), pad4(K))
    out = torch.empty(M, pad4(N), device=x.device)
    epi = {"none": H.EPI_NONE, "relu": H.EPI_RELU, "tanh": H.EPI_TANH}[activation]
    b = None if bias is None else bias.float().contiguous()
    H.gemm_group([H.gemm_desc(H.GEMM_NT, xp, pad4(K), wp, pad4(K), out, pad4(N), M, N, K, bias=b, epi=epi)])
    return out[:, :N]


def mlp3_forward(x, w1, b1, w2, b2, w3, b3, final: str = "none") -> torch.Tensor:
    """Three-layer MLP of the actor / critic (ReLU, ReLU, ``final`` in {none, tanh, softmax})."""
    h = linear(linear(x, w1, b1, "relu"), w2, b2, "relu")
    if final == "softmax":
        return torch.softmax(linear(h, w3, b3), dim=1)
    return linear(h, w3, b3, final)


class DeviceOUNoise:
    """OU exploration noise resident on the GPU (Philox), for batched GPU rollouts."""

    def __init__(self, num_envs: int, action_dim: int, low: float, high: float, theta: float = 0.15,
                 mu: float = 0.0, sigma: float = 0.3, seed: int = 0, device="cuda"):
        self.E, self.A, self.low, self.high = num_envs, action_dim, float(low), float(high)
        self.theta, self.mu, self.sigma, self.seed = theta, mu, sigma, int(seed) & 0xFFFFFFFFFFFFFFFF
        self.x = torch.full((num_envs, action_dim), mu, dtype=torch.float32, device=device)
        self.counter = torch.zeros(1, dtype=torch.int64, device=device)

    def reset(self, env: Optional[int] = None) -> None:
        if env is None:
            self.x.fill_(self.mu)
        else:
            self.x[env].fill_(self.mu)

    def get_action(self, action: torch.Tensor, sigma: Optional[torch.Tensor] = None) -> torch.Tensor:
        H = _H()
        act = action.float().contiguous()
        out = torch.empty_like(act)
        a = H.OuArgs()
        a.x, a.action, a.out = self.x.data_ptr(), act.data_ptr(), out.data_ptr()
        a.sigma = None if sigma is None else sigma.float().contiguous().data_ptr()
        a.counter, a.seed = self.counter.data_ptr(), self.seed
        a.n, a.A = self.E * self.A, self.A
        a.theta, a.mu, a.sigma_const, a.low, a.high = self.theta, self.mu, self.sigma, self.low, self.high
        H.launch("d4pg_ou_noise", a)
        return out
