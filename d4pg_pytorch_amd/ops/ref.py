"""Pure-PyTorch reference implementations ("oracles") of every learner op.

They serve three purposes:

* the numerical ground truth the HIP kernels in ``csrc/`` are tested against,
* the execution path of the CPU-only configuration (no GPU, eager torch),
* the "unfused, same hardware" baseline that ``bench.py --baseline`` times.

Each function is written from the math, in float32, and is device agnostic.
Reference behaviour is cited per function.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

# ---------------------------------------------------------------------------
# Categorical projection
# ---------------------------------------------------------------------------


def l2_project(probs: torch.Tensor, reward: torch.Tensor, done: torch.Tensor,
               gamma: torch.Tensor, v_min: float, v_max: float) -> torch.Tensor:
    """Project the distribution of ``r + g*z`` back onto the fixed support.

    Behaviour of ``models/d4pg/l2_projection.py:7-43`` (a host-side numpy loop
    over atoms there), expressed in its dense form

        m[b, i] = sum_j p[b, j] * max(0, 1 - |beta[b, j] - i|),
        beta[b, j] = (clip(r_b + g_b z_j, v_min, v_max) - v_min) / dz

    which is identical to the lower/upper split (weights ``u - beta`` and
    ``beta - l``; when ``beta`` is integral the single weight is 1).  Terminal
    rows collapse to the projection of the scalar ``clip(r_b)``.

    ``gamma`` is per-sample (``gamma**k`` of the stored transition) or a
    0-dim tensor / float for the reference's single scalar.
    """
    B, N = probs.shape
    dz = (v_max - v_min) / (N - 1)
    dev, dt = probs.device, torch.float32
    probs = probs.to(dt)
    reward = reward.to(dt).reshape(B, 1)
    done = done.to(dt).reshape(B, 1)
    gamma = torch.as_tensor(gamma, dtype=dt, device=dev)
    gamma = gamma.reshape(-1, 1) if gamma.ndim > 0 else gamma.reshape(1, 1)
    z = v_min + dz * torch.arange(N, device=dev, dtype=dt)             # [N]
    # terminal rows: every source atom maps to the reward itself (mass sums to 1)
    tz = reward + (1.0 - done) * gamma * z.reshape(1, N)                # [B, N]
    tz = tz.clamp(v_min, v_max)
    beta = (tz - v_min) / dz                                            # [B, N] in [0, N-1]
    idx = torch.arange(N, device=dev, dtype=dt).reshape(1, 1, N)        # target atom i
    w = (1.0 - (beta.unsqueeze(2) - idx).abs()).clamp_(min=0.0)         # [B, Nsrc, Ndst]
    return torch.einsum("bj,bji->bi", probs, w)


def l2_project_loop(probs, reward, done, gamma, v_min, v_max):
    """Second, independent oracle: explicit floor/ceil scatter (slow, tests only)."""
    B, N = probs.shape
    dz = (v_max - v_min) / (N - 1)
    out = torch.zeros(B, N, dtype=torch.float64)
    p = probs.double().cpu()
    r = reward.double().cpu().reshape(-1)
    d = done.double().cpu().reshape(-1)
    g = torch.as_tensor(gamma, dtype=torch.float64).cpu()
    g = g.expand(B) if g.ndim == 0 else g.reshape(-1)
    for b in range(B):
        if d[b] > 0.5:
            tz = min(v_max, max(v_min, float(r[b])))
            bj = (tz - v_min) / dz
            l, u = math.floor(bj), math.ceil(bj)
            if l == u:
                out[b, l] = 1.0
            else:
                out[b, l] = u - bj
                out[b, u] = bj - l
            continue
        for j in range(N):
            tz = min(v_max, max(v_min, float(r[b]) + float(g[b]) * (v_min + j * dz)))
            bj = (tz - v_min) / dz
            l, u = math.floor(bj), math.ceil(bj)
            if l == u:
                out[b, l] += p[b, j]
            else:
                out[b, l] += p[b, j] * (u - bj)
                out[b, u] += p[b, j] * (bj - l)
    return out.float()


# ---------------------------------------------------------------------------
# Losses
# ---------------------------------------------------------------------------


def bce_per_sample(q: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """``BCELoss(reduction='none')(q, m).mean(1)`` with torch's log clamp at -100.

    This is the reference critic loss (``d4pg.py:58,102-103``).
    """
    logq = torch.log(q).clamp(min=-100.0)
    log1mq = torch.log1p(-q).clamp(min=-100.0)
    return -(m * logq + (1.0 - m) * log1mq).mean(dim=1)


def ce_per_sample(logits: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """Categorical cross-entropy ``-sum_i m_i log softmax(logits)_i`` (the D4PG paper's loss)."""
    return -(m * torch.log_softmax(logits, dim=1)).sum(dim=1)


def d4pg_critic_loss(logits: torch.Tensor, target_proj: torch.Tensor,
                     is_weights: Optional[torch.Tensor], kind: str = "bce"
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns ``(scalar_loss, per_sample_loss, priorities)``.

    ``priorities = |per_sample| + 1e-4`` (``d4pg.py:106-109``); the scalar is
    ``mean(w * L)`` (``d4pg.py:111-114``).
    """
    if kind == "bce":
        per = bce_per_sample(torch.softmax(logits, dim=1), target_proj)
    elif kind == "ce":
        per = ce_per_sample(logits, target_proj)
    else:
        raise ValueError(kind)
    prio = per.detach().abs() + 1e-4
    weighted = per if is_weights is None else per * is_weights
    return weighted.mean(), per, prio


def d4pg_dlogits(logits: torch.Tensor, m: torch.Tensor, is_weights: Optional[torch.Tensor],
                 kind: str = "bce") -> torch.Tensor:
    """Closed-form gradient of :func:`d4pg_critic_loss` w.r.t. the logits.

    BCE through softmax: with ``g_i = dL/dq_i = (1/N) * (-(m_i/q_i) [log q_i > -100]
    + ((1-m_i)/(1-q_i)) [log(1-q_i) > -100])`` the logit gradient is
    ``q_i * (g_i - sum_k g_k q_k)``.  CE: ``q * sum(m) - m``.
    Everything is scaled by ``w_b / B``.
    """
    B, N = logits.shape
    q = torch.softmax(logits, dim=1)
    if kind == "bce":
        live0 = (torch.log(q) > -100.0).to(q.dtype)
        live1 = (torch.log1p(-q) > -100.0).to(q.dtype)
        g = (-(m / q) * live0 + ((1.0 - m) / (1.0 - q)) * live1) / N
        g = torch.nan_to_num(g, nan=0.0, posinf=0.0, neginf=0.0)
        dl = q * (g - (g * q).sum(dim=1, keepdim=True))
    else:
        dl = q * m.sum(dim=1, keepdim=True) - m
    scale = torch.full((B, 1), 1.0 / B, dtype=logits.dtype, device=logits.device)
    if is_weights is not None:
        scale = scale * is_weights.reshape(B, 1)
    return dl * scale


def expected_q(logits: torch.Tensor, v_min: float, v_max: float) -> torch.Tensor:
    """``sum_i softmax(logits)_i z_i`` (``d4pg.py:121-123``)."""
    N = logits.shape[1]
    z = torch.linspace(v_min, v_max, N, device=logits.device, dtype=logits.dtype)
    return (torch.softmax(logits, dim=1) * z).sum(dim=1)


def d3pg_td_target(reward, done, gamma, q_next, min_value=-math.inf, max_value=math.inf):
    """``clip(r + (1-d) g Q'(s', pi'(s')))`` (``d3pg.py:67-70``)."""
    y = reward.reshape(-1, 1) + (1.0 - done.reshape(-1, 1)) * torch.as_tensor(
        gamma, dtype=q_next.dtype, device=q_next.device).reshape(-1, 1) * q_next.reshape(-1, 1)
    return y.clamp(min_value, max_value)


# ---------------------------------------------------------------------------
# Optimiser / target update
# ---------------------------------------------------------------------------


def adam_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
               lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """In-place Adam with torch.optim.Adam's default arithmetic (no amsgrad / weight decay).

    ``step`` is the 1-based step count *after* this update.
    """
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def polyak_(target: torch.Tensor, online: torch.Tensor, tau: float) -> None:
    """``theta' <- (1-tau) theta' + tau theta`` (``d4pg.py:130-138``)."""
    target.mul_(1.0 - tau).add_(online, alpha=tau)


# ---------------------------------------------------------------------------
# N-step returns (vectorised over time x env; mirrors models/agent.py:89-123)
# ---------------------------------------------------------------------------


def nstep_assemble(reward: torch.Tensor, done: torch.Tensor, n: int, gamma: float,
                   truncated: Optional[torch.Tensor] = None):
    """N-step targets for a ``[T, E]`` rollout segment.

    For step ``t`` of env ``e`` the window is ``t .. min(t+n-1, episode_end)``
    where an episode ends at the first ``done`` or ``truncated`` flag at or after
    ``t`` (or at the end of the segment).  Returns ``(R, gk, next_idx, d)``:

    * ``R[t,e]   = sum_{i<k} gamma**i * r[t+i,e]``
    * ``gk[t,e]  = gamma**k``                      (k = window length actually used)
    * ``next_idx[t,e] = t+k-1``  index of the step whose *next* observation is the bootstrap state
    * ``d[t,e]   = done[t+k-1,e]``                  (truncation does not set it, so it bootstraps)
    """
    T, E = reward.shape
    if truncated is None:
        truncated = torch.zeros_like(done)
    R = torch.zeros_like(reward, dtype=torch.float32)
    gk = torch.zeros_like(R)
    nxt = torch.zeros(T, E, dtype=torch.long, device=reward.device)
    d = torch.zeros_like(R)
    end = (done > 0.5) | (truncated > 0.5)
    for e in range(E):
        for t in range(T):
            acc, g, k = 0.0, 1.0, 0
            last = t
            for i in range(n):
                if t + i >= T:
                    break
                acc += g * float(reward[t + i, e])
                g *= gamma
                k += 1
                last = t + i
                if bool(end[t + i, e]):
                    break
            R[t, e] = acc
            gk[t, e] = g
            nxt[t, e] = last
            d[t, e] = float(done[last, e] > 0.5)
    return R, gk, nxt, d


# ---------------------------------------------------------------------------
# Prioritised replay math (models/d4pg/replay_buffer.py:89-209)
# ---------------------------------------------------------------------------


def per_is_weights(p_sampled: torch.Tensor, p_total: float, p_min: float, n: int, beta: float) -> torch.Tensor:
    """``w_i = (P(i) n)^-beta / max_w`` with ``max_w = (p_min / total * n)^-beta``."""
    prob = p_sampled.double() / p_total
    w = (prob * n) ** (-beta)
    max_w = (p_min / p_total * n) ** (-beta)
    return (w / max_w).float()


def stratified_prefix_targets(total: float, batch: int, u: torch.Tensor) -> torch.Tensor:
    """One draw per equal segment of ``[0, total)`` (``replay_buffer.py:129-137``)."""
    seg = total / batch
    return (torch.arange(batch, dtype=torch.float64, device=u.device) + u.double()) * seg


# ---------------------------------------------------------------------------
# 3-layer MLPs (networks.py) on explicit weight tensors
# ---------------------------------------------------------------------------


def mlp3(x, w1, b1, w2, b2, w3, b3, final: str = "none"):
    h1 = torch.relu(torch.nn.functional.linear(x, w1, b1))
    h2 = torch.relu(torch.nn.functional.linear(h1, w2, b2))
    y = torch.nn.functional.linear(h2, w3, b3)
    if final == "tanh":
        y = torch.tanh(y)
    elif final == "softmax":
        y = torch.softmax(y, dim=1)
    return y
