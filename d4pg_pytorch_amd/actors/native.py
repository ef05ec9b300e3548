"""Vectorised actor-side numerics backed by ``csrc/host_runtime.cpp``.

* :class:`OUNoise` — Ornstein-Uhlenbeck exploration (``utils/utils.py:9-34``):
  ``x <- x + theta (mu - x) + sigma N(0, I)``, ``a = clip(pi(s) + x, low, high)``,
  reset to ``mu`` at episode start; one independent process per environment.
* :class:`NStepAssembler` — n-step transitions with per-sample ``gamma**k`` and
  end-of-episode flush (``models/agent.py:89-123``), emitted directly in the
  replay row format.
* :class:`CpuActor` — tanh-MLP policy inference on the flat padded layout, so
  an explorer needs neither torch dispatch nor a parameter copy per tensor.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..parallel.ipc import host_lib, np_ptr


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class OUNoise:
    def __init__(self, num_envs: int, action_dim: int, low: float, high: float, theta: float = 0.15,
                 mu: float = 0.0, max_sigma: float = 0.3, min_sigma: float = 0.3,
                 decay_period: int = 10_000, seed: int = 0):
        self.E, self.A, self.low, self.high = num_envs, action_dim, float(low), float(high)
        self._h = host_lib().ou_create(num_envs, action_dim, theta, mu, max_sigma, min_sigma,
                                       int(decay_period), int(seed) & 0xFFFFFFFFFFFFFFFF)

    def reset(self, env: int = -1) -> None:
        host_lib().ou_reset(self._h, int(env))

    def get_action(self, action, t=0) -> np.ndarray:
        """Noisy, clipped action(s).  ``t``: step index in the episode (scalar or ``[E]``)."""
        a = _f32(action).reshape(self.E, self.A)
        tt = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.int64), (self.E,)))
        out = np.empty_like(a)
        host_lib().ou_sample(self._h, np_ptr(a), np_ptr(tt), self.low, self.high, np_ptr(out))
        return out

    @property
    def state(self) -> np.ndarray:
        import ctypes as C
        buf = (C.c_float * (self.E * self.A)).from_address(host_lib().ou_state(self._h))
        return np.frombuffer(buf, dtype=np.float32).reshape(self.E, self.A).copy()

    def __del__(self):
        try:
            if self._h:
                host_lib().ou_destroy(self._h)
                self._h = None
        except Exception:
            pass


class NStepAssembler:
    def __init__(self, num_envs: int, n: int, state_dim: int, action_dim: int, gamma: float):
        self.E, self.n, self.S, self.A = num_envs, int(n), state_dim, action_dim
        self._h = host_lib().nstep_create(num_envs, self.n, state_dim, action_dim, float(gamma))
        if not self._h:
            raise ValueError("invalid n-step assembler dimensions")
        self.row_ld = host_lib().nstep_row_ld(self._h)
        self._out = np.zeros((num_envs * self.n + num_envs, self.row_ld), dtype=np.float32)

    def reset(self, env: int = -1) -> None:
        host_lib().nstep_reset(self._h, int(env))

    def push(self, states, actions, rewards, next_states, dones, truncs=None, active=None) -> np.ndarray:
        """Feed one step of every environment; returns the rows that became complete."""
        s, a = _f32(states).reshape(self.E, self.S), _f32(actions).reshape(self.E, self.A)
        r, ns = _f32(rewards).reshape(self.E), _f32(next_states).reshape(self.E, self.S)
        d = np.ascontiguousarray(np.asarray(dones).reshape(self.E), dtype=np.uint8)
        tr = None if truncs is None else np.ascontiguousarray(np.asarray(truncs).reshape(self.E), dtype=np.uint8)
        ac = None if active is None else np.ascontiguousarray(np.asarray(active).reshape(self.E), dtype=np.uint8)
        n = host_lib().nstep_push(self._h, np_ptr(s), np_ptr(a), np_ptr(r), np_ptr(ns), np_ptr(d),
                                  None if tr is None else np_ptr(tr), None if ac is None else np_ptr(ac),
                                  np_ptr(self._out), self._out.shape[0])
        if n < 0:
            raise RuntimeError("n-step output buffer overflow")
        return self._out[:n].copy()

    def __del__(self):
        try:
            if self._h:
                host_lib().nstep_destroy(self._h)
                self._h = None
        except Exception:
            pass


class CpuActor:
    """Actor inference for CPU explorers on the flat parameter vector."""

    def __init__(self, state_dim: int, action_dim: int, hidden: int, action_scale: float = 1.0):
        from ..models.nets import NetLayout
        self.S, self.A, self.H = state_dim, action_dim, hidden
        self.layout = NetLayout.build(state_dim, hidden, action_dim)
        self.flat = np.zeros(self.layout.numel, dtype=np.float32)
        self.action_scale = float(action_scale)

    def load_flat(self, flat) -> None:
        flat = np.asarray(flat, dtype=np.float32).reshape(-1)
        assert flat.size == self.flat.size, (flat.size, self.flat.size)
        np.copyto(self.flat, flat)

    def act(self, states) -> np.ndarray:
        x = _f32(states)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        assert x.shape[1] == self.S
        out = np.empty((x.shape[0], self.A), dtype=np.float32)
        host_lib().actor_forward(np_ptr(self.flat), self.S, self.H, self.A, np_ptr(x), x.shape[0], np_ptr(out),
                                 self.action_scale)
        return out

    # reference-layout (linear1/2/3) export / import for checkpoints
    def _view(self):
        import torch
        from ..models.nets import NetView
        return NetView(torch.from_numpy(self.flat), self.layout)

    def state_dict(self):
        return self._view().state_dict()

    def load_state_dict(self, sd, strict: bool = True) -> None:
        self._view().load_state_dict(sd, strict)
