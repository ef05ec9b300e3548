"""The built-in pendulum (``envs/pendulum.py``, the dynamics of gym's ``Pendulum-v0``) resident on the GPU.

``E`` environments advance with one kernel launch (``pendulum_step_kernel`` in ``csrc/rollout.hip``): state,
observations, rewards and the time-limit resets (Philox stream) never leave the device, so a policy that also
runs there (``actors/gpu_actor.py``) steps thousands of environments without a host round trip.  Rewards come
out divided by 100 like the reference's wrapper (``env/pendulum.py:13-14``); the unscaled episode returns are
accumulated on the device and read back on demand.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ..ops import hip as H


class DevicePendulum:
    state_dim, action_dim = 3, 1
    reward_scale = 100.0

    def __init__(self, num_envs: int, device="cuda", seed: int = 0, time_limit: int = 200, limit_terminal: bool = True):
        H.lib()
        self.E, self.device, self.time_limit = int(num_envs), torch.device(device), int(time_limit)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.device)   # noqa: E731
        self.theta, self.theta_dot, self.ep_return = z(self.E), z(self.E), z(self.E)
        self.t = z(self.E, dt=torch.int32)
        self.stats = z(2, dt=torch.float64)
        self.counter = z(1, dt=torch.int64)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._seen = (0.0, 0)
        self.limit_terminal = bool(limit_terminal)      # like PendulumVec: the 200-step limit reads as done
        self.noise: Optional[torch.Tensor] = None       # [E, ld_act] exploration-noise state to zero at episode ends

    def _args(self, init: int) -> H.PendulumArgs:
        a = H.PendulumArgs()
        a.theta, a.theta_dot, a.t, a.ep_return = (self.theta.data_ptr(), self.theta_dot.data_ptr(), self.t.data_ptr(),
                                                  self.ep_return.data_ptr())
        a.stats, a.counter, a.seed = self.stats.data_ptr(), self.counter.data_ptr(), self.seed
        a.E, a.time_limit, a.init, a.reward_scale = self.E, self.time_limit, init, self.reward_scale
        a.limit_terminal = int(self.limit_terminal)
        if self.noise is not None:
            a.noise, a.ld_act = self.noise.data_ptr(), self.noise.shape[1]
        return a

    def reset(self, obs_out: torch.Tensor) -> torch.Tensor:
        """Reset every environment; writes the first observations ``[E, 3]``.  Enqueue only."""
        a = self._args(1)
        a.obs = obs_out.data_ptr()
        H.launch("d4pg_pendulum_step", a)
        return obs_out

    def step(self, action: torch.Tensor, obs_out: torch.Tensor, next_obs_out: torch.Tensor, reward_out: torch.Tensor,
             done_out: torch.Tensor, trunc_out: torch.Tensor) -> None:
        """One step of all environments (enqueue only).  ``action [E, ld]`` (torque in column 0);
        ``next_obs_out`` receives the observation that followed the action, ``obs_out`` what the policy sees next
        (the first observation of a new episode where the time limit ended one)."""
        a = self._args(0)
        a.action, a.ld_act = action.data_ptr(), action.shape[1]
        a.obs, a.next_obs = obs_out.data_ptr(), next_obs_out.data_ptr()
        a.reward, a.done, a.trunc = reward_out.data_ptr(), done_out.data_ptr(), trunc_out.data_ptr()
        H.launch("d4pg_pendulum_step", a)

    def episode_stats(self) -> Tuple[Optional[float], int]:
        """(mean unscaled return, count) of the episodes finished since the last call; synchronises.  The device
        totals only ever grow (a rollout may be stepping on another stream), the difference is taken here."""
        s = self.stats.cpu()
        total, count = float(s[0].item()), int(s[1].item())
        d_total, d_count = total - self._seen[0], count - self._seen[1]
        self._seen = (total, count)
        return (d_total / d_count if d_count else None), d_count
