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


class StepCounter:
    """Philox stream position of a device environment (its resets, and the exploration noise of a rollout on it).

    Unpaired (the default): one slot, incremented by a one-thread launch behind every environment launch.
    Paired: two slots; a step reads the current one and a thread of the environment's own kernel leaves the value + 1
    in the other (``next_counter`` in ``csrc/common.h``), which the launches of the next step read.  No launch of its
    own; which slot is current is tracked here, so a captured sequence must end on the slot it started from
    (:meth:`rewind`).
    """

    def __init__(self, device):
        self.slots = torch.zeros(2, dtype=torch.int64, device=device)
        self.cur, self.paired = 0, False

    @property
    def ptr(self) -> int:
        """Address of the slot the next launches read."""
        return self.slots.data_ptr() + 8 * self.cur

    def fill(self, a) -> None:
        """Counter fields of an environment's argument struct; call once per environment launch."""
        a.counter = self.ptr
        if self.paired:
            a.counter_next = self.slots.data_ptr() + 8 * (self.cur ^ 1)
            self.cur ^= 1

    def rewind(self) -> None:
        """Bring the position back to slot 0 (one tiny launch after an odd number of paired steps, none otherwise)."""
        if self.cur == 1:
            base = self.slots.data_ptr()
            H._check(H.lib().d4pg_counter_copy(base, base + 8, H.stream_ptr()), "d4pg_counter_copy")
            self.cur = 0

    def value(self) -> int:
        """Synchronises."""
        return int(self.slots[self.cur].item())


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
        self.counter = StepCounter(self.device)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._seen = (0.0, 0)
        self.limit_terminal = bool(limit_terminal)      # like PendulumVec: the 200-step limit reads as done
        self.noise: Optional[torch.Tensor] = None       # [E, ld_act] exploration-noise state to zero at episode ends

    def _args(self, init: int) -> H.PendulumArgs:
        a = H.PendulumArgs()
        a.theta, a.theta_dot, a.t, a.ep_return = (self.theta.data_ptr(), self.theta_dot.data_ptr(), self.t.data_ptr(),
                                                  self.ep_return.data_ptr())
        a.stats, a.seed = self.stats.data_ptr(), self.seed
        self.counter.fill(a)
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
