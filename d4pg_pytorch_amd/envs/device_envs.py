"""More environments resident on the GPU, next to ``envs/device_pendulum.py`` (same interface: ``reset``,
``step``, ``episode_stats``, fields ``E``, ``state_dim``, ``action_dim``, ``time_limit``, ``noise``).

* :class:`DeviceCartPole` — ``envs/synthetic.py::CartPoleContinuousVec`` (the stand-in with the shape of
  ``InvertedPendulum-v2``) as one kernel per step: an environment with TERMINAL states, so the device-side
  n-step assembly sees ``done`` flags that stop the bootstrap, not only time limits.
* :class:`DeviceLunarLander` — ``envs/lunar_lander.py::LunarLanderVec`` (the task of ``LunarLanderContinuous-v2``:
  engines, spring-damper feet, shaped reward, crash / landing terminals), one thread per environment.
* :class:`DeviceAnalytic` — ``envs/analytic.py`` (Reacher, inverted double pendulum: closed-form dynamics).
* :class:`DevicePlanar` — ``envs/planar.py`` (Hopper, Walker2d, HalfCheetah, BipedalWalker: articulated bodies
  with ground contact), one thread per environment, mass matrix and Cholesky solve in registers
  (``csrc/planar_env.hip``).
* :class:`DeviceAnt` — ``envs/spatial.py::AntVec`` (3-D articulated body, 14 x 14 mass matrix, implicit contact;
  ``csrc/ant_env.hip``).
* :class:`DeviceSynthetic` — ``envs/synthetic.py::SyntheticVec`` for any observation / action shape (the same
  fixed dynamics matrices), so that the throughput configurations (Ant: 111 / 8, ...) can run their explorers
  on the GPU as well.

:func:`make_device_env` picks the one that stands for ``config['env']``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from ..ops import hip as H
from .device_pendulum import DevicePendulum, StepCounter


class _DeviceEnvBase:
    reward_scale = 1.0

    def __init__(self, num_envs: int, state_dim: int, action_dim: int, device, seed: int, time_limit: int):
        H.lib()
        self.E, self.state_dim, self.action_dim = int(num_envs), int(state_dim), int(action_dim)
        self.device, self.time_limit = torch.device(device), int(time_limit)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.device)   # noqa: E731
        self.state, self.ep_return = z(self.E, self.state_dim), z(self.E)
        self.t = z(self.E, dt=torch.int32)
        self.stats = z(2, dt=torch.float64)
        self.counter = StepCounter(self.device)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.noise: Optional[torch.Tensor] = None
        self._seen = (0.0, 0)

    def _fill(self, a, init: int) -> None:
        a.state, a.t, a.ep_return = self.state.data_ptr(), self.t.data_ptr(), self.ep_return.data_ptr()
        a.stats, a.seed = self.stats.data_ptr(), self.seed
        self.counter.fill(a)
        a.E, a.time_limit, a.init = self.E, self.time_limit, init
        if self.noise is not None:
            a.noise, a.ld_act = self.noise.data_ptr(), self.noise.shape[1]

    def _launch(self, a) -> None:
        raise NotImplementedError

    def _args(self, init: int):
        raise NotImplementedError

    def reset(self, obs_out: torch.Tensor) -> torch.Tensor:
        a = self._args(1)
        a.obs = obs_out.data_ptr()
        self._launch(a)
        return obs_out

    def step(self, action: torch.Tensor, obs_out: torch.Tensor, next_obs_out: torch.Tensor, reward_out: torch.Tensor,
             done_out: torch.Tensor, trunc_out: torch.Tensor) -> None:
        a = self._args(0)
        a.action, a.ld_act = action.data_ptr(), action.shape[1]
        a.obs, a.next_obs = obs_out.data_ptr(), next_obs_out.data_ptr()
        a.reward, a.done, a.trunc = reward_out.data_ptr(), done_out.data_ptr(), trunc_out.data_ptr()
        self._launch(a)

    def episode_stats(self) -> Tuple[Optional[float], int]:
        s = self.stats.cpu()
        total, count = float(s[0].item()), int(s[1].item())
        d_total, d_count = total - self._seen[0], count - self._seen[1]
        self._seen = (total, count)
        return (d_total / d_count if d_count else None), d_count


class DeviceCartPole(_DeviceEnvBase):
    def __init__(self, num_envs: int, device="cuda", seed: int = 0, time_limit: int = 1000, limit_terminal: bool = True):
        super().__init__(num_envs, 4, 1, device, seed, time_limit)
        self.limit_terminal = bool(limit_terminal)      # the host class reports its time limit as done

    def _args(self, init: int) -> H.CartPoleArgs:
        a = H.CartPoleArgs()
        self._fill(a, init)
        a.limit_terminal = int(self.limit_terminal)
        return a

    def _launch(self, a) -> None:
        H.launch("d4pg_cartpole_step", a)


class DeviceLunarLander(_DeviceEnvBase):
    reward_scale = 100.0         # env/lunar_lander_continous.py:12-13

    def __init__(self, num_envs: int, device="cuda", seed: int = 0, time_limit: int = 1000, limit_terminal: bool = True):
        super().__init__(num_envs, 8, 2, device, seed, time_limit)
        # x, y, vx, vy, theta, omega, contacts, previous shaping, rest steps (csrc/rollout.hip, LanderArgs)
        self.state = torch.zeros(self.E, 12, dtype=torch.float32, device=self.device)
        self.limit_terminal = bool(limit_terminal)

    def _args(self, init: int) -> H.LanderArgs:
        a = H.LanderArgs()
        self._fill(a, init)
        a.limit_terminal, a.reward_scale = int(self.limit_terminal), self.reward_scale
        return a

    def _launch(self, a) -> None:
        H.launch("d4pg_lander_step", a)


class DeviceAnalytic(_DeviceEnvBase):
    """``envs/analytic.py`` on the device: ``kind`` 0 = Reacher (2 actions), 1 = inverted double pendulum (1 action);
    11 observations each (``csrc/analytic_env.hip``)."""

    def __init__(self, kind: int, num_envs: int, device="cuda", seed: int = 0, time_limit: int = 1000, limit_terminal: bool = True):
        super().__init__(num_envs, 11, 2 if kind == 0 else 1, device, seed, time_limit)
        self.kind = int(kind)
        self.state = torch.zeros(self.E, 8, dtype=torch.float32, device=self.device)
        self.limit_terminal = bool(limit_terminal)

    def _args(self, init: int) -> H.AnalyticEnvArgs:
        a = H.AnalyticEnvArgs()
        self._fill(a, init)
        a.kind, a.limit_terminal, a.reward_scale = self.kind, int(self.limit_terminal), self.reward_scale
        return a

    def _launch(self, a) -> None:
        H.launch("d4pg_analytic_env_step", a)


class DevicePlanar(_DeviceEnvBase):
    """``envs/planar.py`` (Hopper, Walker2d, HalfCheetah, BipedalWalker) on the device: the mechanism's topology
    is compiled into ``csrc/planar_env.hip``; every number (masses, lengths, gains, task coefficients) is read
    from the host class, so the two cannot drift apart."""
    STRIDE = 24

    def __init__(self, host_cls, num_envs: int, device="cuda", seed: int = 0, time_limit: int = 1000, limit_terminal: bool = True,
                 terrain: bool = True):
        host = host_cls(1)
        super().__init__(num_envs, host.state_dim, host.action_dim, device, seed, time_limit)
        from .native_planar import planar_params
        self.params, self.kind = planar_params(host)
        self.state = torch.zeros(self.E, self.STRIDE, dtype=torch.float32, device=self.device)
        self.limit_terminal = bool(limit_terminal)
        self.n = host.model.n
        # the walker's ground: [200][E] heights (sample-major: coalesced over the environments of a wavefront), drawn by
        # the kernel at every reset (terrain=False: all zero)
        self.terrain = torch.zeros(200, self.E, dtype=torch.float32, device=self.device) if self.kind == 3 else None
        self.hilly = bool(terrain)
        self._params_dev = torch.frombuffer(bytearray(bytes(self.params)), dtype=torch.uint8).to(self.device)

    def _args(self, init: int) -> H.PlanarEnvArgs:
        a = H.PlanarEnvArgs()
        self._fill(a, init)
        a.P, a.kind = self._params_dev.data_ptr(), self.kind
        a.terrain, a.hilly = (self.terrain.data_ptr() if self.terrain is not None else None), int(self.hilly)
        a.limit_terminal, a.reward_scale = int(self.limit_terminal), self.reward_scale
        return a

    def _launch(self, a) -> None:
        H._check(H.lib().d4pg_planar_env_step(H.C.byref(a), H.C.byref(self.params), H.stream_ptr(None)), "d4pg_planar_env_step")

    # state hand-over with the host class (tests, warm starts)
    def set_state(self, q: np.ndarray, qd: np.ndarray) -> None:
        st = torch.zeros(self.E, self.STRIDE, dtype=torch.float32)
        st[:, :self.n] = torch.as_tensor(q, dtype=torch.float32)
        st[:, 9:9 + self.n] = torch.as_tensor(qd, dtype=torch.float32)
        if self.kind == 3:
            st[:, 20] = 130.0 * st[:, 0] / 30.0 - 5.0 * st[:, 2].abs()
        self.state.copy_(st)
        self.t.zero_()
        self.ep_return.zero_()


class DeviceAnt(_DeviceEnvBase):
    """``envs/spatial.py::AntVec`` on the device (``csrc/ant_env.hip``): the ant's topology is compiled in, every
    number is read from the host class."""
    STRIDE = 32

    def __init__(self, num_envs: int, device="cuda", seed: int = 0, time_limit: int = 1000, limit_terminal: bool = True):
        from .spatial import AntVec
        host = AntVec(1)
        super().__init__(num_envs, 111, 8, device, seed, time_limit)
        self.state = torch.zeros(self.E, self.STRIDE, dtype=torch.float32, device=self.device)
        self.limit_terminal = bool(limit_terminal)
        from .native_planar import ant_params
        self.params = ant_params(host)
        self._params_dev = torch.frombuffer(bytearray(bytes(self.params)), dtype=torch.uint8).to(self.device)

    def _args(self, init: int) -> H.AntEnvArgs:
        a = H.AntEnvArgs()
        self._fill(a, init)
        a.P, a.limit_terminal, a.reward_scale = self._params_dev.data_ptr(), int(self.limit_terminal), self.reward_scale
        return a

    def _launch(self, a) -> None:
        H._check(H.lib().d4pg_ant_env_step(H.C.byref(a), H.C.byref(self.params), H.stream_ptr(None)), "d4pg_ant_env_step")


class DeviceSynthetic(_DeviceEnvBase):
    def __init__(self, env_name: str, num_envs: int, state_dim: int, action_dim: int, device="cuda", seed: int = 0,
                 time_limit: int = 1000, terminal_prob: float = 0.001, noise_std: float = 0.05):
        super().__init__(num_envs, state_dim, action_dim, device, seed, time_limit)
        from .synthetic import SyntheticVec
        host = SyntheticVec(env_name, 1, state_dim, action_dim)          # the same dynamics matrices as on the host
        self.M = torch.from_numpy(np.ascontiguousarray(host.M)).to(self.device)
        self.N = torch.from_numpy(np.ascontiguousarray(host.N)).to(self.device)
        self.terminal_prob, self.noise_std = float(terminal_prob), float(noise_std)

    def _args(self, init: int) -> H.SyntheticEnvArgs:
        a = H.SyntheticEnvArgs()
        self._fill(a, init)
        a.M, a.N, a.S, a.A = self.M.data_ptr(), self.N.data_ptr(), self.state_dim, self.action_dim
        a.terminal_prob, a.noise_std = self.terminal_prob, self.noise_std
        return a

    def _launch(self, a) -> None:
        H.launch("d4pg_synthetic_env_step", a)


def make_device_env(cfg, num_envs: int, device, seed: int = 0):
    """Device-resident environment standing for ``cfg['env']`` — what ``envs.create_env_wrapper`` would build on
    the host when no gym is installed — or ``None`` (a simulator that only exists on the host)."""
    name = str(cfg["env"])
    limit = int(cfg["max_ep_length"])
    if cfg.get("synthetic_env"):
        return DeviceSynthetic(name, num_envs, cfg["state_dim"], cfg["action_dim"], device, seed, time_limit=limit)
    if name.startswith("Pendulum"):
        return DevicePendulum(num_envs, device, seed, time_limit=min(200, limit))
    from .gym_adapter import gym_available
    if gym_available() and int(cfg.get("gpu_rollout_envs") or 0) <= 0 and cfg.get("env_backend") != "native":
        # the host agents of this job would play gym's dynamics (envs.create_env_wrapper): no native twin to offer
        return None
    if name == "InvertedPendulum-v2":
        return DeviceCartPole(num_envs, device, seed, time_limit=limit)
    if name.startswith("LunarLanderContinuous"):
        return DeviceLunarLander(num_envs, device, seed, time_limit=limit)
    from . import planar
    for prefix, cls in (("Hopper-", planar.HopperVec), ("Walker2d-", planar.Walker2dVec), ("HalfCheetah-", planar.HalfCheetahVec),
                        ("BipedalWalker-", planar.BipedalWalkerVec)):
        if name.startswith(prefix):
            return DevicePlanar(cls, num_envs, device, seed, time_limit=limit)
    if name.startswith("Reacher-"):
        return DeviceAnalytic(0, num_envs, device, seed, time_limit=min(50, limit))
    if name.startswith("InvertedDoublePendulum-"):
        return DeviceAnalytic(1, num_envs, device, seed, time_limit=limit)
    if name.startswith("Ant-"):
        return DeviceAnt(num_envs, device, seed, time_limit=limit)
    return DeviceSynthetic(name, num_envs, cfg["state_dim"], cfg["action_dim"], device, seed, time_limit=limit)
