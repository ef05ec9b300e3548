"""Shape-only and analytic stand-in environments.

* :class:`SyntheticVec` — an environment with the observation/action shape of
  any configuration (e.g. Ant: 111/8) whose dynamics are a fixed random
  contraction ``s' = tanh(s M + a N + noise)`` with reward ``-|s'|^2/S - 0.01|a|^2``.
  It exists to exercise the full actor -> ring -> replay -> learner pipeline and
  to measure env-steps/s for shapes whose simulator (Box2D, MuJoCo) is not
  installed.  Episodes end by time limit only, with a small random terminal rate.
* :class:`CartPoleContinuousVec` — continuous-force cart-pole balancing with the
  observation/action shape of ``InvertedPendulum-v2`` (4/1): +1 reward per step,
  terminal when the pole leaves +-0.2 rad.  A second task with real learning signal.
"""
from __future__ import annotations

import numpy as np

from .base import VecEnv


class SyntheticVec(VecEnv):
    def __init__(self, env_name: str, num_envs: int, state_dim: int, action_dim: int,
                 action_low=-1.0, action_high=1.0, seed=None, episode_len: int = 1000,
                 terminal_prob: float = 0.001):
        super().__init__(env_name, num_envs, state_dim, action_dim, action_low, action_high, seed)
        fixed = np.random.default_rng(12345)     # dynamics are the same in every process
        self.M = (fixed.standard_normal((state_dim, state_dim)) * (0.9 / np.sqrt(state_dim))).astype(np.float32)
        self.N = (fixed.standard_normal((action_dim, state_dim)) * (0.5 / np.sqrt(action_dim))).astype(np.float32)
        self.episode_len = int(episode_len)
        self.terminal_prob = float(terminal_prob)
        self.s = np.zeros((num_envs, state_dim), np.float32)
        self.t = np.zeros(num_envs, np.int64)

    def _reset_env(self, e: int) -> np.ndarray:
        self.s[e] = self.rng.standard_normal(self.state_dim).astype(np.float32) * 0.5
        self.t[e] = 0
        return self.s[e].copy()

    def _step(self, actions: np.ndarray):
        noise = self.rng.standard_normal(self.s.shape).astype(np.float32) * 0.05
        self.s = np.tanh(self.s @ self.M + actions @ self.N + noise)
        self.t += 1
        reward = -(self.s ** 2).mean(axis=1) - 0.01 * (actions ** 2).sum(axis=1)
        done = (self.t >= self.episode_len) | (self.rng.random(self.num_envs) < self.terminal_prob)
        return self.s.copy(), reward, done


class CartPoleContinuousVec(VecEnv):
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    theta_limit, x_limit = 0.2, 2.4

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1000, env_name: str = "InvertedPendulum-v2"):
        super().__init__(env_name, num_envs, 4, 1, -1.0, 1.0, seed)
        self.state = np.zeros((num_envs, 4))
        self.t = np.zeros(num_envs, np.int64)
        self.time_limit = int(time_limit)

    def _reset_env(self, e: int) -> np.ndarray:
        self.state[e] = self.rng.uniform(-0.05, 0.05, size=4)
        self.t[e] = 0
        return self.state[e].astype(np.float32)

    def _step(self, actions: np.ndarray):
        x, x_dot, th, th_dot = self.state.T
        force = np.clip(actions[:, 0], -1.0, 1.0) * self.force_mag
        total_mass = self.masscart + self.masspole
        pml = self.masspole * self.length
        cos, sin = np.cos(th), np.sin(th)
        temp = (force + pml * th_dot ** 2 * sin) / total_mass
        th_acc = (self.gravity * sin - cos * temp) / (self.length * (4.0 / 3.0 - self.masspole * cos ** 2 / total_mass))
        x_acc = temp - pml * th_acc * cos / total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * x_acc
        th = th + self.tau * th_dot
        th_dot = th_dot + self.tau * th_acc
        self.state = np.stack([x, x_dot, th, th_dot], axis=1)
        self.t += 1
        fell = (np.abs(th) > self.theta_limit) | (np.abs(x) > self.x_limit)
        done = fell | (self.t >= self.time_limit)
        return self.state.astype(np.float32), np.ones(self.num_envs, np.float32), done
