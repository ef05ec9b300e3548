"""Built-in inverted-pendulum swing-up (the dynamics of gym's ``Pendulum-v0``).

gym is not a dependency of this framework, so the one environment the
learning tests need is implemented natively (and vectorised):

* state ``(theta, theta_dot)``, observation ``[cos theta, sin theta, theta_dot]``
* torque ``u = clip(a, -2, 2)``; cost ``angle_norm(theta)^2 + 0.1 theta_dot^2 + 0.001 u^2``
* ``theta_dot' = clip(theta_dot + (-3g/(2l) sin(theta + pi) + 3/(m l^2) u) dt, -8, 8)``,
  ``theta' = theta + theta_dot' dt`` with ``g=10, m=1, l=1, dt=0.05``
* reset: ``theta ~ U(-pi, pi)``, ``theta_dot ~ U(-1, 1)``
* the registered gym environment carries a 200-step time limit whose expiry is
  reported as ``done=True`` through the old 4-tuple API the reference consumes
  (``env/env_wrapper.py:17-19``); that behaviour is kept (``time_limit``).

Rewards are divided by 100 for the learner like the reference's wrapper
(``env/pendulum.py:13-14``).
"""
from __future__ import annotations

import numpy as np

from .base import VecEnv


class PendulumVec(VecEnv):
    reward_scale = 100.0
    max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 200, env_name: str = "Pendulum-v0"):
        super().__init__(env_name, num_envs, 3, 1, -self.max_torque, self.max_torque, seed)
        self.time_limit = int(time_limit)
        self.theta = np.zeros(num_envs)
        self.theta_dot = np.zeros(num_envs)
        self.t = np.zeros(num_envs, dtype=np.int64)
        self.last_u = np.zeros(num_envs)

    def _obs(self, idx=slice(None)) -> np.ndarray:
        th, thd = self.theta[idx], self.theta_dot[idx]
        return np.stack([np.cos(th), np.sin(th), thd], axis=-1).astype(np.float32)

    def _reset_env(self, e: int) -> np.ndarray:
        self.theta[e] = self.rng.uniform(-np.pi, np.pi)
        self.theta_dot[e] = self.rng.uniform(-1.0, 1.0)
        self.t[e] = 0
        self.last_u[e] = 0.0
        return self._obs(e)

    def _step(self, actions: np.ndarray):
        u = np.clip(actions[:, 0].astype(np.float64), -self.max_torque, self.max_torque)
        th, thd = self.theta, self.theta_dot
        ang = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = ang ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
        new_thd = thd + (-3 * self.g / (2 * self.l) * np.sin(th + np.pi) + 3.0 / (self.m * self.l ** 2) * u) * self.dt
        new_th = th + new_thd * self.dt
        self.theta_dot = np.clip(new_thd, -self.max_speed, self.max_speed)
        self.theta = new_th
        self.t += 1
        self.last_u = u
        done = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        return self._obs(), -cost, done

    def render(self, e: int = 0) -> np.ndarray:
        size = 128
        img = np.full((size, size, 3), 255, dtype=np.uint8)
        c = size // 2
        L = size * 0.4
        # theta = 0 is upright; draw the rod as a thick line
        x1, y1 = c + L * np.sin(self.theta[e]), c - L * np.cos(self.theta[e])
        for s in np.linspace(0.0, 1.0, 200):
            x, y = int(round(c + s * (x1 - c))), int(round(c + s * (y1 - c)))
            img[max(0, y - 2): y + 3, max(0, x - 2): x + 3] = (204, 77, 77)
        img[c - 3: c + 4, c - 3: c + 4] = 0
        return img
