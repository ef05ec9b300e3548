"""Two more of the reference's tasks as closed-form rigid-body dynamics (no simulator dependency).

The reference runs ``Reacher-v2`` and ``InvertedDoublePendulum-v2`` through gym / MuJoCo
(``configs/reacher_d4pg.yml``, ``configs/inverted_double_pendulum_d4pg.yml``; any id without a wrapper of its
own goes through ``env/utils.py:15``).  MuJoCo is not a dependency here; both systems are small enough to
write their equations of motion down:

* :class:`ReacherVec` — a planar two-link arm without gravity (links 0.1 and 0.11 long, joint damping 1,
  rotor inertia 1 per joint, torque 200 x action, 0.02 s per step, 50-step episodes), a target drawn in a disk of
  radius 0.2.  Observation ``[cos q1, cos q2, sin q1, sin q2, target x, target y, q1', q2', tip - target (x, y, 0)]``,
  reward ``-|tip - target| - |a|^2``: the layout and reward of the gym task.
* :class:`InvertedDoublePendulumVec` — a cart (mass 1) carrying two uniform rods (0.6 long, mass 0.3 each) on a
  frictionless rail, force 20 x action, 0.05 s per step.  Observation ``[x, sin q1, sin q2, cos q1, cos q2, x', q1',
  q2', 0, 0, 0]`` with q2 relative to the first rod (the three constraint-force entries of the gym task are zero
  here), reward ``10 - 0.01 x_tip^2 - (y_tip - 2)^2 - 0.001 q1'^2 - 0.005 q2'^2``, terminal when the tip drops to
  ``y <= 1``: the layout, reward and termination of the gym task.

Same task definitions, own parameters and integrator: returns are comparable in kind with the gym tasks', not
trajectory by trajectory.
"""
from __future__ import annotations

import numpy as np

from .base import VecEnv


class ReacherVec(VecEnv):
    L1, L2 = 0.1, 0.11
    M1, M2 = 0.0356, 0.0392            # capsule links of radius 0.01, density 1000 (+ the fingertip sphere)
    ARMATURE, DAMPING, GEAR, DT, SUBSTEPS = 1.0, 1.0, 200.0, 0.02, 4

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 50, env_name: str = "Reacher-v2"):
        super().__init__(env_name, num_envs, 11, 2, -1.0, 1.0, seed)
        self.time_limit = int(time_limit)
        E = self.num_envs
        self.q = np.zeros((E, 2))
        self.qd = np.zeros((E, 2))
        self.target = np.zeros((E, 2))
        self.t = np.zeros(E, np.int64)

    def tip(self, idx=slice(None)) -> np.ndarray:
        q = self.q[idx]
        a1, a12 = q[..., 0], q[..., 0] + q[..., 1]
        return np.stack([self.L1 * np.cos(a1) + self.L2 * np.cos(a12), self.L1 * np.sin(a1) + self.L2 * np.sin(a12)], axis=-1)

    def _obs(self, idx=slice(None)) -> np.ndarray:
        q, qd, tg = self.q[idx], self.qd[idx], self.target[idx]
        d = self.tip(idx) - tg
        return np.concatenate([np.cos(q), np.sin(q), tg, qd, d, np.zeros_like(d[..., :1])], axis=-1).astype(np.float32)

    def _reset_env(self, e: int) -> np.ndarray:
        self.q[e] = self.rng.uniform(-0.1, 0.1, 2)
        self.qd[e] = self.rng.uniform(-0.005, 0.005, 2)
        while True:
            g = self.rng.uniform(-0.2, 0.2, 2)
            if np.linalg.norm(g) < 0.2:
                break
        self.target[e] = g
        self.t[e] = 0
        return self._obs(e)

    def _accel(self, q, qd, tau):
        l1, lc1, lc2, m1, m2 = self.L1, self.L1 / 2, self.L2 / 2, self.M1, self.M2
        i1, i2 = m1 * self.L1 ** 2 / 12, m2 * self.L2 ** 2 / 12
        c2, s2 = np.cos(q[:, 1]), np.sin(q[:, 1])
        m11 = i1 + i2 + m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * c2) + self.ARMATURE
        m12 = i2 + m2 * (lc2 ** 2 + l1 * lc2 * c2)
        m22 = i2 + m2 * lc2 ** 2 + self.ARMATURE
        h = -m2 * l1 * lc2 * s2
        b1 = tau[:, 0] - h * qd[:, 1] * (2 * qd[:, 0] + qd[:, 1]) - self.DAMPING * qd[:, 0]
        b2 = tau[:, 1] + h * qd[:, 0] ** 2 - self.DAMPING * qd[:, 1]
        det = m11 * m22 - m12 * m12
        return np.stack([(m22 * b1 - m12 * b2) / det, (m11 * b2 - m12 * b1) / det], axis=1)

    def render(self, e: int = 0) -> np.ndarray:
        from .planar import draw_segments
        w, h, sc = 300, 200, 380.0
        px = lambda x, y: (w / 2 + x * sc, h / 2 - y * sc)                                    # noqa: E731
        a1, a12 = self.q[e, 0], self.q[e, 0] + self.q[e, 1]
        elbow = (self.L1 * np.cos(a1), self.L1 * np.sin(a1))
        tip = (elbow[0] + self.L2 * np.cos(a12), elbow[1] + self.L2 * np.sin(a12))
        tx, ty = self.target[e]
        segs = [(px(0, 0), px(*elbow), (128, 102, 230)), (px(*elbow), px(*tip), (230, 180, 60)),
                (px(tx - 0.01, ty), px(tx + 0.01, ty), (230, 60, 60)), (px(tx, ty - 0.01), px(tx, ty + 0.01), (230, 60, 60))]
        return draw_segments(segs, w, h)

    def _step(self, actions: np.ndarray):
        a = np.clip(actions.astype(np.float64), -1.0, 1.0)
        before = self.tip() - self.target
        reward = -np.linalg.norm(before, axis=1) - np.sum(np.square(actions.astype(np.float64)), axis=1)
        h = self.DT / self.SUBSTEPS
        for _ in range(self.SUBSTEPS):
            self.qd = self.qd + h * self._accel(self.q, self.qd, self.GEAR * a)
            self.q = self.q + h * self.qd
        self.t += 1
        done = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        return self._obs(), reward, done


class InvertedDoublePendulumVec(VecEnv):
    M0, M1, M2, L1, L2, G = 1.0, 0.3, 0.3, 0.6, 0.6, 9.81
    FORCE, DT, SUBSTEPS, X_LIMIT = 20.0, 0.05, 10, 1.0

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1000, env_name: str = "InvertedDoublePendulum-v2"):
        super().__init__(env_name, num_envs, 11, 1, -1.0, 1.0, seed)
        self.time_limit = int(time_limit)
        E = self.num_envs
        self.q = np.zeros((E, 3))          # x, absolute angle of rod 1, absolute angle of rod 2 (0 = upright)
        self.qd = np.zeros((E, 3))
        self.t = np.zeros(E, np.int64)
        m0, m1, m2, l1, l2 = self.M0, self.M1, self.M2, self.L1, self.L2
        self._d = (m0 + m1 + m2, (m1 / 2 + m2) * l1, m2 * l2 / 2, (m1 / 3 + m2) * l1 ** 2, m2 * l1 * l2 / 2, m2 * l2 ** 2 / 3,
                   (m1 / 2 + m2) * l1 * self.G, m2 * l2 / 2 * self.G)

    def tip(self, idx=slice(None)):
        q = self.q[idx]
        return (q[..., 0] + self.L1 * np.sin(q[..., 1]) + self.L2 * np.sin(q[..., 2]),
                self.L1 * np.cos(q[..., 1]) + self.L2 * np.cos(q[..., 2]))

    def _obs(self, idx=slice(None)) -> np.ndarray:
        q, qd = self.q[idx], self.qd[idx]
        rel = np.stack([q[..., 1], q[..., 2] - q[..., 1]], axis=-1)
        vel = np.clip(np.stack([qd[..., 0], qd[..., 1], qd[..., 2] - qd[..., 1]], axis=-1), -10.0, 10.0)
        return np.concatenate([q[..., :1], np.sin(rel), np.cos(rel), vel, np.zeros_like(vel)], axis=-1).astype(np.float32)

    def _reset_env(self, e: int) -> np.ndarray:
        self.q[e] = self.rng.uniform(-0.1, 0.1, 3)
        self.q[e, 2] += self.q[e, 1]
        self.qd[e] = self.rng.standard_normal(3) * 0.1
        self.t[e] = 0
        return self._obs(e)

    def render(self, e: int = 0) -> np.ndarray:
        from .planar import draw_segments
        w, h, sc = 300, 200, 70.0
        px = lambda x, y: (w / 2 + x * sc, h * 0.8 - y * sc)                                  # noqa: E731
        x, a1, a2 = self.q[e]
        j = (x + self.L1 * np.sin(a1), self.L1 * np.cos(a1))
        tip = (j[0] + self.L2 * np.sin(a2), j[1] + self.L2 * np.cos(a2))
        segs = [(px(-self.X_LIMIT, 0), px(self.X_LIMIT, 0), (90, 90, 90)), (px(x - 0.1, 0), px(x + 0.1, 0), (204, 204, 0)),
                (px(x, 0), px(*j), (128, 102, 230)), (px(*j), px(*tip), (230, 180, 60))]
        return draw_segments(segs, w, h)

    def _accel(self, q, qd, u):
        d1, d2, d3, d4, d5, d6, f1, f2 = self._d
        c1, s1, c2, s2 = np.cos(q[:, 1]), np.sin(q[:, 1]), np.cos(q[:, 2]), np.sin(q[:, 2])
        c12, s12 = np.cos(q[:, 1] - q[:, 2]), np.sin(q[:, 1] - q[:, 2])
        E = q.shape[0]
        D = np.empty((E, 3, 3))
        D[:, 0, 0], D[:, 0, 1], D[:, 0, 2] = d1, d2 * c1, d3 * c2
        D[:, 1, 0], D[:, 1, 1], D[:, 1, 2] = d2 * c1, d4, d5 * c12
        D[:, 2, 0], D[:, 2, 1], D[:, 2, 2] = d3 * c2, d5 * c12, d6
        rhs = np.stack([u + d2 * s1 * qd[:, 1] ** 2 + d3 * s2 * qd[:, 2] ** 2,
                        f1 * s1 - d5 * s12 * qd[:, 2] ** 2,
                        f2 * s2 + d5 * s12 * qd[:, 1] ** 2], axis=1)
        return np.linalg.solve(D, rhs[..., None])[..., 0]

    def _step(self, actions: np.ndarray):
        u = np.clip(actions[:, 0].astype(np.float64), -1.0, 1.0) * self.FORCE
        h = self.DT / self.SUBSTEPS
        for _ in range(self.SUBSTEPS):
            self.qd = self.qd + h * self._accel(self.q, self.qd, u)
            self.q = self.q + h * self.qd
        hit = np.abs(self.q[:, 0]) > self.X_LIMIT           # the rail ends: the cart stops there
        self.q[:, 0] = np.clip(self.q[:, 0], -self.X_LIMIT, self.X_LIMIT)
        self.qd[hit, 0] = 0.0
        self.t += 1
        x_tip, y_tip = self.tip()
        v1, v2 = self.qd[:, 1], self.qd[:, 2] - self.qd[:, 1]
        reward = 10.0 - (0.01 * x_tip ** 2 + (y_tip - 2.0) ** 2) - (1e-3 * v1 ** 2 + 5e-3 * v2 ** 2)
        fell = y_tip <= 1.0
        limit = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        return self._obs(), reward, fell | limit
