"""Native continuous-control lunar lander with the task definition of gym's ``LunarLanderContinuous-v2``.

The reference trains on the gym / Box2D environment (``env/lunar_lander_continous.py:1-13``,
``configs/lunar_lander_continuous_d4pg.yml``).  Neither gym nor Box2D is a dependency of this framework, so
the task is implemented here as its own vectorised planar rigid-body simulation — same observation and action
layout, same reward shaping, same termination rules, same units — with a penalty-spring contact model in
place of Box2D's constraint solver and flat ground in place of the random hills beside the pad.  It is the
same TASK, not the same simulator: returns are comparable in kind (crash −100, landing +100, ≥ 200 is a good
landing), trajectories are not bit-comparable.

Units and constants (world units = gym's pixels / 30; 50 simulation steps per second):

* world 20 wide, 13.33 high; pad surface at ``y = 3.33``; the hull starts at ``(10, 13.33)`` with a random
  velocity of up to ±4 per axis (gym: a random force of ±1000 for one step on a body of mass ≈ 5)
* hull: mass 4.96, moment of inertia 0.864, gravity −10
* ``a[0] > 0`` fires the main engine at 50 % … 100 % power: acceleration ``17.5 · power`` along the body's
  up axis ``(−sin θ, cos θ)`` (gym: impulse ``13 · 4/30 · power`` per step)
* ``|a[1]| > 0.5`` fires a side engine at ``|a[1]|`` power: lateral acceleration ``2.42 · power`` along
  ``sign(a[1]) · (cos θ, sin θ)`` and angular acceleration ``−sign(a[1]) · 6.48 · power``
* two feet at body coordinates ``(±0.667, −0.6)``: spring-damper normal force, viscous friction; the hull's
  own corners touching the ground is a crash, and so is a foot pushed in by more than 0.04 (the leg gives way:
  touching down faster than about 3 units/s; free fall arrives at 13.7, half throttle at about 5)
* observation ``[(x−10)/10, (y−3.93)/6.67, vx·10/50, vy·6.67/50, θ, 20·ω/50, left contact, right contact]``
* reward: change of ``−100·|pos| − 100·|vel| − 100·|θ| + 10·contacts`` minus ``0.30·main power`` and
  ``0.03·side power`` per step; −100 and done on a crash or when ``|obs[0]| ≥ 1``; +100 and done once the
  lander has been at rest on both feet for 0.4 s

Rewards are divided by 100 for the learner like the reference's wrapper (``env/lunar_lander_continous.py:12-13``).
The device-resident twin is ``envs/device_envs.py::DeviceLunarLander`` (``csrc/rollout.hip``), same arithmetic in fp32.
"""
from __future__ import annotations

import numpy as np

from .base import VecEnv

# shared with csrc/rollout.hip (lander_step_kernel): keep the two in step
FPS = 50.0
W, H = 20.0, 400.0 / 30.0
PAD_Y = H / 4.0
LEG_DOWN = 0.6                   # hull centre above the feet
FOOT_X = 20.0 / 30.0
HULL_HALF_W, HULL_BOTTOM, HULL_TOP, HULL_TOP_HALF_W = 17.0 / 30.0, -10.0 / 30.0, 17.0 / 30.0, 14.0 / 30.0
MASS, INERTIA, GRAVITY = 4.96, 0.864, -10.0
MAIN_ACC = 13.0 * (4.0 / 30.0) / MASS * FPS              # 17.47
SIDE_ACC = 0.6 * (12.0 / 30.0) / MASS * FPS              # 2.42
SIDE_ANG_ACC = 0.6 * (12.0 / 30.0) * (14.0 / 30.0) / INERTIA * FPS      # 6.48
SUBSTEPS = 4
K_SPRING, C_DAMP, C_FRICTION = 4000.0, 120.0, 60.0       # per foot: N / unit, N s / unit
REST_STEPS = 20
LEG_TRAVEL = 0.04                # a foot pushed in further than this: the leg gives way, the hull hits the ground
INIT_SPEED = 4.0


def lander_obs(x, y, vx, vy, th, om, c0, c1):
    return np.stack([(x - W / 2) / (W / 2), (y - (PAD_Y + LEG_DOWN)) / (H / 2), vx * (W / 2) / FPS, vy * (H / 2) / FPS,
                     th, 20.0 * om / FPS, c0, c1], axis=-1)


def lander_shaping(o):
    return (-100.0 * np.sqrt(o[..., 0] ** 2 + o[..., 1] ** 2) - 100.0 * np.sqrt(o[..., 2] ** 2 + o[..., 3] ** 2)
            - 100.0 * np.abs(o[..., 4]) + 10.0 * o[..., 6] + 10.0 * o[..., 7])


class LunarLanderVec(VecEnv):
    reward_scale = 100.0

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1000,
                 env_name: str = "LunarLanderContinuous-v2"):
        super().__init__(env_name, num_envs, 8, 2, -1.0, 1.0, seed)
        self.time_limit = int(time_limit)
        E = self.num_envs
        # x, y, vx, vy, theta, omega
        self.q = np.zeros((E, 6))
        self.contact = np.zeros((E, 2))
        self.prev_shaping = np.zeros(E)
        self.rest = np.zeros(E, np.int64)
        self.t = np.zeros(E, np.int64)
        self.last_power = np.zeros((E, 2))

    # ---- helpers -------------------------------------------------------------------------------
    def _obs(self, idx=slice(None)) -> np.ndarray:
        q, c = self.q[idx], self.contact[idx]
        return lander_obs(q[..., 0], q[..., 1], q[..., 2], q[..., 3], q[..., 4], q[..., 5], c[..., 0], c[..., 1]).astype(np.float32)

    def _reset_env(self, e: int) -> np.ndarray:
        self.q[e] = (W / 2, H, self.rng.uniform(-INIT_SPEED, INIT_SPEED), self.rng.uniform(-INIT_SPEED, INIT_SPEED), 0.0, 0.0)
        self.contact[e] = 0.0
        self.rest[e] = 0
        self.t[e] = 0
        self.prev_shaping[e] = lander_shaping(self._obs(e).astype(np.float64))
        return self._obs(e)

    @staticmethod
    def engine_powers(actions: np.ndarray):
        """Main-engine power in {0} ∪ [0.5, 1], signed side-engine power in {0} ∪ ±[0.5, 1]."""
        a = np.clip(np.asarray(actions, np.float64), -1.0, 1.0)
        main = np.where(a[:, 0] > 0.0, 0.5 + 0.5 * np.clip(a[:, 0], 0.0, 1.0), 0.0)
        side = np.where(np.abs(a[:, 1]) > 0.5, a[:, 1], 0.0)
        return main, side

    def _step(self, actions: np.ndarray):
        main, side = self.engine_powers(actions)
        x, y, vx, vy, th, om = (self.q[:, i].copy() for i in range(6))
        h = 1.0 / (FPS * SUBSTEPS)
        c0 = np.zeros(self.num_envs)
        c1 = np.zeros(self.num_envs)
        broke = np.zeros(self.num_envs, bool)
        for _ in range(SUBSTEPS):
            s, c = np.sin(th), np.cos(th)
            ax = -s * MAIN_ACC * main + c * SIDE_ACC * side
            ay = c * MAIN_ACC * main + s * SIDE_ACC * side + GRAVITY
            alpha = -SIDE_ANG_ACC * side
            for k, fx_body in enumerate((-FOOT_X, FOOT_X)):
                rx = c * fx_body + s * LEG_DOWN                 # foot relative to the centre, world frame
                ry = s * fx_body - c * LEG_DOWN
                pen = PAD_Y - (y + ry)
                on = pen > 0.0
                broke |= pen > LEG_TRAVEL
                fvx = vx - om * ry                              # foot velocity
                fvy = vy + om * rx
                fn = np.where(on, np.maximum(0.0, K_SPRING * pen - C_DAMP * fvy), 0.0)
                ft = np.where(on, -C_FRICTION * fvx, 0.0)
                ax += ft / MASS
                ay += fn / MASS
                alpha += (rx * fn - ry * ft) / INERTIA
                if k == 0:
                    c0 = np.maximum(c0, on)
                else:
                    c1 = np.maximum(c1, on)
            vx += h * ax
            vy += h * ay
            om += h * alpha
            x += h * vx
            y += h * vy
            th += h * om
        self.q = np.stack([x, y, vx, vy, th, om], axis=1)
        self.contact = np.stack([c0, c1], axis=1).astype(np.float64)
        self.last_power = np.stack([main, np.abs(side)], axis=1)
        self.t += 1
        # the hull itself on the ground (either bottom corner or, upside down, a top corner)
        s, c = np.sin(th), np.cos(th)
        lowest = np.minimum.reduce([
            y + s * (-HULL_HALF_W) + c * HULL_BOTTOM, y + s * HULL_HALF_W + c * HULL_BOTTOM,
            y + s * (-HULL_TOP_HALF_W) + c * HULL_TOP, y + s * HULL_TOP_HALF_W + c * HULL_TOP])
        crashed = (lowest < PAD_Y) | broke
        obs = lander_obs(x, y, vx, vy, th, om, c0, c1)
        shaping = lander_shaping(obs)
        reward = shaping - self.prev_shaping - 0.30 * main - 0.03 * np.abs(side)
        self.prev_shaping = shaping
        still = (c0 > 0) & (c1 > 0) & (np.hypot(vx, vy) < 0.05) & (np.abs(om) < 0.05)
        self.rest = np.where(still, self.rest + 1, 0)
        landed = self.rest >= REST_STEPS
        failed = crashed | (np.abs(obs[:, 0]) >= 1.0)
        reward = np.where(failed, -100.0, np.where(landed, 100.0, reward))
        limit = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        done = failed | landed | limit
        return obs.astype(np.float32), reward, done

    def render(self, e: int = 0) -> np.ndarray:
        w, h = 300, 200
        img = np.zeros((h, w, 3), np.uint8)
        sx, sy = w / W, h / H
        gy = int(h - PAD_Y * sy)
        img[gy:, :] = (90, 90, 90)
        img[gy - 2:gy, int((W / 2 - 1.3) * sx): int((W / 2 + 1.3) * sx)] = (204, 204, 0)
        x, y, _, _, th, _ = self.q[e]
        s, c = np.sin(th), np.cos(th)

        def px(bx, by):
            return int(round((x + c * bx - s * by) * sx)), int(round(h - (y + s * bx + c * by) * sy))

        def line(p, q_, col):
            for u in np.linspace(0.0, 1.0, 60):
                xx, yy = int(p[0] + u * (q_[0] - p[0])), int(p[1] + u * (q_[1] - p[1]))
                if 0 <= xx < w and 0 <= yy < h:
                    img[max(0, yy - 1): yy + 1, max(0, xx - 1): xx + 1] = col
        hull = [(-HULL_TOP_HALF_W, HULL_TOP), (-HULL_HALF_W, 0.0), (-HULL_HALF_W, HULL_BOTTOM), (HULL_HALF_W, HULL_BOTTOM),
                (HULL_HALF_W, 0.0), (HULL_TOP_HALF_W, HULL_TOP)]
        for i in range(len(hull)):
            line(px(*hull[i]), px(*hull[(i + 1) % len(hull)]), (128, 102, 230))
        for fx in (-FOOT_X, FOOT_X):
            line(px(np.sign(fx) * HULL_HALF_W, HULL_BOTTOM), px(fx, -LEG_DOWN), (230, 230, 230))
        if self.last_power[e, 0] > 0:
            line(px(0.0, HULL_BOTTOM), px(0.0, HULL_BOTTOM - 0.8 * self.last_power[e, 0]), (255, 120, 30))
        return img


def heuristic_action(obs: np.ndarray) -> np.ndarray:
    """A hand-written PD landing controller (hover-height and attitude targets from the position error):
    it lands this simulation reliably, which is the check that the task is well posed."""
    o = np.asarray(obs, np.float64)
    angle_targ = np.clip(o[:, 0] * 0.5 + o[:, 2] * 1.0, -0.4, 0.4)
    hover_targ = 0.55 * np.abs(o[:, 0])
    angle_todo = (angle_targ - o[:, 4]) * 0.5 - o[:, 5] * 1.0
    hover_todo = (hover_targ - o[:, 1]) * 0.5 - o[:, 3] * 0.5
    legs = (o[:, 6] > 0) | (o[:, 7] > 0)
    angle_todo = np.where(legs, 0.0, angle_todo)
    hover_todo = np.where(legs, -o[:, 3] * 0.5, hover_todo)
    return np.clip(np.stack([hover_todo * 20.0 - 1.0, -angle_todo * 20.0], axis=1), -1.0, 1.0).astype(np.float32)
