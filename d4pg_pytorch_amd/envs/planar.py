"""Planar articulated-body tasks: BipedalWalker, Hopper, Walker2d and HalfCheetah without Box2D / MuJoCo.

The reference runs ``BipedalWalker-v2`` through gym / Box2D (``env/bipedal.py``, ``configs/bipedal_d4pg.yml``) and
``Hopper-v2``, ``Walker2d-v2``, ``HalfCheetah-v2`` through gym / MuJoCo (``env/utils.py:15``).  All four are PLANAR
mechanisms, so one small engine covers them: a tree of rigid links joined by hinges, a free-floating root, in
reduced coordinates ``q = (x, y, root angle, joint angles...)``:

* mass matrix ``M(q) = sum_i m_i Jv_i' Jv_i + I_i Jw_i' Jw_i`` (+ rotor inertia on the joint diagonal) from the
  links' centre-of-mass Jacobians, bias forces from the centripetal accelerations of the chain, gravity, joint
  damping, joint limits as stiff springs;
* contact points (feet, heels, the torso) against the ground (flat, or a piecewise-linear terrain with the force
  along its local normal and tangent) as spring-dampers with a friction cone
  (``|f_t| <= mu f_n``, viscous inside it);
* actuators: torque ``gear x action`` (MuJoCo tasks) or a torque-limited speed servo (the walker's Box2D motors:
  target speed ``sign(a) x SPEED``, torque limit ``80 |a|``);
* symplectic Euler with sub-steps; everything vectorised over the ``E`` environments of a :class:`VecEnv`
  (``M`` is ``[E, n, n]`` with ``n <= 9``, solved with one batched ``numpy.linalg.solve`` per sub-step).

The TASK definitions follow the gym ones — observation layout and scaling, reward terms and coefficients,
termination rules, episode limits, units — with link dimensions and masses taken from the published models'
geometry (capsule volumes x density 1000 for the MuJoCo bodies, polygon areas x the fixture densities for the
walker).  Same tasks, own simulator: returns are comparable in kind (a Hopper that hops forward collects about
its forward speed + 1 per step, a walker that reaches the far end about 300), not trajectory by trajectory, and
the walker's ground is the gym task's grass (a damped random walk of the slope after a flat start pad; contact
forces along the local normal and tangent; the 10 lidar readings are the exact ray / ground intersections), without the
obstacles of the hardcore variant.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .base import VecEnv


@dataclass
class Link:
    name: str
    parent: int                       # index of the parent link (-1: the root)
    anchor: Tuple[float, float]       # where this link's joint sits, in the parent's frame
    com: Tuple[float, float]          # centre of mass in the link's own frame (origin = its joint)
    mass: float
    inertia: float                    # about the centre of mass
    rest: float = 0.0                 # link angle relative to the parent at joint angle 0
    limits: Tuple[float, float] = (-np.inf, np.inf)
    damping: float = 0.0
    armature: float = 0.0
    stiffness: float = 0.0            # spring towards joint angle 0


@dataclass
class Model:
    links: List[Link]
    contacts: List[Tuple[int, Tuple[float, float]]]      # (link, point in the link's frame)
    gravity: float = 9.81
    k_contact: float = 20000.0
    c_contact: float = 300.0
    c_friction: float = 300.0
    mu: float = 1.0
    k_limit: float = 400.0
    c_limit: float = 4.0
    chains: List[List[int]] = field(default_factory=list)    # per link: the links from the root down to it

    def __post_init__(self):
        self.chains = []
        for i, l in enumerate(self.links):
            self.chains.append((self.chains[l.parent] if l.parent >= 0 else []) + [i])
        self.n = 2 + len(self.links)            # x, y, then one angle per link (the root's is absolute)


class PlanarSim:
    """State and dynamics of ``E`` copies of a :class:`Model`."""

    def __init__(self, model: Model, num_envs: int):
        self.m, self.E = model, int(num_envs)
        self.q = np.zeros((self.E, model.n))
        self.qd = np.zeros((self.E, model.n))
        self.contact_on = np.zeros((self.E, len(model.contacts)), bool)
        self.terrain: Optional[np.ndarray] = None        # [E, J] ground heights at x = j * terrain_step (None: flat, y = 0)
        self.terrain_step = 1.0

    def ground(self, x: np.ndarray, env: Optional[np.ndarray] = None):
        """Height and slope of the ground under ``x [E]`` (piecewise linear between the terrain's samples)."""
        if self.terrain is None:
            return np.zeros_like(x), np.zeros_like(x)
        J = self.terrain.shape[1]
        rows = np.arange(self.terrain.shape[0]) if env is None else env
        j = np.clip(np.floor(x / self.terrain_step), 0, J - 2).astype(np.int64)
        h0, h1 = self.terrain[rows, j], self.terrain[rows, j + 1]
        slope = (h1 - h0) / self.terrain_step
        return h0 + slope * (x - j * self.terrain_step), slope

    # ---- kinematics -----------------------------------------------------------------------------
    def frames(self, q: Optional[np.ndarray] = None):
        """World angle ``[E, L]`` and joint origin ``[E, L, 2]`` of every link."""
        q = self.q if q is None else q
        L = len(self.m.links)
        ang = np.zeros((q.shape[0], L))
        org = np.zeros((q.shape[0], L, 2))
        for i, l in enumerate(self.m.links):
            if l.parent < 0:
                ang[:, i] = q[:, 2] + l.rest
                org[:, i] = q[:, :2]
            else:
                pa = ang[:, l.parent]
                c, s = np.cos(pa), np.sin(pa)
                org[:, i, 0] = org[:, l.parent, 0] + c * l.anchor[0] - s * l.anchor[1]
                org[:, i, 1] = org[:, l.parent, 1] + s * l.anchor[0] + c * l.anchor[1]
                ang[:, i] = pa + l.rest + q[:, 2 + i]
        return ang, org

    @staticmethod
    def _world(ang_i, org_i, local):
        c, s = np.cos(ang_i), np.sin(ang_i)
        return np.stack([org_i[:, 0] + c * local[0] - s * local[1], org_i[:, 1] + s * local[0] + c * local[1]], axis=1)

    def point(self, link: int, local, q: Optional[np.ndarray] = None) -> np.ndarray:
        ang, org = self.frames(q)
        return self._world(ang[:, link], org[:, link], local)

    def _jacobian(self, link: int, p: np.ndarray, org: np.ndarray) -> np.ndarray:
        """``d p / d q`` ``[E, 2, n]`` of a point ``p`` rigidly attached to ``link``."""
        J = np.zeros((p.shape[0], 2, self.m.n))
        J[:, 0, 0] = 1.0
        J[:, 1, 1] = 1.0
        for k in self.m.chains[link]:          # the angle coordinate of link k turns everything below its joint
            r = p - org[:, k]
            J[:, 0, 2 + k] = -r[:, 1]
            J[:, 1, 2 + k] = r[:, 0]
        return J

    def _bias_acc(self, link: int, local, ang, org, w) -> np.ndarray:
        """Acceleration of a point of ``link`` at zero generalised acceleration (centripetal terms of the chain)."""
        a = np.zeros((ang.shape[0], 2))
        chain = self.m.chains[link]
        for idx, k in enumerate(chain):
            nxt = org[:, chain[idx + 1]] if idx + 1 < len(chain) else self._world(ang[:, k], org[:, k], local)
            a -= (w[:, k] ** 2)[:, None] * (nxt - org[:, k])
        return a

    # ---- dynamics ---------------------------------------------------------------------------------
    def accelerations(self, tau: np.ndarray) -> np.ndarray:
        """``q''`` for joint torques ``tau [E, L]`` (entry 0, the root's, is ignored)."""
        m, q, qd = self.m, self.q, self.qd
        E, n, L = q.shape[0], m.n, len(m.links)
        ang, org = self.frames()
        w = np.zeros((E, L))
        for i, l in enumerate(m.links):
            w[:, i] = (w[:, l.parent] if l.parent >= 0 else 0.0) + qd[:, 2 + i]
        M = np.zeros((E, n, n))
        Q = np.zeros((E, n))
        for i, l in enumerate(m.links):
            p = self._world(ang[:, i], org[:, i], l.com)
            J = self._jacobian(i, p, org)
            M += l.mass * np.einsum("eki,ekj->eij", J, J)
            Jw = np.zeros((E, n))
            Jw[:, [2 + k for k in m.chains[i]]] = 1.0
            M += l.inertia * Jw[:, :, None] * Jw[:, None, :]
            a0 = self._bias_acc(i, l.com, ang, org, w)
            Q -= l.mass * np.einsum("eki,ek->ei", J, a0)
            Q[:, :] += np.einsum("eki,k->ei", J, np.array([0.0, -l.mass * m.gravity]))
            if i > 0:
                M[:, 2 + i, 2 + i] += l.armature
                lo, hi = l.limits
                ja, jv = q[:, 2 + i], qd[:, 2 + i]
                t = tau[:, i] - l.damping * jv - l.stiffness * ja
                over, under = ja - hi, lo - ja
                t = t - np.where(over > 0, m.k_limit * over + m.c_limit * jv, 0.0)
                t = t + np.where(under > 0, m.k_limit * under - m.c_limit * jv, 0.0)
                Q[:, 2 + i] += t
        for ci, (link, local) in enumerate(m.contacts):
            p = self._world(ang[:, link], org[:, link], local)
            J = self._jacobian(link, p, org)
            v = np.einsum("eki,ei->ek", J, qd)
            g, slope = self.ground(p[:, 0])
            r = np.sqrt(1.0 + slope * slope)                   # ground normal (-slope, 1) / r, tangent (1, slope) / r
            pen = (g - p[:, 1]) / r
            on = pen > 0.0
            vn, vt = (v[:, 1] - slope * v[:, 0]) / r, (v[:, 0] + slope * v[:, 1]) / r
            fn = np.where(on, np.maximum(0.0, m.k_contact * pen - m.c_contact * vn), 0.0)
            ft = np.where(on, np.clip(-m.c_friction * vt, -m.mu * fn, m.mu * fn), 0.0)
            Q += np.einsum("eki,ek->ei", J, np.stack([(ft - slope * fn) / r, (fn + slope * ft) / r], axis=1))
            self.contact_on[:, ci] |= on
        return np.linalg.solve(M, Q[..., None])[..., 0]

    def step(self, torque_fn, dt: float, substeps: int) -> None:
        """Advance by ``dt``; ``torque_fn(sim) -> tau [E, L]`` is evaluated at every sub-step (servo motors)."""
        h = dt / substeps
        self.contact_on[:] = False
        for _ in range(substeps):
            self.qd = self.qd + h * self.accelerations(torque_fn(self))
            self.q = self.q + h * self.qd

    def energy(self) -> np.ndarray:
        """Kinetic + potential energy (diagnostics / tests)."""
        ang, org = self.frames()
        E, L = self.q.shape[0], len(self.m.links)
        w = np.zeros((E, L))
        e = np.zeros(E)
        for i, l in enumerate(self.m.links):
            w[:, i] = (w[:, l.parent] if l.parent >= 0 else 0.0) + self.qd[:, 2 + i]
            p = self._world(ang[:, i], org[:, i], l.com)
            v = np.einsum("eki,ei->ek", self._jacobian(i, p, org), self.qd)
            e += 0.5 * l.mass * (v ** 2).sum(1) + 0.5 * l.inertia * w[:, i] ** 2 + l.mass * self.m.gravity * p[:, 1]
        return e


def draw_segments(segments, w: int = 300, h: int = 200, ground_y: Optional[int] = None) -> np.ndarray:
    """RGB frame with ``segments`` ``[((x0, y0), (x1, y1), colour), ...]`` in pixel coordinates (y down)."""
    img = np.zeros((h, w, 3), np.uint8)
    if ground_y is not None and ground_y < h:
        img[max(0, ground_y):, :] = (90, 90, 90)
    for (x0, y0), (x1, y1), col in segments:
        n = int(max(abs(x1 - x0), abs(y1 - y0), 1)) + 1
        xs = np.round(np.linspace(x0, x1, n)).astype(int)
        ys = np.round(np.linspace(y0, y1, n)).astype(int)
        ok = (xs >= 1) & (xs < w - 1) & (ys >= 1) & (ys < h - 1)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                img[ys[ok] + dy, xs[ok] + dx] = col
    return img


def render_planar(sim: "PlanarSim", e: int, ends, view: float = 4.0, w: int = 300, h: int = 200, row: float = 0.55) -> np.ndarray:
    """Stick figure of environment ``e``, camera on the root: every link from its hinge to ``ends[link]`` (a point in
    the link's frame) and on to the hinges of its children."""
    ang, org = sim.frames()
    scale = w / view
    cx, cy = sim.q[e, 0], max(0.35 * view * h / w, sim.q[e, 1])

    def px(p):
        return (w / 2 + (p[0] - cx) * scale, h * row - (p[1] - cy) * scale)
    palette = ((128, 102, 230), (230, 180, 60), (90, 200, 120), (230, 110, 110))
    segs = []
    for i, l in enumerate(sim.m.links):
        tip = sim._world(ang[e:e + 1, i], org[e:e + 1, i], ends[i])[0]
        segs.append((px(org[e, i]), px(tip), palette[i % len(palette)]))
        for j, c in enumerate(sim.m.links):
            if c.parent == i:
                segs.append((px(org[e, i]), px(org[e, j]), palette[i % len(palette)]))
    if sim.terrain is None:
        return draw_segments(segs, w, h, ground_y=int(px((0.0, 0.0))[1]))
    step = sim.terrain_step                                  # the ground as the polyline it is
    j_lo = max(0, int((cx - view / 2) / step))
    j_hi = min(sim.terrain.shape[1] - 2, int((cx + view / 2) / step) + 1)
    for j in range(j_lo, j_hi + 1):
        segs.append((px((j * step, sim.terrain[e, j])), px(((j + 1) * step, sim.terrain[e, j + 1])), (90, 90, 90)))
    return draw_segments(segs, w, h)


def _capsule(length: float, radius: float, density: float = 1000.0):
    """Mass and central inertia (about the axis normal to the plane) of a capsule lying in the plane."""
    m_cyl = density * np.pi * radius ** 2 * length
    m_sph = density * 4.0 / 3.0 * np.pi * radius ** 3
    i_cyl = m_cyl * (length ** 2 / 12.0 + radius ** 2 / 4.0)
    i_sph = m_sph * (0.4 * radius ** 2 + (length / 2) ** 2 + 0.375 * radius * length)
    return m_cyl + m_sph, i_cyl + i_sph


# ======================================================================================= BipedalWalker
class BipedalWalkerVec(VecEnv):
    """``BipedalWalker-v2`` (24 observations, 4 actions: hip and knee of each leg)."""
    FPS, SCALE = 50.0, 30.0
    LEG_W, LEG_H, LEG_DOWN = 8 / 30.0, 34 / 30.0, -8 / 30.0
    MOTORS_TORQUE, SPEED_HIP, SPEED_KNEE = 80.0, 4.0, 6.0
    LIDAR_RANGE = 160 / 30.0
    TRACK_END = (200 - 10) * 14 / 30.0
    START_X = 14 / 30.0 * 20 / 2
    SERVO_GAIN = 6.0
    SUBSTEPS = 8
    TERRAIN_STEP, TERRAIN_LENGTH, TERRAIN_STARTPAD, LIDAR_SEGMENTS = 14 / 30.0, 200, 20, 13

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1600, env_name: str = "BipedalWalker-v2",
                 terrain: bool = True):
        super().__init__(env_name, num_envs, 24, 4, -1.0, 1.0, seed)
        self.time_limit = int(time_limit)
        self.hilly = bool(terrain)
        hull_m, hull_i = 5.42, 2.2                       # polygon area 976 px^2 / 900 x density 5
        th_m = self.LEG_W * self.LEG_H * 1.0
        sh_m = 0.8 * self.LEG_W * self.LEG_H * 1.0
        th_i = th_m * (self.LEG_W ** 2 + self.LEG_H ** 2) / 12
        sh_i = sh_m * ((0.8 * self.LEG_W) ** 2 + self.LEG_H ** 2) / 12
        H2 = self.LEG_H / 2
        links = [Link("hull", -1, (0, 0), (0, 0), hull_m, hull_i)]
        for leg in (0, 1):
            links.append(Link(f"thigh{leg}", 0, (0.0, self.LEG_DOWN), (0.0, -H2), th_m, th_i, limits=(-0.8, 1.1)))
            links.append(Link(f"shin{leg}", len(links) - 1, (0.0, -self.LEG_H), (0.0, -H2), sh_m, sh_i, limits=(-1.6, -0.1)))
        contacts = [(2, (0.0, -self.LEG_H)), (4, (0.0, -self.LEG_H)),                       # the feet
                    (0, (-1.0, -8 / 30.0)), (0, (34 / 30.0, -8 / 30.0)), (0, (-1.0, 9 / 30.0)), (0, (34 / 30.0, 1 / 30.0))]
        self.model = Model(links, contacts, gravity=10.0, k_contact=8000.0, c_contact=150.0, c_friction=150.0, mu=2.5,
                           k_limit=600.0, c_limit=6.0)
        self.sim = PlanarSim(self.model, self.num_envs)
        self.sim.terrain = np.zeros((self.num_envs, self.TERRAIN_LENGTH))
        self.sim.terrain_step = self.TERRAIN_STEP
        self.t = np.zeros(self.num_envs, np.int64)
        self.prev_shaping = np.zeros(self.num_envs)
        self._act = np.zeros((self.num_envs, 4))

    def _new_terrain(self, e: int) -> None:
        """The gym task's grass: a flat start pad, then heights that follow a damped random walk of the slope
        (``velocity = 0.8 velocity + 0.01 sign(-y) + U(-1, 1) / 30`` per 0.467 of track)."""
        h = self.sim.terrain[e]
        h[:] = 0.0
        if not self.hilly:
            return
        noise = self.rng.uniform(-1.0, 1.0, self.TERRAIN_LENGTH) / self.SCALE
        y = velocity = 0.0
        for i in range(self.TERRAIN_LENGTH):
            velocity = 0.8 * velocity + 0.01 * np.sign(-y)
            if i > self.TERRAIN_STARTPAD:
                velocity += noise[i]
            y += velocity
            h[i] = y

    def lidar(self, idx=slice(None)) -> np.ndarray:
        """Fractions ``[E, 10]`` of the lidar range at which the 10 rays from the hull (0 to 1.35 rad from straight
        down, towards the front, world axes) meet the ground; 1 where they do not."""
        sim = self.sim
        q = np.atleast_2d(sim.q[idx])
        ter = np.atleast_2d(sim.terrain[idx])
        E, J, step = q.shape[0], ter.shape[1], self.TERRAIN_STEP
        ang = 1.5 * np.arange(10) / 10.0
        dx, dy = np.sin(ang) * self.LIDAR_RANGE, -np.cos(ang) * self.LIDAR_RANGE             # [10]
        x0, y0 = q[:, 0], q[:, 1]
        j0 = np.clip(np.floor(x0 / step), 0, J - 2).astype(np.int64)
        best = np.ones((E, 10))
        rows = np.arange(E)
        for k in range(self.LIDAR_SEGMENTS):
            j = np.clip(j0 + k, 0, J - 2)
            xj = j * step
            h0, h1 = ter[rows, j], ter[rows, j + 1]
            s = (h1 - h0) / step
            t = (y0 - h0 - s * (x0 - xj))[:, None] / (s[:, None] * dx[None, :] - dy[None, :])
            xh = x0[:, None] + t * dx[None, :]
            lo = np.where((j == 0) | (k == 0), -np.inf, xj)[:, None]          # the first segment also covers what lies behind
            hi = np.where(j == J - 2, np.inf, xj + step)[:, None]
            ok = (t >= 0.0) & (xh >= lo) & (xh <= hi)
            best = np.where(ok & (t < best), t, best)
        return np.clip(best, 0.0, 1.0)

    def _torques(self, sim: PlanarSim) -> np.ndarray:
        a = self._act
        tau = np.zeros((sim.E, 5))
        for j, (link, speed) in enumerate(((1, self.SPEED_HIP), (2, self.SPEED_KNEE), (3, self.SPEED_HIP), (4, self.SPEED_KNEE))):
            target = speed * np.sign(a[:, j])
            limit = self.MOTORS_TORQUE * np.clip(np.abs(a[:, j]), 0.0, 1.0)
            tau[:, link] = np.clip(self.SERVO_GAIN * self.MOTORS_TORQUE * (target - sim.qd[:, 2 + link]) / speed, -limit, limit)
        return tau

    def _obs(self, idx=slice(None)) -> np.ndarray:
        q, qd, on = self.sim.q[idx], self.sim.qd[idx], self.sim.contact_on[idx]
        q, qd, on = np.atleast_2d(q), np.atleast_2d(qd), np.atleast_2d(on)
        lidar = self.lidar(idx)
        o = np.concatenate([
            q[:, 2:3], 2.0 * qd[:, 2:3] / self.FPS, 0.3 * qd[:, 0:1] * (600 / self.SCALE) / self.FPS,
            0.3 * qd[:, 1:2] * (400 / self.SCALE) / self.FPS,
            q[:, 3:4], qd[:, 3:4] / self.SPEED_HIP, q[:, 4:5] + 1.0, qd[:, 4:5] / self.SPEED_KNEE, on[:, 0:1].astype(float),
            q[:, 5:6], qd[:, 5:6] / self.SPEED_HIP, q[:, 6:7] + 1.0, qd[:, 6:7] / self.SPEED_KNEE, on[:, 1:2].astype(float),
            lidar], axis=1).astype(np.float32)
        return o if not isinstance(idx, (int, np.integer)) else o[0]

    def _shaping(self) -> np.ndarray:
        return 130.0 * self.sim.q[:, 0] / self.SCALE - 5.0 * np.abs(self.sim.q[:, 2])

    def render(self, e: int = 0) -> np.ndarray:
        ends = [(34 / 30.0, 0.0)] + [(0.0, -self.LEG_H)] * 4
        return render_planar(self.sim, e, ends, view=8.0, row=0.4)

    def _reset_env(self, e: int) -> np.ndarray:
        s = self.sim
        self._new_terrain(e)
        s.q[e] = 0.0
        s.qd[e] = 0.0
        s.q[e, 0] = self.START_X
        s.q[e, 1] = 2 * self.LEG_H - self.LEG_DOWN + 0.02
        s.q[e, 3], s.q[e, 5] = 0.05, -0.05               # the two legs start slightly apart
        s.q[e, 4] = s.q[e, 6] = -0.1
        s.qd[e, 0] = self.rng.uniform(-5.0, 5.0) / 5.42 / self.FPS * 10     # the initial random push on the hull
        s.contact_on[e] = False
        self.t[e] = 0
        self.prev_shaping[e] = 130.0 * s.q[e, 0] / self.SCALE - 5.0 * abs(s.q[e, 2])
        return self._obs(e)

    def _step(self, actions: np.ndarray):
        self._act = np.clip(actions.astype(np.float64), -1.0, 1.0)
        self.sim.step(self._torques, 1.0 / self.FPS, self.SUBSTEPS)
        self.t += 1
        shaping = self._shaping()
        reward = shaping - self.prev_shaping - 0.00035 * self.MOTORS_TORQUE * np.clip(np.abs(actions), 0, 1).sum(axis=1)
        self.prev_shaping = shaping
        x = self.sim.q[:, 0]
        hull_down = self.sim.contact_on[:, 2:].any(axis=1)
        failed = hull_down | (x < 0.0) | ~np.isfinite(self.sim.q).all(axis=1)
        reward = np.where(failed, -100.0, reward)
        finished = x > self.TRACK_END
        limit = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        for e in np.nonzero(~np.isfinite(self.sim.q).all(axis=1))[0]:      # a blown-up state never reaches the policy
            self.sim.q[e], self.sim.qd[e] = 0.0, 0.0
        return self._obs(), reward, failed | finished | limit


# =================================================================================== MuJoCo-style tasks
class _MujocoPlanarVec(VecEnv):
    """Shared part of Hopper / Walker2d / HalfCheetah: observation ``[qpos[1:], clip(qvel, -10, 10)]``, torque
    actuators ``gear x action``, reward ``forward speed (+ alive bonus) - ctrl_cost |a|^2``."""
    DT, SUBSTEPS = 0.008, 8
    ALIVE, CTRL_COST = 1.0, 1e-3
    RESET_NOISE = 5e-3

    def __init__(self, env_name, num_envs, model: Model, gears: Sequence[float], q0_y: float, seed, time_limit):
        super().__init__(env_name, num_envs, 2 * model.n - 1, len(gears), -1.0, 1.0, seed)
        self.model, self.gears = model, np.asarray(gears, float)
        self.sim = PlanarSim(model, self.num_envs)
        self.q0_y = float(q0_y)
        self.time_limit = int(time_limit)
        self.t = np.zeros(self.num_envs, np.int64)
        self._act = np.zeros((self.num_envs, len(gears)))

    def _torques(self, sim: PlanarSim) -> np.ndarray:
        tau = np.zeros((sim.E, len(self.model.links)))
        tau[:, 1:] = self.gears[None, :] * self._act
        return tau

    def _obs(self, idx=slice(None)) -> np.ndarray:
        q, qd = np.atleast_2d(self.sim.q[idx]), np.atleast_2d(self.sim.qd[idx])
        o = np.concatenate([q[:, 1:], np.clip(qd, -10.0, 10.0)], axis=1).astype(np.float32)
        return o if not isinstance(idx, (int, np.integer)) else o[0]

    def _reset_env(self, e: int) -> np.ndarray:
        n = self.model.n
        self.sim.q[e] = self.rng.uniform(-self.RESET_NOISE, self.RESET_NOISE, n)
        self.sim.q[e, 1] += self.q0_y
        self.sim.qd[e] = self.rng.uniform(-self.RESET_NOISE, self.RESET_NOISE, n)
        self.sim.contact_on[e] = False
        self.t[e] = 0
        return self._obs(e)

    def _unhealthy(self) -> np.ndarray:
        return np.zeros(self.num_envs, bool)

    def render(self, e: int = 0) -> np.ndarray:
        ends = [tuple(2.0 * np.asarray(l.com)) for l in self.model.links]        # centres of mass sit at mid-length
        return render_planar(self.sim, e, ends, view=4.0)

    def _step(self, actions: np.ndarray):
        self._act = np.clip(actions.astype(np.float64), -1.0, 1.0)
        x0 = self.sim.q[:, 0].copy()
        self.sim.step(self._torques, self.DT, self.SUBSTEPS)
        self.t += 1
        bad = ~np.isfinite(self.sim.q).all(axis=1) | ~np.isfinite(self.sim.qd).all(axis=1)
        for e in np.nonzero(bad)[0]:
            self.sim.q[e], self.sim.qd[e] = 0.0, 0.0
            self.sim.q[e, 1] = self.q0_y
        speed = np.where(bad, 0.0, (self.sim.q[:, 0] - x0) / self.DT)
        reward = self.ALIVE + speed - self.CTRL_COST * np.square(self._act).sum(axis=1)
        fell = self._unhealthy() | bad
        limit = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        return self._obs(), reward, fell | limit


def _leg_links(links: List[Link], root: int, hip_y: float, lens, radii, foot, limits, damping=1.0, armature=0.01):
    """Thigh, shin and foot hanging from ``root`` at ``(0, hip_y)``; ``foot = (back, front)`` along x."""
    th_m, th_i = _capsule(lens[0], radii[0])
    sh_m, sh_i = _capsule(lens[1], radii[1])
    ft_len = foot[1] - foot[0]
    ft_m, ft_i = _capsule(ft_len, radii[2])
    links.append(Link("thigh", root, (0.0, hip_y), (0.0, -lens[0] / 2), th_m, th_i, limits=limits[0], damping=damping, armature=armature))
    links.append(Link("shin", len(links) - 1, (0.0, -lens[0]), (0.0, -lens[1] / 2), sh_m, sh_i, limits=limits[1], damping=damping, armature=armature))
    links.append(Link("foot", len(links) - 1, (0.0, -lens[1]), ((foot[0] + foot[1]) / 2, 0.0), ft_m, ft_i, limits=limits[2], damping=damping, armature=armature))
    return len(links) - 1


class HopperVec(_MujocoPlanarVec):
    """``Hopper-v2`` (11 observations, 3 actions; torso 0.4, thigh 0.45, leg 0.5, foot 0.39 long)."""

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1000, env_name: str = "Hopper-v2"):
        tm, ti = _capsule(0.4, 0.05)
        links = [Link("torso", -1, (0, 0), (0.0, 0.2), tm, ti)]
        foot = _leg_links(links, 0, 0.0, (0.45, 0.5), (0.05, 0.04, 0.06), (-0.13, 0.26),
                          ((-2.618, 0.0), (-2.618, 0.0), (-0.785, 0.785)))
        contacts = [(foot, (-0.13, -0.06)), (foot, (0.26, -0.06)), (0, (0.0, 0.4)), (0, (0.0, 0.0))]
        model = Model(links, contacts, k_contact=30000.0, c_contact=500.0, c_friction=500.0, mu=2.0)
        super().__init__(env_name, num_envs, model, (200.0, 200.0, 200.0), 0.45 + 0.5 + 0.06 + 0.005, seed, time_limit)
        self.torso_top = 0.4

    def _unhealthy(self) -> np.ndarray:
        z = self.sim.q[:, 1] + 0.2                       # MuJoCo's rootz is the torso centre: 1.25 when standing
        return (z < 0.7) | (np.abs(self.sim.q[:, 2]) > 0.2)


class Walker2dVec(_MujocoPlanarVec):
    """``Walker2d-v2`` (17 observations, 6 actions; two Hopper-like legs, gear 100)."""

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1000, env_name: str = "Walker2d-v2"):
        tm, ti = _capsule(0.4, 0.05)
        links = [Link("torso", -1, (0, 0), (0.0, 0.2), tm, ti)]
        feet = [_leg_links(links, 0, 0.0, (0.45, 0.5), (0.05, 0.04, 0.06), (-0.0, 0.2),
                           ((-2.618, 0.0), (-2.618, 0.0), (-0.785, 0.785)), damping=0.1) for _ in range(2)]
        contacts = [(f, p) for f in feet for p in ((0.0, -0.06), (0.2, -0.06))] + [(0, (0.0, 0.4)), (0, (0.0, 0.0))]
        model = Model(links, contacts, k_contact=30000.0, c_contact=500.0, c_friction=500.0, mu=0.9)
        super().__init__(env_name, num_envs, model, (100.0,) * 6, 0.45 + 0.5 + 0.06 + 0.005, seed, time_limit)

    def _unhealthy(self) -> np.ndarray:
        z = self.sim.q[:, 1] + 0.2
        return (z < 0.8) | (z > 2.0) | (np.abs(self.sim.q[:, 2]) > 1.0)


class HalfCheetahVec(_MujocoPlanarVec):
    """``HalfCheetah-v2`` (17 observations, 6 actions; no termination, control cost 0.1, no alive bonus)."""
    DT, SUBSTEPS = 0.05, 12
    ALIVE, CTRL_COST = 0.0, 0.1
    RESET_NOISE = 0.1

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1000, env_name: str = "HalfCheetah-v2"):
        tm, ti = _capsule(1.0, 0.046)
        hm, hi = _capsule(0.3, 0.046)
        # torso along x from the back hip (-0.5) to the front hip (+0.5), the head as extra mass in front
        links = [Link("torso", -1, (0, 0), (0.05, 0.02), tm + hm, ti + hm * 0.6 ** 2)]

        def leg(x_hip, lens, radii, rests, limits, dampings, stiff):
            parent = 0
            anchor = (x_hip, 0.0)
            for k in range(3):
                m, i = _capsule(lens[k], radii[k])
                links.append(Link("seg", parent, anchor, (0.0, -lens[k] / 2), m, i, rest=rests[k], limits=limits[k],
                                  damping=dampings[k], armature=0.1, stiffness=stiff[k]))
                parent = len(links) - 1
                anchor = (0.0, -lens[k])
            return parent
        bfoot = leg(-0.5, (0.29, 0.3, 0.188), (0.046,) * 3, (-0.4, 0.9, -0.2),
                    ((-0.52, 1.05), (-0.785, 0.785), (-0.4, 0.785)), (6.0, 4.5, 3.0), (240.0, 180.0, 120.0))
        ffoot = leg(0.5, (0.266, 0.212, 0.14), (0.046,) * 3, (0.5, -1.1, 0.5),
                    ((-1.0, 0.7), (-1.2, 0.87), (-0.5, 0.5)), (4.5, 3.0, 1.5), (180.0, 120.0, 60.0))
        contacts = [(bfoot, (0.0, -0.188)), (ffoot, (0.0, -0.14)), (bfoot - 1, (0.0, -0.3)), (ffoot - 1, (0.0, -0.212)),
                    (0, (-0.5, -0.046)), (0, (0.5, -0.046)), (0, (0.75, 0.1))]
        model = Model(links, contacts, k_contact=30000.0, c_contact=500.0, c_friction=500.0, mu=0.4)
        super().__init__(env_name, num_envs, model, (120.0, 90.0, 60.0, 120.0, 60.0, 30.0), 0.72, seed, time_limit)
