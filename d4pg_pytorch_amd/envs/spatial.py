"""Ant without MuJoCo: a small 3-D articulated-body simulator and the ``Ant-v2`` task on it.

The reference runs ``Ant-v2`` through gym / MuJoCo (``configs/ant_d4pg.yml``, ``env/utils.py:15``).  The ant is
a free-floating torso with four two-joint legs, so the engine is the spatial counterpart of ``planar.py``:

* state: torso position, orientation quaternion ``(w, x, y, z)`` and hinge angles; generalised velocity
  ``u = (v, w, q')``: linear velocity of the torso origin, angular velocity in world axes, hinge rates;
* kinematics down the link tree (rotation about the hinge axis by Rodrigues' formula), Jacobians of all centres
  of mass and contact points in one batch, mass matrix ``M = sum m Jv'Jv + Jw' (R I R') Jw`` (+ rotor inertia),
  bias forces from the accelerations at ``u' = 0`` (centripetal / Coriolis / gyroscopic), gravity, joint damping,
  joint limits as stiff spring-dampers;
* contact: spheres (the capsule ends, the torso) on flat ground, normal spring-damper and viscous friction capped
  by the friction cone, applied at the sphere's lowest point;
* symplectic Euler with sub-steps, quaternion advanced by the world angular velocity and renormalised; vectorised
  over the ``E`` environments (one batched 14 x 14 solve per sub-step).

The TASK follows the gym one: 111 observations ``[qpos[2:] (13), qvel (14, angular part in torso axes),
contact wrench of the 14 bodies clipped to +-1 (84)]``, 8 torques (gear 150) in the order hip / ankle of legs
1..4, reward ``forward speed + 1 - 0.5 |a|^2 - 0.5e-3 |clipped contact forces|^2``, terminal when the torso
height leaves ``[0.2, 1.0]``, 0.05 s per step, 1000-step episodes; link geometry, densities (5), rotor inertia (1),
joint damping (1) and joint ranges from the published model.  Same task, own simulator: returns are comparable in
kind (an ant that walks forward collects about its speed + 1 per step), not trajectory by trajectory.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .base import VecEnv


@dataclass
class Link3:
    name: str
    parent: int                                   # -1: the free root
    anchor: Tuple[float, float, float]            # the hinge, in the parent's frame
    axis: Tuple[float, float, float]              # hinge axis in the link's (= parent's at angle 0) frame
    com: Tuple[float, float, float]               # centre of mass in the link's frame (origin = the hinge)
    mass: float
    inertia: np.ndarray                           # 3 x 3 about the centre of mass, link axes
    limits: Tuple[float, float] = (-np.inf, np.inf)
    damping: float = 0.0
    armature: float = 0.0


@dataclass
class Model3:
    links: List[Link3]
    contacts: List[Tuple[int, Tuple[float, float, float], float]]     # (link, sphere centre in the link's frame, radius)
    gravity: float = 9.81
    k_contact: float = 3000.0
    c_contact: float = 30.0
    c_friction: float = 30.0
    mu: float = 1.0
    k_limit: float = 2500.0
    c_limit: float = 100.0
    depends: np.ndarray = field(default=None)     # [L, L]: link i moves with hinge k
    levels: list = field(default=None)

    def __post_init__(self):
        L = len(self.links)
        self.depends = np.zeros((L, L), bool)
        for i, l in enumerate(self.links):
            if l.parent >= 0:
                self.depends[i] = self.depends[l.parent]
                self.depends[i, i] = True
        self.n = 6 + (L - 1)
        depth = [0] * L
        for i, l in enumerate(self.links):
            if l.parent >= 0:
                assert l.parent < i, "links are listed parents first"
                depth[i] = depth[l.parent] + 1
        # links of equal depth are advanced together: (their indices, their parents' indices) per level below the root
        self.levels = [(np.array([i for i in range(L) if depth[i] == d]), np.array([self.links[i].parent for i in range(L) if depth[i] == d]))
                       for d in range(1, max(depth) + 1)]


def _skew(v: np.ndarray) -> np.ndarray:
    z = np.zeros_like(v[..., 0])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1), np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def _cross(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``a x b`` over the last axis (numpy's own spends most of its time rearranging axes at these sizes)."""
    out = np.empty(np.broadcast_shapes(a.shape, b.shape))
    out[..., 0] = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    out[..., 1] = a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2]
    out[..., 2] = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    return out


def quat_to_matrix(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


class SpatialSim:
    """State and dynamics of ``E`` copies of a :class:`Model3`."""

    def __init__(self, model: Model3, num_envs: int):
        self.m, self.E = model, int(num_envs)
        L = len(model.links)
        self.pos = np.zeros((self.E, 3))
        self.quat = np.zeros((self.E, 4))
        self.quat[:, 0] = 1.0
        self.qj = np.zeros((self.E, L - 1))
        self.u = np.zeros((self.E, model.n))
        self.contact_force = np.zeros((self.E, len(model.contacts), 3))
        self.implicit_contact = True
        self._K = np.stack([_skew(np.asarray(l.axis, float)) for l in model.links])          # [L, 3, 3]
        self._mass = np.array([l.mass for l in model.links])
        self._inertia = np.stack([np.asarray(l.inertia, float) for l in model.links])
        self._com = np.array([l.com for l in model.links], float)
        self._c_link = np.array([c[0] for c in model.contacts], int)
        self._c_local = np.array([c[1] for c in model.contacts], float)
        self._c_rad = np.array([c[2] for c in model.contacts], float)
        self._lo = np.array([l.limits[0] for l in model.links[1:]])
        self._hi = np.array([l.limits[1] for l in model.links[1:]])
        self._damp = np.array([l.damping for l in model.links[1:]])
        self._arm = np.array([l.armature for l in model.links[1:]])
        self._KK = self._K @ self._K
        self._eye = np.eye(3)
        self._anchor = np.array([l.anchor for l in model.links], float)
        self._axis = np.array([l.axis for l in model.links], float)
        self._pt_link = np.concatenate([np.arange(L), self._c_link])
        self._dep = model.depends[None, :, 1:, None].astype(float)
        self._mass3 = np.repeat(self._mass, 3)
        self._jd = np.arange(6, model.n)
        self._g = np.array([0.0, 0.0, -model.gravity])

    # ---- kinematics -----------------------------------------------------------------------------
    def frames(self):
        """Rotation ``[E, L, 3, 3]``, hinge position ``[E, L, 3]`` and world hinge axis ``[E, L, 3]`` of every link."""
        m, E, L = self.m, self.E, len(self.m.links)
        R = np.zeros((E, L, 3, 3))
        o = np.zeros((E, L, 3))
        ax = np.zeros((E, L, 3))
        R[:, 0] = quat_to_matrix(self.quat)
        o[:, 0] = self.pos
        for idx, par in m.levels:
            Rp = R[:, par]                                                            # [E, k, 3, 3]
            th = self.qj[:, idx - 1]
            rot = self._eye + np.sin(th)[:, :, None, None] * self._K[idx] + (1 - np.cos(th))[:, :, None, None] * self._KK[idx]
            R[:, idx] = Rp @ rot
            o[:, idx] = o[:, par] + (Rp @ self._anchor[idx][None, :, :, None])[..., 0]
            ax[:, idx] = (Rp @ self._axis[idx][None, :, :, None])[..., 0]
        return R, o, ax

    def _jacobians(self, pts: np.ndarray, link_of: np.ndarray, o: np.ndarray, ax: np.ndarray) -> np.ndarray:
        """``d p / d u`` ``[E, P, 3, n]`` of points ``pts [E, P, 3]`` attached to links ``link_of [P]``."""
        E, P = pts.shape[:2]
        J = np.zeros((E, P, 3, self.m.n))
        J[:, :, 0, 0] = J[:, :, 1, 1] = J[:, :, 2, 2] = 1.0
        r0 = pts - o[:, None, 0]
        J[:, :, 0, 4], J[:, :, 0, 5] = r0[:, :, 2], -r0[:, :, 1]                      # -skew(r0)
        J[:, :, 1, 3], J[:, :, 1, 5] = -r0[:, :, 2], r0[:, :, 0]
        J[:, :, 2, 3], J[:, :, 2, 4] = r0[:, :, 1], -r0[:, :, 0]
        r = pts[:, :, None, :] - o[:, None, 1:, :]                                    # [E, P, L-1, 3]
        c = _cross(ax[:, None, 1:, :], r) * self.m.depends[link_of][None, :, 1:, None]
        J[:, :, :, 6:] = np.swapaxes(c, 2, 3)
        return J

    def link_velocities(self, ax: np.ndarray) -> np.ndarray:
        """World angular velocity ``[E, L, 3]`` of every link."""
        w = np.zeros((self.E, len(self.m.links), 3))
        w[:, 0] = self.u[:, 3:6]
        for idx, par in self.m.levels:
            w[:, idx] = w[:, par] + ax[:, idx] * self.u[:, 6 + idx - 1, None]
        return w

    # ---- dynamics ---------------------------------------------------------------------------------
    def accelerations(self, tau: np.ndarray, h: float = 0.0) -> np.ndarray:
        """``u'`` for hinge torques ``tau [E, L-1]``.  With a step ``h > 0`` the contact spring-dampers are taken
        implicitly in the velocity (``M + h J'CJ + h^2 J'KJ`` on the left-hand side): light links on stiff ground
        stay stable at a step the free dynamics are happy with."""
        m, E, L = self.m, self.E, len(self.m.links)
        R, o, ax = self.frames()
        w = self.link_velocities(ax)
        com = o + (R @ self._com[None, :, :, None])[..., 0]
        cpt = o[:, self._c_link] + (R[:, self._c_link] @ self._c_local[None, :, :, None])[..., 0]
        cpt[:, :, 2] -= self._c_rad                                                   # the spheres' lowest points
        pts = np.concatenate([com, cpt], axis=1)
        J = self._jacobians(pts, self._pt_link, o, ax)
        Jc, Jk = J[:, :L], J[:, L:]
        Jw = np.zeros((E, L, 3, m.n))
        Jw[:, :, 0, 3] = Jw[:, :, 1, 4] = Jw[:, :, 2, 5] = 1.0
        Jw[:, :, :, 6:] = np.swapaxes(ax[:, None, 1:, :] * self._dep, 2, 3)
        Iw = R @ self._inertia @ np.swapaxes(R, 2, 3)
        Jc2 = Jc.reshape(E, 3 * L, m.n)
        M = np.swapaxes(Jc2, 1, 2) @ (Jc2 * self._mass3[None, :, None])
        IJ = (Iw @ Jw).reshape(E, 3 * L, m.n)
        M += np.swapaxes(Jw.reshape(E, 3 * L, m.n), 1, 2) @ IJ
        M[:, self._jd, self._jd] += self._arm
        # accelerations at u' = 0, down the tree
        a_o = np.zeros((E, L, 3))
        alpha = np.zeros((E, L, 3))
        for idx, par in m.levels:
            r = o[:, idx] - o[:, par]
            wp = w[:, par]
            a_o[:, idx] = a_o[:, par] + _cross(alpha[:, par], r) + _cross(wp, _cross(wp, r))
            alpha[:, idx] = alpha[:, par] + _cross(wp, ax[:, idx]) * self.u[:, 6 + idx - 1, None]
        rc = com - o
        a_c = a_o + _cross(alpha, rc) + _cross(w, _cross(w, rc))
        f_lin = self._mass[None, :, None] * (self._g - a_c)
        Iw_w = (Iw @ w[..., None])[..., 0]
        t_rot = -((Iw @ alpha[..., None])[..., 0] + _cross(w, Iw_w))
        Q = (np.swapaxes(Jc2, 1, 2) @ f_lin.reshape(E, 3 * L, 1))[..., 0]
        Q += (np.swapaxes(Jw.reshape(E, 3 * L, m.n), 1, 2) @ t_rot.reshape(E, 3 * L, 1))[..., 0]
        qd = self.u[:, 6:]
        t = tau - self._damp * qd
        over, under = self.qj - self._hi, self._lo - self.qj
        t = t - np.where(over > 0, m.k_limit * over + m.c_limit * qd, 0.0)
        t = t + np.where(under > 0, m.k_limit * under - m.c_limit * qd, 0.0)
        Q[:, 6:] += t
        # contacts
        v = (Jk @ self.u[:, None, :, None])[..., 0]                                   # [E, C, 3]
        pen = -cpt[:, :, 2]
        on = pen > 0.0
        fn = np.where(on, m.k_contact * pen - m.c_contact * v[:, :, 2], 0.0)
        push = fn > 0.0
        fn = np.where(push, fn, 0.0)
        ft = -m.c_friction * v[:, :, :2]
        norm = np.sqrt((ft * ft).sum(2))
        cap = m.mu * fn
        slide = norm > cap
        ft = np.where(slide[:, :, None], ft * (cap / np.maximum(norm, 1e-12))[:, :, None], ft) * push[:, :, None]
        f = np.concatenate([ft, fn[:, :, None]], axis=2)
        C = Jk.shape[1]
        Q += (np.swapaxes(Jk.reshape(E, 3 * C, m.n), 1, 2) @ f.reshape(E, 3 * C, 1))[..., 0]
        if h > 0.0:
            gain = np.zeros((E, C, 3))
            gain[:, :, :2] = (h * m.c_friction * (push & ~slide))[:, :, None]
            gain[:, :, 2] = (h * m.c_contact + h * h * m.k_contact) * push
            Jk2 = Jk.reshape(E, 3 * C, m.n)
            M += np.swapaxes(Jk2, 1, 2) @ (Jk2 * gain.reshape(E, 3 * C, 1))
        self.contact_force = f
        self._contact_point, self._com_world = cpt, com
        return np.linalg.solve(M, Q[..., None])[..., 0]

    def step(self, tau: np.ndarray, dt: float, substeps: int) -> None:
        h = dt / substeps
        for _ in range(substeps):
            self.u = self.u + h * self.accelerations(tau, h if self.implicit_contact else 0.0)
            self.pos = self.pos + h * self.u[:, :3]
            w = self.u[:, 3:6]
            q = self.quat
            dq = 0.5 * np.stack([-(w * q[:, 1:]).sum(1),
                                 q[:, 0] * w[:, 0] + w[:, 1] * q[:, 3] - w[:, 2] * q[:, 2],
                                 q[:, 0] * w[:, 1] + w[:, 2] * q[:, 1] - w[:, 0] * q[:, 3],
                                 q[:, 0] * w[:, 2] + w[:, 0] * q[:, 2] - w[:, 1] * q[:, 1]], axis=1)
            q = q + h * dq
            self.quat = q / np.linalg.norm(q, axis=1, keepdims=True)
            self.qj = self.qj + h * self.u[:, 6:]

    def energy_and_momentum(self):
        """Kinetic + potential energy ``[E]``, linear momentum ``[E, 3]`` and angular momentum about the world
        origin ``[E, 3]`` (diagnostics / tests)."""
        L = len(self.m.links)
        R, o, ax = self.frames()
        w = self.link_velocities(ax)
        com = o + (R @ self._com[None, :, :, None])[..., 0]
        v = np.einsum("elki,ei->elk", self._jacobians(com, np.arange(L), o, ax), self.u)
        Iw = R @ self._inertia[None] @ np.swapaxes(R, 2, 3)
        Iw_w = np.einsum("elij,elj->eli", Iw, w)
        mass = self._mass[None, :]
        e = (0.5 * mass * (v ** 2).sum(2) + 0.5 * (w * Iw_w).sum(2) + mass * self.m.gravity * com[:, :, 2]).sum(1)
        p = (mass[:, :, None] * v).sum(1)
        ang = (_cross(com, mass[:, :, None] * v) + Iw_w).sum(1)
        return e, p, ang


def capsule3(p0, p1, radius: float, density: float):
    """Mass, centre and central inertia tensor of a capsule from ``p0`` to ``p1``."""
    p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
    length = np.linalg.norm(p1 - p0)
    d = (p1 - p0) / length
    m_cyl = density * np.pi * radius ** 2 * length
    m_sph = density * 4.0 / 3.0 * np.pi * radius ** 3
    axial = m_cyl * radius ** 2 / 2 + m_sph * 0.4 * radius ** 2
    trans = m_cyl * (length ** 2 / 12 + radius ** 2 / 4) + m_sph * (0.4 * radius ** 2 + (length / 2) ** 2 + 0.375 * radius * length)
    dd = np.outer(d, d)
    return m_cyl + m_sph, (p0 + p1) / 2, trans * (np.eye(3) - dd) + axial * dd


def _compose(parts):
    """Mass, centre of mass and central inertia of rigidly joined ``(mass, centre, inertia)`` parts."""
    mass = sum(p[0] for p in parts)
    com = sum(p[0] * np.asarray(p[1], float) for p in parts) / mass
    inertia = np.zeros((3, 3))
    for m, c, i in parts:
        r = np.asarray(c, float) - com
        inertia += i + m * ((r @ r) * np.eye(3) - np.outer(r, r))
    return mass, com, inertia


class AntVec(VecEnv):
    """``Ant-v2`` (111 observations, 8 actions)."""
    DT, SUBSTEPS = 0.05, 5                       # sub-steps of 0.01 s: the published model's time step
    GEAR, DENSITY, RADIUS, TORSO_R = 150.0, 5.0, 0.08, 0.25
    Z_INIT, RESET_NOISE = 0.75, 0.1
    CTRL_COST, CONTACT_COST, ALIVE = 0.5, 0.5e-3, 1.0
    HEALTHY_Z = (0.2, 1.0)
    #            hip anchor        leg vector       ankle axis        ankle range (rad)
    LEGS = (((0.2, 0.2, 0.0), (0.2, 0.2, 0.0), (-1.0, 1.0, 0.0), (0.5236, 1.2217)),
            ((-0.2, 0.2, 0.0), (-0.2, 0.2, 0.0), (1.0, 1.0, 0.0), (-1.2217, -0.5236)),
            ((-0.2, -0.2, 0.0), (-0.2, -0.2, 0.0), (-1.0, 1.0, 0.0), (-1.2217, -0.5236)),
            ((0.2, -0.2, 0.0), (0.2, -0.2, 0.0), (1.0, 1.0, 0.0), (0.5236, 1.2217)))
    # slot of every simulated link among the published model's 14 bodies (world, torso, then per leg: the fixed
    # strut, the hip link, the ankle link)
    BODY_SLOT = (1, 3, 4, 6, 7, 9, 10, 12, 13)

    def __init__(self, num_envs: int = 1, seed=None, time_limit: int = 1000, env_name: str = "Ant-v2"):
        super().__init__(env_name, num_envs, 111, 8, -1.0, 1.0, seed)
        self.time_limit = int(time_limit)
        r, rho = self.RADIUS, self.DENSITY
        sphere_m = rho * 4.0 / 3.0 * np.pi * self.TORSO_R ** 3
        parts = [(sphere_m, (0, 0, 0), 0.4 * sphere_m * self.TORSO_R ** 2 * np.eye(3))]
        parts += [capsule3((0, 0, 0), leg[0], r, rho) for leg in self.LEGS]                 # struts, rigid with the torso
        tm, tc, ti = _compose(parts)
        links = [Link3("torso", -1, (0, 0, 0), (0, 0, 1), tuple(tc), tm, ti)]
        contacts = [(0, (0.0, 0.0, 0.0), self.TORSO_R)]
        for k, (anchor, vec, ankle_axis, ankle_range) in enumerate(self.LEGS):
            m1, c1, i1 = capsule3((0, 0, 0), vec, r, rho)
            links.append(Link3(f"hip{k + 1}", 0, anchor, (0, 0, 1), tuple(c1), m1, i1, limits=(-0.5236, 0.5236), damping=1.0, armature=1.0))
            hip = len(links) - 1
            shin = tuple(2 * np.asarray(vec))
            m2, c2, i2 = capsule3((0, 0, 0), shin, r, rho)
            a = np.asarray(ankle_axis) / np.linalg.norm(ankle_axis)
            links.append(Link3(f"ankle{k + 1}", hip, vec, tuple(a), tuple(c2), m2, i2, limits=ankle_range, damping=1.0, armature=1.0))
            contacts += [(hip, (0.0, 0.0, 0.0), r), (hip + 1, (0.0, 0.0, 0.0), r), (hip + 1, shin, r)]
        self.model = Model3(links, contacts)
        self.sim = SpatialSim(self.model, self.num_envs)
        self.t = np.zeros(self.num_envs, np.int64)
        self._ankle0 = np.array([0.5 * (leg[3][0] + leg[3][1]) for leg in self.LEGS])

    # ---- observation ------------------------------------------------------------------------------
    def contact_wrenches(self) -> np.ndarray:
        """``[E, 14, 6]``: (moment about the body's centre of mass, force) of the ground on each body, clipped to +-1."""
        s = self.sim
        out = np.zeros((self.num_envs, 14, 6))
        if not hasattr(s, "_contact_point"):
            return out
        f = s.contact_force
        arm = s._contact_point - s._com_world[:, s._c_link]
        mom = _cross(arm, f)
        for c, link in enumerate(s._c_link):
            out[:, self.BODY_SLOT[link], :3] += mom[:, c]
            out[:, self.BODY_SLOT[link], 3:] += f[:, c]
        return np.clip(out, -1.0, 1.0)

    def _obs(self, idx=slice(None)) -> np.ndarray:
        s = self.sim
        R = quat_to_matrix(s.quat)
        w_local = np.einsum("eji,ej->ei", R, s.u[:, 3:6])
        o = np.concatenate([s.pos[:, 2:3], s.quat, s.qj, s.u[:, :3], w_local, s.u[:, 6:],
                            self.contact_wrenches().reshape(self.num_envs, 84)], axis=1).astype(np.float32)
        return o[idx]

    def render(self, e: int = 0) -> np.ndarray:
        """Two views of ant ``e`` side by side: from the side (x, z) and from above (x, y)."""
        from .planar import draw_segments
        R, o, ax = self.sim.frames()
        w, h, scale = 300, 200, 60.0
        c = self.sim.pos[e]
        side = lambda p: (w * 0.25 + (p[0] - c[0]) * scale, h * 0.7 - p[2] * scale)         # noqa: E731
        top = lambda p: (w * 0.75 + (p[0] - c[0]) * scale, h * 0.5 - (p[1] - c[1]) * scale)  # noqa: E731
        segs = []
        for k, leg in enumerate(self.LEGS):
            hip, ank = 1 + 2 * k, 2 + 2 * k
            tip = o[e, ank] + R[e, ank] @ (2 * np.asarray(leg[1]))
            col = ((230, 180, 60), (90, 200, 120), (230, 110, 110), (110, 160, 230))[k]
            for view in (side, top):
                segs += [(view(o[e, 0]), view(o[e, hip]), (128, 102, 230)), (view(o[e, hip]), view(o[e, ank]), col),
                         (view(o[e, ank]), view(tip), col)]
        return draw_segments(segs, w, h, ground_y=int(h * 0.7))

    def _reset_env(self, e: int) -> np.ndarray:
        s, n = self.sim, self.RESET_NOISE
        s.pos[e] = self.rng.uniform(-n, n, 3) + (0.0, 0.0, self.Z_INIT)
        q = np.array([1.0, 0.0, 0.0, 0.0]) + self.rng.uniform(-n, n, 4)
        s.quat[e] = q / np.linalg.norm(q)
        s.qj[e] = self.rng.uniform(-n, n, 8)
        s.qj[e, 1::2] += self._ankle0                   # the ankles start inside their range (the published model lets
        s.u[e] = self.rng.standard_normal(14) * n      # the joint limits push them there during the first steps)
        s.contact_force[e] = 0.0
        self.t[e] = 0
        return self._obs(e)

    def _step(self, actions: np.ndarray):
        a = np.clip(actions.astype(np.float64), -1.0, 1.0)
        s = self.sim
        x0 = s.pos[:, 0].copy()
        s.step(self.GEAR * a, self.DT, self.SUBSTEPS)
        self.t += 1
        state = np.concatenate([s.pos, s.quat, s.qj, s.u], axis=1)
        bad = ~np.isfinite(state).all(axis=1)
        for e in np.nonzero(bad)[0]:                    # a blown-up state never reaches the policy
            s.pos[e], s.quat[e], s.qj[e], s.u[e] = (0.0, 0.0, self.Z_INIT), (1.0, 0.0, 0.0, 0.0), 0.0, 0.0
            s.contact_force[e] = 0.0
        speed = np.where(bad, 0.0, (s.pos[:, 0] - x0) / self.DT)
        wrench = self.contact_wrenches()
        reward = speed + self.ALIVE - self.CTRL_COST * np.square(a).sum(axis=1) - self.CONTACT_COST * np.square(wrench).sum(axis=(1, 2))
        z = s.pos[:, 2]
        fell = (z < self.HEALTHY_Z[0]) | (z > self.HEALTHY_Z[1]) | bad
        limit = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        return self._obs(), reward, fell | limit
