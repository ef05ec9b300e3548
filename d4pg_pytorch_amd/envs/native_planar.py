"""The articulated-body tasks (planar: Hopper, Walker2d, HalfCheetah, BipedalWalker; 3-D: Ant) stepped by the C++ host
runtime (``csrc/planar_host.cpp``).

``envs/planar.py`` and ``envs/spatial.py`` are the simulators written out in numpy (fp64; the oracles the other two
builds are tested against) and run 1 - 3 k environment steps/s per process.  The classes here keep its task definition, reset logic and state
(``sim.q``, ``sim.qd``) and replace the step by one call into the host library, which is built from the same
source as the device kernels (``csrc/planar_sim.h``, ``csrc/ant_sim.h``, fp32): two orders of magnitude more steps per second for the
CPU explorer processes of the reference's topology (``models/agent.py:60-145`` steps a Box2D / MuJoCo simulator
there).  ``create_env_wrapper`` picks these classes whenever the host library is built.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import planar

KINDS = {"HopperVec": 0, "Walker2dVec": 1, "HalfCheetahVec": 2, "BipedalWalkerVec": 3}
STRIDE, QD0, FLAG0, PREV = 24, 9, 18, 20          # csrc/planar_sim.h


def _pack(values) -> int:
    out = 0
    for i, v in enumerate(values):
        out |= (int(v) + 1) << (4 * i)
    return out


def planar_params(host):
    """``PlanarParams`` (``csrc/planar_sim.h``) of a task of ``envs/planar.py``: every number the compiled
    simulators use, read from the Python model."""
    from ..ops import hip as H
    kind = KINDS[[c.__name__ for c in type(host).__mro__ if c.__name__ in KINDS][0]]
    m, p = host.model, H.PlanarParams()
    if len(m.links) > 8 or len(m.contacts) > 8:
        raise ValueError("the compiled planar simulators hold at most 8 links and 8 contact points")
    for i, l in enumerate(m.links):
        p.anchor[i][0], p.anchor[i][1] = l.anchor
        p.com[i][0], p.com[i][1] = l.com
        p.mass[i], p.inertia[i], p.rest[i] = l.mass, l.inertia, l.rest
        p.lo[i], p.hi[i] = max(l.limits[0], -1e30), min(l.limits[1], 1e30)
        p.damping[i], p.armature[i], p.stiffness[i] = l.damping, l.armature, l.stiffness
    for c, (link, local) in enumerate(m.contacts):
        p.c_local[c][0], p.c_local[c][1] = local
    p.gravity, p.k_contact, p.c_contact, p.c_friction, p.mu = m.gravity, m.k_contact, m.c_contact, m.c_friction, m.mu
    p.k_limit, p.c_limit = m.k_limit, m.c_limit
    p.n_links, p.n_contacts, p.n_act = len(m.links), len(m.contacts), host.action_dim
    p.parents = _pack(l.parent for l in m.links)
    p.contact_links = _pack(link for link, _ in m.contacts)
    big = 1e30
    if kind == 3:
        p.dt, p.substeps = 1.0 / host.FPS, host.SUBSTEPS
    else:
        p.dt, p.substeps = host.DT, host.SUBSTEPS
        for j, g in enumerate(host.gears):
            p.gear[j] = g
        p.alive, p.ctrl_cost, p.reset_noise, p.q0_y = host.ALIVE, host.CTRL_COST, host.RESET_NOISE, host.q0_y
        # healthy: z_min <= q[1] + z_off <= z_max and |root angle| <= ang_max (the rules of _unhealthy() of each class)
        p.z_off, p.z_min, p.z_max, p.ang_max = {0: (0.2, 0.7, big, 0.2), 1: (0.2, 0.8, 2.0, 1.0), 2: (0.0, -big, big, big)}[kind]
    return p, kind


def ant_params(host):
    """``AntParams`` (``csrc/ant_sim.h``) of ``envs/spatial.py::AntVec``: every number the compiled simulators use."""
    from ..ops import hip as H
    m, p = host.model, H.AntParams()
    if len(m.links) != 9 or len(m.contacts) != 13:
        raise ValueError("the ant kernel is compiled for 9 links and 13 contact spheres")
    for i, l in enumerate(m.links):
        for k in range(3):
            p.anchor[i][k], p.axis[i][k], p.com[i][k] = l.anchor[k], l.axis[k], l.com[k]
        inertia = np.asarray(l.inertia, float).reshape(9)
        for k in range(9):
            p.inertia[i][k] = inertia[k]
        p.mass[i], p.damping[i], p.armature[i] = l.mass, l.damping, l.armature
        p.lo[i], p.hi[i] = max(l.limits[0], -1e30), min(l.limits[1], 1e30)
        p.parent[i] = l.parent
    for c, (link, local, rad) in enumerate(m.contacts):
        for k in range(3):
            p.c_local[c][k] = local[k]
        p.c_rad[c], p.c_link[c] = rad, link
    p.gravity, p.k_contact, p.c_contact, p.c_friction, p.mu = m.gravity, m.k_contact, m.c_contact, m.c_friction, m.mu
    p.k_limit, p.c_limit = m.k_limit, m.c_limit
    p.dt, p.substeps, p.gear, p.z_init, p.reset_noise = host.DT, host.SUBSTEPS, host.GEAR, host.Z_INIT, host.RESET_NOISE
    p.ctrl_cost, p.contact_cost, p.alive = host.CTRL_COST, host.CONTACT_COST, host.ALIVE
    p.z_min, p.z_max = host.HEALTHY_Z
    for k in range(4):
        p.ankle0[k] = host._ankle0[k]
    p.n_links, p.n_contacts = 9, 13
    return p


_fn = None


def host_step_fn():
    """``d4pg_planar_host_step`` of the host library, or ``None`` when the library is not built / too old."""
    global _fn
    if _fn is None:
        try:
            from ..ops import hip as H
            from ..parallel.ipc import host_lib
            lib = host_lib()
            size = lib.d4pg_sizeof_host_PlanarParams
            size.restype = C.c_int
            if size() != C.sizeof(H.PlanarParams):
                raise RuntimeError(f"PlanarParams: {C.sizeof(H.PlanarParams)} bytes in ops/hip.py, {size()} in csrc/planar_host.cpp")
            fn = lib.d4pg_planar_host_step
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p]
            _fn = fn
        except (OSError, AttributeError):
            _fn = False
    return _fn or None


class _NativeStep:
    """Mixin over a class of ``envs/planar.py``: the same task, ``_step`` in the host library."""

    def _native_init(self) -> None:
        self._fn = host_step_fn()
        if self._fn is None:
            raise RuntimeError("the host library (csrc/libd4pg_host.so) is not built: python -m d4pg_pytorch_amd.build")
        self._params, self._kind = planar_params(self)
        E = self.num_envs
        self._state = np.zeros((E, STRIDE), np.float32)
        self._act32 = np.zeros((E, self.action_dim), np.float32)
        self._obs32 = np.zeros((E, self.state_dim), np.float32)
        self._rew32 = np.zeros(E, np.float32)
        self._term = np.zeros(E, np.uint8)
        self._n = self.model.n
        self._terrain32 = np.zeros((E, 200), np.float32) if self._kind == 3 else None

    def _step(self, actions: np.ndarray):
        n, st, sim = self._n, self._state, self.sim
        st[:, :n], st[:, QD0:QD0 + n] = sim.q, sim.qd
        if self._kind == 3:
            st[:, FLAG0:FLAG0 + 2], st[:, PREV] = sim.contact_on[:, :2], self.prev_shaping
            self._terrain32[:] = sim.terrain
        self._act32[:] = actions
        rc = self._fn(C.addressof(self._params), self._kind, self.num_envs, st.ctypes.data, self._act32.ctypes.data,
                      self.action_dim, self._obs32.ctypes.data, self._rew32.ctypes.data, self._term.ctypes.data,
                      self._terrain32.ctypes.data if self._terrain32 is not None else None)
        if rc != 0:
            raise RuntimeError(f"d4pg_planar_host_step failed ({rc})")
        sim.q[:], sim.qd[:] = st[:, :n], st[:, QD0:QD0 + n]
        if self._kind == 3:
            sim.contact_on[:] = False
            sim.contact_on[:, :2] = st[:, FLAG0:FLAG0 + 2] > 0.5
            self.prev_shaping[:] = st[:, PREV]
        self.t += 1
        limit = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
        return self._obs32.copy(), self._rew32.astype(np.float64), (self._term > 0) | limit


def _native(cls):
    class Native(_NativeStep, cls):
        def __init__(self, *args, **kwargs):
            cls.__init__(self, *args, **kwargs)
            self._native_init()
    Native.__name__ = "Native" + cls.__name__
    Native.__qualname__ = Native.__name__
    Native.__doc__ = f"``{cls.__name__}`` stepped by ``csrc/planar_host.cpp``."
    return Native


NativeHopperVec = _native(planar.HopperVec)
NativeWalker2dVec = _native(planar.Walker2dVec)
NativeHalfCheetahVec = _native(planar.HalfCheetahVec)
NativeBipedalWalkerVec = _native(planar.BipedalWalkerVec)
NATIVE = {planar.HopperVec: NativeHopperVec, planar.Walker2dVec: NativeWalker2dVec, planar.HalfCheetahVec: NativeHalfCheetahVec,
          planar.BipedalWalkerVec: NativeBipedalWalkerVec}


# ---------------------------------------------------------------------------------------------- the ant (3-D)
_ant_fn = None


def ant_host_step_fn():
    """``d4pg_ant_host_step`` of the host library, or ``None`` when the library is not built / too old."""
    global _ant_fn
    if _ant_fn is None:
        try:
            from ..ops import hip as H
            from ..parallel.ipc import host_lib
            lib = host_lib()
            size = lib.d4pg_sizeof_host_AntParams
            size.restype = C.c_int
            if size() != C.sizeof(H.AntParams):
                raise RuntimeError(f"AntParams: {C.sizeof(H.AntParams)} bytes in ops/hip.py, {size()} in csrc/planar_host.cpp")
            fn = lib.d4pg_ant_host_step
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            _ant_fn = fn
        except (OSError, AttributeError):
            _ant_fn = False
    return _ant_fn or None


def _native_ant():
    from .spatial import AntVec

    class NativeAntVec(AntVec):
        """``AntVec`` stepped by ``csrc/planar_host.cpp`` (``csrc/ant_sim.h``, the source of the device kernel)."""

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._fn = ant_host_step_fn()
            if self._fn is None:
                raise RuntimeError("the host library (csrc/libd4pg_host.so) is not built: python -m d4pg_pytorch_amd.build")
            self._params = ant_params(self)
            E = self.num_envs
            self._state = np.zeros((E, 32), np.float32)
            self._act32 = np.zeros((E, 8), np.float32)
            self._obs32 = np.zeros((E, 111), np.float32)
            self._rew32 = np.zeros(E, np.float32)
            self._term = np.zeros(E, np.uint8)
            self._fresh = np.ones(E, bool)              # no contact wrench known yet (after a reset)

        def _obs(self, idx=slice(None)) -> np.ndarray:
            # positions and velocities from the state; the contact wrenches are those of the last native step
            s = self.sim
            from .spatial import quat_to_matrix
            w_local = np.einsum("eji,ej->ei", quat_to_matrix(s.quat), s.u[:, 3:6])
            o = self._obs32.copy()
            o[:, :27] = np.concatenate([s.pos[:, 2:3], s.quat, s.qj, s.u[:, :3], w_local, s.u[:, 6:]], axis=1)
            o[self._fresh, 27:] = 0.0
            return o[idx]

        def _reset_env(self, e: int) -> np.ndarray:
            self._fresh[e] = True
            return super()._reset_env(e)

        def _step(self, actions: np.ndarray):
            st, s = self._state, self.sim
            st[:, 0:3], st[:, 3:7], st[:, 7:15], st[:, 15:29] = s.pos, s.quat, s.qj, s.u
            self._act32[:] = actions
            rc = self._fn(C.addressof(self._params), self.num_envs, st.ctypes.data, self._act32.ctypes.data, 8,
                          self._obs32.ctypes.data, self._rew32.ctypes.data, self._term.ctypes.data)
            if rc != 0:
                raise RuntimeError(f"d4pg_ant_host_step failed ({rc})")
            s.pos[:], s.quat[:], s.qj[:], s.u[:] = st[:, 0:3], st[:, 3:7], st[:, 7:15], st[:, 15:29]
            self._fresh[:] = False
            self.t += 1
            limit = self.t >= self.time_limit if self.time_limit > 0 else np.zeros(self.num_envs, bool)
            return self._obs32.copy(), self._rew32.astype(np.float64), (self._term > 0) | limit

    return NativeAntVec


NativeAntVec = _native_ant()
