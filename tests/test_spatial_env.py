"""The 3-D articulated-body simulator and the Ant task on it (envs/spatial.py).  MuJoCo is not installable here, so
what is pinned is the mechanics (energy, linear and angular momentum in free flight, static equilibrium on the
contact springs) and the task definition (observation layout, reward terms, termination rule); trajectory parity
with the reference's simulator is unpinned by construction (``env/utils.py:15``, ``configs/ant_d4pg.yml``)."""
import copy

import numpy as np

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.envs import AntVec, create_env_wrapper
from d4pg_pytorch_amd.envs.spatial import SpatialSim, quat_to_matrix


def _free(model):
    m = copy.deepcopy(model)
    for l in m.links:
        l.damping = l.armature = 0.0
        l.limits = (-np.inf, np.inf)
    return m


def test_free_flight_conserves_energy_and_momenta():
    """A tumbling, flailing ant in the air, no gravity: energy, linear and angular momentum stay put; with gravity
    the energy still does and the vertical momentum falls by m g t."""
    model = _free(AntVec(1).model)
    rng = np.random.default_rng(0)
    for g in (0.0, 9.81):
        model.gravity = g
        sim = SpatialSim(model, 3)
        sim.pos[:, 2] = 50.0
        q = rng.standard_normal((3, 4))
        sim.quat[:] = q / np.linalg.norm(q, axis=1, keepdims=True)
        sim.qj[:] = rng.uniform(-0.5, 0.5, sim.qj.shape)
        sim.u[:] = rng.standard_normal(sim.u.shape) * 2.0
        e0, p0, l0 = sim.energy_and_momentum()
        sim.step(np.zeros((3, 8)), 0.2, 2000)
        e1, p1, l1 = sim.energy_and_momentum()
        np.testing.assert_allclose(e1, e0, rtol=1e-4)
        mass = sim._mass.sum()
        np.testing.assert_allclose(p1[:, :2], p0[:, :2], atol=2e-3)               # first-order integrator: drift, not a bias
        np.testing.assert_allclose(p1[:, 2], p0[:, 2] - mass * g * 0.2, atol=2e-3)
        if g == 0.0:
            np.testing.assert_allclose(l1, l0, rtol=2e-3, atol=2e-3)
        assert not sim.contact_force.any()


def test_rotation_matches_quaternion_and_joint_axes():
    env = AntVec(2, seed=0)
    s = env.sim
    s.quat[:] = (np.cos(0.3), 0.0, 0.0, np.sin(0.3))           # yaw by 0.6
    R = quat_to_matrix(s.quat)
    np.testing.assert_allclose(R[0] @ [1, 0, 0], [np.cos(0.6), np.sin(0.6), 0.0], atol=1e-12)
    s.quat[:] = (1.0, 0.0, 0.0, 0.0)
    s.qj[:] = 0.0
    s.pos[:] = (0.0, 0.0, 1.0)
    R, o, ax = s.frames()
    np.testing.assert_allclose(o[0, 1], [0.2, 0.2, 1.0])        # hip 1
    np.testing.assert_allclose(o[0, 2], [0.4, 0.4, 1.0])        # ankle 1
    # every ankle's positive end of its range folds the shin DOWN (away from the torso plane)
    for leg, (_, vec, _, rng_) in enumerate(env.LEGS):
        s.qj[:] = 0.0
        s.qj[:, 2 * leg + 1] = rng_[0] if abs(rng_[0]) > abs(rng_[1]) else rng_[1]
        R, o, ax = s.frames()
        tip = o[0, 2 + 2 * leg] + R[0, 2 + 2 * leg] @ (2 * np.asarray(vec))
        assert tip[2] < 1.0 - 0.5
    # a hip turns its leg about the vertical
    s.qj[:] = 0.0
    s.qj[:, 0] = 0.5
    R, o, ax = s.frames()
    np.testing.assert_allclose(o[0, 2, 2], 1.0)
    np.testing.assert_allclose(o[0, 2, :2] - o[0, 1, :2], 0.2 * np.sqrt(2) * np.array([np.cos(np.pi / 4 + 0.5), np.sin(np.pi / 4 + 0.5)]), atol=1e-12)


def test_ant_settles_on_four_feet_and_carries_its_weight():
    env = AntVec(3, seed=1)
    env.reset()
    for _ in range(120):
        obs, r, d = env.step(np.zeros((3, 8)))
    s = env.sim
    assert not d.any() and np.abs(s.u).max() < 0.05
    frame = env.render(0)
    assert frame.shape == (200, 300, 3) and frame.dtype == np.uint8 and (frame.sum(2) > 0).sum() > 500
    weight = s._mass.sum() * env.model.gravity
    np.testing.assert_allclose(s.contact_force[:, :, 2].sum(1), weight, rtol=0.02)
    feet = [3, 6, 9, 12]
    assert (s.contact_force[:, feet, 2] > 0.1 * weight).all()                  # all four feet share the load
    np.testing.assert_allclose(r, 1.0, atol=0.02)                              # alive bonus, nothing else
    # unpowered, gravity folds the ankles to the flat end of their range: the torso rests just above the ground
    assert (s.pos[:, 2] > 0.3).all() and (s.pos[:, 2] < 0.45).all()


def test_ant_observation_reward_and_termination():
    cfg = default_config(env="Ant-v2", model="d4pg", device="cpu", env_backend="numpy")
    env = create_env_wrapper(cfg, 2, seed=3)
    assert isinstance(env, AntVec) and env.state_dim == 111 and env.action_dim == 8
    obs = env.reset()
    s = env.sim
    assert obs.shape == (2, 111) and obs.dtype == np.float32
    np.testing.assert_allclose(obs[:, 0], s.pos[:, 2], rtol=1e-6)
    np.testing.assert_allclose(obs[:, 1:5], s.quat, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(obs[:, 5:13], s.qj, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(obs[:, 13:16], s.u[:, :3], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(obs[:, 16:19], np.einsum("eji,ej->ei", quat_to_matrix(s.quat), s.u[:, 3:6]), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(obs[:, 19:27], s.u[:, 6:], rtol=1e-6, atol=1e-7)
    assert not obs[:, 27:].any()                                               # in the air: no contact yet
    a = np.array([[1.0, -1.0, 0.5, 0.0, 0.0, 0.0, -0.5, 1.0], [0.0] * 8])
    for _ in range(15):
        x0 = s.pos[:, 0].copy()
        obs, r, d = env.step(a)
        w = env.contact_wrenches()
        want = (s.pos[:, 0] - x0) / env.DT + 1.0 - 0.5 * np.square(a).sum(1) - 0.5e-3 * np.square(w).sum((1, 2))
        np.testing.assert_allclose(r, want, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(obs[:, 27:], w.reshape(2, 84), rtol=1e-6)
    assert np.abs(obs[:, 27:]).max() == 1.0 and obs[:, 27:].any(axis=1).all()  # on the ground: clipped wrenches
    assert not d.any()
    s.pos[0, 2] = 1.2
    _, _, d = env.step(np.zeros((2, 8)))
    assert d[0] and not d[1]
    env.reset_env(0)
    env.HEALTHY_Z = (0.6, 1.0)                         # the lower bound, seen from a crouching ant
    _, _, d = env.step(np.zeros((2, 8)))
    assert d[1] and not d[0]


def test_ant_is_deterministic_and_survives_random_torques():
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (150, 4, 8))
    outs = []
    for _ in range(2):
        env = AntVec(4, seed=5)
        env.reset()
        tr = []
        for a in acts:
            o, r, d = env.step(a)
            assert np.isfinite(o).all() and np.isfinite(r).all() and np.abs(o).max() < 60.0
            for e in np.nonzero(d)[0]:
                env.reset_env(e)
            tr.append((o, r, d))
        outs.append(tr)
    for (o1, r1, d1), (o2, r2, d2) in zip(*outs):
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_device_ant_follows_the_host_simulator():
    """csrc/ant_env.hip (fp32) against envs/spatial.py (fp64): ONE step from identical states with identical
    actions, then the host follows the device.  Touch-down is a threshold and the friction cone a switch, so a few
    environments per step may disagree within rounding."""
    import torch
    from d4pg_pytorch_amd.envs.device_envs import DeviceAnt, make_device_env
    E, limit = 256, 150
    cfg = default_config(env="Ant-v2", model="d4pg", device="cuda", max_ep_length=limit)
    dev = make_device_env(cfg, E, "cuda", seed=5)
    assert isinstance(dev, DeviceAnt)
    obs = torch.zeros(E, 111, device="cuda")
    dev.reset(obs)
    torch.cuda.synchronize()
    st = dev.state.cpu().double().numpy()
    assert np.abs(st[:, :2]).max() <= 0.1 and np.abs(st[:, 2] - 0.75).max() <= 0.1 + 1e-6
    np.testing.assert_allclose(np.linalg.norm(st[:, 3:7], axis=1), 1.0, atol=1e-6)
    assert 0.07 < st[:, 15:29].std() < 0.13                                       # normal deviates x 0.1
    host = AntVec(E, seed=0, time_limit=limit)
    host.reset()

    def follow():
        st = dev.state.cpu().double().numpy()
        s = host.sim
        s.pos[:], s.quat[:], s.qj[:], s.u[:] = st[:, 0:3], st[:, 3:7], st[:, 7:15], st[:, 15:29]
        host.t[:] = dev.t.cpu().numpy()
    follow()
    np.testing.assert_allclose(obs.cpu().numpy()[:, :27], host._obs()[:, :27], rtol=1e-5, atol=1e-6)
    assert not obs[:, 27:].any()
    nobs, rew, done, trunc = (torch.zeros(E, 111, device="cuda"), torch.zeros(E, device="cuda"),
                              torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"))
    g = torch.Generator().manual_seed(2)
    ended_total, touched = 0, 0
    for t in range(200):
        act = (torch.rand(E, 8, generator=g) * 2.6 - 1.3)
        act[: E // 4] = 0.0
        act[E // 4: E // 2] *= 0.3
        act = act.cuda()
        dev.step(act, obs, nobs, rew, done, trunc)
        torch.cuda.synchronize()
        o_h, r_h, d_h = host._step(act.cpu().numpy())
        o_d = nobs.cpu().numpy()
        # positions, angles, velocities; the clipped contact wrenches switch between -1 and 1 at a touch: compared in the mean
        close = np.all(np.abs(o_d[:, :27] - o_h[:, :27]) <= 5e-3 + 5e-3 * np.abs(o_h[:, :27]), axis=1)
        assert close.mean() > 0.95, (t, np.abs(o_d[:, :27] - o_h[:, :27]).max(axis=1))
        assert np.abs(o_d[close, 27:] - o_h[close, 27:]).mean() < 0.02
        touched += int((np.abs(o_d[:, 27:]).max(axis=1) > 0).sum())
        d_d = done.cpu().numpy() > 0.5
        agree = (d_d == d_h) & close
        assert agree.mean() > 0.95
        np.testing.assert_allclose(rew.cpu().numpy()[agree], r_h[agree], rtol=5e-3, atol=0.05)
        assert not trunc.any()
        ended = torch.from_numpy(d_d).cuda()
        ended_total += int(d_d.sum())
        assert torch.equal(obs[~ended], nobs[~ended]) and (dev.t[ended] == 0).all()
        follow()
    mean_ret, count = dev.episode_stats()
    assert count == ended_total and count >= E and np.isfinite(mean_ret)
    assert touched > 50 * E and torch.isfinite(dev.state).all()


def test_host_runtime_ant_follows_the_numpy_oracle():
    """csrc/planar_host.cpp + csrc/ant_sim.h (the source of the device kernel built for the CPU, fp32) against
    envs/spatial.py (fp64): one step from identical states and actions."""
    from d4pg_pytorch_amd.envs import native_planar
    if native_planar.ant_host_step_fn() is None:
        pytest.skip("host library not built")
    cfg = default_config(env="Ant-v2", model="d4pg", device="cpu")
    nat = create_env_wrapper(cfg, 16, seed=1)
    assert type(nat) is native_planar.NativeAntVec
    assert type(create_env_wrapper(dict(cfg, env_backend="numpy"), 2, seed=1)) is AntVec
    ref = AntVec(16, seed=1)
    np.testing.assert_array_equal(nat.reset(), ref.reset())
    rng = np.random.default_rng(0)
    rows = close_rows = ended = 0
    for t in range(200):
        act = rng.uniform(-1.3, 1.3, (16, 8))
        act[:4] = 0.0
        s, b = ref.sim, nat.sim
        b.pos[:], b.quat[:], b.qj[:], b.u[:], nat.t[:] = s.pos, s.quat, s.qj, s.u, ref.t
        o_r, r_r, d_r = ref._step(act)
        o_n, r_n, d_n = nat._step(act)
        close = np.all(np.abs(o_n[:, :27] - o_r[:, :27]) <= 2e-3 + 2e-3 * np.abs(o_r[:, :27]), axis=1)
        rows, close_rows = rows + 16, close_rows + int(close.sum())
        ok = close & (d_r == d_n)
        np.testing.assert_allclose(r_n[ok], r_r[ok], rtol=2e-3, atol=0.02)
        assert np.abs(o_n[ok, 27:] - o_r[ok, 27:]).mean() < 0.01
        for e in np.nonzero(d_r)[0]:
            ref.reset_env(e)
            ended += 1
    assert close_rows >= 0.99 * rows and ended > 3
    env = create_env_wrapper(dict(cfg, max_ep_length=30), 4, seed=2)
    obs = env.reset()
    assert not obs[:, 27:].any()
    dones = 0
    for t in range(70):
        obs, r, d = env.step(np.zeros((4, 8)))
        assert np.isfinite(obs).all() and np.isfinite(r).all()
        for e in np.nonzero(d)[0]:
            fresh = env.reset_env(e)
            assert not fresh[27:].any() and env.t[e] == 0        # a fresh episode starts without contact wrenches
            dones += 1
    assert dones >= 8
