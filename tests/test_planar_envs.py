"""The planar articulated-body tasks (envs/planar.py): BipedalWalker, Hopper, Walker2d, HalfCheetah on their own
reduced-coordinate simulator.  Box2D and MuJoCo are not installable here, so what is pinned is the mechanics
(energy, momentum, static equilibrium on the contact springs) and the task definitions (observation layout and
scaling, reward terms, termination rules); trajectory parity with the reference's simulators is unpinned by
construction (``env/bipedal.py``, ``env/utils.py:15``)."""
import copy

import numpy as np
import pytest

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.envs import BipedalWalkerVec, HalfCheetahVec, HopperVec, Walker2dVec, create_env_wrapper
from d4pg_pytorch_amd.envs.planar import PlanarSim

ALL = [(BipedalWalkerVec, "BipedalWalker-v2", 24, 4), (HopperVec, "Hopper-v2", 11, 3), (Walker2dVec, "Walker2d-v2", 17, 6),
       (HalfCheetahVec, "HalfCheetah-v2", 17, 6)]


@pytest.mark.parametrize("cls,name,S,A", ALL)
def test_factory_shapes_and_determinism(cls, name, S, A):
    cfg = default_config(env=name, model="d4pg", device="cpu", state_dim=S, action_dim=A)
    env = create_env_wrapper(cfg, 3, seed=7)
    assert isinstance(env, cls) and env.state_dim == S and env.action_dim == A
    obs = env.reset()
    assert obs.shape == (3, S) and obs.dtype == np.float32 and np.isfinite(obs).all()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (12, 3, A))
    traj = [env.step(a) for a in acts]
    frame = env.render(1)                      # what Agent.save_replay_gif records (models/agent.py:147-163)
    assert frame.shape == (200, 300, 3) and frame.dtype == np.uint8 and (frame.sum(2) > 0).sum() > 500
    env2 = create_env_wrapper(cfg, 3, seed=7)
    env2.reset()
    for (o, r, d), a in zip(traj, acts):
        o2, r2, d2 = env2.step(a)
        assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2)


@pytest.mark.parametrize("cls", [c for c, *_ in ALL])
def test_free_flight_conserves_energy_and_momentum(cls):
    """No contact, no joint friction / springs / limits, no torques: the mechanism tumbles through the air and its
    total energy, horizontal momentum and the parabola of its centre of mass must come out of the equations."""
    env = cls(2, seed=1)
    model = copy.deepcopy(env.model)
    for l in model.links:
        l.damping = l.stiffness = l.armature = 0.0
        l.limits = (-np.inf, np.inf)
    sim = PlanarSim(model, 2)
    rng = np.random.default_rng(3)
    sim.q[:] = rng.uniform(-0.5, 0.5, sim.q.shape)
    sim.q[:, 1] += 20.0
    sim.qd[:] = rng.standard_normal(sim.qd.shape) * 2.0
    mass = sum(l.mass for l in model.links)

    def momentum_x(s):
        ang, org = s.frames()
        px = np.zeros(2)
        for i, l in enumerate(model.links):
            p = s._world(ang[:, i], org[:, i], l.com)
            px += l.mass * np.einsum("ei,ei->e", s._jacobian(i, p, org)[:, 0, :], s.qd)
        return px
    e0, p0 = sim.energy(), momentum_x(sim)
    sim.step(lambda s: np.zeros((2, len(model.links))), 0.4, 800)
    np.testing.assert_allclose(sim.energy(), e0, rtol=2e-3)
    np.testing.assert_allclose(momentum_x(sim), p0, rtol=1e-3, atol=1e-3 * mass)       # first-order integrator: drift, not a bias
    assert not sim.contact_on.any()


def test_hopper_stands_on_its_contact_springs_then_falls():
    env = HopperVec(4, seed=2)
    obs = env.reset()
    weight = sum(l.mass for l in env.model.links) * env.model.gravity
    total = np.zeros(4)
    steps = np.zeros(4, int)
    fell_at = np.full(4, -1)
    for t in range(400):
        obs, r, d = env.step(np.zeros((4, 3)))
        total += np.where(fell_at < 0, r, 0.0)
        if t == 10:
            # both foot contacts carry the weight: the foot sinks into the springs by about weight / (2 k), no more
            ang, org = env.sim.frames()
            foot = len(env.model.links) - 1
            ys = [env.sim._world(ang[:, foot], org[:, foot], p)[:, 1] for p in ((-0.13, -0.06), (0.26, -0.06))]
            assert all((-3 * weight / env.model.k_contact < y).all() and (y < 0.002).all() for y in ys)
            assert env.sim.contact_on[:, :2].all() and (np.abs(env.sim.q[:, 2]) < 0.05).all()
        for e in np.nonzero(d)[0]:
            if fell_at[e] < 0:
                fell_at[e] = t
            obs[e] = env.reset_env(e)
    # an unactuated hopper is an inverted pendulum: it stays up for a while (alive bonus 1 per step), then falls
    assert (fell_at > 30).all() and (fell_at < 399).all()
    np.testing.assert_allclose(total, fell_at + 1, rtol=0.15)


def test_hopper_reward_and_termination_rules():
    env = HopperVec(2, seed=3)
    env.reset()
    x0 = env.sim.q[:, 0].copy()
    a = np.array([[0.5, -0.5, 0.25], [0.0, 0.0, 0.0]])
    obs, r, d = env.step(a)
    speed = (env.sim.q[:, 0] - x0) / env.DT
    np.testing.assert_allclose(r, 1.0 + speed - 1e-3 * np.square(a).sum(1), rtol=1e-5, atol=1e-6)       # alive + forward speed - control cost
    np.testing.assert_allclose(obs[:, :5], env.sim.q[:, 1:6], rtol=1e-6)                       # qpos[1:]: height, pitch, joints
    np.testing.assert_allclose(obs[:, 5:], np.clip(env.sim.qd, -10, 10), rtol=1e-5, atol=1e-6)
    assert not d.any()
    env.sim.q[0, 2] = 0.3                              # pitched over: unhealthy
    env.sim.q[1, 1] = 0.4                              # too low: unhealthy
    _, _, d = env.step(np.zeros((2, 3)))
    assert d.all()


def test_walker_and_cheetah_rules():
    w = Walker2dVec(2, seed=4)
    w.reset()
    w.sim.q[0, 2] = 1.2
    _, _, d = w.step(np.zeros((2, 6)))
    assert d[0] and not d[1]
    c = HalfCheetahVec(2, seed=5, time_limit=30)
    c.reset()
    rng = np.random.default_rng(0)
    for t in range(30):
        x0 = c.sim.q[:, 0].copy()
        a = rng.uniform(-1, 1, (2, 6))
        obs, r, d = c.step(a)
        np.testing.assert_allclose(r, (c.sim.q[:, 0] - x0) / c.DT - 0.1 * np.square(a).sum(1), rtol=1e-5, atol=1e-6)
        assert np.isfinite(obs).all() and d.all() == (t == 29)                              # no termination but the limit
    assert (c.sim.q[:, 1] > 0.2).all()                                                      # joint springs hold it off the ground


def test_bipedal_walker_observation_reward_and_terminals():
    env = BipedalWalkerVec(3, seed=6, terrain=False)
    obs = env.reset().astype(np.float64)
    q = env.sim.q
    np.testing.assert_allclose(obs[:, 0], q[:, 2], atol=1e-7)                                 # hull angle
    np.testing.assert_allclose(obs[:, 4], q[:, 3], atol=1e-7)                                 # hip 1
    np.testing.assert_allclose(obs[:, 6], q[:, 4] + 1.0, atol=1e-7)                           # knee 1 (+1 like the gym task)
    ang = 1.5 * np.arange(10) / 10
    np.testing.assert_allclose(obs[:, 14:], np.clip(q[:, 1:2] / (np.cos(ang) * env.LIDAR_RANGE), 0, 1), rtol=1e-6)
    assert (obs[:, 14] < obs[:, 23]).all() and (obs[:, 23] <= 1.0).all()                      # straight down is the nearest
    x0, th0 = q[:, 0].copy(), q[:, 2].copy()
    a = np.array([[1.0, -1.0, 0.5, 0.0]] * 3)
    obs, r, d = env.step(a)
    shaping = 130 * (env.sim.q[:, 0] - x0) / 30 - 5 * (np.abs(env.sim.q[:, 2]) - np.abs(th0))
    np.testing.assert_allclose(r, shaping - 0.00035 * 80 * 2.5, rtol=1e-6, atol=1e-9)
    assert not d.any()
    # the servo drives each joint towards sign(a) x SPEED within its torque limit
    assert (env.sim.qd[:, 3] > 0).all() and (env.sim.qd[:, 4] < 0).all()
    # terminals: the hull on the ground, walking off the back of the track, reaching the far end
    env.sim.q[0, 1], env.sim.q[0, 2] = 0.2, 1.5
    env.sim.q[1, 0] = -0.1
    env.sim.q[2, 0] = env.TRACK_END + 0.1
    _, r, d = env.step(np.zeros((3, 4)))
    assert d.all() and r[0] == -100.0 and r[1] == -100.0 and r[2] != -100.0
    # an unactuated walker folds up and its hull hits the ground: about -100
    env = BipedalWalkerVec(4, seed=7)
    env.reset()
    total = np.zeros(4)
    done_at = np.full(4, -1)
    for t in range(120):
        _, r, d = env.step(np.zeros((4, 4)))
        total += np.where(done_at < 0, r, 0.0)
        done_at = np.where((done_at < 0) & d, t, done_at)
        if (done_at >= 0).all():
            break
    assert (done_at > 5).all() and (total < -95).all() and (total > -115).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cls,name,S,A", ALL)
def test_device_planar_follows_the_host_simulator(cls, name, S, A):
    """The device kernels (csrc/planar_env.hip, fp32) against the host simulator (fp64): ONE step from identical
    states with identical actions, then the host follows the device.  Contacts are stiff and touch-down is a
    threshold, so a few environments per step may disagree on a contact made within rounding."""
    import torch
    from d4pg_pytorch_amd.envs.device_envs import DevicePlanar, make_device_env
    E = 256
    limit = 120
    cfg = default_config(env=name, model="d4pg", device="cuda", max_ep_length=limit, state_dim=S, action_dim=A)
    dev = make_device_env(cfg, E, "cuda", seed=11)
    assert isinstance(dev, DevicePlanar) and dev.state_dim == S and dev.action_dim == A
    obs = torch.zeros(E, S, device="cuda")
    dev.reset(obs)
    torch.cuda.synchronize()
    host = cls(E, seed=0, time_limit=limit)
    host.reset()
    n = host.model.n
    st0 = dev.state.cpu().double().numpy()
    if cls is BipedalWalkerVec:
        assert np.all(st0[:, 0] == np.float32(host.START_X)) and st0[:, 9].std() > 0.01       # the random push
    else:
        assert np.abs(st0[:, 0]).max() <= host.RESET_NOISE and st0[:, :n].std() > 0.2 * host.RESET_NOISE

    def follow():
        st = dev.state.cpu().double().numpy()
        host.sim.q[:], host.sim.qd[:] = st[:, :n], st[:, 9:9 + n]
        host.t[:] = dev.t.cpu().numpy()
        if cls is BipedalWalkerVec:
            host.sim.contact_on[:] = False
            host.sim.contact_on[:, :2] = st[:, 18:20] > 0.5
            host.prev_shaping[:] = st[:, 20]
            host.sim.terrain[:] = dev.terrain.cpu().double().numpy().T           # the device keeps it sample-major
    follow()
    np.testing.assert_allclose(obs.cpu().numpy(), host._obs(), rtol=1e-5, atol=1e-6)
    nobs, rew, done, trunc = (torch.zeros(E, S, device="cuda"), torch.zeros(E, device="cuda"),
                              torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"))
    g = torch.Generator().manual_seed(4)
    ended_total, worst = 0, 1.0
    for t in range(300):
        act = (torch.rand(E, A, generator=g) * 2.6 - 1.3)                     # beyond [-1, 1]: clipped on both sides
        act[: E // 4] = 0.0
        act = act.cuda()
        dev.step(act, obs, nobs, rew, done, trunc)
        torch.cuda.synchronize()
        o_h, r_h, d_h = host._step(act.cpu().numpy())
        o_d = nobs.cpu().numpy()
        close = np.all(np.abs(o_d - o_h) <= 5e-3 + 5e-3 * np.abs(o_h), axis=1)
        worst = min(worst, close.mean())
        assert close.mean() > 0.95, (t, np.abs(o_d - o_h).max(axis=1))
        d_d = done.cpu().numpy() > 0.5
        agree = (d_d == d_h) & close
        assert agree.mean() > 0.95
        np.testing.assert_allclose(rew.cpu().numpy()[agree], r_h[agree], rtol=2e-3, atol=0.05)
        assert not trunc.any()
        ended = torch.from_numpy(d_d).cuda()
        ended_total += int(d_d.sum())
        assert torch.equal(obs[~ended], nobs[~ended]) and (dev.t[ended] == 0).all()
        follow()
    mean_ret, count = dev.episode_stats()
    assert count == ended_total and count >= E and np.isfinite(mean_ret)
    assert torch.isfinite(dev.state).all()


@pytest.mark.parametrize("cls,name,S,A", ALL)
def test_host_runtime_simulator_follows_the_numpy_oracle(cls, name, S, A):
    """csrc/planar_host.cpp (the source of the device kernels built for the CPU, fp32) against envs/planar.py
    (fp64): one step from identical states and actions.  A touch-down within rounding of a sub-step boundary
    changes a contact force by its damper term, so single environments may differ for a step."""
    from d4pg_pytorch_amd.envs import native_planar
    if native_planar.host_step_fn() is None:
        pytest.skip("host library not built")
    cfg = default_config(env=name, model="d4pg", device="cpu", state_dim=S, action_dim=A)
    nat = create_env_wrapper(cfg, 32, seed=1)
    assert type(nat) is native_planar.NATIVE[cls]
    assert type(create_env_wrapper(dict(cfg, env_backend="numpy"), 2, seed=1)) is cls
    ref = cls(32, seed=1)
    np.testing.assert_array_equal(nat.reset(), ref.reset())          # the same reset logic and random stream
    rng = np.random.default_rng(0)
    rows, close_rows, ended = 0, 0, 0
    for t in range(250):
        act = rng.uniform(-1.3, 1.3, (32, A))
        act[:8] = 0.0
        nat.sim.q[:], nat.sim.qd[:], nat.t[:] = ref.sim.q, ref.sim.qd, ref.t
        nat.sim.contact_on[:] = ref.sim.contact_on
        if cls is BipedalWalkerVec:
            nat.prev_shaping[:] = ref.prev_shaping
            nat.sim.terrain[:] = ref.sim.terrain
        o_r, r_r, d_r = ref._step(act)
        o_n, r_n, d_n = nat._step(act)
        close = np.all(np.abs(o_n - o_r) <= 2e-3 + 2e-3 * np.abs(o_r), axis=1)
        rows, close_rows = rows + 32, close_rows + int(close.sum())
        ok = close & (d_r == d_n)
        np.testing.assert_allclose(r_n[ok], r_r[ok], rtol=2e-3, atol=0.02)
        assert (d_r == d_n)[close].all()
        for e in np.nonzero(d_r)[0]:
            ref.reset_env(e)
            ended += 1
    assert close_rows >= 0.99 * rows and (ended > 5 or cls is HalfCheetahVec)          # the cheetah has no terminal state
    # through the public step(): resets included, finite, and the time limit ends episodes
    env = create_env_wrapper(dict(cfg, max_ep_length=40), 4, seed=2)
    env.reset()
    dones = 0
    for t in range(100):
        o, r, d = env.step(np.zeros((4, A)))
        assert np.isfinite(o).all() and np.isfinite(r).all()
        for e in np.nonzero(d)[0]:
            env.reset_env(e)
            dones += 1
    assert dones >= 8


def test_bipedal_walker_terrain_and_lidar():
    """The gym task's grass: a flat start pad, gentle random hills after it; contact follows the ground; the lidar
    readings are the exact ray / ground intersections (checked against a brute-force march along the rays)."""
    env = BipedalWalkerVec(4, seed=0)
    env.reset()
    ter, step = env.sim.terrain, env.TERRAIN_STEP
    assert not ter[:, :21].any() and np.abs(ter[:, 21:]).max() > 0.1 and np.abs(ter).max() < 2.5
    assert np.abs(np.diff(ter, axis=1)).max() / step < 0.5 and not np.array_equal(ter[0], ter[1])
    g, s = env.sim.ground(np.array([25.3, 40.0, 60.1, 80.7]))
    j = np.floor(np.array([25.3, 40.0, 60.1, 80.7]) / step).astype(int)
    np.testing.assert_allclose(s, (ter[np.arange(4), j + 1] - ter[np.arange(4), j]) / step)
    env.sim.q[:, 0] = [25.3, 40.0, 60.1, 80.7]
    env.sim.q[:, 1] = g + 1.9
    got = env.lidar()
    ang = 1.5 * np.arange(10) / 10
    for e in range(4):
        for i in range(10):
            ts = np.linspace(0.0, 1.0, 100001)
            xs = env.sim.q[e, 0] + ts * np.sin(ang[i]) * env.LIDAR_RANGE
            ys = env.sim.q[e, 1] - ts * np.cos(ang[i]) * env.LIDAR_RANGE
            gs, _ = env.sim.ground(xs, env=np.full(xs.shape, e))
            hit = np.nonzero(ys <= gs)[0]
            assert abs((ts[hit[0]] if len(hit) else 1.0) - got[e, i]) < 3e-4
    # a walker set down on a slope rests ON the slope: feet within the contact springs' sag of the local ground
    env = BipedalWalkerVec(2, seed=3)
    env.reset()
    x = 50.0
    g, s = env.sim.ground(np.array([x, x]))
    env.sim.q[:, 0], env.sim.q[:, 1] = x, g + 2 * env.LEG_H - env.LEG_DOWN + 0.02
    env.prev_shaping[:] = env._shaping()
    touched = np.zeros(2, bool)
    for _ in range(12):
        env.step(np.zeros((2, 4)))
        touched |= env.sim.contact_on[:, :2].any(axis=1)
        for link in (2, 4):                                      # no foot ever sinks through the local ground
            foot = env.sim.point(link, (0.0, -env.LEG_H))
            gf, _ = env.sim.ground(foot[:, 0])
            assert (foot[:, 1] - gf > -0.03).all()
    assert touched.all()
    new = create_env_wrapper(default_config(env="BipedalWalker-v2", model="d4pg", device="cpu"), 2, seed=1)
    new.reset()
    assert np.abs(new.sim.terrain).max() > 0.05                  # the factory builds the hilly task
