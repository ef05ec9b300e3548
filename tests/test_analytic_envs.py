"""Reacher and the inverted double pendulum as closed-form dynamics (envs/analytic.py): the task definitions of
``Reacher-v2`` / ``InvertedDoublePendulum-v2`` that the reference runs through gym + MuJoCo.  MuJoCo is not
installable here, so what is pinned is the task (observation layout, reward, termination), the physics itself
(energy, equilibria) and that the tasks are well posed (simple controllers solve them); trajectory parity with
MuJoCo is unpinned by construction."""
import numpy as np
import pytest

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.envs import InvertedDoublePendulumVec, ReacherVec, create_env_wrapper


def test_factory_builds_the_native_tasks_for_the_reference_ids():
    r = create_env_wrapper(default_config(env="Reacher-v2", model="d4pg", device="cpu"), 2, seed=0)
    p = create_env_wrapper(default_config(env="InvertedDoublePendulum-v2", model="d4pg", device="cpu"), 2, seed=0)
    assert isinstance(r, ReacherVec) and isinstance(p, InvertedDoublePendulumVec)
    assert r.reset().shape == (2, 11) and p.reset().shape == (2, 11) and r.action_dim == 2 and p.action_dim == 1
    assert r.time_limit == 50


def test_reacher_observation_reward_and_time_limit():
    env = ReacherVec(5, seed=1)
    obs = env.reset().astype(np.float64)
    assert np.all(np.linalg.norm(env.target, axis=1) < 0.2)
    np.testing.assert_allclose(obs[:, 0:2] ** 2 + obs[:, 2:4] ** 2, 1.0, atol=1e-6)          # cos, cos, sin, sin
    np.testing.assert_allclose(obs[:, 4:6], env.target, atol=1e-7)
    np.testing.assert_allclose(obs[:, 8:10], env.tip() - env.target, atol=1e-7)
    assert np.all(obs[:, 10] == 0)
    # stretched along x at q = 0: tip at (0.21, 0)
    env.q[:] = 0.0
    np.testing.assert_allclose(env.tip(), [[0.21, 0.0]] * 5, atol=1e-12)
    a = np.full((5, 2), 0.5)
    d0 = np.linalg.norm(env.tip() - env.target, axis=1)
    _, r, done = env.step(a)
    np.testing.assert_allclose(r, -d0 - 0.5, atol=1e-6)                                       # -distance - |a|^2
    assert not done.any()
    for _ in range(49):
        _, _, done = env.step(np.zeros((5, 2)))
    assert done.all()


def test_reacher_free_arm_loses_energy_only_to_damping_and_a_pd_controller_reaches():
    env = ReacherVec(1, seed=2)
    env.reset()
    env.q[:], env.qd[:] = [[0.3, -0.5]], [[3.0, -2.0]]

    def kinetic():      # 1/2 qd' M qd
        acc0 = env._accel(env.q, np.zeros((1, 2)), np.array([[1.0, 0.0]]))[0]
        acc1 = env._accel(env.q, np.zeros((1, 2)), np.array([[0.0, 1.0]]))[0]
        M = np.linalg.inv(np.stack([acc0, acc1], axis=1))
        return 0.5 * env.qd[0] @ M @ env.qd[0]
    e = [kinetic()]
    for _ in range(20):
        env.step(np.zeros((1, 2)))
        e.append(kinetic())
    assert all(b < a for a, b in zip(e, e[1:])) and e[-1] > 0.3 * e[0]          # decays smoothly, time constant ~ 0.5 s
    env = ReacherVec(16, seed=3)
    obs = env.reset()
    d0 = np.linalg.norm(obs[:, 8:10], axis=1)
    for _ in range(50):
        # move the tip along the error with the transposed Jacobian
        q = env.q
        a1, a12 = q[:, 0], q[:, 0] + q[:, 1]
        J = np.stack([np.stack([-env.L1 * np.sin(a1) - env.L2 * np.sin(a12), -env.L2 * np.sin(a12)], 1),
                      np.stack([env.L1 * np.cos(a1) + env.L2 * np.cos(a12), env.L2 * np.cos(a12)], 1)], 1)
        err = env.target - env.tip()
        act = np.clip(2.0 * np.einsum("eij,ei->ej", J, err) * 50 - 0.05 * env.qd, -1, 1)
        obs, r, done = env.step(act)
        if done.all():
            break
    assert np.linalg.norm(obs[:, 8:10], axis=1).mean() < 0.35 * d0.mean()


def test_double_pendulum_energy_equilibrium_reward_and_termination():
    env = InvertedDoublePendulumVec(3, seed=4)
    env.reset()
    env.q[:], env.qd[:] = 0.0, 0.0
    obs, r, done = env.step(np.zeros((3, 1)))
    np.testing.assert_allclose(env.q, 0.0, atol=1e-12)                          # upright is an equilibrium
    np.testing.assert_allclose(r, 10.0 - (1.2 - 2.0) ** 2, atol=1e-9)           # alive bonus - tip height penalty
    np.testing.assert_allclose(obs[0], [0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0], atol=1e-7)

    def energy(e):
        m0, m1, m2, l1, l2, g = e.M0, e.M1, e.M2, e.L1, e.L2, e.G
        x, t1, t2 = e.q.T
        xd, w1, w2 = e.qd.T
        v1 = (xd + l1 / 2 * np.cos(t1) * w1) ** 2 + (l1 / 2 * np.sin(t1) * w1) ** 2
        v2 = (xd + l1 * np.cos(t1) * w1 + l2 / 2 * np.cos(t2) * w2) ** 2 + (l1 * np.sin(t1) * w1 + l2 / 2 * np.sin(t2) * w2) ** 2
        kin = 0.5 * m0 * xd ** 2 + 0.5 * m1 * v1 + 0.5 * m2 * v2 + m1 * l1 ** 2 / 24 * w1 ** 2 + m2 * l2 ** 2 / 24 * w2 ** 2
        pot = m1 * g * l1 / 2 * np.cos(t1) + m2 * g * (l1 * np.cos(t1) + l2 / 2 * np.cos(t2))
        return kin + pot
    env.SUBSTEPS = 200                                                            # fine steps: the integrator is symplectic Euler
    env.q[:] = [[0.0, 0.05, -0.03]] * 3
    e0 = energy(env)
    fell_at = None
    for t in range(40):
        _, _, done = env.step(np.zeros((3, 1)))
        if fell_at is None:
            np.testing.assert_allclose(energy(env), e0, rtol=2e-3)                # no force, no friction: energy stays
        if done.any() and fell_at is None:
            fell_at = t
    assert fell_at is not None and 5 < fell_at < 40                               # it falls, and the episode ends at y_tip <= 1


def test_double_pendulum_is_balanced_by_linear_feedback():
    scipy_linalg = pytest.importorskip("scipy.linalg")
    env = InvertedDoublePendulumVec(1, seed=5)
    # linearise q'' = f(q, q', u) around the upright rest state and solve the LQR problem
    eps = 1e-5
    z0 = np.zeros((1, 3))
    f0 = env._accel(z0, z0, np.zeros(1))[0]
    A = np.zeros((6, 6))
    A[:3, 3:] = np.eye(3)
    for j in range(3):
        dq = z0.copy()
        dq[0, j] = eps
        A[3:, j] = (env._accel(dq, z0, np.zeros(1))[0] - f0) / eps
    Bm = np.zeros((6, 1))
    Bm[3:, 0] = (env._accel(z0, z0, np.array([eps]))[0] - f0) / eps
    P = scipy_linalg.solve_continuous_are(A, Bm, np.diag([1.0, 10.0, 10.0, 0.1, 0.1, 0.1]), np.eye(1) * 0.01)
    K = np.linalg.solve(np.eye(1) * 0.01, Bm.T @ P)
    env.reset()
    total = 0.0
    for t in range(400):
        state = np.concatenate([env.q[0], env.qd[0]])
        u = -(K @ state)[0] / env.FORCE
        _, r, done = env.step(np.array([[u]]))
        total += r[0]
        assert not done[0], t
    assert total / 400 > 9.2


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Reacher-v2", "InvertedDoublePendulum-v2"])
def test_device_twins_follow_the_host_dynamics(name):
    """csrc/analytic_env.hip (fp32) against envs/analytic.py (fp64): one step from identical states and actions,
    then the host follows the device."""
    import torch
    from d4pg_pytorch_amd.envs.device_envs import DeviceAnalytic, make_device_env
    E = 512
    reach = name.startswith("Reacher")
    limit = 50 if reach else 200
    cfg = default_config(env=name, model="d4pg", device="cuda", max_ep_length=limit)
    dev = make_device_env(cfg, E, "cuda", seed=3)
    assert isinstance(dev, DeviceAnalytic) and dev.time_limit == limit
    A = dev.action_dim
    obs = torch.zeros(E, 11, device="cuda")
    dev.reset(obs)
    torch.cuda.synchronize()
    host = (ReacherVec if reach else InvertedDoublePendulumVec)(E, seed=0, time_limit=limit)
    host.reset()
    st = dev.state.cpu().double().numpy()
    if reach:
        assert np.abs(st[:, :2]).max() <= 0.1 and np.abs(st[:, 2:4]).max() <= 0.005
        rad = np.linalg.norm(st[:, 4:6], axis=1)
        assert rad.max() <= 0.2 and 0.1 < rad.mean() < 0.16                     # uniform in the disk: mean radius 2/3 x 0.2
    else:
        assert np.abs(st[:, :2]).max() <= 0.1 and np.abs(st[:, 2] - st[:, 1]).max() <= 0.1 + 1e-6
        assert 0.07 < st[:, 3:6].std() < 0.13                                   # normal deviates x 0.1

    def follow():
        st = dev.state.cpu().double().numpy()
        if reach:
            host.q[:], host.qd[:], host.target[:] = st[:, 0:2], st[:, 2:4], st[:, 4:6]
        else:
            host.q[:], host.qd[:] = st[:, 0:3], st[:, 3:6]
        host.t[:] = dev.t.cpu().numpy()
    follow()
    np.testing.assert_allclose(obs.cpu().numpy(), host._obs(), rtol=1e-5, atol=1e-6)
    nobs, rew, done, trunc = (torch.zeros(E, 11, device="cuda"), torch.zeros(E, device="cuda"),
                              torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"))
    g = torch.Generator().manual_seed(1)
    ended_total = 0
    for t in range(260):
        act = (torch.rand(E, A, generator=g) * 2.6 - 1.3).cuda()               # beyond [-1, 1]: clipped (reacher: cost on the raw action)
        dev.step(act, obs, nobs, rew, done, trunc)
        torch.cuda.synchronize()
        o_h, r_h, d_h = host._step(act.cpu().numpy())
        np.testing.assert_allclose(nobs.cpu().numpy(), o_h, rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(rew.cpu().numpy(), r_h, rtol=2e-3, atol=2e-3)
        d_d = done.cpu().numpy() > 0.5
        assert (d_d == d_h).mean() > 0.995 and not trunc.any()                 # a tip within rounding of y = 1 may differ
        ended = torch.from_numpy(d_d).cuda()
        ended_total += int(d_d.sum())
        assert torch.equal(obs[~ended], nobs[~ended]) and (dev.t[ended] == 0).all()
        follow()
    mean_ret, count = dev.episode_stats()
    assert count == ended_total and count >= E
    if reach:
        assert -120.0 < mean_ret < -20.0          # 50 steps of -|a|^2 (about 1.1 at random) - distance
    else:
        assert 20.0 < mean_ret < 200.0            # a few steps of about 9 before the tip drops below 1


def test_render_frames_for_the_replay_gif():
    for cls in (ReacherVec, InvertedDoublePendulumVec):
        env = cls(2, seed=0)
        env.reset()
        frame = env.render(1)
        assert frame.shape == (200, 300, 3) and frame.dtype == np.uint8 and (frame.sum(2) > 0).sum() > 200
