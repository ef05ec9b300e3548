"""The native lunar lander (envs/lunar_lander.py): the task of ``LunarLanderContinuous-v2`` that the reference
trains on through gym / Box2D (``env/lunar_lander_continous.py``, ``tests/test_lunar.py``).  gym is not
installable here, so what is pinned is the task definition itself (observation layout, engine thresholds,
shaped reward, terminal rules and their ±100) and that the task is well posed (a hand-written controller
lands it, a random policy crashes it); trajectory parity with Box2D is unpinned by construction."""
import numpy as np
import pytest

from d4pg_pytorch_amd.config import default_config
from d4pg_pytorch_amd.envs import LunarLanderVec, create_env_wrapper
from d4pg_pytorch_amd.envs import lunar_lander as L


def test_factory_builds_the_native_lander_for_the_reference_id():
    env = create_env_wrapper(default_config(env="LunarLanderContinuous-v2", model="d4pg", device="cpu"), 3, seed=0)
    assert isinstance(env, LunarLanderVec) and env.state_dim == 8 and env.action_dim == 2
    obs = env.reset()
    assert obs.shape == (3, 8) and obs.dtype == np.float32
    # starts above the pad centre at the top of the world, upright, feet in the air
    np.testing.assert_allclose(obs[:, 0], 0.0, atol=1e-6)
    np.testing.assert_allclose(obs[:, 1], (L.H - L.PAD_Y - L.LEG_DOWN) / (L.H / 2), rtol=1e-6)
    assert np.all(obs[:, 4:] == 0.0) and np.all(np.abs(obs[:, 2]) <= L.INIT_SPEED * 10 / 50 + 1e-6)
    assert env.normalise_reward(50.0) == 0.5                                  # env/lunar_lander_continous.py:12-13
    assert env.render().shape == (200, 300, 3)


def test_engine_thresholds_and_directions():
    main, side = LunarLanderVec.engine_powers(np.array([[-0.3, 0.4], [0.0, 0.5], [0.2, -0.7], [3.0, 2.0]]))
    np.testing.assert_allclose(main, [0.0, 0.0, 0.6, 1.0])                   # off at <= 0, 50 % .. 100 % above
    np.testing.assert_allclose(side, [0.0, 0.0, -0.7, 1.0])                  # off inside +-0.5
    env = LunarLanderVec(3, seed=1)
    env.reset()
    env.q[:, 2:] = 0.0
    _, _, _ = env.step(np.array([[-1.0, 0.0], [1.0, 0.0], [-1.0, 1.0]]))
    dt = 1.0 / L.FPS
    # free fall; full main engine beats gravity by 17.47 - 10; a side engine pushes along +x and turns clockwise
    np.testing.assert_allclose(env.q[0, 3], L.GRAVITY * dt, rtol=1e-9)
    np.testing.assert_allclose(env.q[1, 3], (L.MAIN_ACC + L.GRAVITY) * dt, rtol=1e-9)
    assert env.q[2, 2] > 0 and env.q[2, 5] == pytest.approx(-L.SIDE_ANG_ACC * dt)
    assert env.q[0, 2] == 0.0 and env.q[0, 5] == 0.0


def test_reward_is_the_change_of_the_shaping_minus_fuel():
    env = LunarLanderVec(2, seed=2)
    o0 = env.reset().astype(np.float64)
    o1, r, d = env.step(np.array([[1.0, 0.0], [-1.0, -1.0]]))
    want = L.lander_shaping(env._obs().astype(np.float64)) - L.lander_shaping(o0)
    np.testing.assert_allclose(r, want - np.array([0.30, 0.03]), atol=2e-3)    # (observations are float32)
    assert not d.any()


def test_crash_flyaway_landing_and_time_limit():
    env = LunarLanderVec(4, seed=3, time_limit=50)
    env.reset()
    # 0: dropped from low height at high speed, tilted -> hull hits the ground; 1: far to the side;
    # 2: placed at rest on its feet; 3: hovering around (time limit)
    env.q[0] = (10.0, L.PAD_Y + 1.0, 0.0, -12.0, 1.2, 0.0)
    env.q[1] = (19.9, 8.0, 6.0, 0.0, 0.0, 0.0)
    env.q[2] = (10.0, L.PAD_Y + L.LEG_DOWN - L.MASS * 10.0 / (2 * L.K_SPRING), 0.0, 0.0, 0.0, 0.0)
    env.q[3] = (10.0, 9.0, 0.0, 0.0, 0.0, 0.0)
    env.prev_shaping[:] = L.lander_shaping(env._obs().astype(np.float64))
    seen = {}
    for t in range(50):
        hover = 0.3 if env.q[3, 3] < 0 else -1.0
        obs, r, d = env.step(np.array([[-1.0, 0.0], [-1.0, 0.0], [-1.0, 0.0], [hover, 0.0]]))
        for e in np.nonzero(d)[0]:
            seen.setdefault(int(e), (t, float(r[e])))
            env.q[e] = (10.0, 9.0, 0.0, 0.0, 0.0, 0.0)                        # park it in the air again
            env.rest[e] = 0
    assert seen[0][1] == -100.0 and seen[0][0] < 10                           # crash
    assert seen[1][1] == -100.0 and seen[1][0] < 5                            # left the viewport
    assert seen[2] == (L.REST_STEPS - 1, 100.0)                               # at rest on both feet for 0.4 s
    assert seen[3][0] == 49 and abs(seen[3][1]) < 20                          # time limit, an ordinary reward


def test_heuristic_controller_lands_and_random_policy_does_not():
    E = 48
    env = LunarLanderVec(E, seed=4)
    obs = env.reset()
    ret, finished, landings = np.zeros(E), [], 0
    for _ in range(600):
        obs, r, d = env.step(L.heuristic_action(obs))
        ret += r
        for e in np.nonzero(d)[0]:
            finished.append(ret[e])
            landings += int(r[e] == 100.0)
            ret[e] = 0.0
            obs[e] = env.reset_env(e)
    assert len(finished) >= E and landings == len(finished) and np.mean(finished) > 200.0
    env = LunarLanderVec(E, seed=5)
    obs, ret, finished = env.reset(), np.zeros(E), []
    rng = np.random.default_rng(0)
    for _ in range(400):
        obs, r, d = env.step(rng.uniform(-1, 1, (E, 2)))
        ret += r
        for e in np.nonzero(d)[0]:
            finished.append(ret[e])
            ret[e] = 0.0
            obs[e] = env.reset_env(e)
    assert len(finished) >= E and np.mean(finished) < -100.0


@pytest.mark.gpu
def test_device_lander_follows_the_host_lander():
    import torch
    from d4pg_pytorch_amd.envs.device_envs import DeviceLunarLander, make_device_env
    E = 256
    cfg = default_config(env="LunarLanderContinuous-v2", model="d4pg", device="cuda", max_ep_length=250)
    dev = make_device_env(cfg, E, "cuda", seed=7)
    assert isinstance(dev, DeviceLunarLander) and dev.time_limit == 250
    obs = torch.zeros(E, 8, device="cuda")
    dev.reset(obs)
    torch.cuda.synchronize()
    st = dev.state.cpu().double().numpy()
    assert np.all(st[:, 0] == 10.0) and np.all(np.abs(st[:, 2:4]) <= L.INIT_SPEED) and st[:, 2].std() > 1.0
    host = LunarLanderVec(E, seed=0, time_limit=250)
    host.reset()

    def follow():
        st = dev.state.cpu().double().numpy()
        host.q[:], host.contact[:], host.prev_shaping[:] = st[:, :6], st[:, 6:8], st[:, 8]
        host.rest[:], host.t[:] = st[:, 9].astype(np.int64), dev.t.cpu().numpy()
    follow()
    nobs, rew, done, trunc = (torch.zeros(E, 8, device="cuda"), torch.zeros(E, device="cuda"),
                              torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"))
    g = torch.Generator().manual_seed(4)
    outcomes = {"crash": 0, "land": 0, "limit": 0}
    for t in range(520):
        # half the environments fly the landing controller (so that landings happen), half act at random
        act = torch.from_numpy(L.heuristic_action(obs.cpu().numpy()))
        act[E // 2:] = torch.rand(E - E // 2, 2, generator=g) * 3 - 1.5          # beyond [-1, 1]: clipped on both sides
        act = act.cuda()
        dev.step(act, obs, nobs, rew, done, trunc)
        torch.cuda.synchronize()
        o_h, r_h, d_h = host._step(act.cpu().numpy())
        # contact springs are stiff and touch-down is a threshold: compare ONE step from identical states (then
        # follow the device), and let a few environments per step disagree on a contact made within rounding
        close = np.all(np.abs(nobs.cpu().numpy() - o_h) <= 2e-3 + 2e-3 * np.abs(o_h), axis=1)
        assert close.mean() > 0.98, (t, np.abs(nobs.cpu().numpy() - o_h).max(axis=1))
        d_d = done.cpu().numpy() > 0.5
        agree = (d_d == d_h) & close
        assert agree.mean() > 0.98
        np.testing.assert_allclose(rew.cpu().numpy()[agree] * 100.0, r_h[agree], rtol=1e-3, atol=0.06)
        outcomes["crash"] += int(((r_h == -100.0) & d_h & agree).sum())
        outcomes["land"] += int(((r_h == 100.0) & d_h & agree).sum())
        outcomes["limit"] += int(((np.abs(r_h) != 100.0) & d_h & agree).sum())
        assert not trunc.any()
        ended = torch.from_numpy(d_d).cuda()
        assert torch.equal(obs[~ended], nobs[~ended]) and (dev.t[ended] == 0).all()
        assert (obs[ended][:, 0] == 0).all() and (obs[ended][:, 4:] == 0).all()   # fresh start above the pad
        follow()
    mean_ret, n = dev.episode_stats()
    assert n > E and outcomes["crash"] > 20 and outcomes["land"] > 20, (n, outcomes)
    assert -400.0 < mean_ret < 300.0


def test_falling_or_sinking_at_half_throttle_is_a_crash():
    """Touch-down speed matters: with the engine off (13.7 units/s at the pad) or held at its minimum power (about
    5 units/s) the legs give way; only a controlled descent lands."""
    for a0 in (-0.5, 0.001):
        env = LunarLanderVec(64, seed=3, time_limit=1000)
        env.reset()
        total, alive = np.zeros(64), np.ones(64, bool)
        for _ in range(1000):
            _, r, d = env._step(np.tile([a0, 0.0], (64, 1)))
            total += r * alive
            alive &= ~d
            if not alive.any():
                break
        assert not alive.any() and (total < 0).mean() > 0.9 and (total > 200).sum() == 0
