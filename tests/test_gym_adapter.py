"""GymVec against stub ``gymnasium`` / ``gym`` modules of both API generations (neither is installed here;
the adapter only touches ``make``, ``reset``, ``step``, ``seed``, ``render``, ``close`` and the two spaces)."""
import sys
import types

import numpy as np
import pytest


class _Space:
    def __init__(self, n):
        self.shape = (n,)


class _NewApiEnv:
    """gymnasium style: reset(seed=...) -> (obs, info); step -> 5-tuple; 4-step time limit (truncated)."""
    made = []

    def __init__(self, name):
        self.observation_space, self.action_space = _Space(3), _Space(2)
        self.rng = np.random.default_rng(12345)
        self.seeds, self.t, self.closed = [], 0, False
        _NewApiEnv.made.append(self)

    def reset(self, seed=None, options=None):
        self.seeds.append(seed)
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        self.t = 0
        return self.rng.standard_normal(3), {"info": 1}

    def step(self, a):
        assert a.shape == (2,)
        self.t += 1
        return self.rng.standard_normal(3), float(a.sum()), False, self.t >= 4, {}

    def render(self):
        return np.zeros((4, 4, 3), np.uint8)

    def close(self):
        self.closed = True


class _OldApiEnv:
    """gym < 0.26 style (what the reference calls): seed(), reset() -> obs, step -> 4-tuple."""
    made = []

    def __init__(self, name):
        self.observation_space, self.action_space = _Space(3), _Space(2)
        self.rng = np.random.default_rng(999)
        self.seed_calls, self.t = [], 0
        _OldApiEnv.made.append(self)

    def seed(self, s):
        self.seed_calls.append(s)
        self.rng = np.random.default_rng(s)

    def reset(self):
        self.t = 0
        return self.rng.standard_normal(3)

    def step(self, a):
        self.t += 1
        return self.rng.standard_normal(3), 1.0, self.t >= 3, {}

    def render(self, mode="human"):
        assert mode == "rgb_array"
        return np.ones((4, 4, 3), np.uint8)

    def close(self):
        pass


def _install(monkeypatch, name, env_cls):
    env_cls.made = []
    mod = types.ModuleType(name)
    mod.make = lambda env_name: env_cls(env_name)
    monkeypatch.setitem(sys.modules, name, mod)
    if name == "gym":          # make sure the new-API module is not found first
        monkeypatch.setitem(sys.modules, "gymnasium", None)


def test_new_api_seeds_once_and_truncation_reads_as_done(monkeypatch):
    from d4pg_pytorch_amd.envs.gym_adapter import GymVec, gym_available
    _install(monkeypatch, "gymnasium", _NewApiEnv)
    assert gym_available()
    env = GymVec("Stub-v0", 2, 3, 2, seed=7)
    first = env.reset()
    assert first.shape == (2, 3) and first.dtype == np.float32
    assert [e.seeds for e in _NewApiEnv.made] == [[7], [8]]          # seed + e, at the first reset
    again = env.reset_env(0)
    assert _NewApiEnv.made[0].seeds == [7, None]                      # never reseeded: episodes differ
    assert not np.allclose(again, first[0])
    done_seen = []
    for _ in range(4):
        obs, rew, done = env.step(np.ones((2, 2), np.float32))
        done_seen.append(done.copy())
        assert obs.shape == (2, 3) and rew.tolist() == [2.0, 2.0]
    assert not done_seen[2].any() and done_seen[3].all()             # truncated (time limit) reads as done
    env.set_random_seed(100)                                          # re-arms the one-shot seeding
    env.reset()
    assert _NewApiEnv.made[0].seeds[-1] == 100 and _NewApiEnv.made[1].seeds[-1] == 101
    assert env.render(0).shape == (4, 4, 3)
    env.close()
    assert all(e.closed for e in _NewApiEnv.made)


def test_old_api_uses_seed_method_once(monkeypatch):
    from d4pg_pytorch_amd.envs.gym_adapter import GymVec
    _install(monkeypatch, "gym", _OldApiEnv)
    env = GymVec("Stub-v0", 2, 3, 2, seed=3)
    a = env.reset()
    assert [e.seed_calls for e in _OldApiEnv.made] == [[3], [4]]
    b = env.reset()
    assert [e.seed_calls for e in _OldApiEnv.made] == [[3], [4]] and not np.allclose(a, b)
    for t in range(3):
        obs, rew, done = env.step(np.zeros((2, 2), np.float32))
    assert done.all() and rew.tolist() == [1.0, 1.0]
    assert env.render(1).sum() == 48


def test_same_seed_same_first_episode_and_unseeded(monkeypatch):
    from d4pg_pytorch_amd.envs.gym_adapter import GymVec
    _install(monkeypatch, "gymnasium", _NewApiEnv)
    a, b = GymVec("Stub-v0", 1, 3, 2, seed=5).reset(), GymVec("Stub-v0", 1, 3, 2, seed=5).reset()
    np.testing.assert_array_equal(a, b)
    env = GymVec("Stub-v0", 1, 3, 2, seed=None)
    env.reset()
    assert _NewApiEnv.made[-1].seeds == [None]


def test_dim_mismatch_and_missing_module(monkeypatch):
    from d4pg_pytorch_amd.envs import gym_adapter
    _install(monkeypatch, "gymnasium", _NewApiEnv)
    with pytest.raises(ValueError, match="env dims"):
        gym_adapter.GymVec("Stub-v0", 1, 8, 2)
    monkeypatch.setitem(sys.modules, "gymnasium", None)
    monkeypatch.setitem(sys.modules, "gym", None)
    assert not gym_adapter.gym_available()
    with pytest.raises(ImportError):
        gym_adapter.GymVec("Stub-v0", 1, 3, 2)
