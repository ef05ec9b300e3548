"""Adapter for OpenAI gym / gymnasium environments, when one is installed.

Handles both API generations: the reference targets the old one
(``reset() -> obs``, ``step() -> (obs, r, done, info)``, ``env.seed``;
``env/env_wrapper.py:9-22``); gymnasium returns ``(obs, info)`` and a
5-tuple with separate ``terminated`` / ``truncated`` flags.  With the new API a
truncated episode reports ``done=True`` here too, matching what the old
``TimeLimit`` wrapper handed to the reference.
"""
from __future__ import annotations

import numpy as np

from .base import VecEnv


def _import_gym():
    try:
        import gymnasium as gym   # type: ignore
        return gym
    except Exception:
        pass
    try:
        import gym                # type: ignore
        return gym
    except Exception:
        return None


def gym_available() -> bool:
    return _import_gym() is not None


class GymVec(VecEnv):
    def __init__(self, env_name: str, num_envs: int, state_dim: int, action_dim: int,
                 action_low=-1.0, action_high=1.0, seed=None, reward_scale: float = 1.0):
        gym = _import_gym()
        if gym is None:
            raise ImportError("neither gymnasium nor gym is installed")
        super().__init__(env_name, num_envs, state_dim, action_dim, action_low, action_high, seed)
        self.reward_scale = float(reward_scale)
        self.envs = [gym.make(env_name) for _ in range(num_envs)]
        self._seed = seed
        obs_dim = int(np.prod(self.envs[0].observation_space.shape))
        act_dim = int(np.prod(self.envs[0].action_space.shape))
        if (obs_dim, act_dim) != (state_dim, action_dim):
            raise ValueError(f"{env_name}: env dims ({obs_dim}, {act_dim}) != config ({state_dim}, {action_dim})")

    def _reset_env(self, e: int):
        env = self.envs[e]
        try:
            out = env.reset(seed=None if self._seed is None else self._seed + e)
        except TypeError:
            if self._seed is not None and hasattr(env, "seed"):
                env.seed(self._seed + e)
            out = env.reset()
        obs = out[0] if isinstance(out, tuple) else out
        return np.asarray(obs, np.float32).reshape(-1)

    def _step(self, actions):
        obs, rew, done = [], [], []
        for env, a in zip(self.envs, actions):
            out = env.step(a.ravel())
            if len(out) == 5:
                o, r, term, trunc, _ = out
                d = bool(term) or bool(trunc)
            else:
                o, r, d, _ = out
            obs.append(np.asarray(o, np.float32).reshape(-1))
            rew.append(r)
            done.append(d)
        return np.stack(obs), np.asarray(rew, np.float32), np.asarray(done, bool)

    def render(self, e: int = 0):
        env = self.envs[e]
        try:
            return env.render(mode="rgb_array")
        except TypeError:
            return env.render()

    def close(self):
        for env in self.envs:
            env.close()
