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
        obs_dim = int(np.prod(self.envs[0].observation_space.shape))
        act_dim = int(np.prod(self.envs[0].action_space.shape))
        if (obs_dim, act_dim) != (state_dim, action_dim):
            raise ValueError(f"{env_name}: env dims ({obs_dim}, {act_dim}) != config ({state_dim}, {action_dim})")
        self._arm_seeds(seed)

    def _arm_seeds(self, seed) -> None:
        """Environment ``e`` is seeded with ``seed + e`` ONCE, at its next reset (the reference seeds once at
        construction, ``env/env_wrapper.py:36-38``); later resets continue that stream, so episodes differ."""
        self._pending_seed = [None if seed is None else int(seed) + e for e in range(self.num_envs)]

    def set_random_seed(self, seed: int) -> None:
        super().set_random_seed(seed)
        self._arm_seeds(seed)

    def _reset_env(self, e: int):
        env = self.envs[e]
        seed, self._pending_seed[e] = self._pending_seed[e], None
        if seed is None:
            out = env.reset()
        else:
            try:
                out = env.reset(seed=seed)           # gymnasium / gym >= 0.26
            except TypeError:
                if hasattr(env, "seed"):
                    env.seed(seed)                   # the old API the reference uses
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
