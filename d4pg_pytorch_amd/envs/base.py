"""Vectorised environment interface.

The reference wraps one gym environment per agent process
(``env/env_wrapper.py:4-38``) and calls it one step at a time.  Here an agent
process owns ``E`` environments stepped together so the policy runs batched
(``E = envs_per_agent``; ``E = 1`` reproduces the reference behaviour).

Method names follow the reference wrapper — ``reset``, ``step``,
``normalise_state``, ``normalise_reward``, ``render``, ``set_random_seed``,
``get_random_action``, ``get_action_space``, ``close`` — with array-valued
arguments:  ``step(actions[E, A]) -> (next_obs[E, S], reward[E], done[E])``.
Finished environments are reset individually with ``reset_env(e)``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class BoxSpace:
    low: np.ndarray
    high: np.ndarray

    @property
    def shape(self):
        return self.low.shape

    def sample(self, rng: Optional[np.random.Generator] = None) -> np.ndarray:
        rng = rng or np.random.default_rng()
        return rng.uniform(self.low, self.high).astype(np.float32)


class VecEnv:
    """Base class; subclasses implement ``_reset_env`` and ``_step``."""

    #: divisor applied by ``normalise_reward`` (``env/pendulum.py:13-14`` uses 100)
    reward_scale: float = 1.0

    def __init__(self, env_name: str, num_envs: int, state_dim: int, action_dim: int,
                 action_low: float = -1.0, action_high: float = 1.0, seed: Optional[int] = None):
        self.env_name = env_name
        self.num_envs = int(num_envs)
        self.state_dim, self.action_dim = int(state_dim), int(action_dim)
        self.action_space = BoxSpace(np.full(action_dim, action_low, np.float32),
                                     np.full(action_dim, action_high, np.float32))
        self.rng = np.random.default_rng(seed)

    # ---- reference-compatible surface ----
    def reset(self) -> np.ndarray:
        return np.stack([self.reset_env(e) for e in range(self.num_envs)]).astype(np.float32)

    def reset_env(self, e: int) -> np.ndarray:
        return np.asarray(self._reset_env(e), dtype=np.float32)

    def step(self, actions) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.action_dim)
        obs, rew, done = self._step(a)
        return (np.asarray(obs, np.float32), np.asarray(rew, np.float32).reshape(self.num_envs),
                np.asarray(done, bool).reshape(self.num_envs))

    def set_random_seed(self, seed: int) -> None:
        self.rng = np.random.default_rng(seed)

    def get_random_action(self) -> np.ndarray:
        return np.stack([self.action_space.sample(self.rng) for _ in range(self.num_envs)])

    def get_action_space(self) -> BoxSpace:
        return self.action_space

    def normalise_state(self, state):
        return state

    def normalise_reward(self, reward):
        return reward / self.reward_scale if self.reward_scale != 1.0 else reward

    def render(self, e: int = 0) -> np.ndarray:
        """RGB frame ``[H, W, 3]`` uint8 of environment ``e``."""
        return np.zeros((64, 64, 3), dtype=np.uint8)

    def close(self) -> None:
        pass

    # ---- to implement ----
    def _reset_env(self, e: int) -> np.ndarray:
        raise NotImplementedError

    def _step(self, actions: np.ndarray):
        raise NotImplementedError
