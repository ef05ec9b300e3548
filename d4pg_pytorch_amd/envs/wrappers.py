"""Single-environment wrappers with the reference's class names and call signatures
(``env/env_wrapper.py:4-38``, ``env/pendulum.py``, ``env/lunar_lander_continous.py``,
``env/bipedal.py``): ``reset() -> obs``, ``step(action) -> (obs, reward, done)`` with
un-batched arrays.  Each one is a view of a one-environment :class:`VecEnv` obtained
from the same factory the agents use, so it resolves to gym / gymnasium when one is
installed and to the native or synthetic implementation otherwise.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .base import VecEnv

_SHAPES = {   # env id -> (state_dim, action_dim, action_low, action_high)
    "Pendulum-v0": (3, 1, -2.0, 2.0), "Pendulum-v1": (3, 1, -2.0, 2.0),
    "LunarLanderContinuous-v2": (8, 2, -1.0, 1.0), "BipedalWalker-v2": (24, 4, -1.0, 1.0),
}


class EnvWrapper:
    reward_scale = 1.0

    def __init__(self, env_name: str, config: Optional[Dict] = None, seed: Optional[int] = None):
        from . import create_env_wrapper
        self.env_name = env_name
        cfg = dict(config or {})
        cfg["env"] = env_name
        if "state_dim" not in cfg:
            if env_name not in _SHAPES:
                raise ValueError(f"state_dim / action_dim of '{env_name}' must be given in the config")
            cfg["state_dim"], cfg["action_dim"], cfg["action_low"], cfg["action_high"] = _SHAPES[env_name]
        cfg.setdefault("action_low", -1.0)
        cfg.setdefault("action_high", 1.0)
        cfg.setdefault("max_ep_length", 1000)
        self.env: VecEnv = create_env_wrapper(cfg, num_envs=1, seed=seed)
        if self.reward_scale != 1.0:
            self.env.reward_scale = self.reward_scale

    def reset(self) -> np.ndarray:
        return self.env.reset()[0]

    def get_random_action(self) -> np.ndarray:
        return self.env.get_random_action()[0]

    def step(self, action):
        obs, rew, done = self.env.step(np.asarray(action, np.float32).reshape(1, -1))
        return obs[0], float(rew[0]), bool(done[0])

    def set_random_seed(self, seed: int) -> None:
        self.env.set_random_seed(seed)

    def render(self) -> np.ndarray:
        return self.env.render(0)

    def close(self) -> None:
        self.env.close()

    def get_action_space(self):
        return self.env.get_action_space()

    def normalise_state(self, state):
        return self.env.normalise_state(state)

    def normalise_reward(self, reward):
        return self.env.normalise_reward(reward)


class PendulumWrapper(EnvWrapper):
    reward_scale = 100.0    # env/pendulum.py:13-14

    def __init__(self, config: Optional[Dict] = None, seed: Optional[int] = None):
        super().__init__("Pendulum-v0", config, seed)


class LunarLanderContinous(EnvWrapper):      # (sic) the reference's spelling
    reward_scale = 100.0    # env/lunar_lander_continous.py:12-13

    def __init__(self, config: Optional[Dict] = None, seed: Optional[int] = None):
        super().__init__("LunarLanderContinuous-v2", config, seed)


class BipedalWalker(EnvWrapper):
    def __init__(self, config: Optional[Dict] = None, seed: Optional[int] = None):
        super().__init__("BipedalWalker-v2", config, seed)
