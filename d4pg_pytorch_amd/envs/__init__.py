"""Environment factory (``env/utils.py:7-15`` ``create_env_wrapper``).

Resolution order for ``config['env']``:

1. ``synthetic_env: true``  -> shape-only :class:`SyntheticVec`
2. built-in native dynamics -> ``Pendulum-v0/v1`` (:class:`PendulumVec`)
3. gym / gymnasium, if installed -> :class:`GymVec` (reward /100 for Pendulum and
   LunarLanderContinuous like ``env/pendulum.py:13-14`` and
   ``env/lunar_lander_continous.py:12-13``; identity otherwise)
4. native implementation of the task (``LunarLanderContinuous-v2`` -> :class:`LunarLanderVec`, own planar
   rigid-body simulation; ``Reacher-v2`` -> :class:`ReacherVec` and ``InvertedDoublePendulum-v2`` ->
   :class:`InvertedDoublePendulumVec`, closed-form dynamics; ``BipedalWalker-v2``, ``Hopper-v2``, ``Walker2d-v2``,
   ``HalfCheetah-v2`` -> the planar articulated-body engine of ``envs/planar.py``, stepped by the C++ host runtime
   where it is built (``envs/native_planar.py``, ``env_backend``); ``Ant-v2`` -> the 3-D one of ``envs/spatial.py``) or analytic stand-in of the same shape (``InvertedPendulum-v2`` -> cart-pole)
5. shape-only synthetic environment (with a warning), so that every shipped
   id can at least run the full pipeline (every id the reference ships a configuration for resolves above).
"""
from __future__ import annotations

import warnings

from .analytic import InvertedDoublePendulumVec, ReacherVec
from .base import BoxSpace, VecEnv
from .gym_adapter import GymVec, gym_available
from .lunar_lander import LunarLanderVec
from .pendulum import PendulumVec
from .planar import BipedalWalkerVec, HalfCheetahVec, HopperVec, Walker2dVec
from .spatial import AntVec
from .synthetic import CartPoleContinuousVec, SyntheticVec
from .wrappers import BipedalWalker, EnvWrapper, LunarLanderContinous, PendulumWrapper  # noqa: E402,F401

REWARD_SCALE = {"Pendulum-v0": 100.0, "Pendulum-v1": 100.0, "LunarLanderContinuous-v2": 100.0}


def create_env_wrapper(config, num_envs=None, seed=None) -> VecEnv:
    name = config["env"]
    E = int(num_envs if num_envs is not None else config.get("envs_per_agent", 1))
    S, A = config["state_dim"], config["action_dim"]
    lo, hi = config["action_low"], config["action_high"]
    if config.get("synthetic_env"):
        return SyntheticVec(name, E, S, A, lo, hi, seed, episode_len=config["max_ep_length"])
    if name in ("Pendulum-v0", "Pendulum-v1"):
        return PendulumVec(E, seed, env_name=name)
    # Device explorers (gpu_rollout_envs > 0) step the NATIVE task; the host agents of such a job must play the same
    # dynamics, so gym is not used then (nor with env_backend: native) even where it is installed.
    native_only = int(config.get("gpu_rollout_envs") or 0) > 0 or config.get("env_backend") == "native"
    if gym_available() and not native_only:
        try:
            return GymVec(name, E, S, A, lo, hi, seed, REWARD_SCALE.get(name, 1.0))
        except Exception as exc:   # unknown id, missing simulator backend, ...
            warnings.warn(f"gym could not create '{name}' ({exc}); using a stand-in", stacklevel=2)
    if name.startswith("LunarLanderContinuous"):
        return LunarLanderVec(E, seed, time_limit=config["max_ep_length"], env_name=name)
    if name.startswith("Reacher-"):
        return ReacherVec(E, seed, time_limit=min(50, config["max_ep_length"]), env_name=name)
    if name.startswith("InvertedDoublePendulum-"):
        return InvertedDoublePendulumVec(E, seed, time_limit=config["max_ep_length"], env_name=name)
    planar = {"BipedalWalker-": BipedalWalkerVec, "Hopper-": HopperVec, "Walker2d-": Walker2dVec,
              "HalfCheetah-": HalfCheetahVec, "Ant-": AntVec}
    for prefix, cls in planar.items():
        if name.startswith(prefix) and not name.startswith("BipedalWalkerHardcore"):
            backend = config.get("env_backend", "auto")
            if backend not in ("auto", "native", "numpy"):
                raise ValueError(f"env_backend must be auto, native or numpy, got {backend!r}")
            from . import native_planar
            if cls in native_planar.NATIVE and (backend == "native" or (backend == "auto" and native_planar.host_step_fn())):
                cls = native_planar.NATIVE[cls]            # the same task, stepped by the C++ host runtime
            elif cls is AntVec and (backend == "native" or (backend == "auto" and native_planar.ant_host_step_fn())):
                cls = native_planar.NativeAntVec
            return cls(E, seed, time_limit=config["max_ep_length"], env_name=name)
    if name == "InvertedPendulum-v2":
        return CartPoleContinuousVec(E, seed, time_limit=config["max_ep_length"], env_name=name)
    warnings.warn(f"no simulator for '{name}': using the shape-only synthetic environment "
                  f"(state_dim={S}, action_dim={A})", stacklevel=2)
    env = SyntheticVec(name, E, S, A, lo, hi, seed, episode_len=config["max_ep_length"])
    env.reward_scale = REWARD_SCALE.get(name, 1.0)
    return env


__all__ = ["create_env_wrapper", "VecEnv", "BoxSpace", "PendulumVec", "LunarLanderVec", "ReacherVec", "InvertedDoublePendulumVec", "BipedalWalkerVec", "HopperVec",
           "Walker2dVec", "HalfCheetahVec", "AntVec", "SyntheticVec",
           "CartPoleContinuousVec", "GymVec", "gym_available",
           "EnvWrapper", "PendulumWrapper", "LunarLanderContinous", "BipedalWalker"]
