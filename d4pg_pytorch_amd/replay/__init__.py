"""Replay storage.

* :class:`HbmReplay` (``hbm_replay.py``)  — rows and the priority tree live in the
  learner GPU's HBM; sampling, gathering and priority write-back are kernels.
* :class:`HostReplay` (``host_replay.py``) — numpy twin for the CPU topology.
* :class:`ReplayBuffer`, :class:`PrioritizedReplayBuffer`, :class:`D3PGReplayBuffer`,
  :func:`create_replay_buffer` — the reference's buffer classes and factory
  (``models/d4pg/replay_buffer.py``, ``models/d3pg/utils.py``) with their call
  signatures, implemented on :class:`HostReplay`.
* ``segment_tree.py`` — ``SumSegmentTree`` / ``MinSegmentTree`` with the reference interface.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Optional

import numpy as np

from .host_replay import HostReplay
from .segment_tree import MinSegmentTree, SegmentTree, SumSegmentTree


def _cfg(state_dim: int, action_dim: int, size: int, prioritized: bool, alpha: float, seed) -> Dict:
    return {"state_dim": state_dim, "action_dim": action_dim, "replay_mem_size": size,
            "replay_memory_prioritized": int(prioritized), "priority_alpha": alpha,
            "random_seed": 0 if seed is None else seed}


class ReplayBuffer:
    """Uniform replay with the reference's D4PG interface (``replay_buffer.py:15-86``):
    ``add(obs_t, action, reward, obs_tp1, done, gamma)``, ``sample(batch_size)`` returning
    the 8-element batch, ``remove(n)``, ``dump(save_dir)``, ``len()``.

    The row storage is allocated on the first ``add`` (dimensions come from the data) and —
    unlike the reference, whose ``_maxsize`` is never enforced — overwrites the oldest
    row once ``size`` rows are stored."""

    prioritized = False

    def __init__(self, size: int, alpha: float = 0.6, seed: Optional[int] = None):
        self._maxsize = int(size)
        self._alpha = float(alpha)
        self._seed = seed
        self._impl: Optional[HostReplay] = None

    def _ensure(self, obs, action) -> HostReplay:
        if self._impl is None:
            S, A = int(np.size(obs)), int(np.size(action))
            self._impl = HostReplay(_cfg(S, A, self._maxsize, self.prioritized, self._alpha, self._seed))
        return self._impl

    def __len__(self) -> int:
        return 0 if self._impl is None else len(self._impl)

    def add(self, obs_t, action, reward, obs_tp1, done, gamma) -> None:
        self._ensure(obs_t, action).add(np.ravel(obs_t), np.ravel(action), reward, np.ravel(obs_tp1), done, gamma)

    def sample(self, batch_size: int, **kwargs):
        """``[obs, act, rew, next_obs, done, gamma, weights, idxes]`` (weights are ones here)."""
        assert self._impl is not None and len(self._impl) > 0, "sampling from an empty buffer"
        return list(self._impl.sample(batch_size, **kwargs))

    def remove(self, num_samples: int) -> None:
        """Forget the ``num_samples`` oldest transitions (``replay_buffer.py:37-39``)."""
        im = self._impl
        if im is None or num_samples <= 0:
            return
        n = min(int(num_samples), im.size)
        oldest = (im.head - im.size) % im.capacity
        order = (oldest + np.arange(im.size)) % im.capacity
        keep = im.rows[order[n:]].copy()
        prio = im.tree.sum[im.tree.cap + order[n:]].copy() if im.prioritized else None
        im.rows[: keep.shape[0]] = keep
        im.size, im.head = keep.shape[0], keep.shape[0] % im.capacity
        if im.prioritized:
            im.tree.set(np.arange(im.capacity), np.concatenate([prio, np.zeros(im.capacity - prio.size)]))

    def dump(self, save_dir: str) -> str:
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, "replay_buffer.pkl")
        if self._impl is None:
            with open(path, "wb") as fh:
                pickle.dump({"format": "d4pg_pytorch_amd.replay.v1", "rows": np.zeros((0, 0), np.float32)}, fh)
        else:
            self._impl.dump(path)
        print(f"Buffer dumped to {path}")
        return path


class PrioritizedReplayBuffer(ReplayBuffer):
    """Proportional prioritised replay (``replay_buffer.py:89-215``): new rows get
    ``max_priority ** alpha``, ``sample(batch_size, beta)`` draws one index per equal-mass
    stratum and returns importance weights normalised by the largest possible weight,
    ``update_priorities(idxes, priorities)`` stores ``priority ** alpha``.
    (The reference constructor also demands an unused ``save_dir`` and cannot be built by
    its own factory; here it is optional.)"""

    prioritized = True

    def __init__(self, size: int, alpha: float, save_dir: Optional[str] = None, seed: Optional[int] = None):
        assert alpha >= 0
        super().__init__(size, alpha, seed)
        self.save_dir = save_dir

    def sample(self, batch_size: int, beta: float = 0.4):
        assert beta > 0
        return super().sample(batch_size, beta=beta)

    def update_priorities(self, idxes, priorities) -> None:
        assert len(idxes) == len(priorities)
        assert np.all(np.asarray(priorities) > 0)
        assert self._impl is not None and np.all((0 <= np.asarray(idxes)) & (np.asarray(idxes) < len(self._impl)))
        self._impl.update_priorities(idxes, priorities)


class D3PGReplayBuffer:
    """The D3PG/DDPG ring buffer interface (``models/d3pg/utils.py:6-53``):
    ``add(state, action, next_state, reward, done)`` and ``sample(batch_size)`` returning
    ``(state, action, reward[B,1], next_state, not_done[B,1])``."""

    def __init__(self, state_dim: int, action_dim: int, max_size: int = int(1e6), save_dir: Optional[str] = None,
                 seed: Optional[int] = None):
        self.max_size = int(max_size)
        self.save_dir = save_dir
        self._impl = HostReplay(_cfg(state_dim, action_dim, self.max_size, False, 0.6, seed))

    def __len__(self) -> int:
        return len(self._impl)

    def add(self, state, action, next_state, reward, done, gamma: float = 1.0) -> None:
        self._impl.add(np.ravel(state), np.ravel(action), reward, np.ravel(next_state), done, gamma)

    def sample(self, batch_size: int):
        s, a, r, s2, d, _, _, _ = self._impl.sample(batch_size)
        return s, a, r.reshape(-1, 1), s2, (1.0 - d).reshape(-1, 1)

    def dump(self, save_dir: Optional[str] = None) -> str:
        save_dir = save_dir or self.save_dir or "."
        os.makedirs(save_dir, exist_ok=True)
        path = os.path.join(save_dir, "replay_buffer.pkl")
        self._impl.dump(path)
        return path


def create_replay_buffer(config: Dict):
    """``models/d4pg/replay_buffer.py:218-223``: prioritised iff ``replay_memory_prioritized``."""
    size = int(config["replay_mem_size"])
    seed = config.get("random_seed")
    if config.get("replay_memory_prioritized"):
        return PrioritizedReplayBuffer(size=size, alpha=float(config["priority_alpha"]),
                                       save_dir=config.get("results_path"), seed=seed)
    return ReplayBuffer(size, seed=seed)


__all__ = ["HostReplay", "ReplayBuffer", "PrioritizedReplayBuffer", "D3PGReplayBuffer", "create_replay_buffer",
           "SegmentTree", "SumSegmentTree", "MinSegmentTree"]
