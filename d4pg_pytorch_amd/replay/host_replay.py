"""Host-memory replay for the CPU-only topology (sampler process).

Same capabilities as the reference buffers — uniform ring
(``models/d3pg/utils.py:6-53``) and proportional prioritised replay with
stratified sampling and importance weights
(``models/d4pg/replay_buffer.py:89-209``, ``models/d4pg/segment_tree.py``) — on
the same row format as the HBM replay, with vectorised numpy instead of Python
lists: the sum/min trees are flat arrays updated level by level for a whole
batch of indices at once, and a batch is gathered with one fancy index.

Capacity IS enforced (ring overwrite); ``beta`` is passed per call by the
sampler, which anneals it with the learner step.
"""
from __future__ import annotations

import pickle
from typing import Dict, Optional, Tuple

import numpy as np

from .hbm_replay import pack_rows, row_ld


class SumMinTree:
    """Array-embedded binary trees (sum + min) with batched updates."""

    def __init__(self, capacity: int):
        self.cap = 1
        while self.cap < capacity:
            self.cap *= 2
        self.sum = np.zeros(2 * self.cap, dtype=np.float64)
        self.min = np.full(2 * self.cap, np.inf, dtype=np.float64)

    def set(self, idx: np.ndarray, value: np.ndarray) -> None:
        idx = np.asarray(idx, dtype=np.int64)
        pos = idx + self.cap
        self.sum[pos] = value
        self.min[pos] = np.where(np.asarray(value) > 0, value, np.inf)
        pos = np.unique(pos >> 1)
        while pos.size and pos[0] >= 1:
            self.sum[pos] = self.sum[2 * pos] + self.sum[2 * pos + 1]
            self.min[pos] = np.minimum(self.min[2 * pos], self.min[2 * pos + 1])
            if pos[0] == 1:
                break
            pos = np.unique(pos >> 1)

    @property
    def total(self) -> float:
        return float(self.sum[1])

    @property
    def minimum(self) -> float:
        return float(self.min[1])

    def find_prefix(self, mass: np.ndarray) -> np.ndarray:
        """Highest ``i`` with ``sum(leaf[:i]) <= mass`` for every entry (vectorised descent)."""
        mass = np.asarray(mass, dtype=np.float64).copy()
        pos = np.ones(mass.shape, dtype=np.int64)
        while pos[0] < self.cap:
            left = 2 * pos
            go_right = mass >= self.sum[left]
            mass = np.where(go_right, mass - self.sum[left], mass)
            pos = np.where(go_right, left + 1, left)
        return pos - self.cap


class HostReplay:
    def __init__(self, cfg: Dict, capacity: Optional[int] = None, seed: Optional[int] = None):
        self.S, self.A = int(cfg["state_dim"]), int(cfg["action_dim"])
        self.capacity = int(capacity if capacity is not None else cfg["replay_mem_size"])
        self.row_ld = row_ld(self.S, self.A)
        self.rows = np.zeros((self.capacity, self.row_ld), dtype=np.float32)
        self.size = 0
        self.head = 0
        self.prioritized = bool(cfg["replay_memory_prioritized"])
        self.alpha = float(cfg["priority_alpha"])
        self.max_priority = 1.0
        self.tree = SumMinTree(self.capacity) if self.prioritized else None
        self.rng = np.random.default_rng(cfg["random_seed"] if seed is None else seed)

    def __len__(self) -> int:
        return self.size

    def insert_rows(self, rows: np.ndarray) -> None:
        rows = np.asarray(rows, dtype=np.float32).reshape(-1, self.row_ld)
        n = rows.shape[0]
        if n == 0:
            return
        if n > self.capacity:
            rows = rows[-self.capacity:]
            n = self.capacity
        slots = (self.head + np.arange(n)) % self.capacity
        self.rows[slots] = rows
        self.head = int((self.head + n) % self.capacity)
        self.size = min(self.capacity, self.size + n)
        if self.prioritized:
            self.tree.set(slots, np.full(n, self.max_priority ** self.alpha))

    def add(self, state, action, reward, next_state, done, gamma) -> None:
        self.insert_rows(pack_rows(np.atleast_2d(state), np.atleast_2d(action), np.atleast_1d(reward),
                                   np.atleast_2d(next_state), np.atleast_1d(done), np.atleast_1d(gamma)))

    def sample(self, batch_size: int, beta: float = 0.4) -> Tuple[np.ndarray, ...]:
        """Returns ``(state, action, reward, next_state, done, gamma, weights, indices)``
        — the reference's 8-element batch (``replay_buffer.py:41-53``)."""
        assert self.size > 0
        if self.prioritized:
            total = self.tree.total
            seg = total / batch_size
            mass = (np.arange(batch_size) + self.rng.random(batch_size)) * seg
            idx = np.minimum(self.tree.find_prefix(mass), self.size - 1)
            p = self.tree.sum[idx + self.tree.cap]
            p_min = self.tree.minimum
            w = (p_min / np.maximum(p, 1e-300)) ** beta       # == (P(i) n)^-beta / max_w
            weights = w.astype(np.float32)
        else:
            idx = self.rng.integers(0, self.size, size=batch_size)
            weights = np.ones(batch_size, dtype=np.float32)
        r = self.rows[idx]
        S, A = self.S, self.A
        return (r[:, :S], r[:, S:S + A], r[:, S + A], r[:, S + A + 3:S + A + 3 + S], r[:, S + A + 1],
                r[:, S + A + 2], weights, idx.astype(np.int64))

    def update_priorities(self, idx, priorities) -> None:
        if not self.prioritized:
            return
        idx = np.asarray(idx, dtype=np.int64).reshape(-1)
        pr = np.asarray(priorities, dtype=np.float64).reshape(-1)
        pr = np.where(np.isfinite(pr) & (pr > 0), pr, 1e-4)
        # duplicates: the last occurrence wins, deterministically
        _, last = np.unique(idx[::-1], return_index=True)
        keep = idx.size - 1 - last
        self.tree.set(idx[keep], pr[keep] ** self.alpha)
        self.max_priority = max(self.max_priority, float(pr.max()))

    def dump(self, path: str) -> None:
        with open(path, "wb") as fh:
            pickle.dump({"format": "d4pg_pytorch_amd.replay.v1", "state_dim": self.S, "action_dim": self.A,
                         "rows": self.rows[: self.size].copy(), "head": self.head}, fh)
