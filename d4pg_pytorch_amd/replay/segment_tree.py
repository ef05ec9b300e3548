"""Array-embedded segment trees with the reference's interface
(``models/d4pg/segment_tree.py:10-154``): ``tree[i] = v``, ``tree[i]``,
``reduce(start, end)``, ``SumSegmentTree.sum`` / ``find_prefixsum_idx``,
``MinSegmentTree.min``, ``remove_items``.

The reference keeps the nodes in a Python list and recurses per query; here the
nodes are one float64 numpy array, point updates and queries walk the
``log2(capacity)`` levels iteratively, and every operation also accepts index
*arrays* (one vectorised pass per level for the whole batch), which is what the
host replay uses.  The GPU twin is the fan-out-64 tree in ``csrc/replay.hip``.
"""
from __future__ import annotations

import operator
from typing import Callable

import numpy as np


class SegmentTree:
    def __init__(self, capacity: int, operation: Callable, neutral_element: float):
        assert capacity > 0 and capacity & (capacity - 1) == 0, "capacity must be positive and a power of 2."
        self._capacity = int(capacity)
        self._operation = operation
        self._neutral = float(neutral_element)
        self._value = np.full(2 * capacity, neutral_element, dtype=np.float64)
        if operation is operator.add:
            self._vec = np.add
        elif operation is min:
            self._vec = np.minimum
        elif operation is max:
            self._vec = np.maximum
        else:
            self._vec = np.frompyfunc(operation, 2, 1)

    # ------------------------------------------------------------------ queries
    def reduce(self, start: int = 0, end=None) -> float:
        """``operation`` over the contiguous leaves ``[start, end)`` (negative / ``None`` ``end``
        count from the back like a slice)."""
        if end is None:
            end = self._capacity
        if end < 0:
            end += self._capacity
        assert 0 <= start <= end <= self._capacity
        res = self._neutral
        lo, hi = start + self._capacity, end + self._capacity
        while lo < hi:
            if lo & 1:
                res = self._operation(res, self._value[lo])
                lo += 1
            if hi & 1:
                hi -= 1
                res = self._operation(res, self._value[hi])
            lo >>= 1
            hi >>= 1
        return float(res)

    def __getitem__(self, idx):
        idx = np.asarray(idx)
        assert np.all((0 <= idx) & (idx < self._capacity))
        out = self._value[self._capacity + idx]
        return float(out) if out.ndim == 0 else out

    # ------------------------------------------------------------------ updates
    def __setitem__(self, idx, val) -> None:
        """Point update; ``idx`` may be an array (duplicates: the last value wins)."""
        idx = np.asarray(idx, dtype=np.int64).reshape(-1)
        val = np.broadcast_to(np.asarray(val, dtype=np.float64).reshape(-1), idx.shape)
        assert np.all((0 <= idx) & (idx < self._capacity))
        pos = idx + self._capacity
        self._value[pos] = val
        pos = np.unique(pos >> 1)
        while pos.size and pos[-1] >= 1:
            self._value[pos] = self._vec(self._value[2 * pos], self._value[2 * pos + 1])
            if pos[-1] == 1:
                break
            pos = np.unique(pos >> 1)

    def remove_items(self, num_items: int) -> None:
        """Drop the first ``num_items`` leaves: the rest shift left, the tail becomes neutral
        (``segment_tree.py:92-102``), followed by a full bottom-up rebuild."""
        n = int(num_items)
        assert 0 <= n <= self._capacity
        leaves = self._value[self._capacity:].copy()
        leaves[: self._capacity - n] = leaves[n:]
        leaves[self._capacity - n:] = self._neutral
        self._value[self._capacity:] = leaves
        for level_start in (1 << np.arange(int(np.log2(self._capacity)) - 1, -1, -1)):
            pos = np.arange(level_start, 2 * level_start)
            self._value[pos] = self._vec(self._value[2 * pos], self._value[2 * pos + 1])


class SumSegmentTree(SegmentTree):
    def __init__(self, capacity: int):
        super().__init__(capacity, operator.add, 0.0)

    def sum(self, start: int = 0, end=None) -> float:
        return super().reduce(start, end)

    def find_prefixsum_idx(self, prefixsum):
        """Highest ``i`` with ``sum(leaves[:i]) <= prefixsum``; scalar or array argument."""
        scalar = np.ndim(prefixsum) == 0
        mass = np.atleast_1d(np.asarray(prefixsum, dtype=np.float64)).copy()
        assert np.all((0 <= mass) & (mass <= self.sum() + 1e-5))
        pos = np.ones(mass.shape, dtype=np.int64)
        while pos[0] < self._capacity:
            left = 2 * pos
            right = mass >= self._value[left]
            mass = np.where(right, mass - self._value[left], mass)
            pos = np.where(right, left + 1, left)
        out = pos - self._capacity
        return int(out[0]) if scalar else out


class MinSegmentTree(SegmentTree):
    def __init__(self, capacity: int):
        super().__init__(capacity, min, float("inf"))

    def min(self, start: int = 0, end=None) -> float:
        return super().reduce(start, end)
