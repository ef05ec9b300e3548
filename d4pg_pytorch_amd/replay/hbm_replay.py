"""Replay shard resident in the learner GPU's HBM.

Capabilities of the reference's ``ReplayBuffer`` / ``PrioritizedReplayBuffer``
(``models/d4pg/replay_buffer.py:15-215``) and the D3PG ring
(``models/d3pg/utils.py:6-53``): ``add`` / ``sample`` / ``update_priorities`` /
``__len__`` / ``dump``, ring overwrite of the oldest row, proportional
stratified sampling with importance weights.  What differs is where it lives:
rows, the priority tree and the sampling kernel are all on the device, a batch
is gathered straight into the learner's input buffers, and the host only ever
sees whole staged chunks on their way in.

Row layout (floats): ``[state S | action A | reward | done | gamma**k | next_state S]``
padded to a multiple of 4 — also the record format of the explorer rings.
"""
from __future__ import annotations

import ctypes as C
import pickle
from typing import Dict, Optional

import numpy as np
import torch

from ..models.nets import pad4
from ..ops import hip as H

FANOUT = 64


def row_width(S: int, A: int) -> int:
    return 2 * S + A + 3


def row_ld(S: int, A: int) -> int:
    return pad4(row_width(S, A))


def pack_rows(state, action, reward, next_state, done, gamma) -> np.ndarray:
    """Pack numpy transition columns into the row format (``[n, row_ld]`` float32)."""
    state = np.asarray(state, dtype=np.float32)
    n, S = state.shape
    action = np.asarray(action, dtype=np.float32).reshape(n, -1)
    A = action.shape[1]
    out = np.zeros((n, row_ld(S, A)), dtype=np.float32)
    out[:, :S] = state
    out[:, S:S + A] = action
    out[:, S + A] = np.asarray(reward, dtype=np.float32).reshape(n)
    out[:, S + A + 1] = np.asarray(done, dtype=np.float32).reshape(n)
    out[:, S + A + 2] = np.asarray(gamma, dtype=np.float32).reshape(n)
    out[:, S + A + 3: S + A + 3 + S] = np.asarray(next_state, dtype=np.float32)
    return out


def tree_level_sizes(capacity: int):
    """Entries per level of the fan-out-64 tree, each padded to a multiple of 64."""
    sizes = [(capacity + FANOUT - 1) // FANOUT * FANOUT]
    while sizes[-1] > FANOUT:
        nxt = sizes[-1] // FANOUT
        sizes.append(FANOUT if nxt <= FANOUT else (nxt + FANOUT - 1) // FANOUT * FANOUT)
        if nxt <= FANOUT:
            break
    return sizes


class HbmReplay:
    def __init__(self, cfg: Dict, device, dev_state: torch.Tensor, capacity: Optional[int] = None):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.S, self.A = int(cfg["state_dim"]), int(cfg["action_dim"])
        self.capacity = int(capacity if capacity is not None else cfg["replay_mem_size"])
        self.row_ld = row_ld(self.S, self.A)
        self.prioritized = bool(cfg["replay_memory_prioritized"])
        self.alpha = float(cfg["priority_alpha"])
        self.dev_state = dev_state
        self.rows = torch.zeros(self.capacity, self.row_ld, dtype=torch.float32, device=self.device)
        self.size = 0            # host mirror of the device cursor (inserts are host-ordered)
        self.head = 0
        self.chunk_max = int(cfg.get("ingest_chunk", 1024))
        self._slot_out = torch.zeros(self.chunk_max, dtype=torch.int64, device=self.device)
        self._prio_out = torch.zeros(self.chunk_max, dtype=torch.float32, device=self.device)
        # ---- priority tree ----
        self.level_sizes = tree_level_sizes(self.capacity)
        self.L = len(self.level_sizes) - 1
        assert self.L <= H.MAXL and self.level_sizes[-1] <= FANOUT or self.L == 0
        self.leaf = torch.zeros(self.level_sizes[0], dtype=torch.float32, device=self.device)
        self.sums = [None] + [torch.zeros(n, dtype=torch.float64, device=self.device)
                              for n in self.level_sizes[1:]]
        self.mins = [None] + [torch.full((n,), float("inf"), dtype=torch.float32, device=self.device)
                              for n in self.level_sizes[1:]]
        self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device)      # multi-workgroup tree updates

    # ------------------------------------------------------------ descriptors
    def _replay_desc(self) -> H.ReplayDesc:
        d = H.ReplayDesc()
        d.rows, d.capacity = self.rows.data_ptr(), self.capacity
        d.S, d.A, d.row_ld = self.S, self.A, self.row_ld
        return d

    def _tree_desc(self) -> H.TreeDesc:
        t = H.TreeDesc()
        t.leaf = self.leaf.data_ptr()
        for l in range(1, self.L + 1):
            t.sum[l] = self.sums[l].data_ptr()
            t.mn[l] = self.mins[l].data_ptr()
        for l, n in enumerate(self.level_sizes):
            t.n[l] = n
        t.L = self.L
        return t

    def __len__(self) -> int:
        return self.size

    # ------------------------------------------------------------------ insert
    def insert(self, staged: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Append ``staged`` (``[n, row_ld]`` float32 on this device) at the ring cursor.

        New rows enter the tree with the running max priority
        (``replay_buffer.py:116-122``)."""
        assert staged.device == self.device and staged.dtype == torch.float32
        assert staged.ndim == 2 and staged.shape[1] == self.row_ld and staged.is_contiguous()
        n_total = staged.shape[0]
        off = 0
        while off < n_total:
            # uniform replay: one launch for the whole block; prioritised: chunks the tree update's staging holds
            n = min(n_total - off, self.capacity, self.chunk_max if self.prioritized else n_total)
            part = staged[off: off + n]
            a = H.InsertArgs()
            a.rp = self._replay_desc()
            a.staged = part.data_ptr()
            a.head = self.head
            self.head = (self.head + n) % self.capacity
            self.size = min(self.capacity, self.size + n)
            a.new_size, a.new_head, a.n = self.size, self.head, n
            a.write_tree_inputs = 1 if self.prioritized else 0
            a.slot_out, a.prio_out = self._slot_out.data_ptr(), self._prio_out.data_ptr()
            a.st = self.dev_state.data_ptr()
            H.launch("d4pg_replay_insert", a, stream)
            if self.prioritized:
                self.enqueue_priority_update(self._slot_out, self._prio_out, n, stream)
            off += n

    def add(self, state, action, reward, next_state, done, gamma) -> None:
        """Reference-style single/batched add from host arrays (``replay_buffer.py:30-35``)."""
        rows = pack_rows(np.atleast_2d(state), np.atleast_2d(action), np.atleast_1d(reward),
                         np.atleast_2d(next_state), np.atleast_1d(done), np.atleast_1d(gamma))
        self.insert(torch.from_numpy(rows).to(self.device))

    # ------------------------------------------------------------------ sample
    def enqueue_sample(self, core, do_tick: bool = True, stream=None) -> None:
        """Sample ``core.B`` rows into the core's static input buffers (enqueue only)."""
        H.launch("d4pg_replay_sample", self.sample_args(core, do_tick), stream)

    def sample_args(self, core, do_tick: bool = True) -> H.SampleArgs:
        """Launch descriptor of :meth:`enqueue_sample`."""
        cfg = self.cfg
        a = H.SampleArgs()
        a.rp, a.t, a.st = self._replay_desc(), self._tree_desc(), self.dev_state.data_ptr()
        a.sa, a.s_pi, a.s2_a2 = core.sa.data_ptr(), core.s_pi.data_ptr(), core.s2a2.data_ptr()
        a.reward, a.done, a.gamma = core.reward.data_ptr(), core.done.data_ptr(), core.gamma.data_ptr()
        a.weights, a.idx = core.weights.data_ptr(), core.idx.data_ptr()
        a.B, a.ld_x = core.B, core.ldx
        a.prioritized = 1 if self.prioritized else 0
        a.do_tick = 1 if do_tick else 0
        a.beta1, a.beta2 = float(cfg["adam_beta1"]), float(cfg["adam_beta2"])
        a.per_b0, a.per_b1 = float(cfg["priority_beta_start"]), float(cfg["priority_beta_end"])
        a.total_steps = int(cfg["num_steps_train"])
        return a

    def packed_width(self) -> int:
        """Floats per row of a packed batch: the replay row, the importance weight and the slot."""
        return self.row_ld + H.PACK_EXTRA

    def sample_rows_args(self, out: torch.Tensor, n: int, chunk: int = 1) -> H.SampleRowsArgs:
        """Launch descriptor of :meth:`enqueue_sample_rows` (reusable: set ``.step`` per call)."""
        assert out.dtype == torch.float32 and out.is_contiguous()
        assert out.numel() >= chunk * n * self.packed_width()
        cfg = self.cfg
        a = H.SampleRowsArgs()
        a.rp, a.t, a.st = self._replay_desc(), self._tree_desc(), self.dev_state.data_ptr()
        a.out, a.B, a.K, a.step = out.data_ptr(), int(n), int(chunk), 0
        a.prioritized = 1 if self.prioritized else 0
        a.per_b0, a.per_b1 = float(cfg["priority_beta_start"]), float(cfg["priority_beta_end"])
        a.total_steps = int(cfg["num_steps_train"])
        return a

    def enqueue_sample_rows(self, out: torch.Tensor, n: int, step: int, chunk: int = 1, stream=None) -> None:
        """Draw ``chunk`` batches of ``n`` rows into the packed buffer ``out[chunk, n, packed_width()]``
        (enqueue only).  Batch ``j`` is the one update ``step + j`` consumes: ``step`` positions
        the RNG stream and the importance-exponent schedule.  Unlike :meth:`enqueue_sample`
        nothing is read from or written to the device step counter, so batches can be drawn ahead."""
        a = self.sample_rows_args(out, n, chunk)
        a.step = int(step)
        H.launch("d4pg_replay_sample_rows", a, stream)

    def tree_parts(self, n: int) -> int:
        """Workgroups that share a priority update of ``n`` items (``tree_update_body`` in ``csrc/replay_dev.h``:
        subtrees of the level below the top are dealt to the workgroups): one per 512 items, at most 8, and
        only for trees of three levels or more."""
        if self.L < 2 or n < 1024:
            return 1
        return int(min(8, n // 512))

    def priority_update_args(self, idx: torch.Tensor, prio: torch.Tensor, n: int) -> H.TreeUpdateArgs:
        a = H.TreeUpdateArgs()
        a.t = self._tree_desc()
        a.idx, a.prio, a.st = idx.data_ptr(), prio.data_ptr(), self.dev_state.data_ptr()
        a.n, a.alpha = int(n), self.alpha
        a.parts, a.ticket = self.tree_parts(int(n)), self._ticket.data_ptr()
        return a

    def enqueue_priority_update(self, idx: torch.Tensor, prio: torch.Tensor, n: int, stream=None) -> None:
        H.launch("d4pg_tree_update", self.priority_update_args(idx, prio, n), stream)

    def update_priorities(self, idx: torch.Tensor, prio: torch.Tensor) -> None:
        """``PrioritizedReplayBuffer.update_priorities`` (``replay_buffer.py:187-209``)."""
        if not self.prioritized:
            return
        idx = idx.to(self.device, torch.int64).contiguous()
        prio = prio.to(self.device, torch.float32).contiguous()
        self.enqueue_priority_update(idx, prio, idx.numel())

    # -------------------------------------------------------------- inspection
    def total_priority(self) -> float:
        top = self.sums[self.L] if self.L >= 1 else self.leaf.double()
        return float(top.sum().item())

    def min_priority(self) -> float:
        if self.L >= 1:
            return float(self.mins[self.L].min().item())
        lf = self.leaf[self.leaf > 0]
        return float(lf.min().item()) if lf.numel() else float("inf")

    def snapshot_priorities(self):
        return {"leaf": self.leaf.clone(), "sums": [None if s is None else s.clone() for s in self.sums],
                "mins": [None if m is None else m.clone() for m in self.mins]}

    def restore_priorities(self, snap) -> None:
        self.leaf.copy_(snap["leaf"])
        for l in range(1, self.L + 1):
            self.sums[l].copy_(snap["sums"][l])
            self.mins[l].copy_(snap["mins"][l])

    def rebuild_tree(self) -> None:
        """Recompute every internal level from the leaves (used after a restore)."""
        cur = self.leaf.double()
        cur_min = torch.where(self.leaf > 0, self.leaf, torch.full_like(self.leaf, float("inf")))
        for l in range(1, self.L + 1):
            n = self.level_sizes[l]
            s = cur.view(-1, FANOUT).sum(1)
            m = cur_min.view(-1, FANOUT).min(1).values
            self.sums[l].zero_()
            self.mins[l].fill_(float("inf"))
            self.sums[l][: s.numel()] = s
            self.mins[l][: m.numel()] = m
            cur, cur_min = self.sums[l], self.mins[l]

    # -------------------------------------------------------------------- dump
    def dump(self, path: str) -> None:
        """``save_buffer_on_disk`` (``models/d4pg/engine.py:73-74``): valid rows as one array."""
        valid = self.rows[: self.size].cpu().numpy()
        with open(path, "wb") as fh:
            pickle.dump({"format": "d4pg_pytorch_amd.replay.v1", "state_dim": self.S, "action_dim": self.A,
                         "rows": valid, "head": self.head}, fh)
