"""Replicated data parallelism: sharded replay, all-gathered batches, replicated update.

One learner update of the 400-wide MLPs occupies 32 of the 256 compute units at batch 256
(55 us) and takes 61 us at batch 1024 and 67 us at 2048 — it is bound by the length of the
dependent layer chain, not by arithmetic.  Two gradient all-reduces of 0.7-0.9 MB per
update (``attach_data_parallel`` in ``comm.py``) sit on exactly that critical path and are
pure xGMI latency.  This mode removes them:

* every rank keeps its own HBM replay shard and priority tree, fed by its own explorers,
  and draws ``B`` rows per update from it (``HbmReplay.enqueue_sample_rows``);
* the drawn rows (``B x (2S+A+7)`` floats, ~20 KB at batch 256) are all-gathered over RCCL,
  ``K`` batches (a chunk; 32 for uniform replay, 8 with priorities) per collective, on a side stream while the previous
  chunk is being consumed: one ~0.6 MB/rank collective and one cross-stream event pair per
  chunk (a cross-stream event wait + record around every graph replay costs ~14 us);
* every rank then runs the SAME update on the global batch ``world x B`` with the
  single-graph fused pipeline (no gradient exchange, optimiser fused into the
  weight-gradient kernel).  The kernels are deterministic and the inputs identical, so
  the replicas stay bit-identical; ``check_consistency`` verifies that;
* priorities are written back by the rank that owns the row.

The result is the update a single learner would make on the concatenated global batch —
the same semantics as gradient all-reduce — with batches drawn ``K .. 2K`` updates before
they are consumed (with prioritised replay their priorities lag by that much; the
reference's sampler runs up to 64 batches ahead, ``models/d4pg/engine.py:122``).  Uniform
replay is unaffected.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..ops import hip as H
from .comm import Comm


class ReplicatedDP:
    def __init__(self, core, replay, comm: Comm, local_batch: int, chunk: int = 8, reseed: bool = True):
        assert core.device.type == "cuda", "replicated data parallelism runs on the HIP learner cores"
        self.core, self.replay, self.comm = core, replay, comm
        self.world, self.rank = int(comm.world), int(comm.rank)
        self.Bl, self.K = int(local_batch), max(1, int(chunk))
        if core.B != self.Bl * self.world:
            raise ValueError(f"the learner core must be built for the global batch {self.Bl * self.world}, "
                             f"not {core.B}")
        dev = core.device
        W = replay.packed_width()
        # chunk buffers: [K][Bl][W] drawn locally, [world][K][Bl][W] after the all-gather
        self.local = [torch.zeros(self.K, self.Bl, W, device=dev) for _ in range(2)]
        self.bufs = self.local if self.world == 1 else \
            [torch.zeros(self.world, self.K, self.Bl, W, device=dev) for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev)
        self.drawn = [torch.cuda.Event() for _ in range(2)]      # chunk drawn    (main stream)
        self.ready = [torch.cuda.Event() for _ in range(2)]      # chunk gathered (side stream)
        if self.world > 1:
            # identical replicas
            with torch.no_grad():
                for t in (core.online.flat, core.target.flat, core.adam_m, core.adam_v):
                    comm.broadcast_(t.data if hasattr(t, "data") else t, src=0)
            core._after_load()
        if reseed:                   # a sampling stream of its own for every rank
            st = core.read_dev_state()
            core._write_dev_state(step=core.step, seed=(st.seed + 0x9E3779B97F4A7C15 * (self.rank + 1)),
                                  keep_replay=True)
        if self.world > 1:
            core.grad_sync = None        # replicas exchange rows, not gradients
        if core.replay is not replay:
            core.attach_replay(replay)
        self.step0 = int(core.step)
        self.n = 0                   # updates made through this driver
        self._precaptured = False
        self._sample_args = [replay.sample_rows_args(self.local[p], self.Bl, self.K) for p in range(2)]
        prio_args = None
        if replay.prioritized:       # the rank that owns a row writes its priority back
            sl = slice(self.rank * self.Bl, (self.rank + 1) * self.Bl)
            prio_args = replay.priority_update_args(core.idx[sl], core.prio[sl], self.Bl)
        # [unpack +] update [+ priority write-back]: ONE captured graph serves every update
        if core.supports_direct_input and not replay.prioritized:
            # the next chunk buffer is filled on this stream (one rank) or by an all-gather on the side
            # stream that the main stream only waits for when it starts consuming that chunk (several)
            core.register_direct("gathered", self.bufs[0], self.bufs[1], replay.row_ld, self.Bl, self.K,
                                 critic_ahead=2 if self.world == 1 else 1)
        else:
            core.register_packed("gathered", self.bufs[0], self.bufs[1], replay.row_ld, self.Bl, self.K, prio_args)

    # ------------------------------------------------------------------ prefetch
    def _draw_chunk(self, c: int) -> None:
        """Draw this rank's rows of chunk ``c`` (the batches of updates ``c*K .. c*K + K - 1``) and
        all-gather them.

        The draw runs on the MAIN stream, in order with the replay inserts and the priority
        write-backs, so it always sees a consistent tree; it costs one small kernel per K
        updates.  Only the all-gather runs on the side stream, overlapping the updates of the
        previous chunk.  Buffer ``c & 1`` was last read by chunk ``c - 2``, whose updates precede
        this point on the main stream."""
        p = c & 1
        main = torch.cuda.current_stream(self.core.device)
        a = self._sample_args[p]
        a.step = self.step0 + c * self.K
        H.launch("d4pg_replay_sample_rows", a, main)
        if self.world > 1:
            self.drawn[p].record(main)
            self.side.wait_event(self.drawn[p])
            with torch.cuda.stream(self.side):
                self.comm.all_gather_into_(self.bufs[p], self.local[p])
            self.ready[p].record(self.side)

    # -------------------------------------------------------------------- update
    def update(self) -> torch.Tensor:
        """One synchronous update of all replicas on the global batch."""
        core, K = self.core, self.K
        j, c = self.n % K, self.n // K
        if j == 0 and not (c == 0 and getattr(self, "_primed_by_inspection", False)):
            if c == 0:
                self._draw_chunk(0)
            self._draw_chunk(c + 1)                   # gathered while chunk c is consumed
        if j == 0:
            if self.world > 1:
                torch.cuda.current_stream(core.device).wait_event(self.ready[c & 1])
        prio = core.update_from_source("gathered")
        self.n += 1
        return prio

    def update_many(self, n: int) -> int:
        """Up to ``n`` updates; returns how many were made.  The updates are ONE graph replay of the
        largest power-of-two run that fits both ``n`` and the rest of the current chunk (a whole chunk
        when aligned), so that a caller who asks for an odd number of updates, or starts in the middle
        of a chunk, still replays a few long graphs instead of one graph per update (a graph launch
        leaves ~8 us of idle GPU).  The graphs (one per run length) are captured on first use."""
        K = self.K
        j, c = self.n % K, self.n // K
        r = min(int(n), K - j)
        if r < 2:
            self.update()
            return 1
        r = 1 << (r.bit_length() - 1)
        if not self._precaptured:
            self._precapture()
        core = self.core
        if j == 0:
            if not (c == 0 and getattr(self, "_primed_by_inspection", False)):
                if c == 0:
                    self._draw_chunk(0)
                self._draw_chunk(c + 1)
            if self.world > 1:
                torch.cuda.current_stream(core.device).wait_event(self.ready[c & 1])
        core.update_from_source("gathered", r)
        self.n += r
        return r

    def _precapture(self) -> None:
        """Capture the graph of every run length now (each capture makes one throw-away update and
        restores the state), so that none of them falls into a timed or latency-sensitive region."""
        self._precaptured = True
        core = self.core
        if not getattr(core, "use_graph", False):
            return
        if core.grad_sync is not None and not core._sync_in_graph():
            return          # host-issued gradient exchange: the update stays cut into per-update graph segments
        if self.n == 0 and not getattr(self, "_primed_by_inspection", False):
            # the throw-away updates read chunk 0: draw it (and chunk 1, as update() would) first
            self._draw_chunk(0)
            self._draw_chunk(1)
            self._primed_by_inspection = True
            if self.world > 1:
                torch.cuda.current_stream(core.device).wait_event(self.ready[0])
        r = 1 << (self.K.bit_length() - 1)
        while r >= 1:
            core.ensure_captured("gathered", r)
            r >>= 1

    def current_batch(self) -> torch.Tensor:
        """The packed global batch ``[world * Bl, W]`` the NEXT ``update()`` will consume (testing /
        inspection; synchronises)."""
        j, c = self.n % self.K, self.n // self.K
        if self.n == 0:
            self._draw_chunk(0)
            self._draw_chunk(1)
            self._primed_by_inspection = True
        torch.cuda.synchronize(self.core.device)
        buf = self.bufs[c & 1]
        if self.world == 1:
            return buf[j].clone()
        return buf[:, j].reshape(self.world * self.Bl, -1).clone()

    def mark_inserted(self) -> None:
        """Replay inserts are enqueued on the main stream, where the draws run: nothing to order."""

    def check_consistency(self) -> float:
        """Largest difference between this replica's parameter checksum and rank 0's (0.0 when the
        replicas are bit-identical; one tiny collective)."""
        c = self.core
        s = torch.stack([c.online.flat.detach().double().sum(), c.target.flat.detach().double().sum(),
                         c.adam_v.detach().double().sum()])
        ref = s.clone()
        self.comm.broadcast_(ref, src=0)
        return float((s - ref).abs().max().item())

    def close(self) -> None:
        torch.cuda.synchronize(self.core.device)


def attach_replicated(core, replay, comm: Comm, local_batch: Optional[int] = None, chunk: int = 8) -> ReplicatedDP:
    """Wire ``core`` (built for the GLOBAL batch) and the local ``replay`` shard for replicated
    data parallelism over ``comm``; batches are drawn and gathered ``chunk`` updates at a time."""
    return ReplicatedDP(core, replay, comm, local_batch if local_batch is not None else core.B // comm.world, chunk)
