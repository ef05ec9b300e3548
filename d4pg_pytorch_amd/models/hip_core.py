"""The MI355X learner update: hand-written HIP kernels replayed as HIP graphs.

One update of D4PG (``models/d4pg/d4pg.py:60-152``) or D3PG/DDPG
(``models/d3pg/d3pg.py:53-112``) is ~23 kernel launches on flat, resident
buffers; nothing touches the host:

====  =======================================================================
 #    kernel (csrc/)
====  =======================================================================
 1    replay_sample      draw B slots (uniform or fan-out-64 sum-tree), gather rows, tick step
 2-4  gemm_group x3      layer 1..3 of {target actor(s'), actor(s), critic(s,a)} in one launch each
 5-7  gemm_group x3      target critic(s', pi'(s'))
 8    critic_loss        softmax + categorical projection + BCE/CE + dlogits + priorities
 8b   tree_update        (PER) priorities back into the tree, on a forked graph branch
 9-11 gemm_group x3      critic backward: {dW, db, dX} of a layer per launch
 12   adam_polyak        critic Adam + critic target Polyak           [grad all-reduce before]
 13-15 gemm_group x3     critic(s, pi(s)) with the UPDATED critic
 16   actor_loss         expected Q, dlogits = -(1/B) p (z - q)
 17-19 gemm_group x3     critic backward, dX only (no wasted critic wgrads); last one is d/da * tanh'
 20-22 gemm_group x3     actor backward
 23   adam_polyak        actor Adam + actor target Polyak, publishes step_done   [all-reduce before]
====  =======================================================================

Single GPU: the whole list is ONE captured graph.  Data parallel: the list is
cut at the two all-reduce points into three graphs, the RCCL all-reduce of the
flat gradient range runs between them.
"""
from __future__ import annotations

import ctypes as C
import gc
from typing import Callable, Dict, List, Optional

import torch

from ..ops import hip as H
from .core import Batch, CoreBase
from .nets import LayerSlot, NetLayout, pad4


class HipCore(CoreBase):
    def __init__(self, cfg: Dict, device="cuda", seed: Optional[int] = None):
        super().__init__(cfg, device, seed)
        assert self.device.type == "cuda", "HipCore needs a GPU"
        H.lib()   # fail loudly if the kernels were not built
        lay = self.layout
        dev = self.device
        self.B = B = int(cfg["batch_size"])
        self.S, self.A, self.Hd = lay.state_dim, lay.action_dim, lay.hidden
        self.NO = lay.critic_out
        self.ldx = pad4(self.S + self.A)
        self.ldh = pad4(self.Hd)
        self.ldn = pad4(self.NO)
        self.lda = pad4(self.A)
        self.inv_global_batch = 1.0 / B      # DP: set_world() rescales
        self.world = 1
        self.use_graph = bool(cfg.get("use_hip_graph", True))
        self.prioritized = bool(cfg["replay_memory_prioritized"])

        def z(*shape, dtype=torch.float32):
            return torch.zeros(*shape, dtype=dtype, device=dev)

        # inputs
        self.sa, self.s_pi, self.s2a2 = z(B, self.ldx), z(B, self.ldx), z(B, self.ldx)
        self.reward, self.done, self.gamma = z(B), z(B), z(B)
        self.weights = torch.ones(B, dtype=torch.float32, device=dev)
        self.idx = z(B, dtype=torch.int64)
        # activations
        names = ["h1_ta", "h2_ta", "h1_a", "h2_a", "h1_c", "h2_c", "h1_tc", "h2_tc", "h1_q", "h2_q",
                 "dh2", "dh1", "dh2q", "dh1q", "dh2a", "dh1a"]
        for n in names:
            setattr(self, n, z(B, self.ldh))
        self.out_c, self.out_t, self.out_q = z(B, self.ldn), z(B, self.ldn), z(B, self.ldn)
        self.dout_c, self.dout_q = z(B, self.ldn), z(B, self.ldn)
        self.da = z(B, self.lda)
        self.per_loss, self.prio, self.q_val = z(B), z(B), z(B)
        self.grad = torch.zeros_like(self.online.flat)
        # device state
        self.dev_state = torch.zeros(C.sizeof(H.DevStateMirror), dtype=torch.uint8, device=dev)
        self._write_dev_state(step=0, seed=int(cfg["random_seed"]))
        self.replay = None
        self._graphs: Dict[str, List] = {}
        self._segments_cache: Dict[str, List[Callable[[], None]]] = {}
        self._side = torch.cuda.Stream(device=dev)
        self.packed_unpack: Dict[str, H.UnpackArgs] = {}     # batch sources registered by register_packed()
        self.packed_direct: Dict[str, H.PackedIn] = {}       # sources the kernels read in place (bf16 pipeline)
        self.packed_prio = None
        self._fork_join = None       # event of a forked priority write-back still to be joined

    # ------------------------------------------------------------------ state
    def _write_dev_state(self, step: int, seed: Optional[int] = None, keep_replay: bool = False) -> None:
        cur = self.read_dev_state() if keep_replay else None
        m = H.DevStateMirror()
        m.step_done = step
        m.step_cur = step
        m.bc1 = 1.0
        m.bc2 = 1.0
        m.per_beta = float(self.cfg["priority_beta_start"])
        m.max_priority = cur.max_priority if cur is not None else 1.0
        m.replay_size = cur.replay_size if cur is not None else 0
        m.replay_head = cur.replay_head if cur is not None else 0
        m.inserted_total = cur.inserted_total if cur is not None else 0
        m.seed = (cur.seed if (cur is not None and seed is None) else (seed or 0)) & 0xFFFFFFFFFFFFFFFF
        host = torch.frombuffer(bytearray(bytes(m)), dtype=torch.uint8)
        self.dev_state.copy_(host)
        torch.cuda.synchronize(self.device)

    def read_dev_state(self) -> H.DevStateMirror:
        raw = bytes(self.dev_state.cpu().numpy().tobytes())
        return H.DevStateMirror.from_buffer_copy(raw)

    def _after_load(self) -> None:
        self._write_dev_state(step=self.step, keep_replay=True)

    def set_world(self, world: int) -> None:
        """Data-parallel degree: gradients are summed over ranks, so the loss
        kernels scale by 1/(world*B) and every rank applies the same update."""
        self.world = int(world)
        self.inv_global_batch = 1.0 / (self.B * self.world)
        self._graphs.clear()
        self._segments_cache.clear()

    def attach_replay(self, replay) -> None:
        """Use an HBM replay shard (``replay/hbm_replay.py``) as the batch source."""
        self.replay = replay
        self._graphs.clear()
        self._segments_cache.clear()

    # ------------------------------------------------------------- descriptors
    def _sl(self, flat: torch.Tensor, off: int, n: int) -> torch.Tensor:
        return flat[off: off + n]

    def _fwd(self, x, ldx, net_flat, slot: LayerSlot, out, ldo, epi, col_off: int = 0):
        W = self._sl(net_flat, slot.w_off, slot.w_numel)
        b = self._sl(net_flat, slot.b_off, slot.out_dim)
        Cm = out if col_off == 0 else out.view(-1)[col_off:]
        return H.gemm_desc(H.GEMM_NT, x, ldx, W, slot.ld, Cm, ldo, self.B, slot.out_dim, slot.in_dim,
                           bias=b, epi=epi)

    def _dw(self, dy, ldy, x, ldx, slot: LayerSlot):
        gW = self._sl(self.grad, slot.w_off, slot.w_numel)
        gb = self._sl(self.grad, slot.b_off, slot.out_dim)
        return H.gemm_desc(H.GEMM_TN, dy, ldy, x, ldx, gW, slot.ld, slot.out_dim, slot.in_dim, self.B,
                           colsum=gb)

    def _dx(self, dy, ldy, net_flat, slot: LayerSlot, out, ldo, n_cols, epi, aux=None, ldaux=0, w_col_off=0):
        W = self._sl(net_flat, slot.w_off, slot.w_numel)
        if w_col_off:
            W = W[w_col_off:]
        return H.gemm_desc(H.GEMM_NN, dy, ldy, W, slot.ld, out, ldo, self.B, n_cols, slot.out_dim,
                           epi=epi, aux=aux, ldaux=ldaux)

    # ---------------------------------------------------------------- segments
    def _build_segments(self, source: str) -> List[Callable[[], None]]:
        """Returns the update as three enqueue-only callables, cut at the
        gradient all-reduce points.  ``source``: 'replay' or 'external'."""
        cfg, lay = self.cfg, self.layout
        a_on, c_on = lay.actor.layers, lay.critic.layers
        P, T = self.online.flat, self.target.flat
        S, A = self.S, self.A
        ldx, ldh, ldn = self.ldx, self.ldh, self.ldn
        R, TH = H.EPI_RELU, H.EPI_TANH
        dist = self.distributional
        fixed_gamma = -1.0 if cfg["use_per_sample_gamma"] else self.fixed_gamma
        st = self.dev_state.data_ptr()
        b1, b2 = self.betas

        def adam_args(net: NetLayout, lr: float, finish: int) -> H.AdamPolyakArgs:
            a = H.AdamPolyakArgs()
            sl = slice(net.base, net.base + net.numel)
            a.p, a.g = P[sl].data_ptr(), self.grad[sl].data_ptr()
            a.m, a.v = self.adam_m[sl].data_ptr(), self.adam_v[sl].data_ptr()
            a.t, a.st, a.n = T[sl].data_ptr(), st, net.numel
            a.lr, a.beta1, a.beta2, a.eps, a.tau = lr, b1, b2, self.eps, self.tau
            a.gscale, a.finish_step = 1.0, finish
            a.metrics = self.metrics.data_ptr() if finish else None
            a.per_loss, a.weights, a.q_val = self.per_loss.data_ptr(), self.weights.data_ptr(), self.q_val.data_ptr()
            a.B, a.inv_B = self.B, 1.0 / self.B
            return a

        def seg_critic() -> None:
            if source == "replay":
                self.replay.enqueue_sample(self, do_tick=True)
            elif source == "external":
                H._check(H.lib().d4pg_tick(st, b1, b2, cfg["priority_beta_start"], cfg["priority_beta_end"],
                                           int(cfg["num_steps_train"]), H.stream_ptr()), "d4pg_tick")
            else:       # a registered packed batch buffer: unpack it (this also ticks)
                H.launch("d4pg_batch_unpack", self.packed_unpack[source])
            # layers 1..3 of target actor(s'), actor(s), critic(s,a)
            H.gemm_group([self._fwd(self.s2a2, ldx, T, a_on[0], self.h1_ta, ldh, R),
                          self._fwd(self.s_pi, ldx, P, a_on[0], self.h1_a, ldh, R),
                          self._fwd(self.sa, ldx, P, c_on[0], self.h1_c, ldh, R)])
            H.gemm_group([self._fwd(self.h1_ta, ldh, T, a_on[1], self.h2_ta, ldh, R),
                          self._fwd(self.h1_a, ldh, P, a_on[1], self.h2_a, ldh, R),
                          self._fwd(self.h1_c, ldh, P, c_on[1], self.h2_c, ldh, R)])
            H.gemm_group([self._fwd(self.h2_ta, ldh, T, a_on[2], self.s2a2, ldx, TH, col_off=S),
                          self._fwd(self.h2_a, ldh, P, a_on[2], self.s_pi, ldx, TH, col_off=S),
                          self._fwd(self.h2_c, ldh, P, c_on[2], self.out_c, ldn, H.EPI_NONE)])
            # target critic(s', a')
            H.gemm_group([self._fwd(self.s2a2, ldx, T, c_on[0], self.h1_tc, ldh, R)])
            H.gemm_group([self._fwd(self.h1_tc, ldh, T, c_on[1], self.h2_tc, ldh, R)])
            H.gemm_group([self._fwd(self.h2_tc, ldh, T, c_on[2], self.out_t, ldn, H.EPI_NONE)])
            # loss
            if dist:
                la = H.D4pgLossArgs()
                la.logits_t, la.logits = self.out_t.data_ptr(), self.out_c.data_ptr()
                la.reward, la.done, la.gamma = self.reward.data_ptr(), self.done.data_ptr(), self.gamma.data_ptr()
                la.weights = self.weights.data_ptr()
                la.dlogits, la.per_loss, la.priority = self.dout_c.data_ptr(), self.per_loss.data_ptr(), self.prio.data_ptr()
                la.proj_out, la.metrics = None, self.metrics.data_ptr()
                la.B, la.N, la.ld = self.B, self.NO, ldn
                la.use_ce = 1 if cfg["critic_loss"] == "ce" else 0
                la.v_min, la.v_max = self.v_min, self.v_max
                la.fixed_gamma, la.inv_B = fixed_gamma, self.inv_global_batch
                H.launch("d4pg_critic_loss", la)
            else:
                la = H.D3pgLossArgs()
                la.q_t, la.q = self.out_t.data_ptr(), self.out_c.data_ptr()
                la.reward, la.done, la.gamma = self.reward.data_ptr(), self.done.data_ptr(), self.gamma.data_ptr()
                la.weights = self.weights.data_ptr()
                la.dq, la.per_loss, la.priority = self.dout_c.data_ptr(), self.per_loss.data_ptr(), self.prio.data_ptr()
                la.metrics, la.B, la.ldq = self.metrics.data_ptr(), self.B, ldn
                la.fixed_gamma = fixed_gamma
                la.min_value, la.max_value = float(cfg["min_value"]), float(cfg["max_value"])
                la.inv_B = self.inv_global_batch
                H.launch("d3pg_critic_loss", la)
            # PER write-back runs on a forked branch, joined at the end of the segment
            forked = None
            packed_prio = self.packed_prio if source in self.packed_unpack else None
            if (source == "replay" and self.replay.prioritized) or packed_prio is not None:
                main = torch.cuda.current_stream()
                ev = torch.cuda.Event()
                ev.record(main)
                self._side.wait_event(ev)
                with torch.cuda.stream(self._side):
                    if packed_prio is not None:
                        H.launch("d4pg_tree_update", packed_prio)      # this rank's rows of the global batch
                    else:
                        self.replay.enqueue_priority_update(self.idx, self.prio, self.B)
                    forked = torch.cuda.Event()
                    forked.record(self._side)
            # critic backward
            H.gemm_group([self._dw(self.dout_c, ldn, self.h2_c, ldh, c_on[2]),
                          self._dx(self.dout_c, ldn, P, c_on[2], self.dh2, ldh, self.Hd,
                                   H.EPI_MUL_RELU_MASK, self.h2_c, ldh)])
            H.gemm_group([self._dw(self.dh2, ldh, self.h1_c, ldh, c_on[1]),
                          self._dx(self.dh2, ldh, P, c_on[1], self.dh1, ldh, self.Hd,
                                   H.EPI_MUL_RELU_MASK, self.h1_c, ldh)])
            H.gemm_group([self._dw(self.dh1, ldh, self.sa, ldx, c_on[0])])
            if forked is not None:
                if self.grad_sync is None:
                    # one graph holds the whole update: the priority write-back only has to be
                    # finished before the NEXT update samples, so it joins at the very end and
                    # overlaps the actor phase
                    self._fork_join = forked
                else:
                    torch.cuda.current_stream().wait_event(forked)

        def seg_actor() -> None:
            H.launch("d4pg_adam_polyak", adam_args(lay.critic, self.lr_critic, 0))
            # critic(s, pi(s)) with the updated critic
            H.gemm_group([self._fwd(self.s_pi, ldx, P, c_on[0], self.h1_q, ldh, R)])
            H.gemm_group([self._fwd(self.h1_q, ldh, P, c_on[1], self.h2_q, ldh, R)])
            H.gemm_group([self._fwd(self.h2_q, ldh, P, c_on[2], self.out_q, ldn, H.EPI_NONE)])
            if dist:
                aa = H.D4pgActorLossArgs()
                aa.logits, aa.dlogits, aa.q_out = self.out_q.data_ptr(), self.dout_q.data_ptr(), self.q_val.data_ptr()
                aa.metrics, aa.B, aa.N, aa.ld = self.metrics.data_ptr(), self.B, self.NO, ldn
                aa.v_min, aa.v_max, aa.inv_B = self.v_min, self.v_max, self.inv_global_batch
                H.launch("d4pg_actor_loss", aa)
            else:
                aa = H.D3pgActorLossArgs()
                aa.q, aa.dq, aa.q_out = self.out_q.data_ptr(), self.dout_q.data_ptr(), self.q_val.data_ptr()
                aa.B, aa.ldq, aa.inv_B = self.B, ldn, self.inv_global_batch
                H.launch("d3pg_actor_loss", aa)
            # dX-only critic backward down to the action columns
            H.gemm_group([self._dx(self.dout_q, ldn, P, c_on[2], self.dh2q, ldh, self.Hd,
                                   H.EPI_MUL_RELU_MASK, self.h2_q, ldh)])
            H.gemm_group([self._dx(self.dh2q, ldh, P, c_on[1], self.dh1q, ldh, self.Hd,
                                   H.EPI_MUL_RELU_MASK, self.h1_q, ldh)])
            pi_cols = self.s_pi.view(-1)[S:]
            H.gemm_group([self._dx(self.dh1q, ldh, P, c_on[0], self.da, self.lda, A,
                                   H.EPI_MUL_TANH_GRAD, pi_cols, ldx, w_col_off=S)])
            # actor backward
            H.gemm_group([self._dw(self.da, self.lda, self.h2_a, ldh, a_on[2]),
                          self._dx(self.da, self.lda, P, a_on[2], self.dh2a, ldh, self.Hd,
                                   H.EPI_MUL_RELU_MASK, self.h2_a, ldh)])
            H.gemm_group([self._dw(self.dh2a, ldh, self.h1_a, ldh, a_on[1]),
                          self._dx(self.dh2a, ldh, P, a_on[1], self.dh1a, ldh, self.Hd,
                                   H.EPI_MUL_RELU_MASK, self.h1_a, ldh)])
            H.gemm_group([self._dw(self.dh1a, ldh, self.s_pi, ldx, a_on[0])])

        def seg_finish() -> None:
            H.launch("d4pg_adam_polyak", adam_args(lay.actor, self.lr_actor, 1))
            self._join_fork()

        return [seg_critic, seg_actor, seg_finish]

    def _join_fork(self) -> None:
        if self._fork_join is not None:
            torch.cuda.current_stream().wait_event(self._fork_join)
            self._fork_join = None

    # ------------------------------------------------------------------- run
    def _segments(self, source: str) -> List[Callable[[], None]]:
        if source not in self._segments_cache:
            self._segments_cache[source] = self._build_segments(source)
        return self._segments_cache[source]

    def _sync_in_graph(self) -> bool:
        """Can the gradient exchange be captured into the update's graph?  True for collectives that are
        enqueued on the stream (RCCL); host-side exchanges (the in-process test fake, gloo staging) and a
        capture that failed once keep the update cut into three graphs with the exchange between them."""
        return bool(getattr(self.grad_sync, "capturable", False)) and not getattr(self, "_dp_capture_failed", False)

    def _run(self, source: str, reps: int = 1) -> None:
        """Enqueue ``reps`` consecutive updates from ``source`` as ONE graph replay (a graph launch costs
        ~8 us of idle GPU between two replays, so a source that holds several batches is replayed several
        updates at a time).  With data-parallel learners the two gradient all-reduces of every update are
        nodes of that same graph when the collective library can be captured (RCCL): nothing returns to
        the host between the critic phase of the first and the optimiser of the last update."""
        segs = self._segments(source)
        sync = self.grad_sync
        lay = self.layout
        if sync is not None and self.use_graph and self._sync_in_graph():
            key = source + "/dp1g" + (f"/x{reps}" if reps > 1 else "")
            if key not in self._graphs:
                try:
                    self._graphs[key] = self._capture(self._with_sync(segs), fused=True, reps=reps)
                except Exception as exc:      # the collective refused capture: fall back for good
                    import warnings
                    warnings.warn(f"gradient all-reduce could not be captured into the update graph ({exc!r}); "
                                  "using three graph segments with host-issued collectives")
                    self._dp_capture_failed = True
            if key in self._graphs:
                self._graphs[key][0].replay()
                return
        if reps > 1 and sync is not None and self.use_graph:
            for _ in range(reps):          # gradient exchange between the graph segments: one update at a time
                self._run(source, 1)
            return
        if not self.use_graph:
            for _ in range(reps):
                segs[0]()
                if sync is not None:
                    sync(self.grad[lay.critic.base: lay.critic.base + lay.critic.numel])
                segs[1]()
                if sync is not None:
                    sync(self.grad[lay.actor.base: lay.actor.base + lay.actor.numel])
                segs[2]()
            return
        key = source + ("/dp" if sync is not None else "") + (f"/x{reps}" if reps > 1 else "")
        if key not in self._graphs:
            self._graphs[key] = self._capture(segs, fused=sync is None, reps=reps)
        graphs = self._graphs[key]
        if sync is None:
            graphs[0].replay()
        else:
            graphs[0].replay()
            sync(self.grad[lay.critic.base: lay.critic.base + lay.critic.numel])
            graphs[1].replay()
            sync(self.grad[lay.actor.base: lay.actor.base + lay.actor.numel])
            graphs[2].replay()

    def _with_sync(self, segs):
        """The three segments of an update with the two gradient exchanges between them, as one list."""
        lay, sync = self.layout, self.grad_sync
        cr = slice(lay.critic.base, lay.critic.base + lay.critic.numel)
        ac = slice(lay.actor.base, lay.actor.base + lay.actor.numel)
        return [segs[0], lambda: sync(self.grad[cr]), segs[1], lambda: sync(self.grad[ac]), segs[2]]

    def ensure_captured(self, source: str, reps: int = 1) -> None:
        """Capture the graph ``_run(source, reps)`` would replay, if it does not exist yet."""
        sync = self.grad_sync
        if not self.use_graph:
            return
        if sync is not None and self._sync_in_graph():
            key = source + "/dp1g" + (f"/x{reps}" if reps > 1 else "")
            if key not in self._graphs:
                try:
                    self._graphs[key] = self._capture(self._with_sync(self._segments(source)), fused=True, reps=reps)
                except Exception:
                    self._dp_capture_failed = True
            return
        if reps > 1 and sync is not None:
            return
        key = source + ("/dp" if sync is not None else "") + (f"/x{reps}" if reps > 1 else "")
        if key not in self._graphs:
            self._graphs[key] = self._capture(self._segments(source), fused=sync is None, reps=reps)

    def _capture(self, segs, fused: bool, reps: int = 1):
        """Warm up once eagerly (module loading, allocator) on a side stream,
        restore the state it touched, then capture."""
        snap = self._snapshot()
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for f in segs:
                f()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize(self.device)
        self._restore(snap)
        groups = [list(segs) * reps] if fused else [[f] for f in segs]
        graphs = []
        # No garbage collection while a stream is capturing: a collected tensor or event is
        # released through hipFree / hipEventDestroy, which is illegal during capture and
        # aborts the process.
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            for grp in groups:
                g = torch.cuda.CUDAGraph()
                # thread_local: other threads (the collective library's watchdog, the ring ingest thread) may
                # make HIP calls while this thread captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for f in grp:
                        f()
                graphs.append(g)
        finally:
            if gc_was_enabled:
                gc.enable()
        # capture does not execute; nothing to restore
        return graphs

    def _snapshot(self):
        snap = {
            "online": self.online.flat.clone(), "target": self.target.flat.clone(),
            "m": self.adam_m.clone(), "v": self.adam_v.clone(),
            "metrics": self.metrics.clone(), "state": self.dev_state.clone(),
        }
        if self.replay is not None:
            snap["replay"] = self.replay.snapshot_priorities()
        return snap

    def _restore(self, snap) -> None:
        with torch.no_grad():
            self.online.flat.copy_(snap["online"])
            self.target.flat.copy_(snap["target"])
            self.adam_m.copy_(snap["m"])
            self.adam_v.copy_(snap["v"])
            self.metrics.copy_(snap["metrics"])
            self.dev_state.copy_(snap["state"])
        if self.replay is not None and "replay" in snap:
            self.replay.restore_priorities(snap["replay"])
        torch.cuda.synchronize(self.device)

    # -------------------------------------------------------------- public API
    def load_batch(self, batch: Batch) -> None:
        """Copy an externally sampled batch into the static input buffers."""
        assert batch.size == self.B, f"batch has {batch.size} rows, learner was built for {self.B}"
        a = H.ScatterBatchArgs()
        st = batch.state.contiguous().float()
        ac = batch.action.contiguous().float()
        ns = batch.next_state.contiguous().float()
        a.state, a.action, a.next_state = st.data_ptr(), ac.data_ptr(), ns.data_ptr()
        a.sa, a.s_pi, a.s2_a2 = self.sa.data_ptr(), self.s_pi.data_ptr(), self.s2a2.data_ptr()
        a.B, a.S, a.A, a.ld_x = self.B, self.S, self.A, self.ldx
        H.launch("d4pg_scatter_batch", a)
        self.reward.copy_(batch.reward.reshape(-1))
        self.done.copy_(batch.done.reshape(-1).float())
        self.gamma.copy_(batch.gamma.reshape(-1))
        if batch.weights is not None:
            self.weights.copy_(batch.weights.reshape(-1))
        else:
            self.weights.fill_(1.0)
        if batch.indices is not None:
            self.idx.copy_(batch.indices.reshape(-1))

    def update(self, batch: Batch) -> torch.Tensor:
        """One update on an externally provided batch; returns the priorities."""
        self.load_batch(batch)
        self._run("external")
        self.step += 1
        return self.prio

    def unpack_args(self, chunk0: torch.Tensor, chunk1: torch.Tensor, row_ld: int, local_batch: int,
                    chunk: int) -> H.UnpackArgs:
        """Launch descriptor that spreads one batch out of the gathered chunk buffers
        (``[world][chunk][local_batch][row_ld + 4]``, ``HbmReplay.enqueue_sample_rows`` rows) over
        the static input buffers and advances the device step.  Which chunk buffer and which
        batch follow from the device step counter, counted from the learner's current step."""
        need = self.B * chunk * (row_ld + H.PACK_EXTRA)
        for t in (chunk0, chunk1):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == need, "chunk buffer size"
        assert self.B % local_batch == 0
        cfg = self.cfg
        a = H.UnpackArgs()
        a.in0, a.in1, a.st = chunk0.data_ptr(), chunk1.data_ptr(), self.dev_state.data_ptr()
        a.sa, a.s_pi, a.s2_a2 = self.sa.data_ptr(), self.s_pi.data_ptr(), self.s2a2.data_ptr()
        a.reward, a.done, a.gamma = self.reward.data_ptr(), self.done.data_ptr(), self.gamma.data_ptr()
        a.weights, a.idx = self.weights.data_ptr(), self.idx.data_ptr()
        a.B, a.Bl, a.K = self.B, int(local_batch), int(chunk)
        a.S, a.A, a.row_ld, a.ld_x = self.S, self.A, int(row_ld), self.ldx
        a.do_tick = 1
        a.step0 = int(self.step)
        a.beta1, a.beta2 = float(cfg["adam_beta1"]), float(cfg["adam_beta2"])
        a.per_b0, a.per_b1 = float(cfg["priority_beta_start"]), float(cfg["priority_beta_end"])
        a.total_steps = int(cfg["num_steps_train"])
        return a

    def register_packed(self, name: str, chunk0: torch.Tensor, chunk1: torch.Tensor, row_ld: int,
                        local_batch: int, chunk: int = 1, prio_args=None) -> None:
        """Make the gathered chunk buffers a batch source called ``name``: the unpack of the batch
        that belongs to the update in flight (and, with ``prio_args``, the priority write-back of
        this rank's rows) become part of the captured update, so ``update_from_source(name)`` is
        a single graph replay."""
        self.packed_unpack[name] = self.unpack_args(chunk0, chunk1, row_ld, local_batch, chunk)
        self.packed_prio = prio_args
        self._graphs = {k: v for k, v in self._graphs.items() if not k.startswith(name)}
        self._segments_cache = {k: v for k, v in self._segments_cache.items() if not k.startswith(name)}

    supports_direct_input = False     # the fused bf16 pipeline reads packed chunk rows in place

    def update_from_source(self, name: str, reps: int = 1) -> torch.Tensor:
        """``reps`` consecutive updates on the registered packed batch source ``name`` (which must
        hold that many batches from the current step on); returns the last priorities."""
        self._run(name, reps)
        self.step += reps
        return self.prio

    def update_from_replay_many(self, n: int) -> int:
        """Up to ``n`` updates from the attached replay in one call; returns how many were made.
        The base implementation makes one."""
        self.update_from_replay()
        return 1

    def update_from_packed(self, packed: torch.Tensor, row_ld: int) -> torch.Tensor:
        """One update on an arbitrary packed batch ``[B, row_ld + 4]`` (the rows are copied into a
        registered one-batch staging buffer first)."""
        if getattr(self, "_packed_stage", None) is None or self._packed_stage.shape != packed.shape:
            self._packed_stage = torch.zeros_like(packed)
            # a one-batch "chunk": the step arithmetic of the unpack always selects batch 0
            self.register_packed("packed", self._packed_stage, self._packed_stage, row_ld, self.B, 1, None)
        self._packed_stage.copy_(packed)
        return self.update_from_source("packed")

    def update_from_replay(self) -> torch.Tensor:
        """One update whose batch is sampled on the device from the attached replay."""
        assert self.replay is not None, "attach_replay() first"
        self._run("replay")
        self.step += 1
        return self.prio
