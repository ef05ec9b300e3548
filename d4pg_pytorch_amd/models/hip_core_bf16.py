"""Fused row-local bf16 learner update (``compute_dtype: bf16``), and the launch structure the fp32 twin
(``models/hip_core_f32.py``) shares.

Four launches per update instead of 23 (see ``csrc/fused.hip``), 32 updates per graph replay:

====  ===================================================================
 1    critic_phase    target actor -> target critic -> critic -> loss -> dX, per row tile
                      (extra workgroups: the online actor's forward pass, L2 warmers)
 2    wgrad           dW/db of the three critic layers in one grouped launch; every tile applies Adam, Polyak and
                      the refresh of the packed copies to the weights it has just reduced
                      (extra workgroups: the priority write-back of prioritised replay)
 3    actor_phase     actor -> critic -> expected Q -> dX critic -> dX actor, per row tile
                      (extra workgroups: the online critic's forward pass for the NEXT batch; with prioritised replay
                      they draw that batch first)
 4    wgrad           the same for the three actor layers; publishes the step
====  ===================================================================

Batches are drawn ahead (uniform replay: a chunk of 32 per sampling launch, read in place), so no sampling launch
belongs to an update.  With a gradient exchange (``grad_sync``) launches 2 and 4 only WRITE the gradients, the
all-reduce follows (a node of the same graph with RCCL), and a second launch of the same kernel applies the summed
gradients tile by tile (``apply_only``; ``dp_optimizer: flat`` selects the element-wise kernel of ``csrc/optim.hip``).

Master weights, Adam state, gradients, logits and losses are fp32; only the MFMA
operands (weights and activations) are rounded to bf16.  The public API, state
layout and checkpoints are those of :class:`HipCore`.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, List, Optional

import torch

from ..ops import hip as H
from .hip_core import HipCore
from .nets import LayerSlot, NetLayout, ShadowLayout, ShadowSlot, pad16, pad32, refresh_shadows

LDS_PAD = 8
LDS_LIMIT = 160 * 1024


class ShapeLimit(ValueError):
    """The configuration is outside what a fused pipeline was built for (the caller falls back)."""


class HipCoreBf16(HipCore):
    # what differs between the fused pipelines of the two precisions (models/hip_core_f32.py)
    _critic_args_cls = H.CriticPhaseArgs
    _actor_args_cls = H.ActorPhaseArgs
    _fn_critic, _fn_actor = "d4pg_critic_phase", "d4pg_actor_phase"
    rows_per_tile_unit = 16          # a phase workgroup owns rows_per_tile_unit * mt batch rows
    #: prioritised replay: the actor-phase launch can also draw the next batch (behind its own tree update) and
    #: make the online critic's forward pass on it (``CriticAhead`` mode 3, csrc/fused32.hip)
    supports_per_ahead = False
    ahead_tiles = 128        # row tiles up to which an actor-phase launch leaves room for the critic's forward pass
    role_map = 0             # how the two roles of a phase launch are dealt to its workgroups (csrc/fused32.hip, role_of_block)
    #: gradient-exchange mode: apply the summed gradients with a tile-shaped launch of the weight-gradient kernel
    #: (``apply_only``) instead of the flat element-wise optimiser kernel
    supports_apply_only = True

    def __init__(self, cfg: Dict, device="cuda", seed: Optional[int] = None):
        super().__init__(cfg, device, seed)
        lay, dev = self.layout, self.device
        B, S, A, Hd, NO = self.B, self.S, self.A, self.Hd, self.NO
        if B % 16:
            raise ShapeLimit("the fused bf16 pipeline needs batch_size to be a multiple of 16")
        self.sl = ShadowLayout.build(lay)
        self.shadow_online = torch.zeros(self.sl.numel, dtype=torch.bfloat16, device=dev)
        self.shadow_target = torch.zeros(self.sl.numel, dtype=torch.bfloat16, device=dev)
        self.Bp = pad32(B)              # K of the weight-gradient GEMMs
        self.zero_bias = torch.zeros(pad16(max(Hd, S + A, NO)), dtype=torch.float32, device=dev)   # bias of the dX layers

        def stash(rows):
            return torch.zeros(pad32(rows), self.Bp, dtype=torch.bfloat16, device=dev)   # fragment order, see stash_off
        # critic phase stashes
        self.saT, self.c_h1T, self.c_h2T = stash(S + A), stash(Hd), stash(Hd)
        self.c_dlT, self.c_dh2T, self.c_dh1T = stash(NO), stash(Hd), stash(Hd)
        # actor phase stashes
        self.sT, self.a_h1T, self.a_h2T = stash(S), stash(Hd), stash(Hd)
        self.a_daT, self.a_dh2T, self.a_dh1T = stash(A), stash(Hd), stash(Hd)
        self.dims = self._phase_dims()
        # row tiles of the actor forward pass, handed from the critic phase launch to the actor phase
        ldpi = 16 * ((A + 15) // 16)
        self.g_xin = torch.zeros(B, self.dims.ldx, dtype=torch.bfloat16, device=dev)
        self.g_h1 = torch.zeros(B, self.dims.ldh, dtype=torch.bfloat16, device=dev)
        self.g_h2 = torch.zeros(B, self.dims.ldh, dtype=torch.bfloat16, device=dev)
        self.g_pi = torch.zeros(B, ldpi, dtype=torch.float32, device=dev)
        # row tiles of the online critic's forward pass on the NEXT batch, made alongside the actor phase
        self.gc_h1 = torch.zeros(B, self.dims.ldh, dtype=torch.bfloat16, device=dev)
        self.gc_h2 = torch.zeros(B, self.dims.ldh, dtype=torch.bfloat16, device=dev)
        self.gc_lg = torch.zeros(B, self.dims.nop, dtype=torch.float32, device=dev)
        self.direct_ahead: Dict[str, int] = {}
        # rows per workgroup: a phase is a latency-bound chain whose length does not depend on the
        # number of workgroups until they outnumber the 256 CUs, so 16-row tiles (more, smaller
        # workgroups) win up to B = 4096; beyond that 32-row tiles halve the weight traffic
        self.mt = 1
        want_rows = str(cfg.get("phase_tile_rows", "auto"))
        if B % 32 == 0 and (want_rows == "32" or (want_rows == "auto" and B // 16 > 256)):
            need = max(H.lib().d4pg_phase_lds_bytes(self.dims, 2, 0), H.lib().d4pg_phase_lds_bytes(self.dims, 2, 1))
            if need <= LDS_LIMIT:
                self.mt = 2
        need = max(H.lib().d4pg_phase_lds_bytes(self.dims, self.mt, 0),
                   H.lib().d4pg_phase_lds_bytes(self.dims, self.mt, 1))
        if need > LDS_LIMIT:
            raise ShapeLimit(f"dense_size={Hd} needs {need} bytes of LDS per workgroup (> {LDS_LIMIT}); "
                             f"use compute_dtype: fp32")
        self.refresh_shadows()

    supports_direct_input = True

    def register_direct(self, name: str, chunk0: torch.Tensor, chunk1: torch.Tensor, row_ld: int,
                        local_batch: int, chunk: int = 1, critic_ahead: int = 0) -> None:
        """Like :meth:`register_packed`, but the phase kernels read the rows of the batch in flight
        straight out of the gathered chunk buffers (no unpack kernel, no copy into the input
        matrices; the critic phase performs the tick).  Slots are not extracted, so this is for
        uniform replay; prioritised replay goes through :meth:`register_packed`.

        ``critic_ahead``: let the actor phase launch of every update also make the online critic's
        forward pass for the NEXT batch (``CriticAhead`` in ``csrc/fused.hip``): 2 = always (the
        next chunk buffer is complete whenever an update of this chunk runs), 1 = only within a
        chunk buffer, 0 = never."""
        u = self.unpack_args(chunk0, chunk1, row_ld, local_batch, chunk)      # validates the buffers
        pk = H.PackedIn()
        pk.in0, pk.in1, pk.st, pk.step0 = u.in0, u.in1, u.st, u.step0
        pk.K, pk.Bl, pk.row_ld, pk.enabled, pk.do_tick = u.K, u.Bl, u.row_ld, 1, 1
        pk.beta1, pk.beta2, pk.per_b0, pk.per_b1, pk.total_steps = u.beta1, u.beta2, u.per_b0, u.per_b1, u.total_steps
        self.packed_direct[name] = pk
        # moving the pass to the other launch only pays while that launch leaves compute units idle
        # (batch 4096: 93.9 us per update without, 98.6 us with)
        fits = self.B // (self.rows_per_tile_unit * self.mt) <= self.ahead_tiles
        self.direct_ahead[name] = int(critic_ahead) if fits and int(os.environ.get("D4PG_CRITIC_AHEAD", "1")) else 0
        self.packed_unpack.pop(name, None)
        self._graphs = {k: v for k, v in self._graphs.items() if not k.startswith(name)}
        self._segments_cache = {k: v for k, v in self._segments_cache.items() if not k.startswith(name)}

    # --------------------------------------------------------------- draw ahead
    def attach_replay(self, replay) -> None:
        super().attach_replay(replay)
        self._ahead = None
        self._presampled_step = -1
        self._smp_dev = None
        if replay is not None and replay.prioritized:
            # descriptor of "sample the next batch into the input buffers", in device memory for the
            # sampling role of the finishing weight-gradient launch (WgradGroup::smp)
            raw = bytes(replay.sample_args(self, do_tick=False))
            self._smp_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self._graphs = {k: v for k, v in self._graphs.items() if not k.startswith("replay")}
        self._segments_cache = {k: v for k, v in self._segments_cache.items() if not k.startswith("replay")}

    def _per_sampling_rides_along(self) -> bool:
        """Prioritised replay without gradient exchange: every update draws the batch of the NEXT one in
        extra workgroups of its last launch (after its priorities are written back), instead of a
        sampling kernel at the start of every update."""
        return (self.replay is not None and self.replay.prioritized and self.grad_sync is None
                and getattr(self, "_smp_dev", None) is not None and bool(int(os.environ.get("D4PG_PER_SAMPLE_ROLE", "1"))))

    def _per_ahead(self) -> bool:
        """Prioritised replay, no gradient exchange, a grid that leaves compute units idle: the update's priorities
        go into the tree in extra workgroups of the critic's weight-gradient launch, and the workgroups of the
        actor-phase launch that make the online critic's forward pass for the next update draw its rows themselves
        (instead of a tree update beside the actor phase, a sampling role in the update's last launch and a
        critic phase that makes its own forward pass).  ``D4PG_PER_AHEAD=0`` keeps the older scheme."""
        return (self.supports_per_ahead and self._per_sampling_rides_along()
                and self.B // (self.rows_per_tile_unit * self.mt) <= self.ahead_tiles
                and bool(int(os.environ.get("D4PG_PER_AHEAD", "1"))) and bool(int(os.environ.get("D4PG_CRITIC_AHEAD", "1"))))

    def _write_dev_state(self, *args, **kw) -> None:
        super()._write_dev_state(*args, **kw)
        self._presampled_step = -1          # step counter / RNG stream changed: the batch in the input buffers is void

    def update_from_replay(self) -> torch.Tensor:
        """With uniform replay the batches of the next ``sample_ahead`` updates are drawn by ONE
        kernel launch (same RNG stream positions, hence the same rows, as drawing them one
        update at a time) and the phase kernels read them in place: no per-update sampling
        kernel.  Prioritised replay keeps the draw inside every update, so that it sees the
        priorities the previous update wrote."""
        ahead = getattr(self, "_ahead", None)
        # (also with a gradient exchange between the graph segments: the draws are local to the rank)
        if ahead is None and self.replay is not None and not self.replay.prioritized \
                and int(self.cfg.get("sample_ahead", 8)) > 1:
            from ..parallel.comm import LocalComm
            from ..parallel.replicated import ReplicatedDP
            ahead = self._ahead = ReplicatedDP(self, self.replay, LocalComm(), self.B,
                                               int(self.cfg.get("sample_ahead", 8)), reseed=False)
        if ahead is not None:
            return ahead.update()
        if self._per_sampling_rides_along():
            self._per_run(1)
            return self.prio
        return super().update_from_replay()

    def _per_run(self, reps: int) -> None:
        """``reps`` prioritised updates as one graph replay.  Every update draws the batch of the next one in its
        last launch, behind its own priority write-back, so a chain of updates needs nothing from the host."""
        key = "replay" + (f"/x{reps}" if reps > 1 else "")
        if self.use_graph and key not in self._graphs:
            # the warm-up run of the capture works on (and overwrites) the input buffers
            self.replay.enqueue_sample(self, do_tick=False)
            self._graphs[key] = self._capture(self._segments("replay"), fused=True, reps=reps)
            self._presampled_step = -1
        if self._presampled_step != self.step:        # first update, or the state was reset / reloaded
            self.replay.enqueue_sample(self, do_tick=False)
        self._run("replay", reps)
        self.step += reps
        self._presampled_step = self.step

    def update_from_replay_many(self, n: int) -> int:
        """Up to ``n`` updates in one call: a whole drawn-ahead chunk of batches is ONE graph replay."""
        ahead = getattr(self, "_ahead", None)
        if ahead is None and self._per_sampling_rides_along():
            chunk = max(1, int(self.cfg.get("sample_ahead", 32)))
            reps = chunk if (n >= chunk and self.use_graph) else 1
            self._per_run(reps)
            return reps
        if ahead is None:
            self.update_from_replay()      # creates the draw-ahead driver when it applies
            return 1
        return ahead.update_many(n)

    # ------------------------------------------------------------------ shadows
    def refresh_shadows(self) -> None:
        refresh_shadows(self.online.flat.detach(), self.shadow_online, self.layout, self.sl)
        refresh_shadows(self.target.flat.detach(), self.shadow_target, self.layout, self.sl)
        # the weights were changed from the host: a critic forward pass made ahead is stale
        off = H.DevStateMirror.pad_.offset
        self.dev_state[off:off + 16].zero_()

    def _after_load(self) -> None:
        super()._after_load()
        self.refresh_shadows()

    def set_world(self, world: int) -> None:
        super().set_world(world)
        self.dims = self._phase_dims()
        self.refresh_shadows()          # parameters were just broadcast

    def _restore(self, snap) -> None:
        super()._restore(snap)
        self._presampled_step = -1
        # the run that is being undone may have overwritten the tiles of a critic forward pass made
        # ahead, while the restored device state still carries the tag of the old ones
        off = H.DevStateMirror.pad_.offset
        self.dev_state[off:off + 16].zero_()
        self.refresh_shadows()

    def _phase_dims(self) -> H.PhaseDims:
        cfg = self.cfg
        d = H.PhaseDims()
        d.B, d.Bp, d.S, d.A, d.H, d.NO = self.B, self.Bp, self.S, self.A, self.Hd, self.NO
        d.ldx_in = self.ldx
        d.ldx = pad32(self.S + self.A) + LDS_PAD
        d.ldh = pad32(self.Hd) + LDS_PAD
        d.ldd = pad32(self.NO) + LDS_PAD
        d.lda_ = pad32(self.A) + LDS_PAD
        d.nop = pad16(self.NO)
        d.distributional = 1 if self.distributional else 0
        d.use_ce = 1 if cfg["critic_loss"] == "ce" else 0
        d.v_min, d.v_max = self.v_min, self.v_max
        d.fixed_gamma = -1.0 if cfg["use_per_sample_gamma"] else self.fixed_gamma
        d.inv_B = self.inv_global_batch
        d.min_value = max(float(cfg["min_value"]), -3.0e38)
        d.max_value = min(float(cfg["max_value"]), 3.0e38)
        return d

    # ------------------------------------------------------------- descriptors
    def _lw(self, shadow: torch.Tensor, sh: ShadowSlot, flat: torch.Tensor, slot: LayerSlot, back: bool = False) -> H.LayerW:
        L = H.LayerW()
        if not back:
            L.W = shadow[sh.f_off:].data_ptr()
            L.bias = flat[slot.b_off:].data_ptr()
            L.ks, L.ntiles, L.N = sh.f_ld // 32, sh.f_rows // 16, slot.out_dim
        else:
            assert sh.b_off >= 0
            L.W = shadow[sh.b_off:].data_ptr()
            L.bias = self.zero_bias.data_ptr()
            L.ks, L.ntiles, L.N = sh.b_ld // 32, sh.b_rows // 16, sh.b_c1 - sh.b_c0
        return L

    def _wgrad(self, triples, opt=None, sample_next: bool = False, tree_update=None, apply_only: bool = False) -> None:
        """dW/db of up to three layers in one launch.  ``opt=(lr, finish)`` additionally applies
        Adam + Polyak + the shadow refresh inside the same kernel.  ``apply_only`` (gradient-exchange mode): the
        gradients are in ``self.grad`` already (a launch without ``opt``, then the all-reduce); the tiles apply them."""

        g = H.WgradGroup()
        for i, (dyT, xT, slot, sh) in enumerate(triples):
            d = g.p[i]
            d.AT, d.BT = dyT.data_ptr(), xT.data_ptr()
            d.C = self.grad[slot.w_off:].data_ptr()
            d.rowsum = self.grad[slot.b_off:].data_ptr()
            d.M, d.N, d.rowsA, d.rowsB, d.ldc = slot.out_dim, slot.in_dim, dyT.shape[0], xT.shape[0], slot.ld
            d.w_off, d.b_off = slot.w_off, slot.b_off
            d.sf_off, d.sb_off = sh.f_off, sh.b_off
            d.sf_ld, d.sb_ld, d.sb_c0, d.sb_c1 = sh.f_ld, sh.b_ld, sh.b_c0, sh.b_c1
        g.count, g.Bp, g.apply_only = len(triples), self.Bp, int(apply_only)
        if opt is not None:
            lr, finish = opt
            o = g.opt
            o.p, o.t = self.online.flat.data_ptr(), self.target.flat.data_ptr()
            o.m, o.v = self.adam_m.data_ptr(), self.adam_v.data_ptr()
            o.p_shadow, o.t_shadow = self.shadow_online.data_ptr(), self.shadow_target.data_ptr()
            o.st = self.dev_state.data_ptr()
            o.metrics = self.metrics.data_ptr() if finish else None
            # the critic phase stores the importance-weighted per-row loss: the finishing role must not read
            # `weights`, which the sampling role of the same launch is refilling for the next update
            o.per_loss, o.weights, o.q_val = self.per_loss.data_ptr(), None, self.q_val.data_ptr()
            o.B, o.inv_B = self.B, 1.0 / self.B
            o.lr, o.beta1, o.beta2, o.eps, o.tau = lr, self.betas[0], self.betas[1], self.eps, self.tau
            o.finish_step = 1 if finish else 0
        if sample_next:
            g.smp, g.smp_rows = self._smp_dev.data_ptr(), self.B
        if tree_update is not None:        # (arguments in device memory, workgroups): the priority write-back rides along
            g.tree, g.tree_parts = tree_update[0].data_ptr(), tree_update[1]
        H._check(H.lib().d4pg_wgrad(g, H.stream_ptr()), "d4pg_wgrad")

    def _segments(self, source: str):
        # without data parallelism the optimiser is fused into the weight-gradient kernel
        key = f"{source}/{'fusedopt' if self.grad_sync is None else 'dp'}"
        if key not in self._segments_cache:
            self._segments_cache[key] = self._build_segments(source)
        return self._segments_cache[key]

    def _shadow_args(self, net: NetLayout, slots, base: int) -> H.ShadowArgs:
        sa = H.ShadowArgs()
        sa.p_shadow, sa.t_shadow = self.shadow_online.data_ptr(), self.shadow_target.data_ptr()
        for i, (s, sh) in enumerate(zip(net.layers, slots)):
            m = sa.m[i]
            m.w_off = s.w_off - base
            m.f_off, m.b_off = sh.f_off, sh.b_off
            m.out_dim, m.in_dim, m.ld = s.out_dim, s.in_dim, s.ld
            m.f_ld, m.b_ld, m.b_c0, m.b_c1 = sh.f_ld, sh.b_ld, sh.b_c0, sh.b_c1
        sa.count = 3
        return sa

    # ---------------------------------------------------------------- segments
    def _build_segments(self, source: str) -> List[Callable[[], None]]:
        cfg, lay, sl = self.cfg, self.layout, self.sl
        P, T = self.online.flat, self.target.flat
        SP, ST = self.shadow_online, self.shadow_target
        al, cl = lay.actor.layers, lay.critic.layers
        st = self.dev_state.data_ptr()
        b1, b2 = self.betas
        dims = self.dims
        fuse_opt = self.grad_sync is None
        apply_tiles = self.supports_apply_only and str(cfg.get("dp_optimizer", "tiles")) != "flat"

        def adam_args(net: NetLayout, slots, lr: float, finish: int) -> H.AdamPolyakArgs:
            a = H.AdamPolyakArgs()
            r = slice(net.base, net.base + net.numel)
            a.p, a.g = P[r].data_ptr(), self.grad[r].data_ptr()
            a.m, a.v = self.adam_m[r].data_ptr(), self.adam_v[r].data_ptr()
            a.t, a.st, a.n = T[r].data_ptr(), st, net.numel
            a.lr, a.beta1, a.beta2, a.eps, a.tau = lr, b1, b2, self.eps, self.tau
            a.gscale, a.finish_step = 1.0, finish
            a.metrics = self.metrics.data_ptr() if finish else None
            a.per_loss, a.weights, a.q_val = self.per_loss.data_ptr(), None, self.q_val.data_ptr()   # per_loss is weighted
            a.B, a.inv_B = self.B, 1.0 / self.B
            a.sh = self._shadow_args(net, slots, net.base)
            return a

        per_role = source == "replay" and fuse_opt and self._per_sampling_rides_along()
        # Prioritised replay with the fused optimiser: where the update's priorities go into the tree.  Beside the
        # actor phase (extra workgroups of that launch) the write-back is free as long as the phase's grid leaves
        # compute units idle, and it must be over before the next batch is drawn in the update's last launch.  In
        # extra workgroups of the critic's weight-gradient launch it lengthens that launch (a 256-item write-back
        # is 12.7 us, the launch 6-8 us) but the tree is complete when the actor phase starts, which lets that
        # launch draw the next batch and make the critic's forward pass on it (`_per_ahead`), and it does not
        # queue behind a phase grid that already fills the chip.  Measured (us per update, in the weight-gradient
        # launch / beside the actor phase): fp32 batch 256: 84.7 / 90.1, 2048: 163 / 170, Ant 4096: 321 / 337;
        # bf16 batch 256: 67.6 / 61.6, 2048: 90.3 / 78.8, Ant 4096: 130 / 134.
        want = os.environ.get("D4PG_TREE_IN_WGRAD")
        grid_full = self.B // (self.rows_per_tile_unit * self.mt) >= 256
        in_wgrad = per_role and (bool(int(want)) if want is not None else (self._per_ahead() or grid_full))
        per_ahead = in_wgrad and self._per_ahead()
        tree_dev = None
        if in_wgrad:
            targs = self.replay.priority_update_args(self.idx, self.prio, self.B)
            tree_dev = (torch.frombuffer(bytearray(bytes(targs)), dtype=torch.uint8).to(self.device), int(targs.parts))
            self._tree_dev = tree_dev      # (kept alive with the segments)

        def seg_critic() -> None:
            if source == "replay" and not per_role:
                self.replay.enqueue_sample(self, do_tick=True)
            elif source == "external":
                H._check(H.lib().d4pg_tick(st, b1, b2, cfg["priority_beta_start"], cfg["priority_beta_end"],
                                           int(cfg["num_steps_train"]), H.stream_ptr()), "d4pg_tick")
            elif source in self.packed_unpack:      # a registered packed batch buffer: unpack it (this also ticks)
                H.launch("d4pg_batch_unpack", self.packed_unpack[source])
            a = self._critic_args_cls()
            a.d = dims
            if source in self.packed_direct:        # rows are read in place; the kernel ticks
                a.pk = self.packed_direct[source]
            elif per_role:                          # the batch was drawn by the previous update: only the tick is left
                pk = a.pk
                pk.enabled, pk.do_tick, pk.st = 0, 1, st
                pk.beta1, pk.beta2 = b1, b2
                pk.per_b0, pk.per_b1 = float(cfg["priority_beta_start"]), float(cfg["priority_beta_end"])
                pk.total_steps = int(cfg["num_steps_train"])
            a.sa, a.s2 = self.sa.data_ptr(), self.s2a2.data_ptr()
            a.reward, a.done, a.gamma = self.reward.data_ptr(), self.done.data_ptr(), self.gamma.data_ptr()
            a.weights = self.weights.data_ptr()
            for i in range(3):
                a.ta[i] = self._lw(ST, sl.actor[i], T, al[i])
                a.tc[i] = self._lw(ST, sl.critic[i], T, cl[i])
                a.cf[i] = self._lw(SP, sl.critic[i], P, cl[i])
            a.cb3 = self._lw(SP, sl.critic[2], P, cl[2], back=True)
            a.cb2 = self._lw(SP, sl.critic[1], P, cl[1], back=True)
            a.saT, a.h1T, a.h2T = self.saT.data_ptr(), self.c_h1T.data_ptr(), self.c_h2T.data_ptr()
            a.dlT, a.dh2T, a.dh1T = self.c_dlT.data_ptr(), self.c_dh2T.data_ptr(), self.c_dh1T.data_ptr()
            a.per_loss, a.priority = self.per_loss.data_ptr(), self.prio.data_ptr()
            # the online actor's forward pass rides along in extra workgroups of this launch
            f = a.fwd
            for i in range(3):
                f.af[i] = self._lw(SP, sl.actor[i], P, al[i])
            f.s = self.s_pi.data_ptr()
            f.sT, f.h1T, f.h2T = self.sT.data_ptr(), self.a_h1T.data_ptr(), self.a_h2T.data_ptr()
            f.g_xin, f.g_h1, f.g_h2 = self.g_xin.data_ptr(), self.g_h1.data_ptr(), self.g_h2.data_ptr()
            if self.g_pi is not None:
                f.g_pi = self.g_pi.data_ptr()
            ah = a.ahead
            ah.mode = 3 if per_ahead else self.direct_ahead.get(source, 0)
            ah.role_map = self.role_map
            ah.h1, ah.h2, ah.lg = self.gc_h1.data_ptr(), self.gc_h2.data_ptr(), self.gc_lg.data_ptr()
            H._check(getattr(H.lib(), self._fn_critic)(a, self.mt, H.stream_ptr()), self._fn_critic)
            # the priority write-back rides along with the actor phase (seg_actor)
            self._wgrad(critic_triples,
                        opt=(self.lr_critic, False) if fuse_opt else None, tree_update=tree_dev)

        critic_triples = [(self.c_dlT, self.c_h2T, cl[2], sl.critic[2]), (self.c_dh2T, self.c_h1T, cl[1], sl.critic[1]),
                          (self.c_dh1T, self.saT, cl[0], sl.critic[0])]
        actor_triples = [(self.a_daT, self.a_h2T, al[2], sl.actor[2]), (self.a_dh2T, self.a_h1T, al[1], sl.actor[1]),
                         (self.a_dh1T, self.sT, al[0], sl.actor[0])]

        def seg_actor() -> None:
            if not fuse_opt:       # the critic's gradients are summed over the ranks by now: apply them
                if apply_tiles:
                    self._wgrad(critic_triples, opt=(self.lr_critic, False), apply_only=True)
                else:
                    H.launch("d4pg_adam_polyak", adam_args(lay.critic, sl.critic, self.lr_critic, 0))
            a = self._actor_args_cls()
            a.d = dims
            a.g_xin, a.g_h1, a.g_h2 = self.g_xin.data_ptr(), self.g_h1.data_ptr(), self.g_h2.data_ptr()
            if self.g_pi is not None:
                a.g_pi = self.g_pi.data_ptr()
            for i in range(3):
                a.cf[i] = self._lw(SP, sl.critic[i], P, cl[i])
            a.cb3 = self._lw(SP, sl.critic[2], P, cl[2], back=True)
            a.cb2 = self._lw(SP, sl.critic[1], P, cl[1], back=True)
            a.cb1 = self._lw(SP, sl.critic[0], P, cl[0], back=True)
            a.ab3 = self._lw(SP, sl.actor[2], P, al[2], back=True)
            a.ab2 = self._lw(SP, sl.actor[1], P, al[1], back=True)
            a.daT, a.dh2T, a.dh1T = self.a_daT.data_ptr(), self.a_dh2T.data_ptr(), self.a_dh1T.data_ptr()
            a.q_val = self.q_val.data_ptr()
            if source == "replay" and self.replay.prioritized and tree_dev is None:
                a.tree = self.replay.priority_update_args(self.idx, self.prio, self.B)
            elif source in self.packed_unpack and self.packed_prio is not None:
                a.tree = self.packed_prio      # this rank's rows of the global batch
            if self.direct_ahead.get(source, 0):      # the critic forward pass of the next update rides along
                a.pk = self.packed_direct[source]
                a.c_saT, a.c_h1T, a.c_h2T = self.saT.data_ptr(), self.c_h1T.data_ptr(), self.c_h2T.data_ptr()
                ah = a.ahead
                ah.mode = self.direct_ahead[source]
                ah.h1, ah.h2, ah.lg = self.gc_h1.data_ptr(), self.gc_h2.data_ptr(), self.gc_lg.data_ptr()
            elif per_ahead:      # ... and with prioritised replay its workgroups draw the next batch first
                a.pk.enabled, a.pk.st = 0, st
                a.smp = self._smp_dev.data_ptr()
                a.c_saT, a.c_h1T, a.c_h2T = self.saT.data_ptr(), self.c_h1T.data_ptr(), self.c_h2T.data_ptr()
                ah = a.ahead
                ah.mode = 3
                ah.h1, ah.h2, ah.lg = self.gc_h1.data_ptr(), self.gc_h2.data_ptr(), self.gc_lg.data_ptr()
            a.ahead.role_map = self.role_map if (a.ahead.mode != 0 and a.tree.n == 0) else 0
            H._check(getattr(H.lib(), self._fn_actor)(a, self.mt, H.stream_ptr()), self._fn_actor)
            self._wgrad(actor_triples,
                        opt=(self.lr_actor, True) if fuse_opt else None, sample_next=per_role and not per_ahead)

        def seg_finish() -> None:
            if not fuse_opt:
                if apply_tiles:
                    self._wgrad(actor_triples, opt=(self.lr_actor, True), apply_only=True)
                else:
                    H.launch("d4pg_adam_polyak", adam_args(lay.actor, sl.actor, self.lr_actor, 1))
            self._join_fork()

        return [seg_critic, seg_actor, seg_finish]
