"""Fused row-local learner update in reference precision (``compute_dtype: fp32``, the default).

The reference computes in fp32 end to end (``models/d4pg/d4pg.py:73-77`` ``.float()``,
``models/d4pg/networks.py:23-25``).  This core keeps that precision — every product is an exact
f32 multiply-add on the matrix cores — and the launch structure of :class:`HipCoreBf16`:

====  ===================================================================
 1    critic_phase32  target actor -> target critic -> critic -> loss -> dX, per tile of 4/8/16 rows
                      (+ the online actor's forward pass in extra workgroups)
 2    wgrad32         dW/db of the three critic layers + Adam + Polyak + packed-copy refresh
 3    actor_phase32   critic(s, pi(s)) -> expected Q -> dX critic -> dX actor
                      (+ the critic's forward pass for the NEXT batch, + the PER tree update)
 4    wgrad32         the same for the actor; publishes the step
====  ===================================================================

``csrc/fused32.hip``: two sets of phase kernels.  ``csrc/mlp32_dev.h``: ``v_mfma_f32_4x4x1_16b_f32`` on tiles of
4 / 8 / 16 rows, one compiled instance per hidden-width class of 32 (64, 100 / 128, 200, 256, 300, 400, 512) — the fast
one while an update is a latency-bound chain.  ``csrc/mlp16_dev.h``: ``v_mfma_f32_16x16x4_f32`` on tiles of 8 / 16 rows,
every width at run time — for batches beyond 2048 rows (in a lean build that keeps two workgroups on every CU:
``phase_occupancy``) and for every other hidden width.
``v_mfma_f32_32x32x2_f32`` in the weight gradients; weights are read from fp32 copies packed in the fragment order of
the kernel set in use (``Shadow32Layout``), which the optimiser inside ``wgrad32`` keeps current.
Everything else (drawn-ahead batches read in place, 32 updates per graph replay, critic forward made
ahead, prioritised draw riding in the last launch, checkpoints, data-parallel wiring) is inherited.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from ..ops import hip as H
from .hip_core import HipCore
from .hip_core_bf16 import LDS_LIMIT, HipCoreBf16, ShapeLimit
from .nets import LayerSlot, Packed32, Shadow32Layout, Shadow32Slot, pad16, pad32, refresh_shadows32

LDS_PAD32 = 4
#: batch size from which the 16x16x4 kernels (``csrc/mlp16_dev.h``, tiles of 16 rows) replace the 4x4x1 ones for the
#: compiled widths (other widths always take them): beyond 2048 rows, where 8-row tiles no longer fit the 256 CUs in one
#: round.  Measured, us per update, 4x4x1 / 16x16x4: Walker2d shape batch 1024 107 / 144, 2048 145 / 158 (128 tiles of
#: 16 rows leave half the chip to the short roles of each launch), 2064 205 / 176, 2304 207 / 177, 2560 209 / 180,
#: 3072 217 / 186, 3584 243 / 212 (193 with two workgroups per CU), 4096 253 / 221 (202); Ant shape 3072 254 / 198,
#: 4096 301 / 238 (216).  With the 4 input chunks of the Ant shape (119 inputs) the 4x4x1 kernels lose at 2048 rows
#: already (1024: 123 / 152, 2048: 175 / 165): WIDE_FROM_BATCH_WIDE_INPUT.
WIDE_FROM_BATCH = 2049
WIDE_FROM_BATCH_WIDE_INPUT = 2048
#: row tiles from which the lean build of the 16x16x4 kernels runs (two workgroups per CU: at most 128 registers per
#: lane and 80 KB of LDS per workgroup, tiles that are never live together sharing a buffer): as soon as a launch holds
#: more workgroups than the chip has CUs.  The second workgroup of a CU multiplies while the first loads rows, reduces
#: K-slices or computes loss rows.  Measured, one / two workgroups per CU, width 400: 128 tiles (Ant) 165.7 / 173.3 us,
#: 129 tiles 175.9 / 175.1, 160 tiles 179.7 / 179.5, 192 tiles 185.9 / 186.1, 200 tiles 207.8 / 188.5, 224 tiles
#: 211.9 / 193.3, 256 tiles 222.9 / 201.8, 512 tiles 416 / 358; width 300: 129 tiles 139.5 / 132.9, 256 tiles
#: 177.5 / 159.8; width 256: 160 tiles 127.2 / 115.7, 256 tiles 160.3 / 134.9; width 128: 160 tiles 103.4 / 91.1,
#: 256 tiles 134.0 / 107.7.
LEAN_FROM_TILES = 129
#: 8-row tiles beyond which a network too wide for four 16-row tiles of LDS (widths from ~530: the one-workgroup build
#: then runs 8-row tiles, half of every MFMA spent on copies) runs the lean build on 16-row tiles, as one workgroup per CU
#: (its two shared tiles fit up to ~1000 columns).  Measured, us per update, 8-row / 16-row tiles, Walker2d shape:
#: width 768: 256 rows 322 / 330, 1024: 349 / 354, 2048: 482 / 390, 4096: 926 / 558; width 1024: 256 rows 499 / 531,
#: 1024: 553 / 582, 2048: 766 / 644, 4096: 1474 / 922.
LEAN_ONLY_FROM_TILES8 = 255


class HipCoreF32(HipCoreBf16):
    _critic_args_cls = H.CriticPhaseArgs32
    _actor_args_cls = H.ActorPhaseArgs32
    _fn_critic, _fn_actor = "d4pg_critic_phase32", "d4pg_actor_phase32"
    rows_per_tile_unit = 4
    supports_per_ahead = True
    wide = False
    lean = False

    def __init__(self, cfg: Dict, device="cuda", seed: Optional[int] = None):
        HipCore.__init__(self, cfg, device, seed)
        lay, dev = self.layout, self.device
        B, S, A, Hd, NO = self.B, self.S, self.A, self.Hd, self.NO
        waves = H.lib().d4pg_phase32_waves()
        ksh = pad32(Hd) // 32
        # 4-row-tile kernels are compiled per hidden-width class of 32 (D4PG_KSH_LIST in csrc/fused32.hip: 64, 100 / 128,
        # 200, 256, 300, 400, 512); one 64-column tile per wave
        narrow_ok = bool(H.lib().d4pg_phase32_has_width(ksh)) and (Hd + 63) // 64 <= waves and pad32(S + A) <= 128
        if NO > 64 or A > 64:
            raise ShapeLimit("the fused fp32 pipeline needs <= 64 atoms and <= 64 actions")
        # Two sets of phase kernels.  4x4x1 MFMA on tiles of 4 / 8 / 16 rows, one compiled instance per hidden width
        # class: the fast one while an update is a latency-bound chain.  16x16x4 MFMA on tiles of 8 / 16 rows, every
        # width at run time: for batches that fill the chip (a 4x4x1 kernel with 8- or 16-row tiles issues as many
        # other instructions as MFMAs and keeps the matrix pipe 44 % busy) and for every other hidden width.
        want = str(cfg.get("phase_tile_rows", "auto"))
        mfma = str(cfg.get("phase_mfma", "auto"))          # auto | 4x4 | 16x16
        if mfma == "auto":
            wide_from = WIDE_FROM_BATCH_WIDE_INPUT if pad32(S + A) >= 128 else WIDE_FROM_BATCH
            self.wide = (not narrow_ok) or (B >= wide_from and B % 16 == 0 and want in ("auto", "16"))
        else:
            self.wide = mfma == "16x16"
        if not self.wide and not narrow_ok:
            raise ShapeLimit(f"dense_size={Hd}: no 4x4x1 phase kernels are compiled for this hidden width "
                             f"(D4PG_KSH_LIST in csrc/fused32.hip); phase_mfma: auto runs the 16x16x4 kernels")
        if B % (8 if self.wide else 4):
            raise ShapeLimit(f"the fused fp32 pipeline needs batch_size to be a multiple of {8 if self.wide else 4}")
        if self.wide:
            self._fn_critic, self._fn_actor = "d4pg_critic_phase32w", "d4pg_actor_phase32w"
        self.sl = Shadow32Layout.build(lay, waves, wide=self.wide)
        self.shadow_online = torch.zeros(self.sl.numel, dtype=torch.float32, device=dev)
        self.shadow_target = torch.zeros(self.sl.numel, dtype=torch.float32, device=dev)
        self.Bp = pad32(B)
        self.zero_bias = torch.zeros(pad16(max(Hd, S + A, NO)), dtype=torch.float32, device=dev)

        def stash(rows):          # fp32, operand-fragment order of the weight-gradient kernel (stash32_off)
            return torch.zeros(pad32(rows), self.Bp, dtype=torch.float32, device=dev)
        self.saT, self.c_h1T, self.c_h2T = stash(S + A), stash(Hd), stash(Hd)
        self.c_dlT, self.c_dh2T, self.c_dh1T = stash(NO), stash(Hd), stash(Hd)
        self.sT, self.a_h1T, self.a_h2T = stash(S), stash(Hd), stash(Hd)
        self.a_daT, self.a_dh2T, self.a_dh1T = stash(A), stash(Hd), stash(Hd)
        self.dims = self._phase_dims()
        d = self.dims
        self.g_xin = torch.zeros(B, d.ldx, dtype=torch.float32, device=dev)
        self.g_h1 = torch.zeros(B, d.ldh, dtype=torch.float32, device=dev)
        self.g_h2 = torch.zeros(B, d.ldh, dtype=torch.float32, device=dev)
        self.g_pi = None          # pi(s) stays in the action columns of g_xin (fp32)
        self.gc_h1 = torch.zeros(B, d.ldh, dtype=torch.float32, device=dev)
        self.gc_h2 = torch.zeros(B, d.ldh, dtype=torch.float32, device=dev)
        self.gc_lg = torch.zeros(B, d.nop, dtype=torch.float32, device=dev)
        self.direct_ahead: Dict[str, int] = {}
        # Rows per workgroup.  A layer costs the time to stream its weights through one CU whatever the tile
        # height, until the MFMA work of the tile (2.3 us per 400 x 400 layer and 4 rows) exceeds it: 4-row
        # tiles while their workgroups (critic + actor forward) fit the 256 CUs, taller tiles beyond.  Measured
        # (us per update, 4 / 8 / 16 rows): batch 512: 86 / 96 / -, 1024: 117 / 105 / -, 2048: 205 / 147 / 186,
        # 4096 (Ant): 451 / 329 / 311  (bench/exp/ab_tile32.sh).  In between, Walker2d shape, 4 / 8 rows: 640: 97.7 / 103.8,
        # 768: 99.3 / 104.4, 896: 112.8 / 106.1 (bench/exp/r3_tile_sweep.sh): 4-row tiles up to 192 of them per role.
        lds_fn = H.lib().d4pg_phase32w_lds_bytes if self.wide else H.lib().d4pg_phase32_lds_bytes

        def lds_need(mt: int) -> int:
            return max(lds_fn(self.dims, mt, 0), lds_fn(self.dims, mt, 1))
        occ = str(cfg.get("phase_occupancy", "auto"))      # the lean build: auto | 1 | 2 (workgroups per CU)
        lean_need, lean_only = 0, False
        if self.wide:
            # 16 rows use every row of the MFMA; 8 rows (half of each instruction spent on copies) where 16 do not
            # divide the batch or four 16-row hidden tiles do not fit the LDS (widths beyond ~470).  The lean build
            # keeps two such tiles, not four: widths up to ~1000 fit on 16-row tiles there, as one workgroup per CU.
            lean_need = max(H.lib().d4pg_phase32wl_lds_bytes(self.dims, 0), H.lib().d4pg_phase32wl_lds_bytes(self.dims, 1))
            rows16 = want != "8" and B % 16 == 0
            lean_only = rows16 and lds_need(4) > LDS_LIMIT and lean_need <= LDS_LIMIT and occ != "1" and \
                (occ == "2" or B // 8 > int(os.environ.get("D4PG_LEAN_ONLY_FROM_TILES8", LEAN_ONLY_FROM_TILES8)))
            self.mt = 4 if rows16 and (lds_need(4) <= LDS_LIMIT or lean_only) else 2
        elif want in ("4", "8", "16"):
            self.mt = int(want) // 4
        else:
            self.mt = 1 if B // 4 <= 192 else (2 if B // 8 <= 256 else 4)
        while not self.wide and self.mt > 1 and B % (4 * self.mt):
            self.mt //= 2
        need = lean_need if lean_only else lds_need(self.mt)
        if need > LDS_LIMIT:
            raise ShapeLimit(f"dense_size={Hd} needs {need} bytes of LDS per workgroup (> {LDS_LIMIT})")
        lean_two = self.wide and self.mt == 4 and lean_need <= H.lib().d4pg_phase32wl_lds_limit()      # two workgroups fit a CU
        if occ == "2" and not (lean_two or lean_only):
            raise ShapeLimit("phase_occupancy: 2 needs the 16x16x4 kernels on 16-row tiles of at most 160 KB of LDS")
        self.lean = lean_only or (lean_two and (occ == "2" or (occ == "auto" and B // 16 >= LEAN_FROM_TILES)))
        if self.lean:
            self._fn_critic, self._fn_actor = "d4pg_critic_phase32wl", "d4pg_actor_phase32wl"
            if lean_two:
                self.ahead_tiles = 256        # the critic's forward pass rides in the actor launch while both fit the chip twice
                self.role_map = int(os.environ.get("D4PG_ROLE_MAP", "1")) if (B // 16) % 256 == 0 else 0
        self.refresh_shadows()

    # ------------------------------------------------------------------ shadows
    def refresh_shadows(self) -> None:
        refresh_shadows32(self.online.flat.detach(), self.shadow_online, self.layout, self.sl)
        refresh_shadows32(self.target.flat.detach(), self.shadow_target, self.layout, self.sl)
        off = H.DevStateMirror.pad_.offset        # a critic forward pass made ahead is stale
        self.dev_state[off:off + 16].zero_()

    def _phase_dims(self) -> H.PhaseDims:
        d = super()._phase_dims()
        sl = self.sl
        if self.wide:      # K padded to groups of 16; + 4 floats: consecutive rows start 4 banks apart (mod 16 rows)
            d.ldx = pad16(self.S + self.A) + LDS_PAD32
            d.ldh = pad16(self.Hd) + LDS_PAD32
            d.ldd = pad16(self.NO) + LDS_PAD32
            d.lda_ = pad16(self.A) + LDS_PAD32
            d.nop = pad16(self.NO)
            return d
        hidden_ks = max(sl.actor[1].f.ks, sl.actor[2].f.ks, sl.critic[2].f.ks, sl.critic[0].b.ks)
        d.ldx = pad32(self.S + self.A) + LDS_PAD32
        d.ldh = 32 * hidden_ks + LDS_PAD32        # K-sliced output layers pad K up to a multiple of the slices
        d.ldd = pad32(self.NO) + LDS_PAD32
        d.lda_ = pad32(self.A) + LDS_PAD32
        d.nop = pad16(self.NO)
        return d

    # ------------------------------------------------------------- descriptors
    def _lw(self, shadow: torch.Tensor, sh: Shadow32Slot, flat: torch.Tensor, slot: LayerSlot, back: bool = False) -> H.LayerW32:
        L = H.LayerW32()
        p: Packed32 = sh.b if back else sh.f
        assert p is not None
        L.W = shadow[p.off:].data_ptr()
        L.bias = self.zero_bias.data_ptr() if back else flat[slot.b_off:].data_ptr()
        L.ks, L.ntiles, L.N, L.ksplit, L.nch = p.ks, p.ntiles, p.N, p.ksplit, p.nch
        L.kg = (p.K + 3) // 4 if p.ksplit == 1 else p.nch * 8       # 16-byte k-groups that hold weights
        return L

    def _wgrad(self, triples, opt=None, sample_next: bool = False, tree_update=None, apply_only: bool = False) -> None:
        """``apply_only``: the gradients are in ``self.grad`` already (an earlier launch without ``opt``, then the
        all-reduce); the tiles apply them (Adam, Polyak, packed copies) instead of computing them."""
        g = H.WgradGroup32()
        for i, (dyT, xT, slot, sh) in enumerate(triples):
            d = g.p[i]
            d.AT, d.BT = dyT.data_ptr(), xT.data_ptr()
            d.C = self.grad[slot.w_off:].data_ptr()
            d.rowsum = self.grad[slot.b_off:].data_ptr()
            d.M, d.N, d.rowsA, d.rowsB, d.ldc = slot.out_dim, slot.in_dim, dyT.shape[0], xT.shape[0], slot.ld
            d.w_off, d.b_off = slot.w_off, slot.b_off
            d.sf_off, d.sf_ks = sh.f.off, sh.f.ks
            d.sb_off = sh.b.off if sh.b is not None else -1
            d.sb_ks = sh.b.ks if sh.b is not None else 0
            d.sb_c0, d.sb_c1 = sh.b_c0, sh.b_c1
        g.count, g.Bp, g.wide = len(triples), self.Bp, int(self.wide)
        g.apply_only = int(apply_only)
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
        H._check(H.lib().d4pg_wgrad32(g, H.stream_ptr()), "d4pg_wgrad32")

    def _shadow_args(self, net, slots, base: int) -> H.ShadowArgs:
        """Packed-copy refresh inside the stand-alone optimiser kernel (gradient-exchange mode)."""
        sa = H.ShadowArgs()
        sa.p_shadow, sa.t_shadow = self.shadow_online.data_ptr(), self.shadow_target.data_ptr()
        for i, (s, sh) in enumerate(zip(net.layers, slots)):
            m = sa.m[i]
            m.w_off = s.w_off - base
            m.f_off = sh.f.off
            m.b_off = sh.b.off if sh.b is not None else -1
            m.out_dim, m.in_dim, m.ld = s.out_dim, s.in_dim, s.ld
            unit = 16 if self.wide else 32
            m.f_ld = sh.f.ks * unit
            m.b_ld = sh.b.ks * unit if sh.b is not None else 0
            m.b_c0, m.b_c1 = sh.b_c0, sh.b_c1
        sa.count = 3
        sa.f32 = 2 if self.wide else 1
        return sa
