"""Batched policy inference on the GPU for exploiter / explorer agents.

The reference's exploiter runs its target policy on ``device: cuda`` one state at a time: an H2D
copy, six kernels and a D2H copy per environment step (``models/d4pg/networks.py:78-81``,
``models/agent.py:75-80``).  Here an agent steps ``E`` environments and evaluates the policy for all
of them with ONE launch (``actor_infer32_kernel`` in ``csrc/fused32.hip``: the three layers on tiles
of 4-16 environments per workgroup, exact f32 MFMA, activations in LDS) between one pinned H2D and one
pinned D2H copy, i.e. one stream synchronisation per step.

Weights live in the packed fp32 layout that kernel streams (``models/nets.py::Packed32``).  They are
rebuilt on the device (``pack_actor32_kernel``) from flat fp32 parameters that arrive

* from the host (``load_flat``: the seq-locked weight slab every agent can read), or
* from the learner's device memory, mapped into this process through HIP IPC (``attach_shared`` /
  ``refresh_shared``: zero host copies, what ``share_memory()`` gives the reference's exploiter,
  ``models/d4pg/engine.py:129-144``).

Hidden widths outside the fused kernel's set fall back to three launches of the per-layer fp32 MFMA
GEMM (``csrc/gemm.hip``).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import os

import torch

from ..models.nets import NetLayout, NetView, Packed32, pad4, pad16, pad32
from ..ops import hip as H

LDS_PAD32 = 4
#: states per launch from which the policy runs on the 16x16x4 kernel, as the learner's phases do (models/hip_core_f32.py):
#: beyond 2048 the 4x4x1 kernel is on 16-row tiles.  Measured, whole rollout steps (policy + noise + environment) of the
#: lunar lander, 4x4x1 / 16x16x4 policy kernel: 2048 environments 27.4 / 29.1 us, 4096: 40.4 / 30.4, 8192: 68.0 / 47.4;
#: Walker2d 4096: 72.7 / 63.1 (bench/exp/r3_infer_wide.py).  D4PG_INFER_WIDE_FROM overrides it for A/B runs.
WIDE_INFER_FROM_ROWS = 2049


def has_narrow_kernels(hidden: int) -> bool:
    """Whether the library holds 4x4x1 kernels compiled for this hidden width (``D4PG_KSH_LIST``, csrc/fused32.hip)."""
    return bool(H.lib().d4pg_phase32_has_width(pad32(hidden) // 32))


class PackedActor32:
    """Packed fp32 copies of the three actor layers + biases in one device buffer, and the launch
    descriptor of the single-launch inference kernel."""

    def __init__(self, state_dim: int, action_dim: int, hidden: int, device, rows: int = 1):
        """``rows``: how many states a launch will usually take (decides between the two kernel sets)."""
        self.S, self.A, self.Hd = state_dim, action_dim, hidden
        self.device = torch.device(device)
        self.layout = NetLayout.build(state_dim, hidden, action_dim)
        waves = H.lib().d4pg_phase32_waves()
        # hidden widths the 4x4x1 kernels are compiled for take them; every other width the 16x16x4 kernel
        # (csrc/mlp16_dev.h: any width at run time, tiles of 8 / 16 environments)
        self.wide = not (has_narrow_kernels(hidden) and pad32(state_dim) <= 128) or \
            (rows >= int(os.environ.get("D4PG_INFER_WIDE_FROM", WIDE_INFER_FROM_ROWS)) and rows % 16 == 0)
        off = 0
        self.packed = []
        for i, s in enumerate(self.layout.layers):
            if self.wide:
                nt, ks, ksplit, nch = Packed32.geometry_wide(s.out_dim, s.in_dim, waves)
            else:
                nt, ks, ksplit, nch = Packed32.geometry(s.out_dim, s.in_dim, i == 2, waves)
            p = Packed32(off, s.out_dim, s.in_dim, nt, ks, ksplit, nch, self.wide)
            off += p.numel
            self.packed.append(p)
        self.bias_off = []
        for s in self.layout.layers:
            self.bias_off.append(off)
            off += pad16(s.out_dim)
        self.buf = torch.zeros(off, dtype=torch.float32, device=self.device)
        if self.wide:
            self.ldx = pad16(state_dim) + LDS_PAD32
            self.ldh = pad16(hidden) + LDS_PAD32
        else:
            self.ldx = pad32(state_dim) + LDS_PAD32
            self.ldh = 32 * max(self.packed[1].ks, self.packed[2].ks) + LDS_PAD32

    @staticmethod
    def supported(state_dim: int, action_dim: int, hidden: int) -> bool:
        """One launch per batch of states: any hidden width whose two activation tiles of 8 rows fit the LDS."""
        return action_dim <= 64 and 8 * 4 * (pad16(state_dim) + 2 * pad16(hidden) + 12) + 7 * 1024 <= 160 * 1024

    def pack_from(self, src_ptr: int, actor_base: int = 0, stream=None) -> None:
        """Rebuild the packed copies from flat actor parameters at device address ``src_ptr`` (floats, the
        ``NetLayout`` of this actor starting ``actor_base`` floats in).  Enqueue only."""
        a = H.PackActorArgs32()
        a.src, a.dst = src_ptr, self.buf.data_ptr()
        for i, (s, p) in enumerate(zip(self.layout.layers, self.packed)):
            m = a.m[i]
            m.src_off, m.bias_off = actor_base + s.w_off, actor_base + s.b_off
            m.dst_off, m.dst_bias = p.off, self.bias_off[i]
            m.out_dim, m.in_dim, m.ld, m.ks = s.out_dim, s.in_dim, s.ld, p.ks
        a.wide = int(self.wide)
        H.launch("d4pg_pack_actor32", a, stream)

    def infer_args(self, obs: torch.Tensor, act: torch.Tensor, n: int, action_scale: float = 1.0) -> H.ActorInferArgs32:
        a = H.ActorInferArgs32()
        for i, p in enumerate(self.packed):
            L = a.l[i]
            L.W = self.buf[p.off:].data_ptr()
            L.bias = self.buf[self.bias_off[i]:].data_ptr()
            L.ks, L.ntiles, L.N, L.ksplit, L.nch = p.ks, p.ntiles, p.N, p.ksplit, p.nch
            L.kg = (p.K + 3) // 4 if p.ksplit == 1 else p.nch * 8
        a.obs, a.act = obs.data_ptr(), act.data_ptr()
        a.E, a.S, a.A, a.H = int(n), self.S, self.A, self.Hd
        a.ld_obs, a.ld_act, a.ldx, a.ldh = obs.shape[1], act.shape[1], self.ldx, self.ldh
        a.action_scale = float(action_scale)
        return a

    @staticmethod
    def tile_rows(n: int) -> int:
        """Rows per workgroup / 4: 4-row tiles while their workgroups fit the chip (see models/hip_core_f32.py)."""
        return 1 if n <= 1024 else (2 if n <= 2048 else 4)

    def infer(self, a: "H.ActorInferArgs32", n: int) -> None:
        """Enqueue the inference launch described by ``a`` (:meth:`infer_args`) for ``n`` rows."""
        if self.wide:
            lds16 = 16 * 4 * (self.ldx + 2 * self.ldh) + 7 * 1024
            mt = 4 if (n > 8 and lds16 <= 160 * 1024) else 2
            H._check(H.lib().d4pg_actor_infer32w(a, mt, H.stream_ptr()), "d4pg_actor_infer32w")
        else:
            H._check(H.lib().d4pg_actor_infer32(a, self.tile_rows(n), H.stream_ptr()), "d4pg_actor_infer32")


class GpuActor:
    def __init__(self, state_dim: int, action_dim: int, hidden: int, max_batch: int = 1,
                 device="cuda", action_scale: float = 1.0):
        H.lib()
        self.S, self.A, self.Hd, self.E = state_dim, action_dim, hidden, int(max_batch)
        self.device = torch.device(device)
        self.layout = NetLayout.build(state_dim, hidden, action_dim)
        self.flat = torch.zeros(self.layout.numel, dtype=torch.float32, device=self.device)
        self.view = NetView(self.flat, self.layout)
        self.action_scale = float(action_scale)
        ldx, ldh, lda = pad4(state_dim), pad4(hidden), pad4(action_dim)
        self.ldx, self.ldh, self.lda = ldx, ldh, lda
        self.x = torch.zeros(self.E, ldx, device=self.device)
        self.out = torch.zeros(self.E, lda, device=self.device)
        self._x_host = torch.zeros(self.E, ldx).pin_memory()
        self._out_host = torch.zeros(self.E, lda).pin_memory()
        self._flat_host = torch.zeros(self.layout.numel).pin_memory()
        self.fused = PackedActor32(state_dim, action_dim, hidden, self.device, rows=self.E) \
            if PackedActor32.supported(state_dim, action_dim, hidden) else None
        if self.fused is None:         # per-layer path: two hidden activation buffers
            self.h1 = torch.zeros(self.E, ldh, device=self.device)
            self.h2 = torch.zeros(self.E, ldh, device=self.device)
        self.launches_per_act = 1 if self.fused is not None else 3
        self._shared = None
        self._shared_half = 0

    # ------------------------------------------------------------------ weights
    def load_flat(self, flat) -> None:
        self._flat_host.copy_(torch.as_tensor(np.asarray(flat, dtype=np.float32)))
        self.flat.copy_(self._flat_host, non_blocking=True)
        if self.fused is not None:
            self.fused.pack_from(self.flat.data_ptr())

    def attach_shared(self, shared, actor_index: int) -> None:
        """Follow the learner's weights in place.  ``shared`` is the learner's ``SharedDeviceBuffer``: two
        slots of ``[online actor | target actor]`` written alternately (``models/learner.py::WeightPublisher``);
        ``actor_index`` 0 / 1 selects the online / target half."""
        assert shared.numel == 4 * self.layout.numel
        self._shared, self._shared_half = shared, int(actor_index)

    def refresh_shared(self, version: int) -> bool:
        """Repack the policy from the slot that holds publication ``version`` (1-based count of completed
        publications).  Returns False without a shared buffer or before the first publication."""
        if self._shared is None or version < 1:
            return False
        n = self.layout.numel
        base = ((version - 1) % 2) * 2 * n + self._shared_half * n
        self.flat.copy_(self._shared.tensor[base: base + n], non_blocking=True)      # for state_dict() / checkpoints
        if self.fused is not None:
            self.fused.pack_from(self._shared.ptr, base)      # straight out of the learner's memory
        torch.cuda.current_stream(self.device).synchronize()
        return True

    def _slot(self, i):
        s = self.layout.layers[i]
        return (self.flat[s.w_off: s.w_off + s.w_numel], self.flat[s.b_off: s.b_off + s.out_dim], s)

    # ---------------------------------------------------------------------- act
    @torch.no_grad()
    def act_device(self, obs: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``obs [n, >= S]`` on the device -> actions ``[n, lda]`` on the device; enqueue only."""
        n = obs.shape[0]
        out = self.out[:n] if out is None else out
        if self.fused is not None:
            self.fused.infer(self.fused.infer_args(obs, out, n, self.action_scale), n)
            return out
        (w1, b1, s1), (w2, b2, s2), (w3, b3, s3) = self._slot(0), self._slot(1), self._slot(2)
        H.gemm_group([H.gemm_desc(H.GEMM_NT, obs, obs.shape[1], w1, s1.ld, self.h1, self.ldh, n, self.Hd, self.S,
                                  bias=b1, epi=H.EPI_RELU)])
        H.gemm_group([H.gemm_desc(H.GEMM_NT, self.h1, self.ldh, w2, s2.ld, self.h2, self.ldh, n, self.Hd, self.Hd,
                                  bias=b2, epi=H.EPI_RELU)])
        H.gemm_group([H.gemm_desc(H.GEMM_NT, self.h2, self.ldh, w3, s3.ld, out, out.shape[1], n, self.A, self.Hd,
                                  bias=b3, epi=H.EPI_TANH)])
        if self.action_scale != 1.0:
            out.mul_(self.action_scale)
        return out

    @torch.no_grad()
    def act(self, states) -> np.ndarray:
        x = np.asarray(states, dtype=np.float32)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        n = x.shape[0]
        assert n <= self.E and x.shape[1] == self.S
        self._x_host[:n, : self.S] = torch.from_numpy(x)
        self.x.copy_(self._x_host, non_blocking=True)
        self.act_device(self.x[:n], self.out[:n])
        self._out_host.copy_(self.out, non_blocking=True)
        torch.cuda.current_stream().synchronize()          # the only synchronisation of a step
        return self._out_host[:n, : self.A].numpy().copy()

    def state_dict(self):
        return self.view.state_dict()

    def load_state_dict(self, sd, strict: bool = True) -> None:
        self.view.load_state_dict(sd, strict)
        if self.fused is not None:
            self.fused.pack_from(self.flat.data_ptr())
