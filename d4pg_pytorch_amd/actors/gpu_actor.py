"""Batched policy inference on the GPU for exploiter agents.

The reference's exploiter runs its target policy on ``device: cuda`` one state
at a time: an H2D copy, six kernels and a D2H copy per environment step
(``models/d4pg/networks.py:78-81``, ``models/agent.py:75-80``).  Here an agent
steps ``E`` environments and evaluates the policy for all of them with three
launches of the grouped MFMA GEMM kernel (``csrc/gemm.hip``, inference use: no
activation stash beyond the two hidden buffers), through pinned staging buffers.
"""
from __future__ import annotations

import numpy as np
import torch

from ..models.nets import NetLayout, NetView, pad4
from ..ops import hip as H


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
        self.h1 = torch.zeros(self.E, ldh, device=self.device)
        self.h2 = torch.zeros(self.E, ldh, device=self.device)
        self.out = torch.zeros(self.E, lda, device=self.device)
        self._x_host = torch.zeros(self.E, ldx).pin_memory()
        self._out_host = torch.zeros(self.E, lda).pin_memory()
        self._flat_host = torch.zeros(self.layout.numel).pin_memory()

    def load_flat(self, flat) -> None:
        self._flat_host.copy_(torch.as_tensor(np.asarray(flat, dtype=np.float32)))
        self.flat.copy_(self._flat_host, non_blocking=True)

    def _slot(self, i):
        s = self.layout.layers[i]
        return (self.flat[s.w_off: s.w_off + s.w_numel], self.flat[s.b_off: s.b_off + s.out_dim], s)

    @torch.no_grad()
    def act(self, states) -> np.ndarray:
        x = np.asarray(states, dtype=np.float32)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        n = x.shape[0]
        assert n <= self.E and x.shape[1] == self.S
        self._x_host[:n, : self.S] = torch.from_numpy(x)
        self.x.copy_(self._x_host, non_blocking=True)
        (w1, b1, s1), (w2, b2, s2), (w3, b3, s3) = self._slot(0), self._slot(1), self._slot(2)
        H.gemm_group([H.gemm_desc(H.GEMM_NT, self.x, self.ldx, w1, s1.ld, self.h1, self.ldh, n, self.Hd, self.S,
                                  bias=b1, epi=H.EPI_RELU)])
        H.gemm_group([H.gemm_desc(H.GEMM_NT, self.h1, self.ldh, w2, s2.ld, self.h2, self.ldh, n, self.Hd, self.Hd,
                                  bias=b2, epi=H.EPI_RELU)])
        H.gemm_group([H.gemm_desc(H.GEMM_NT, self.h2, self.ldh, w3, s3.ld, self.out, self.lda, n, self.A, self.Hd,
                                  bias=b3, epi=H.EPI_TANH)])
        self._out_host.copy_(self.out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        a = self._out_host[:n, : self.A].numpy().copy()
        return a * self.action_scale if self.action_scale != 1.0 else a

    def state_dict(self):
        return self.view.state_dict()

    def load_state_dict(self, sd, strict: bool = True) -> None:
        self.view.load_state_dict(sd, strict)
