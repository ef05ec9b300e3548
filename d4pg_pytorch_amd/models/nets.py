"""Actor / critic parameter storage.

The networks are the reference's 3-layer MLPs (``models/d4pg/networks.py:6-81``,
``models/d3pg/networks.py:6-74``):

* actor  ``S -> H -> H -> A`` with ReLU, ReLU, tanh
* critic ``S+A -> H -> H -> N`` (D4PG, logits over N atoms) or ``-> 1`` (D3PG/DDPG)

but they are not ``nn.Module`` objects.  All parameters of one learner live in
ONE flat fp32 buffer (``ParamPack.flat``) so that the optimiser, the Polyak
update, the gradient all-reduce and the weight publication each touch a single
contiguous range.  Every matrix is stored row-major ``[out, ld]`` with the
leading dimension padded to a multiple of 4 floats so the HIP kernels can use
16-byte loads; pad columns are zero and stay zero (their gradient is zero).

``state_dict`` import/export uses the reference key layout
(``linear{1,2,3}.{weight,bias}``) with unpadded, contiguous tensors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


@dataclass(frozen=True)
class LayerSlot:
    """Location of one linear layer inside a flat buffer (units: floats)."""
    w_off: int
    b_off: int
    out_dim: int
    in_dim: int
    ld: int            # padded leading dimension of the weight matrix

    @property
    def w_numel(self) -> int:
        return self.out_dim * self.ld

    @property
    def b_numel(self) -> int:
        return pad4(self.out_dim)


@dataclass(frozen=True)
class NetLayout:
    """Three linear layers packed back to back, 16-byte aligned."""
    in_dim: int
    hidden: int
    out_dim: int
    base: int
    layers: Tuple[LayerSlot, LayerSlot, LayerSlot]
    numel: int

    @staticmethod
    def build(in_dim: int, hidden: int, out_dim: int, base: int = 0) -> "NetLayout":
        assert base % 4 == 0
        dims = [(hidden, in_dim), (hidden, hidden), (out_dim, hidden)]
        off = base
        slots: List[LayerSlot] = []
        for out_d, in_d in dims:
            ld = pad4(in_d)
            w_off = off
            off += out_d * ld
            off = pad4(off)
            b_off = off
            off += pad4(out_d)
            slots.append(LayerSlot(w_off, b_off, out_d, in_d, ld))
        return NetLayout(in_dim, hidden, out_dim, base, tuple(slots), off - base)


@dataclass(frozen=True)
class LearnerLayout:
    """``[actor | critic]`` in one flat buffer."""
    state_dim: int
    action_dim: int
    hidden: int
    critic_out: int
    actor: NetLayout
    critic: NetLayout
    numel: int

    @staticmethod
    def build(state_dim: int, action_dim: int, hidden: int, critic_out: int) -> "LearnerLayout":
        actor = NetLayout.build(state_dim, hidden, action_dim, base=0)
        critic = NetLayout.build(state_dim + action_dim, hidden, critic_out, base=pad4(actor.numel))
        return LearnerLayout(state_dim, action_dim, hidden, critic_out, actor, critic,
                             critic.base + critic.numel)

    @staticmethod
    def from_config(cfg) -> "LearnerLayout":
        out = cfg["num_atoms"] if cfg["model"] == "d4pg" else 1
        return LearnerLayout.build(cfg["state_dim"], cfg["action_dim"], cfg["dense_size"], out)


STATE_DICT_KEYS = ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                   "linear3.weight", "linear3.bias")


class NetView:
    """Unpadded tensor views of one network inside a flat buffer."""

    def __init__(self, flat: torch.Tensor, layout: NetLayout):
        self.layout = layout
        self.flat = flat[layout.base: layout.base + layout.numel]
        self.weights: List[torch.Tensor] = []
        self.biases: List[torch.Tensor] = []
        for s in layout.layers:
            w = flat[s.w_off: s.w_off + s.w_numel].view(s.out_dim, s.ld)[:, : s.in_dim]
            b = flat[s.b_off: s.b_off + s.out_dim]
            self.weights.append(w)
            self.biases.append(b)

    def tensors(self) -> List[torch.Tensor]:
        """``[W1, b1, W2, b2, W3, b3]`` — the order of ``nn.Module.parameters()`` in the reference."""
        out = []
        for w, b in zip(self.weights, self.biases):
            out += [w, b]
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: t.detach().clone().contiguous().cpu() for k, t in zip(STATE_DICT_KEYS, self.tensors())}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        """Accepts reference checkpoints; ``bn1.*`` entries of the reference's
        D3PG critic (``models/d3pg/networks.py:19``, never used in its forward)
        are ignored."""
        extra = [k for k in sd if k not in STATE_DICT_KEYS and not k.startswith("bn1.")]
        if strict and extra:
            raise KeyError(f"unexpected keys in state_dict: {extra}")
        with torch.no_grad():
            for k, t in zip(STATE_DICT_KEYS, self.tensors()):
                if k not in sd:
                    raise KeyError(f"missing key '{k}' in state_dict")
                src = torch.as_tensor(sd[k], dtype=torch.float32)
                if tuple(src.shape) != tuple(t.shape):
                    raise ValueError(f"{k}: shape {tuple(src.shape)} != expected {tuple(t.shape)}")
                t.copy_(src.to(t.device))

    def forward(self, x: torch.Tensor, final: str) -> torch.Tensor:
        from ..ops import ref
        w1, b1, w2, b2, w3, b3 = self.tensors()
        return ref.mlp3(x, w1, b1, w2, b2, w3, b3, final)


def init_net_(view: NetView, final_init: float, gen: Optional[torch.Generator] = None) -> None:
    """Reference initialisation (``networks.py:23-28,59-64``): layers 1-2 use the
    ``nn.Linear`` default (U(+-1/sqrt(fan_in)) for weight and bias), layer 3 weight
    and bias are U(+-final_init).  Draws on the CPU for device-independent seeds."""
    with torch.no_grad():
        for i, (w, b) in enumerate(zip(view.weights, view.biases)):
            bound = final_init if i == 2 else 1.0 / math.sqrt(w.shape[1])
            w.copy_((torch.rand(w.shape, generator=gen) * 2 - 1) * bound)
            b.copy_((torch.rand(b.shape, generator=gen) * 2 - 1) * bound)


class ParamPack:
    """Flat parameter buffer of a learner (actor + critic) with named views."""

    def __init__(self, layout: LearnerLayout, device="cpu", flat: Optional[torch.Tensor] = None):
        self.layout = layout
        if flat is None:
            flat = torch.zeros(layout.numel, dtype=torch.float32, device=device)
        assert flat.numel() == layout.numel and flat.dtype == torch.float32
        self.flat = flat
        self.actor = NetView(flat, layout.actor)
        self.critic = NetView(flat, layout.critic)

    @property
    def device(self):
        return self.flat.device

    def init_(self, final_init: float = 3e-3, seed: Optional[int] = None) -> "ParamPack":
        gen = None
        if seed is not None:
            gen = torch.Generator().manual_seed(int(seed))
        self.flat.zero_()
        init_net_(self.actor, final_init, gen)
        init_net_(self.critic, final_init, gen)
        return self

    def clone(self) -> "ParamPack":
        return ParamPack(self.layout, flat=self.flat.detach().clone())

    def actor_flat(self) -> torch.Tensor:
        a = self.layout.actor
        return self.flat[a.base: a.base + a.numel]

    def critic_flat(self) -> torch.Tensor:
        c = self.layout.critic
        return self.flat[c.base: c.base + c.numel]


class ActorPolicy:
    """Stand-alone actor used by agents (``PolicyNetwork.get_action``, ``networks.py:78-81``)."""

    def __init__(self, state_dim: int, action_dim: int, hidden: int, device="cpu"):
        self.layout = NetLayout.build(state_dim, hidden, action_dim)
        self.flat = torch.zeros(self.layout.numel, dtype=torch.float32, device=device)
        self.view = NetView(self.flat, self.layout)
        self.device = torch.device(device)

    def load_flat(self, flat) -> None:
        with torch.no_grad():
            self.flat.copy_(torch.as_tensor(flat, dtype=torch.float32).to(self.flat.device))

    @torch.no_grad()
    def act(self, states) -> torch.Tensor:
        x = torch.as_tensor(states, dtype=torch.float32, device=self.device)
        if x.ndim == 1:
            x = x.unsqueeze(0)
        return self.view.forward(x, "tanh")

    def state_dict(self):
        return self.view.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.view.load_state_dict(sd, strict)


# ---------------------------------------------------------------------------
# bf16 shadow weights for the fused bf16 pipeline (csrc/fused.hip)
# ---------------------------------------------------------------------------

def pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def pad32(n: int) -> int:
    return (n + 31) // 32 * 32


@dataclass(frozen=True)
class ShadowSlot:
    """bf16 copies of one weight matrix ``W[out, in]``.

    forward   ``Wf[pad16(out), pad32(in)]``      — B operand of ``y = x W^T`` (K = in)
    backward  ``Wb[pad16(c1-c0), pad32(out)]``   — B operand of ``dX = dY W`` restricted to input
                                                   columns ``[c0, c1)`` (K = out)
    Both are stored in MFMA fragment order (:func:`pack_fragments`): the 64 lanes x 8 elements
    a wave feeds to ``v_mfma_f32_16x16x32_bf16`` for one (16-column tile, 32-wide K-step) are one
    contiguous 1 KB block, so the kernels stream weights as whole cache lines.
    Offsets are in bf16 elements, 16-byte aligned; ``b_off = -1`` when no transpose is kept.
    """
    f_off: int
    f_rows: int
    f_ld: int
    b_off: int
    b_rows: int
    b_ld: int
    b_c0: int
    b_c1: int


@dataclass(frozen=True)
class ShadowLayout:
    actor: Tuple[ShadowSlot, ShadowSlot, ShadowSlot]
    critic: Tuple[ShadowSlot, ShadowSlot, ShadowSlot]
    numel: int

    @staticmethod
    def build(lay: "LearnerLayout") -> "ShadowLayout":
        off = 0

        def slot(s: LayerSlot, back):
            nonlocal off
            f_rows, f_ld = pad16(s.out_dim), pad32(s.in_dim)
            f_off = off
            off += f_rows * f_ld
            if back is None:
                return ShadowSlot(f_off, f_rows, f_ld, -1, 0, 0, 0, 0)
            c0, c1 = back
            b_rows, b_ld = pad16(c1 - c0), pad32(s.out_dim)
            b_off = off
            off += b_rows * b_ld
            return ShadowSlot(f_off, f_rows, f_ld, b_off, b_rows, b_ld, c0, c1)

        a, c = lay.actor.layers, lay.critic.layers
        S, A, Hd = lay.state_dim, lay.action_dim, lay.hidden
        actor = (slot(a[0], None), slot(a[1], (0, Hd)), slot(a[2], (0, Hd)))
        # critic layer 1 only ever back-propagates into the action columns
        critic = (slot(c[0], (S, S + A)), slot(c[1], (0, Hd)), slot(c[2], (0, Hd)))
        return ShadowLayout(actor, critic, off)


def pack_fragments(m: torch.Tensor) -> torch.Tensor:
    """``[rows (multiple of 16), K (multiple of 32)]`` -> flat fragment order
    ``[rows/16][K/32][(k%32)/8][row%16][k%8]`` (``packed_off`` in ``csrc/common.h``)."""
    rows, k = m.shape
    assert rows % 16 == 0 and k % 32 == 0
    return m.reshape(rows // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


def unpack_fragments(flat: torch.Tensor, rows: int, k: int) -> torch.Tensor:
    """Inverse of :func:`pack_fragments`."""
    return flat.reshape(rows // 16, k // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(rows, k)


def refresh_shadows(flat: torch.Tensor, shadow: torch.Tensor, lay: "LearnerLayout", sl: ShadowLayout) -> None:
    """Rebuild every bf16 shadow from the fp32 masters with torch ops (initialisation,
    checkpoint load, data-parallel broadcast; the steady state is maintained by the
    optimiser kernel)."""
    with torch.no_grad():
        shadow.zero_()
        for net, slots in ((lay.actor, sl.actor), (lay.critic, sl.critic)):
            for s, sh in zip(net.layers, slots):
                W = flat[s.w_off: s.w_off + s.w_numel].view(s.out_dim, s.ld)[:, : s.in_dim]
                f = torch.zeros(sh.f_rows, sh.f_ld, dtype=torch.bfloat16, device=shadow.device)
                f[: s.out_dim, : s.in_dim] = W.to(torch.bfloat16)
                shadow[sh.f_off: sh.f_off + sh.f_rows * sh.f_ld] = pack_fragments(f)
                if sh.b_off >= 0:
                    b = torch.zeros(sh.b_rows, sh.b_ld, dtype=torch.bfloat16, device=shadow.device)
                    b[: sh.b_c1 - sh.b_c0, : s.out_dim] = W[:, sh.b_c0: sh.b_c1].t().to(torch.bfloat16)
                    shadow[sh.b_off: sh.b_off + sh.b_rows * sh.b_ld] = pack_fragments(b)


# ---------------------------------------------------------------------------
# packed fp32 weight copies for the fused fp32 pipeline (csrc/fused32.hip, csrc/mlp32_dev.h)
# ---------------------------------------------------------------------------

def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


@dataclass(frozen=True)
class Packed32:
    """One B operand ``(N outputs, K)`` in fp32 fragment order ``[n/64][k/4][n%64][k%4]``
    (``packed32_off`` in ``csrc/mlp32_dev.h``): ``ntiles`` column tiles of 64, ``ks`` chunks of 32 k.
    ``ksplit`` > 1: a single-tile layer (N <= 64) whose K is dealt to the waves of a workgroup in
    ``ksplit`` slices of ``nch`` chunks (``ks = nch * ksplit``, zero padded)."""
    off: int
    N: int
    K: int
    ntiles: int
    ks: int
    ksplit: int
    nch: int
    #: fragment order of ``v_mfma_f32_16x16x4_f32`` (``packed16_off`` in ``csrc/mlp16_dev.h``):
    #: ``[n/16][k/16][(k%16)/4][n%16][k%4]``, ``ntiles`` column tiles of 16, ``ks`` groups of 16 k
    wide: bool = False

    @property
    def numel(self) -> int:
        return self.ntiles * self.ks * (256 if self.wide else 2048)

    @staticmethod
    def geometry_wide(N: int, K: int, waves: int):
        """Tiles of 16 columns, groups of 16 k.  A layer with at most ``waves / 2`` column tiles is dealt to the
        waves in K-slices (one slice of one tile per wave, no slice empty)."""
        ntiles, ks = (N + 15) // 16, (K + 15) // 16
        want = 1
        while ntiles * want * 2 <= waves:
            want *= 2
        nch = (ks + want - 1) // want
        ksplit = (ks + nch - 1) // nch
        return ntiles, ks, ksplit, nch

    @staticmethod
    def geometry(N: int, K: int, split: bool, waves: int):
        ntiles, ks = (N + 63) // 64, (K + 31) // 32
        if split:
            assert ntiles == 1
            nch = (ks + waves - 1) // waves
            ksplit = (ks + nch - 1) // nch
            return ntiles, nch * ksplit, ksplit, nch
        return ntiles, ks, 1, ks


@dataclass(frozen=True)
class Shadow32Slot:
    """Packed fp32 copies of one weight matrix ``W[out, in]``: ``f`` = B operand of ``y = x W^T``
    (n = out, k = in); ``b`` = B operand of ``dX = dY W`` restricted to input columns ``[c0, c1)``
    (n = in - c0, k = out), or ``None``."""
    f: Packed32
    b: Optional[Packed32]
    b_c0: int
    b_c1: int


@dataclass(frozen=True)
class Shadow32Layout:
    actor: Tuple[Shadow32Slot, Shadow32Slot, Shadow32Slot]
    critic: Tuple[Shadow32Slot, Shadow32Slot, Shadow32Slot]
    numel: int

    @staticmethod
    def build(lay: "LearnerLayout", waves: int, wide: bool = False) -> "Shadow32Layout":
        off = 0

        def packed(N, K, split):
            nonlocal off
            if wide:
                nt, ks, ksplit, nch = Packed32.geometry_wide(N, K, waves)
            else:
                nt, ks, ksplit, nch = Packed32.geometry(N, K, split, waves)
            p = Packed32(off, N, K, nt, ks, ksplit, nch, wide)
            off += p.numel
            return p

        def slot(s: LayerSlot, f_split: bool, back, b_split: bool = False):
            f = packed(s.out_dim, s.in_dim, f_split)
            if back is None:
                return Shadow32Slot(f, None, 0, 0)
            c0, c1 = back
            return Shadow32Slot(f, packed(c1 - c0, s.out_dim, b_split), c0, c1)

        a, c = lay.actor.layers, lay.critic.layers
        S, A, Hd = lay.state_dim, lay.action_dim, lay.hidden
        # output layers (N = actions / atoms <= 64, K = hidden) are dealt out in K-slices; so is the transpose
        # of the action columns of critic layer 1, the only part of it that is ever back-propagated through
        actor = (slot(a[0], False, None), slot(a[1], False, (0, Hd)), slot(a[2], True, (0, Hd)))
        critic = (slot(c[0], False, (S, S + A), b_split=True), slot(c[1], False, (0, Hd)), slot(c[2], True, (0, Hd)))
        return Shadow32Layout(actor, critic, off)


def pack32(m: torch.Tensor, p: Packed32) -> torch.Tensor:
    """``m[N, K]`` -> the flat packed copy described by ``p`` (zero padded)."""
    if p.wide:
        full = torch.zeros(p.ntiles * 16, p.ks * 16, dtype=torch.float32, device=m.device)
        full[: m.shape[0], : m.shape[1]] = m
        # [tile][c][group][q][j] -> [tile][group][q][c][j]
        return full.reshape(p.ntiles, 16, p.ks, 4, 4).permute(0, 2, 3, 1, 4).reshape(-1)
    full = torch.zeros(p.ntiles * 64, p.ks * 32, dtype=torch.float32, device=m.device)
    full[: m.shape[0], : m.shape[1]] = m
    return full.reshape(p.ntiles, 64, p.ks * 8, 4).permute(0, 2, 1, 3).reshape(-1)


def unpack32(flat: torch.Tensor, p: Packed32) -> torch.Tensor:
    """Inverse of :func:`pack32` (true rows / columns only)."""
    if p.wide:
        return flat.reshape(p.ntiles, p.ks, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(p.ntiles * 16, p.ks * 16)[: p.N, : p.K]
    return flat.reshape(p.ntiles, p.ks * 8, 64, 4).permute(0, 2, 1, 3).reshape(p.ntiles * 64, p.ks * 32)[: p.N, : p.K]


def refresh_shadows32(flat: torch.Tensor, shadow: torch.Tensor, lay: "LearnerLayout", sl: Shadow32Layout) -> None:
    """Rebuild every packed fp32 copy from the masters with torch ops (initialisation, checkpoint load,
    parameter broadcast; the steady state is maintained by the optimiser kernels)."""
    with torch.no_grad():
        shadow.zero_()
        for net, slots in ((lay.actor, sl.actor), (lay.critic, sl.critic)):
            for s, sh in zip(net.layers, slots):
                W = flat[s.w_off: s.w_off + s.w_numel].view(s.out_dim, s.ld)[:, : s.in_dim]
                shadow[sh.f.off: sh.f.off + sh.f.numel] = pack32(W, sh.f)
                if sh.b is not None:
                    shadow[sh.b.off: sh.b.off + sh.b.numel] = pack32(W[:, sh.b_c0: sh.b_c1].t(), sh.b)
