"""ctypes bindings for the hand-written gfx950 kernels in ``csrc/*.hip``.

The kernels are compiled by ``hipcc --offload-arch=gfx950`` into the in-tree
shared object ``csrc/libd4pg_hip.so`` (see ``build.py``) with a C ABI: every
launcher takes a pointer to an argument struct plus a ``hipStream_t`` and only
enqueues work, so a sequence of calls issued while a torch stream is being
captured becomes one HIP graph.

Importing this module never needs a GPU; calling a launcher does.  If the
shared object is missing the loader raises (there is no silent fallback on a
GPU box).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# D4PG_HIP_LIB: load another build of the library (kernel-variant A/B runs, bench/exp/build_variant.py)
LIB_PATH = os.environ.get("D4PG_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libd4pg_hip.so")

_lib: Optional[C.CDLL] = None

# ---------------------------------------------------------------------------
# struct mirrors (field order and types must match csrc/*.hip exactly)
# ---------------------------------------------------------------------------
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_NONE, EPI_RELU, EPI_TANH, EPI_MUL_RELU_MASK, EPI_MUL_TANH_GRAD = 0, 1, 2, 3, 4
MAX_GROUP = 6
MAXL = 6

vp = C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("aux", vp), ("colsum", vp),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int), ("ldaux", C.c_int),
                ("kind", C.c_int), ("epi", C.c_int),
                ("tiles_m", C.c_int), ("tiles_n", C.c_int), ("tile_base", C.c_int), ("pad_", C.c_int)]


class GemmGroup(C.Structure):
    _fields_ = [("p", GemmDesc * MAX_GROUP), ("count", C.c_int), ("total_tiles", C.c_int)]


class D4pgLossArgs(C.Structure):
    _fields_ = [("logits_t", vp), ("logits", vp), ("reward", vp), ("done", vp), ("gamma", vp),
                ("weights", vp), ("dlogits", vp), ("per_loss", vp), ("priority", vp),
                ("proj_out", vp), ("metrics", vp),
                ("B", C.c_int), ("N", C.c_int), ("ld", C.c_int), ("use_ce", C.c_int),
                ("v_min", C.c_float), ("v_max", C.c_float), ("fixed_gamma", C.c_float),
                ("inv_B", C.c_float)]


class D4pgActorLossArgs(C.Structure):
    _fields_ = [("logits", vp), ("dlogits", vp), ("q_out", vp), ("metrics", vp),
                ("B", C.c_int), ("N", C.c_int), ("ld", C.c_int),
                ("v_min", C.c_float), ("v_max", C.c_float), ("inv_B", C.c_float)]


class D3pgLossArgs(C.Structure):
    _fields_ = [("q_t", vp), ("q", vp), ("reward", vp), ("done", vp), ("gamma", vp), ("weights", vp),
                ("dq", vp), ("per_loss", vp), ("priority", vp), ("metrics", vp),
                ("B", C.c_int), ("ldq", C.c_int),
                ("fixed_gamma", C.c_float), ("min_value", C.c_float), ("max_value", C.c_float),
                ("inv_B", C.c_float)]


class D3pgActorLossArgs(C.Structure):
    _fields_ = [("q", vp), ("dq", vp), ("q_out", vp), ("B", C.c_int), ("ldq", C.c_int),
                ("inv_B", C.c_float)]


class ShadowMat(C.Structure):
    _fields_ = [("w_off", C.c_longlong), ("f_off", C.c_longlong), ("b_off", C.c_longlong),
                ("out_dim", C.c_int), ("in_dim", C.c_int), ("ld", C.c_int),
                ("f_ld", C.c_int), ("b_ld", C.c_int), ("b_c0", C.c_int), ("b_c1", C.c_int), ("pad_", C.c_int)]


class ShadowArgs(C.Structure):
    _fields_ = [("p_shadow", vp), ("t_shadow", vp), ("m", ShadowMat * 3), ("count", C.c_int), ("f32", C.c_int)]


class AdamPolyakArgs(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("t", vp), ("st", vp), ("metrics", vp),
                ("per_loss", vp), ("weights", vp), ("q_val", vp), ("B", C.c_int), ("inv_B", C.c_float),
                ("n", C.c_longlong),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("tau", C.c_float), ("gscale", C.c_float),
                ("finish_step", C.c_int), ("pad_", C.c_int), ("sh", ShadowArgs)]


class TreeDesc(C.Structure):
    _fields_ = [("leaf", vp), ("sum", vp * (MAXL + 1)), ("mn", vp * (MAXL + 1)),
                ("n", C.c_longlong * (MAXL + 1)), ("L", C.c_int), ("pad_", C.c_int)]


class ReplayDesc(C.Structure):
    _fields_ = [("rows", vp), ("capacity", C.c_longlong),
                ("S", C.c_int), ("A", C.c_int), ("row_ld", C.c_int), ("pad_", C.c_int)]


class InsertArgs(C.Structure):
    _fields_ = [("rp", ReplayDesc), ("staged", vp), ("head", C.c_longlong), ("new_size", C.c_longlong),
                ("new_head", C.c_longlong), ("n", C.c_int), ("write_tree_inputs", C.c_int),
                ("slot_out", vp), ("prio_out", vp), ("st", vp)]


class TreeUpdateArgs(C.Structure):
    _fields_ = [("t", TreeDesc), ("idx", vp), ("prio", vp), ("st", vp), ("ticket", vp), ("n", C.c_int),
                ("alpha", C.c_float), ("parts", C.c_int), ("pad_", C.c_int)]


class SampleArgs(C.Structure):
    _fields_ = [("rp", ReplayDesc), ("t", TreeDesc), ("st", vp),
                ("sa", vp), ("s_pi", vp), ("s2_a2", vp),
                ("reward", vp), ("done", vp), ("gamma", vp), ("weights", vp), ("idx", vp),
                ("B", C.c_int), ("ld_x", C.c_int), ("prioritized", C.c_int), ("do_tick", C.c_int),
                ("beta1", C.c_float), ("beta2", C.c_float), ("per_b0", C.c_float), ("per_b1", C.c_float),
                ("total_steps", C.c_longlong)]


class SampleRowsArgs(C.Structure):
    _fields_ = [("rp", ReplayDesc), ("t", TreeDesc), ("st", vp), ("out", vp),
                ("B", C.c_int), ("K", C.c_int), ("prioritized", C.c_int), ("pad_", C.c_int),
                ("step", C.c_longlong),
                ("per_b0", C.c_float), ("per_b1", C.c_float), ("total_steps", C.c_longlong)]


class UnpackArgs(C.Structure):
    _fields_ = [("in0", vp), ("in1", vp), ("st", vp), ("sa", vp), ("s_pi", vp), ("s2_a2", vp),
                ("reward", vp), ("done", vp), ("gamma", vp), ("weights", vp), ("idx", vp),
                ("B", C.c_int), ("Bl", C.c_int), ("K", C.c_int),
                ("S", C.c_int), ("A", C.c_int), ("row_ld", C.c_int), ("ld_x", C.c_int),
                ("do_tick", C.c_int), ("step0", C.c_longlong),
                ("beta1", C.c_float), ("beta2", C.c_float), ("per_b0", C.c_float), ("per_b1", C.c_float),
                ("total_steps", C.c_longlong)]


PACK_EXTRA = 4      # floats appended to a replay row in a packed batch (csrc/replay.hip)


class ScatterBatchArgs(C.Structure):
    _fields_ = [("state", vp), ("action", vp), ("next_state", vp), ("sa", vp), ("s_pi", vp),
                ("s2_a2", vp), ("B", C.c_int), ("S", C.c_int), ("A", C.c_int), ("ld_x", C.c_int)]


class NstepArgs(C.Structure):
    _fields_ = [("reward", vp), ("done", vp), ("trunc", vp), ("R", vp), ("gk", vp), ("nxt", vp),
                ("d", vp), ("T", C.c_int), ("E", C.c_int), ("n", C.c_int), ("gamma", C.c_float)]


class LayerW(C.Structure):
    _fields_ = [("W", vp), ("bias", vp), ("ks", C.c_int), ("ntiles", C.c_int), ("N", C.c_int), ("pad_", C.c_int)]


class PhaseDims(C.Structure):
    _fields_ = [("B", C.c_int), ("Bp", C.c_int), ("S", C.c_int), ("A", C.c_int), ("H", C.c_int), ("NO", C.c_int),
                ("ldx_in", C.c_int), ("ldx", C.c_int), ("ldh", C.c_int), ("ldd", C.c_int), ("lda_", C.c_int),
                ("nop", C.c_int), ("distributional", C.c_int), ("use_ce", C.c_int),
                ("v_min", C.c_float), ("v_max", C.c_float), ("fixed_gamma", C.c_float), ("inv_B", C.c_float),
                ("min_value", C.c_float), ("max_value", C.c_float)]


class PackedIn(C.Structure):
    _fields_ = [("in0", vp), ("in1", vp), ("st", vp), ("step0", C.c_longlong),
                ("K", C.c_int), ("Bl", C.c_int), ("row_ld", C.c_int), ("enabled", C.c_int), ("do_tick", C.c_int),
                ("beta1", C.c_float), ("beta2", C.c_float), ("per_b0", C.c_float), ("per_b1", C.c_float),
                ("pad_", C.c_int), ("total_steps", C.c_longlong)]


class ActorFwd(C.Structure):
    _fields_ = [("af", LayerW * 3), ("s", vp), ("sT", vp), ("h1T", vp), ("h2T", vp),
                ("g_xin", vp), ("g_h1", vp), ("g_h2", vp), ("g_pi", vp)]


class CriticAhead(C.Structure):
    _fields_ = [("mode", C.c_int), ("role_map", C.c_int), ("h1", vp), ("h2", vp), ("lg", vp)]


class CriticPhaseArgs(C.Structure):
    _fields_ = [("d", PhaseDims), ("sa", vp), ("s2", vp), ("reward", vp), ("done", vp), ("gamma", vp),
                ("weights", vp), ("pk", PackedIn), ("ta", LayerW * 3), ("tc", LayerW * 3), ("cf", LayerW * 3),
                ("cb3", LayerW), ("cb2", LayerW),
                ("saT", vp), ("h1T", vp), ("h2T", vp), ("dlT", vp), ("dh2T", vp), ("dh1T", vp),
                ("per_loss", vp), ("priority", vp), ("fwd", ActorFwd), ("ahead", CriticAhead)]


class ActorPhaseArgs(C.Structure):
    _fields_ = [("d", PhaseDims), ("g_xin", vp), ("g_h1", vp), ("g_h2", vp), ("g_pi", vp),
                ("cf", LayerW * 3),
                ("cb3", LayerW), ("cb2", LayerW), ("cb1", LayerW), ("ab3", LayerW), ("ab2", LayerW),
                ("daT", vp), ("dh2T", vp), ("dh1T", vp), ("q_val", vp),
                ("tree", TreeUpdateArgs), ("pk", PackedIn), ("c_saT", vp), ("c_h1T", vp), ("c_h2T", vp),
                ("ahead", CriticAhead)]


class WgradDesc(C.Structure):
    _fields_ = [("AT", vp), ("BT", vp), ("C", vp), ("rowsum", vp), ("M", C.c_int), ("N", C.c_int),
                ("rowsA", C.c_int), ("rowsB", C.c_int), ("ldc", C.c_int),
                ("tiles_m", C.c_int), ("tiles_n", C.c_int), ("tile_base", C.c_int),
                ("w_off", C.c_longlong), ("b_off", C.c_longlong), ("sf_off", C.c_longlong), ("sb_off", C.c_longlong),
                ("sf_ld", C.c_int), ("sb_ld", C.c_int), ("sb_c0", C.c_int), ("sb_c1", C.c_int)]


class WgradAdam(C.Structure):
    _fields_ = [("p", vp), ("m", vp), ("v", vp), ("t", vp), ("p_shadow", vp), ("t_shadow", vp), ("st", vp),
                ("metrics", vp), ("per_loss", vp), ("weights", vp), ("q_val", vp), ("B", C.c_int),
                ("inv_B", C.c_float), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("tau", C.c_float), ("finish_step", C.c_int)]


class WgradGroup(C.Structure):
    _fields_ = [("p", WgradDesc * 3), ("count", C.c_int), ("total_tiles", C.c_int), ("Bp", C.c_int),
                ("apply_only", C.c_int), ("opt", WgradAdam), ("smp", vp), ("smp_rows", C.c_int), ("tree_parts", C.c_int),
                ("tree", vp)]


# ---- fused fp32 pipeline (csrc/fused32.hip) ----
class LayerW32(C.Structure):
    _fields_ = [("W", vp), ("bias", vp), ("ks", C.c_int), ("ntiles", C.c_int), ("N", C.c_int), ("ksplit", C.c_int),
                ("nch", C.c_int), ("kg", C.c_int)]


class ActorFwd32(C.Structure):
    _fields_ = [("af", LayerW32 * 3), ("s", vp), ("sT", vp), ("h1T", vp), ("h2T", vp),
                ("g_xin", vp), ("g_h1", vp), ("g_h2", vp)]


class CriticPhaseArgs32(C.Structure):
    _fields_ = [("d", PhaseDims), ("sa", vp), ("s2", vp), ("reward", vp), ("done", vp), ("gamma", vp),
                ("weights", vp), ("pk", PackedIn), ("ta", LayerW32 * 3), ("tc", LayerW32 * 3), ("cf", LayerW32 * 3),
                ("cb3", LayerW32), ("cb2", LayerW32),
                ("saT", vp), ("h1T", vp), ("h2T", vp), ("dlT", vp), ("dh2T", vp), ("dh1T", vp),
                ("per_loss", vp), ("priority", vp), ("fwd", ActorFwd32), ("ahead", CriticAhead)]


class ActorPhaseArgs32(C.Structure):
    _fields_ = [("d", PhaseDims), ("g_xin", vp), ("g_h1", vp), ("g_h2", vp),
                ("cf", LayerW32 * 3),
                ("cb3", LayerW32), ("cb2", LayerW32), ("cb1", LayerW32), ("ab3", LayerW32), ("ab2", LayerW32),
                ("daT", vp), ("dh2T", vp), ("dh1T", vp), ("q_val", vp),
                ("tree", TreeUpdateArgs), ("pk", PackedIn), ("c_saT", vp), ("c_h1T", vp), ("c_h2T", vp),
                ("ahead", CriticAhead), ("smp", vp)]


class WgradDesc32(C.Structure):
    _fields_ = [("AT", vp), ("BT", vp), ("C", vp), ("rowsum", vp), ("M", C.c_int), ("N", C.c_int),
                ("rowsA", C.c_int), ("rowsB", C.c_int), ("ldc", C.c_int),
                ("tiles_m", C.c_int), ("tiles_n", C.c_int), ("tile_base", C.c_int),
                ("w_off", C.c_longlong), ("b_off", C.c_longlong), ("sf_off", C.c_longlong), ("sb_off", C.c_longlong),
                ("sf_ks", C.c_int), ("sb_ks", C.c_int), ("sb_c0", C.c_int), ("sb_c1", C.c_int)]


class WgradGroup32(C.Structure):
    _fields_ = [("p", WgradDesc32 * 3), ("count", C.c_int), ("total_tiles", C.c_int), ("Bp", C.c_int),
                ("wide", C.c_int), ("opt", WgradAdam), ("smp", vp), ("smp_rows", C.c_int), ("tree_parts", C.c_int),
                ("tree", vp), ("apply_only", C.c_int), ("pad_", C.c_int)]


class OuFuse32(C.Structure):
    _fields_ = [("x", vp), ("counter", vp), ("seed", C.c_ulonglong), ("theta", C.c_float), ("mu", C.c_float),
                ("sigma", C.c_float), ("low", C.c_float), ("high", C.c_float), ("pad_", C.c_int)]


class ActorInferArgs32(C.Structure):
    _fields_ = [("l", LayerW32 * 3), ("obs", vp), ("act", vp), ("E", C.c_int), ("S", C.c_int), ("A", C.c_int),
                ("H", C.c_int), ("ld_obs", C.c_int), ("ld_act", C.c_int), ("ldx", C.c_int), ("ldh", C.c_int),
                ("action_scale", C.c_float), ("pad_", C.c_int), ("ou", OuFuse32)]


# ---- device-resident rollouts, weight hand-off (csrc/rollout.hip) ----
class PackMat32(C.Structure):
    _fields_ = [("src_off", C.c_longlong), ("bias_off", C.c_longlong), ("dst_off", C.c_longlong),
                ("dst_bias", C.c_longlong), ("out_dim", C.c_int), ("in_dim", C.c_int), ("ld", C.c_int), ("ks", C.c_int)]


class PackActorArgs32(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("m", PackMat32 * 3), ("wide", C.c_int), ("pad_", C.c_int)]


class PendulumArgs(C.Structure):
    _fields_ = [("theta", vp), ("theta_dot", vp), ("t", vp), ("ep_return", vp), ("action", vp), ("obs", vp),
                ("next_obs", vp), ("reward", vp), ("done", vp), ("trunc", vp), ("noise", vp), ("stats", vp), ("counter", vp), ("counter_next", vp),
                ("seed", C.c_ulonglong), ("E", C.c_int), ("ld_act", C.c_int), ("time_limit", C.c_int), ("init", C.c_int),
                ("reward_scale", C.c_float), ("limit_terminal", C.c_int)]


class CartPoleArgs(C.Structure):
    _fields_ = [("state", vp), ("t", vp), ("ep_return", vp), ("action", vp), ("obs", vp), ("next_obs", vp),
                ("reward", vp), ("done", vp), ("trunc", vp), ("noise", vp), ("stats", vp), ("counter", vp), ("counter_next", vp),
                ("seed", C.c_ulonglong), ("E", C.c_int), ("ld_act", C.c_int), ("time_limit", C.c_int), ("init", C.c_int),
                ("limit_terminal", C.c_int), ("pad_", C.c_int)]


class LanderArgs(C.Structure):
    _fields_ = [("state", vp), ("t", vp), ("ep_return", vp), ("action", vp), ("obs", vp), ("next_obs", vp),
                ("reward", vp), ("done", vp), ("trunc", vp), ("noise", vp), ("stats", vp), ("counter", vp), ("counter_next", vp),
                ("seed", C.c_ulonglong), ("E", C.c_int), ("ld_act", C.c_int), ("time_limit", C.c_int), ("init", C.c_int),
                ("limit_terminal", C.c_int), ("reward_scale", C.c_float)]


class PlanarParams(C.Structure):
    """Numbers of a planar mechanism and its task (csrc/planar_env.hip); filled from ``envs/planar.py`` models."""
    _f8, _f82 = C.c_float * 8, (C.c_float * 2) * 8
    _fields_ = [("anchor", _f82), ("com", _f82), ("mass", _f8), ("inertia", _f8), ("rest", _f8), ("lo", _f8), ("hi", _f8),
                ("damping", _f8), ("armature", _f8), ("stiffness", _f8), ("c_local", _f82), ("gear", _f8),
                ("gravity", C.c_float), ("k_contact", C.c_float), ("c_contact", C.c_float), ("c_friction", C.c_float),
                ("mu", C.c_float), ("k_limit", C.c_float), ("c_limit", C.c_float),
                ("dt", C.c_float), ("alive", C.c_float), ("ctrl_cost", C.c_float), ("reset_noise", C.c_float),
                ("q0_y", C.c_float), ("z_off", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float), ("ang_max", C.c_float),
                ("substeps", C.c_int), ("n_links", C.c_int), ("n_contacts", C.c_int), ("n_act", C.c_int),
                ("parents", C.c_ulonglong), ("contact_links", C.c_ulonglong)]


class PlanarEnvArgs(C.Structure):
    _fields_ = [("P", vp), ("state", vp), ("t", vp), ("ep_return", vp), ("action", vp), ("obs", vp), ("next_obs", vp),
                ("reward", vp), ("done", vp), ("trunc", vp), ("noise", vp), ("stats", vp), ("counter", vp), ("counter_next", vp), ("terrain", vp),
                ("seed", C.c_ulonglong), ("E", C.c_int), ("ld_act", C.c_int), ("time_limit", C.c_int), ("init", C.c_int),
                ("limit_terminal", C.c_int), ("kind", C.c_int), ("reward_scale", C.c_float), ("hilly", C.c_int)]


class AntParams(C.Structure):
    """Numbers of the ant and its task (csrc/ant_env.hip); filled from ``envs/spatial.py::AntVec``."""
    _f9, _f93, _f99, _f13, _f133 = C.c_float * 9, (C.c_float * 3) * 9, (C.c_float * 9) * 9, C.c_float * 13, (C.c_float * 3) * 13
    _fields_ = [("anchor", _f93), ("axis", _f93), ("com", _f93), ("mass", _f9), ("inertia", _f99), ("lo", _f9), ("hi", _f9),
                ("damping", _f9), ("armature", _f9), ("c_local", _f133), ("c_rad", _f13),
                ("gravity", C.c_float), ("k_contact", C.c_float), ("c_contact", C.c_float), ("c_friction", C.c_float),
                ("mu", C.c_float), ("k_limit", C.c_float), ("c_limit", C.c_float),
                ("dt", C.c_float), ("gear", C.c_float), ("z_init", C.c_float), ("reset_noise", C.c_float),
                ("ctrl_cost", C.c_float), ("contact_cost", C.c_float), ("alive", C.c_float), ("z_min", C.c_float),
                ("z_max", C.c_float), ("ankle0", C.c_float * 4),
                ("substeps", C.c_int), ("n_links", C.c_int), ("n_contacts", C.c_int), ("pad_", C.c_int),
                ("parent", C.c_int * 9), ("c_link", C.c_int * 13)]


class AntEnvArgs(C.Structure):
    _fields_ = [("P", vp), ("state", vp), ("t", vp), ("ep_return", vp), ("action", vp), ("obs", vp), ("next_obs", vp),
                ("reward", vp), ("done", vp), ("trunc", vp), ("noise", vp), ("stats", vp), ("counter", vp), ("counter_next", vp),
                ("seed", C.c_ulonglong), ("E", C.c_int), ("ld_act", C.c_int), ("time_limit", C.c_int), ("init", C.c_int),
                ("limit_terminal", C.c_int), ("reward_scale", C.c_float)]


class AnalyticEnvArgs(C.Structure):
    _fields_ = [("state", vp), ("t", vp), ("ep_return", vp), ("action", vp), ("obs", vp), ("next_obs", vp),
                ("reward", vp), ("done", vp), ("trunc", vp), ("noise", vp), ("stats", vp), ("counter", vp), ("counter_next", vp),
                ("seed", C.c_ulonglong), ("E", C.c_int), ("ld_act", C.c_int), ("time_limit", C.c_int), ("init", C.c_int),
                ("limit_terminal", C.c_int), ("kind", C.c_int), ("reward_scale", C.c_float)]


class SyntheticEnvArgs(C.Structure):
    _fields_ = [("state", vp), ("t", vp), ("ep_return", vp), ("M", vp), ("N", vp), ("action", vp), ("obs", vp),
                ("next_obs", vp), ("reward", vp), ("done", vp), ("trunc", vp), ("noise", vp), ("stats", vp),
                ("counter", vp), ("counter_next", vp), ("seed", C.c_ulonglong), ("E", C.c_int), ("S", C.c_int), ("A", C.c_int),
                ("ld_act", C.c_int), ("time_limit", C.c_int), ("init", C.c_int), ("terminal_prob", C.c_float),
                ("noise_std", C.c_float)]


class RolloutRowsArgs(C.Structure):
    _fields_ = [("obs", vp), ("act", vp), ("next_obs", vp), ("R", vp), ("gk", vp), ("nxt", vp), ("d", vp), ("rows", vp),
                ("T", C.c_int), ("E", C.c_int), ("S", C.c_int), ("A", C.c_int), ("ld_act", C.c_int), ("row_ld", C.c_int)]


class OuArgs(C.Structure):
    _fields_ = [("x", vp), ("action", vp), ("out", vp), ("sigma", vp), ("counter", vp), ("seed", C.c_ulonglong),
                ("n", C.c_int), ("A", C.c_int), ("theta", C.c_float), ("mu", C.c_float), ("sigma_const", C.c_float),
                ("low", C.c_float), ("high", C.c_float)]


class DevStateMirror(C.Structure):
    """Host mirror of ``struct DevState`` (csrc/common.h) for init / read-back."""
    _fields_ = [("step_done", C.c_longlong), ("step_cur", C.c_longlong),
                ("bc1", C.c_float), ("bc2", C.c_float), ("per_beta", C.c_float),
                ("max_priority", C.c_float),
                ("replay_size", C.c_longlong), ("replay_head", C.c_longlong),
                ("seed", C.c_ulonglong), ("inserted_total", C.c_longlong),
                ("pad_", C.c_longlong * 2)]


def _mirrored_structs():
    """Every ctypes mirror of a kernel argument struct defined in this module (``DevStateMirror`` has its own
    check against ``d4pg_devstate_size``)."""
    return [v for v in globals().values()
            if isinstance(v, type) and issubclass(v, C.Structure) and v is not C.Structure and v is not DevStateMirror
            and v.__module__ == __name__]

def check_struct_sizes(lib) -> int:
    """Compare every argument struct mirrored above with the size the kernels were compiled with: a drifted
    field would otherwise corrupt GPU memory silently.  Returns the number of structs checked."""
    n = 0
    for cls in _mirrored_structs():
        fn = getattr(lib, "d4pg_sizeof_" + cls.__name__)        # a mirror without an export fails here, loudly
        fn.restype = C.c_int
        if fn() != C.sizeof(cls):
            raise RuntimeError(f"{cls.__name__}: {C.sizeof(cls)} bytes in ops/hip.py, {fn()} in csrc/")
        n += 1
    return n


# ---------------------------------------------------------------------------
# loader
# ---------------------------------------------------------------------------

class HipLibraryMissing(RuntimeError):
    pass


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib() -> C.CDLL:
    """Load ``libd4pg_hip.so`` (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: run `python -m d4pg_pytorch_amd.build` (hipcc --offload-arch=gfx950)")
        # make sure the HIP runtime torch ships is the one that gets bound
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        for name, argt in (
            ("d4pg_gemm_group", [C.POINTER(GemmGroup), vp]),
            ("d4pg_critic_loss", [C.POINTER(D4pgLossArgs), vp]),
            ("d4pg_actor_loss", [C.POINTER(D4pgActorLossArgs), vp]),
            ("d3pg_critic_loss", [C.POINTER(D3pgLossArgs), vp]),
            ("d3pg_actor_loss", [C.POINTER(D3pgActorLossArgs), vp]),
            ("d4pg_adam_polyak", [C.POINTER(AdamPolyakArgs), vp]),
            ("d4pg_polyak", [vp, vp, C.c_longlong, C.c_float, vp]),
            ("d4pg_tick", [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, vp]),
            ("d4pg_replay_insert", [C.POINTER(InsertArgs), vp]),
            ("d4pg_tree_update", [C.POINTER(TreeUpdateArgs), vp]),
            ("d4pg_replay_sample", [C.POINTER(SampleArgs), vp]),
            ("d4pg_scatter_batch", [C.POINTER(ScatterBatchArgs), vp]),
            ("d4pg_replay_sample_rows", [C.POINTER(SampleRowsArgs), vp]),
            ("d4pg_batch_unpack", [C.POINTER(UnpackArgs), vp]),
            ("d4pg_nstep", [C.POINTER(NstepArgs), vp]),
            ("d4pg_critic_phase", [C.POINTER(CriticPhaseArgs), C.c_int, vp]),
            ("d4pg_actor_phase", [C.POINTER(ActorPhaseArgs), C.c_int, vp]),
            ("d4pg_phase_lds_bytes", [C.POINTER(PhaseDims), C.c_int, C.c_int]),
            ("d4pg_wgrad", [C.POINTER(WgradGroup), vp]),
            ("d4pg_ou_noise", [C.POINTER(OuArgs), vp]),
            ("d4pg_ou_noise_shared_counter", [C.POINTER(OuArgs), vp]),
            ("d4pg_counter_copy", [vp, vp, vp]),
            ("d4pg_graph_kernel_nodes", [vp]),
            ("d4pg_critic_phase32", [C.POINTER(CriticPhaseArgs32), C.c_int, vp]),
            ("d4pg_actor_phase32", [C.POINTER(ActorPhaseArgs32), C.c_int, vp]),
            ("d4pg_phase32_lds_bytes", [C.POINTER(PhaseDims), C.c_int, C.c_int]),
            ("d4pg_critic_phase32w", [C.POINTER(CriticPhaseArgs32), C.c_int, vp]),
            ("d4pg_actor_phase32w", [C.POINTER(ActorPhaseArgs32), C.c_int, vp]),
            ("d4pg_actor_infer32w", [C.POINTER(ActorInferArgs32), C.c_int, vp]),
            ("d4pg_phase32w_lds_bytes", [C.POINTER(PhaseDims), C.c_int, C.c_int]),
            ("d4pg_critic_phase32wl", [C.POINTER(CriticPhaseArgs32), C.c_int, vp]),
            ("d4pg_actor_phase32wl", [C.POINTER(ActorPhaseArgs32), C.c_int, vp]),
            ("d4pg_phase32wl_lds_bytes", [C.POINTER(PhaseDims), C.c_int]),
            ("d4pg_phase32wl_lds_limit", []),
            ("d4pg_wgrad32", [C.POINTER(WgradGroup32), vp]),
            ("d4pg_actor_infer32", [C.POINTER(ActorInferArgs32), C.c_int, vp]),
            ("d4pg_pack_actor32", [C.POINTER(PackActorArgs32), vp]),
            ("d4pg_pendulum_step", [C.POINTER(PendulumArgs), vp]),
            ("d4pg_cartpole_step", [C.POINTER(CartPoleArgs), vp]),
            ("d4pg_synthetic_env_step", [C.POINTER(SyntheticEnvArgs), vp]),
            ("d4pg_lander_step", [C.POINTER(LanderArgs), vp]),
            ("d4pg_planar_env_step", [C.POINTER(PlanarEnvArgs), C.POINTER(PlanarParams), vp]),
            ("d4pg_analytic_env_step", [C.POINTER(AnalyticEnvArgs), vp]),
            ("d4pg_ant_env_step", [C.POINTER(AntEnvArgs), C.POINTER(AntParams), vp]),
            ("d4pg_rollout_rows", [C.POINTER(RolloutRowsArgs), vp]),
            ("d4pg_ipc_alloc", [C.c_size_t, C.POINTER(vp), C.c_char_p]),
            ("d4pg_ipc_open", [C.c_char_p, C.POINTER(vp)]),
            ("d4pg_ipc_close", [vp]),
            ("d4pg_ipc_free", [vp]),
            ("d4pg_memcpy_d2d_async", [vp, vp, C.c_size_t, vp]),
            ("d4pg_memcpy_h2d_async", [vp, vp, C.c_size_t, vp]),
            ("d4pg_host_register", [vp, C.c_size_t]),
            ("d4pg_host_unregister", [vp]),
        ):
            fn = getattr(_lib, name)
            fn.argtypes = argt
            fn.restype = C.c_int
        _lib.d4pg_devstate_size.restype = C.c_int
        _lib.d4pg_pack_extra.restype = C.c_int
        if _lib.d4pg_pack_extra() != PACK_EXTRA:
            raise RuntimeError("packed-batch layout mismatch between csrc/replay.hip and ops/hip.py")
        if _lib.d4pg_devstate_size() != C.sizeof(DevStateMirror):
            raise RuntimeError("DevState layout mismatch between csrc/common.h and ops/hip.py")
        _lib.d4pg_phase32_waves.restype = C.c_int
        _lib.d4pg_phase32_has_width.restype = C.c_int
        _lib.d4pg_phase32_has_width.argtypes = [C.c_int]
        check_struct_sizes(_lib)
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def stream_ptr(stream: Optional[torch.cuda.Stream] = None) -> int:
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------
# GEMM helpers
# ---------------------------------------------------------------------------

def gemm_desc(kind: int, A: torch.Tensor, lda: int, B: torch.Tensor, ldb: int, Cmat: torch.Tensor, ldc: int,
              M: int, N: int, K: int, bias: Optional[torch.Tensor] = None, epi: int = EPI_NONE,
              aux: Optional[torch.Tensor] = None, ldaux: int = 0,
              colsum: Optional[torch.Tensor] = None) -> GemmDesc:
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cmat.data_ptr()
    d.bias = ptr(bias)
    d.aux = ptr(aux)
    d.colsum = ptr(colsum)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldaux = lda, ldb, ldc, ldaux
    d.kind, d.epi = kind, epi
    return d


def gemm_group(descs: Sequence[GemmDesc], stream: Optional[torch.cuda.Stream] = None) -> None:
    assert 1 <= len(descs) <= MAX_GROUP
    g = GemmGroup()
    for i, d in enumerate(descs):
        g.p[i] = d
    g.count = len(descs)
    _check(lib().d4pg_gemm_group(C.byref(g), stream_ptr(stream)), "d4pg_gemm_group")


def graph_kernel_nodes(graph: "torch.cuda.CUDAGraph") -> int:
    """Kernel launches one replay of ``graph`` makes (a graph captured with ``keep_graph=True``)."""
    n = lib().d4pg_graph_kernel_nodes(graph.raw_cuda_graph())
    if n < 0:
        raise RuntimeError("hipGraphGetNodes failed")
    return int(n)


def launch(name: str, args: C.Structure, stream: Optional[torch.cuda.Stream] = None) -> None:
    _check(getattr(lib(), name)(C.byref(args), stream_ptr(stream)), name)
