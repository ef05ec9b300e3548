"""Typed configuration for the actor-learner framework.

The YAML files are flat ``key: value`` maps, exactly as in the reference
(``utils/utils.py:55-66`` ``read_config`` and ``configs/*.yml``): every reference
config loads unchanged.  On top of that this module adds what the reference
lacks: defaults, type coercion, validation, ``--set key=value`` overrides and
the keys that describe the MI355X topology (learner ranks, HBM replay,
HIP-graph capture, compute dtype).

The result is still a plain ``dict`` (it is pickled to every worker process
and indexed as ``config['key']`` like the reference does), but a *complete*
one: after :func:`finalize_config` every key in :data:`DEFAULTS` is present.
"""
from __future__ import annotations

import copy
import math
import warnings
from typing import Any, Dict, Iterable, Optional

import yaml

# ---------------------------------------------------------------------------
# Schema
# ---------------------------------------------------------------------------

#: Keys every reference YAML carries (``configs/bipedal_d4pg.yml``) with the
#: value all 30 shipped files agree on, used as the default when absent.
REFERENCE_DEFAULTS: Dict[str, Any] = {
    "env": "Pendulum-v0",
    "state_dim": None,            # inferred from the env table when missing
    "action_dim": None,
    "action_low": -1.0,
    "action_high": 1.0,
    "num_agents": 4,
    "random_seed": 2019,
    "model": "d4pg",
    "batch_size": 256,
    "num_steps_train": 100_000,
    "max_ep_length": 1000,
    "replay_mem_size": 1_000_000,
    "priority_alpha": 0.6,
    "priority_beta_start": 0.4,
    "priority_beta_end": 1.0,
    "discount_rate": 0.99,
    "n_step_returns": 5,
    "update_agent_ep": 1,
    "replay_queue_size": 64,
    "batch_queue_size": 64,
    "replay_memory_prioritized": 0,
    "num_episode_save": 100,
    "device": "cuda",
    "agent_device": "cpu",
    "save_buffer_on_disk": 0,
    "save_reward_threshold": 1.0,
    "critic_learning_rate": 5e-4,
    "actor_learning_rate": 5e-4,
    "dense_size": 400,
    "final_layer_init": 3e-3,
    "num_atoms": 51,
    "v_min": -100.0,
    "v_max": 300.0,
    "tau": 1e-3,
    "results_path": "results",
}

#: Keys that only exist in this framework.
EXTENSION_DEFAULTS: Dict[str, Any] = {
    # --- algorithm switches for the places where the reference is ambiguous ---
    "critic_loss": "bce",            # "bce" = reference semantics (d4pg.py:58,102), "ce" = categorical CE
    "use_per_sample_gamma": True,    # bootstrap with the stored gamma**k column (False: gamma**n_step_returns)
    "min_value": -math.inf,          # D3PG target clamp (d3pg.py:70)
    "max_value": math.inf,
    "scale_action": False,           # multiply tanh output by action_high before acting
    "adam_beta1": 0.9,
    "adam_beta2": 0.999,
    "adam_eps": 1e-8,
    # --- exploration noise (utils/utils.py:9-34 constants) ---
    "ou_theta": 0.15,
    "ou_mu": 0.0,
    "ou_max_sigma": 0.3,
    "ou_min_sigma": 0.3,
    "ou_decay_period": 10_000,
    # --- topology ---
    "num_learners": 1,               # data-parallel learner ranks (one per GPU, RCCL)
    "learner_devices": None,         # list of device indices; default range(num_learners)
    "dp_mode": "auto",               # num_learners > 1: "replicated" (all-gather the sampled rows, every rank runs the
                                     # update on the global batch) | "grad_allreduce" (all-reduce both gradient buckets)
                                     # | "auto": replicated up to a global batch of 2048 rows in fp32 (8192 in bf16), else grad_allreduce
    "exploiter_device": None,        # default: same as ``device``
    "envs_per_agent": 1,             # vectorised environments per explorer/exploiter process
    "env_backend": "auto",           # articulated-body tasks on the host: "native" = C++ host runtime (csrc/planar_host.cpp),
                                     # "numpy" = the fp64 simulator written out in numpy (envs/planar.py), "auto" = native if built
    "gpu_rollout_envs": 0,           # > 0: that many explorer environments stepped ON the learner's GPU (policy, noise,
                                     # dynamics, n-step rows and replay insert without a host hop; envs/device_pendulum.py)
    "rollout_devices": None,         # list of GPU indices: one ROLLOUT WORKER process per entry (actors/rollout_worker.py) steps
                                     # gpu_rollout_envs device environments on THAT GPU, reads the online actor out of the
                                     # learner's HIP-IPC slots and delivers its rows into the learner's HBM (peer write over
                                     # xGMI + one replay_insert); None / []: the device explorers run inside the learner
    "gpu_rollout_segment": 32,       # environment steps per captured rollout segment
    "gpu_rollout_interval": 32,      # learner updates between two segments
    "share_weights_ipc": True,       # GPU agents read the learner's weights in place through HIP IPC
    "replay_device": "auto",         # "hbm" (inside the learner, on its GPU) | "host" (sampler process) | "auto"
    "weight_publish_interval": 100,  # learner updates between actor-weight publications (d4pg.py:141)
    "lossless_replay_queue": True,   # back-pressure instead of silently dropping transitions
    "min_replay_size": None,         # updates start once the buffer holds this many rows (default batch_size)
    "max_env_steps_per_agent": 1_000_000,  # agent.py:60 bound, kept as a safety stop
    "updates_per_env_step": None,    # optional throttle: learner waits so updates <= ratio * env_steps
    # --- MI355X execution ---
    "compute_dtype": "auto",         # GPU learner: "fp32" = "auto" (reference precision, d4pg.py:73-77: exact f32
                                     # MFMA, fused row-local pipeline, tracks eager PyTorch to rounding) | "bf16"
                                     # (opt-in: the same pipeline with bf16 MFMA operands; fp32 master weights, Adam,
                                     # logits and loss)
    "use_hip_graph": True,           # replay the whole update as one captured HIP graph
    "fused_learner": "auto",         # "auto": HIP kernels on cuda, torch ops on cpu; "torch": force eager torch;
                                     # "layers": the fp32 kernels with one launch per layer (any shape)
    "metrics_interval": 100,         # device-side metric accumulator is read back every N updates
    "control_interval": 1024,        # several learners: updates between two agreements on the stop flag (one 8-byte
                                     # all-reduce; a rank must not leave the collectives of the others on its own)
    "ring_capacity": 4096,           # rows per explorer->replay shared-memory ring
    "ingest_chunk": 1024,            # rows per pinned H2D staging chunk
    "ingest_interval": 32,           # learner updates between two drains of the explorer rings
    "ring_dma": True,                # DMA rows straight out of the explorers' shared-memory rings (registered with the
                                     # HIP runtime); False: pop into a pinned staging buffer first
    "dp_optimizer": "tiles",         # gradient-exchange mode: "tiles" = the weight-gradient kernel applies the summed gradients tile by tile, "flat" = the element-wise optimiser kernel (csrc/optim.hip)
    "phase_mfma": "auto",            # fp32 phase kernels: "auto" | "4x4" (v_mfma_f32_4x4x1, compiled widths) | "16x16" (v_mfma_f32_16x16x4, any width)
    "phase_tile_rows": "auto",       # batch rows per phase workgroup: "auto" | 4 | 8 | 16 (fp32) | 16 | 32 (bf16)
    "phase_occupancy": "auto",       # fp32 16x16x4 phase kernels: "auto" | 1 | 2.  2: the lean build (<= 128 registers, LDS tiles sharing buffers; two workgroups per CU where its LDS fits twice); 1: never; auto: from 129 row tiles of 16, and for networks too wide for four 16-row LDS tiles from 2048 rows
    "sample_ahead": 32,              # uniform replay, fused pipelines: batches drawn per sampling launch = updates per
                                     # graph replay (1: every update); 8: 55.3 us per update, 32: 54.4 us
    # --- lifecycle ---
    "resume": None,                  # path to a learner checkpoint bundle to resume from
    "learner_checkpoint_interval": 0,  # 0 = only at the end
    "heartbeat_timeout_s": 120.0,
    "max_restarts_per_process": 3,   # crashed / hung agents and rollout workers are replaced this many times EACH ...
    "max_restarts_per_hour": 20,     # ... and the job gives up only if replacements come faster than this overall
    "log_backend": "tb",             # "tb" (TensorBoard event files) | "jsonl" | "both"
    "check_rings": False,            # race detection (train.py --check): every ring slot carries the running index of
                                     # its row; the consumer verifies it, the run fails on a mismatch
    "synthetic_env": False,          # force the shape-only synthetic env even when the id is known
}

DEFAULTS: Dict[str, Any] = {**REFERENCE_DEFAULTS, **EXTENSION_DEFAULTS}

#: (state_dim, action_dim) of the environments the reference ships configs
#: for (``configs/*.yml``).  Used to infer missing dims and to repair
#: ``configs/hopper_d4pg.yml:4`` (``state_dim: 1`` for an 11-dim observation).
ENV_DIMS: Dict[str, tuple] = {
    "Pendulum-v0": (3, 1),
    "Pendulum-v1": (3, 1),
    "LunarLanderContinuous-v2": (8, 2),
    "BipedalWalker-v2": (24, 4),
    "BipedalWalker-v3": (24, 4),
    "InvertedPendulum-v2": (4, 1),
    "InvertedDoublePendulum-v2": (11, 1),
    "Reacher-v2": (11, 2),
    "Hopper-v2": (11, 3),
    "Walker2d-v2": (17, 6),
    "HalfCheetah-v2": (17, 6),
    "Ant-v2": (111, 8),
}

VALID_MODELS = ("d4pg", "d3pg", "ddpg")

_INT_KEYS = {
    "state_dim", "action_dim", "num_agents", "random_seed", "batch_size", "num_steps_train",
    "max_ep_length", "replay_mem_size", "n_step_returns", "update_agent_ep", "replay_queue_size",
    "batch_queue_size", "replay_memory_prioritized", "num_episode_save", "save_buffer_on_disk",
    "dense_size", "num_atoms", "num_learners", "envs_per_agent", "weight_publish_interval",
    "max_env_steps_per_agent", "metrics_interval", "ring_capacity", "ingest_chunk", "ingest_interval", "sample_ahead", "control_interval",
    "ou_decay_period", "learner_checkpoint_interval", "gpu_rollout_envs", "gpu_rollout_segment", "gpu_rollout_interval",
    "max_restarts_per_process", "max_restarts_per_hour",
}
_FLOAT_KEYS = {
    "action_low", "action_high", "priority_alpha", "priority_beta_start", "priority_beta_end",
    "discount_rate", "save_reward_threshold", "critic_learning_rate", "actor_learning_rate",
    "final_layer_init", "v_min", "v_max", "tau", "min_value", "max_value", "adam_beta1",
    "adam_beta2", "adam_eps", "ou_theta", "ou_mu", "ou_max_sigma", "ou_min_sigma",
    "heartbeat_timeout_s",
}
_BOOL_KEYS = {
    "use_per_sample_gamma", "scale_action", "lossless_replay_queue", "use_hip_graph",
    "synthetic_env", "check_rings", "share_weights_ipc", "ring_dma",
}


class ConfigError(ValueError):
    """Raised for an invalid configuration (unknown model, bad dims, ...)."""


# ---------------------------------------------------------------------------
# Loading
# ---------------------------------------------------------------------------

def read_config(path: str, overrides: Optional[Iterable[str]] = None, finalize: bool = True) -> Dict[str, Any]:
    """Load a flat YAML file and return a validated, complete config dict.

    Same call shape as the reference ``read_config(path)`` (``utils/utils.py:55``);
    uses ``yaml.safe_load`` (the reference's loader-less ``yaml.load`` fails on
    PyYAML >= 6).
    """
    with open(path, "r") as fh:
        raw = yaml.safe_load(fh) or {}
    if not isinstance(raw, dict):
        raise ConfigError(f"{path}: expected a flat mapping, got {type(raw).__name__}")
    if overrides:
        apply_overrides(raw, overrides)
    return finalize_config(raw) if finalize else raw


def apply_overrides(cfg: Dict[str, Any], overrides: Iterable[str]) -> Dict[str, Any]:
    """Apply ``key=value`` strings (values parsed as YAML scalars)."""
    for item in overrides:
        if "=" not in item:
            raise ConfigError(f"override '{item}' is not of the form key=value")
        key, value = item.split("=", 1)
        cfg[key.strip()] = yaml.safe_load(value)
    return cfg


def _coerce(key: str, value: Any) -> Any:
    if value is None:
        return None
    try:
        if key in _INT_KEYS:
            if isinstance(value, str):
                value = value.replace("_", "")
            iv = int(float(value))
            return iv
        if key in _FLOAT_KEYS:
            return float(value)
        if key in _BOOL_KEYS:
            if isinstance(value, str):
                return value.strip().lower() in ("1", "true", "yes", "on")
            return bool(value)
    except (TypeError, ValueError) as exc:
        raise ConfigError(f"config key '{key}': cannot interpret {value!r}") from exc
    return value


def finalize_config(raw: Dict[str, Any]) -> Dict[str, Any]:
    """Fill defaults, coerce types, validate, and resolve ``auto`` values."""
    cfg: Dict[str, Any] = copy.deepcopy(DEFAULTS)
    unknown = [k for k in raw if k not in DEFAULTS]
    for k, v in raw.items():
        cfg[k] = v
    if unknown:
        warnings.warn(f"unknown config keys kept verbatim: {sorted(unknown)}", stacklevel=2)
    for k in list(cfg):
        cfg[k] = _coerce(k, cfg[k])

    model = str(cfg["model"]).lower()
    if model not in VALID_MODELS:
        # the reference returns None here (models/engine.py:7-10) and crashes later
        raise ConfigError(f"unknown model '{cfg['model']}', expected one of {VALID_MODELS}")
    cfg["model"] = model

    # ---- dims: infer or repair against the table of known environments ----
    env = cfg["env"]
    known = ENV_DIMS.get(env)
    for idx, key in enumerate(("state_dim", "action_dim")):
        if cfg[key] is None or cfg[key] <= 0:
            if known is None:
                raise ConfigError(f"'{key}' missing and env '{env}' is not in the known-dims table")
            cfg[key] = known[idx]
        elif known is not None and cfg[key] != known[idx] and not cfg.get("synthetic_env"):
            warnings.warn(
                f"{key}={cfg[key]} contradicts env '{env}' ({known[idx]}); using {known[idx]}",
                stacklevel=2)
            cfg[key] = known[idx]

    # ---- sanity ----
    def need(cond: bool, msg: str) -> None:
        if not cond:
            raise ConfigError(msg)

    need(cfg["batch_size"] > 0, "batch_size must be positive")
    need(cfg["dense_size"] > 0, "dense_size must be positive")
    need(cfg["n_step_returns"] >= 1, "n_step_returns must be >= 1")
    need(0.0 <= cfg["discount_rate"] <= 1.0, "discount_rate must be in [0, 1]")
    need(cfg["replay_mem_size"] >= cfg["batch_size"] or cfg["replay_mem_size"] >= 1,
         "replay_mem_size must be positive")
    need(cfg["action_high"] > cfg["action_low"], "action_high must exceed action_low")
    need(cfg["num_learners"] >= 1, "num_learners must be >= 1")
    need(cfg["dp_mode"] in ("auto", "replicated", "grad_allreduce"),
         "dp_mode must be 'auto', 'replicated' or 'grad_allreduce'")
    need(cfg["envs_per_agent"] >= 1, "envs_per_agent must be >= 1")
    need(cfg["gpu_rollout_envs"] >= 0 and cfg["gpu_rollout_segment"] >= 1 and cfg["gpu_rollout_interval"] >= 1,
         "gpu_rollout_envs must be >= 0, gpu_rollout_segment and gpu_rollout_interval >= 1")
    need(cfg["gpu_rollout_envs"] == 0 or str(cfg["device"]).startswith("cuda"), "gpu_rollout_envs needs device: cuda")
    if cfg["rollout_devices"] in (None, "", []):
        cfg["rollout_devices"] = []
    elif isinstance(cfg["rollout_devices"], (int, str)):
        cfg["rollout_devices"] = [int(x) for x in str(cfg["rollout_devices"]).strip("[]").split(",") if x.strip()]
    else:
        cfg["rollout_devices"] = [int(x) for x in cfg["rollout_devices"]]
    need(not cfg["rollout_devices"] or cfg["gpu_rollout_envs"] > 0, "rollout_devices needs gpu_rollout_envs > 0")
    need(all(d >= 0 for d in cfg["rollout_devices"]), "rollout_devices must be GPU indices")
    need(cfg["critic_loss"] in ("bce", "ce"), "critic_loss must be 'bce' or 'ce'")
    need(cfg["compute_dtype"] in ("auto", "fp32", "bf16"), "compute_dtype must be 'auto', 'fp32' or 'bf16'")
    need(cfg["replay_device"] in ("auto", "hbm", "host"), "replay_device must be auto|hbm|host")
    need(cfg["log_backend"] in ("tb", "jsonl", "both"), "log_backend must be tb|jsonl|both")
    if model == "d4pg":
        need(cfg["num_atoms"] >= 2, "num_atoms must be >= 2")
        need(cfg["v_max"] > cfg["v_min"], "v_max must exceed v_min")

    # A job needs at least one explorer to produce data: the reference ships
    # ``walker2d_d4pg.yml`` with num_agents: 1, which can never train
    # (only explorers write to the replay queue, models/agent.py:101-105).
    if cfg["num_agents"] < 2 and cfg["gpu_rollout_envs"] > 0:
        cfg["num_agents"] = max(1, cfg["num_agents"])        # the device explorers produce the data
    elif cfg["num_agents"] < 2:
        warnings.warn("num_agents < 2 leaves no explorer to generate data; using 2", stacklevel=2)
        cfg["num_agents"] = 2

    dev = str(cfg["device"])
    if cfg["replay_device"] == "auto":
        cfg["replay_device"] = "hbm" if dev.startswith("cuda") else "host"
    if cfg["exploiter_device"] is None:
        cfg["exploiter_device"] = dev
    if cfg["learner_devices"] is None:
        cfg["learner_devices"] = list(range(cfg["num_learners"]))
    if cfg["min_replay_size"] is None:
        cfg["min_replay_size"] = cfg["batch_size"]
    cfg["min_replay_size"] = int(cfg["min_replay_size"])
    return cfg


def default_config(**overrides: Any) -> Dict[str, Any]:
    """A complete config built from defaults plus keyword overrides."""
    return finalize_config(dict(overrides))


def delta_z(cfg: Dict[str, Any]) -> float:
    """Atom spacing of the categorical support (``d4pg.py:36``)."""
    return (cfg["v_max"] - cfg["v_min"]) / (cfg["num_atoms"] - 1)


def beta_at(cfg: Dict[str, Any], update_step: int) -> float:
    """Linearly annealed PER importance exponent.

    The reference documents this schedule in its YAML comments
    (``configs/bipedal_d4pg.yml:19-20``) but never implements it.
    """
    total = max(1, int(cfg["num_steps_train"]))
    frac = min(1.0, max(0.0, update_step / total))
    return cfg["priority_beta_start"] + frac * (cfg["priority_beta_end"] - cfg["priority_beta_start"])
