"""Checkpoint formats.

What the reference does (``models/agent.py:139-161``): only agent 0 — the
exploiter, which holds the TARGET actor — saves, as a whole-module pickle
``torch.save(self.actor, path)`` under
``{experiment_dir}/agent_0/local_episode_{E}_reward_{best:4f}.pt``; nothing else
is ever saved and nothing can be resumed.

Here:

* **actor checkpoints** keep that directory / file-name convention but contain a
  plain ``state_dict`` with the reference key layout
  (``linear{1,2,3}.{weight,bias}``) plus a little metadata, loadable with
  ``torch.load(..., weights_only=True)``;
* **learner bundles** (``learner/checkpoint_{step}.pt``) hold actor, critic, both
  targets, both Adam states, the update step and the config: real resume;
* :func:`import_legacy_actor` reads a whole-module pickle written by the
  reference and returns its ``state_dict`` without importing the reference's
  modules and without executing anything from the file (``weights_only=True``
  plus an exact allowlist of four stand-in classes and two torch layers).
"""
from __future__ import annotations

import os
import pickle
from collections import OrderedDict
from typing import Any, Dict, Optional

import torch

ACTOR_FORMAT = "d4pg_pytorch_amd.actor.v1"
LEARNER_FORMAT = "d4pg_pytorch_amd.learner.v1"


def actor_checkpoint_path(log_dir: str, n_agent: int, local_episode: int, best_reward: float) -> str:
    return os.path.join(log_dir, f"agent_{n_agent}", f"local_episode_{local_episode}_reward_{best_reward:4f}.pt")


def save_actor(path: str, state_dict: Dict[str, torch.Tensor], meta: Optional[Dict[str, Any]] = None) -> str:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    payload = {"format": ACTOR_FORMAT,
               "state_dict": OrderedDict((k, v.detach().cpu().contiguous()) for k, v in state_dict.items()),
               "meta": dict(meta or {})}
    tmp = path + ".tmp"
    torch.save(payload, tmp)
    os.replace(tmp, path)
    return path


def load_actor(path: str, allow_legacy: bool = False) -> Dict[str, torch.Tensor]:
    """Returns a reference-layout ``state_dict`` from any supported actor file.  A whole-module pickle
    written by the reference is only accepted with ``allow_legacy=True`` (:func:`import_legacy_actor`)."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        if not allow_legacy:
            raise
        return import_legacy_actor(path)
    if isinstance(obj, dict) and obj.get("format") == ACTOR_FORMAT:
        return obj["state_dict"]
    if isinstance(obj, dict) and "linear1.weight" in obj:
        return obj
    if isinstance(obj, dict) and "actor" in obj and isinstance(obj["actor"], dict):
        return obj["actor"]      # a learner bundle
    raise ValueError(f"{path}: not an actor checkpoint")


def save_learner_bundle(path: str, bundle: Dict[str, Any], config: Optional[Dict[str, Any]] = None) -> str:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    payload = dict(bundle)
    if config is not None:
        payload["config"] = {k: v for k, v in config.items()
                             if isinstance(v, (int, float, str, bool, type(None), list))}
    tmp = path + ".tmp"
    torch.save(payload, tmp)
    os.replace(tmp, path)
    return path


def load_learner_bundle(path: str) -> Dict[str, Any]:
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(obj, dict) or obj.get("format") != LEARNER_FORMAT:
        raise ValueError(f"{path}: not a learner bundle")
    return obj


# ---------------------------------------------------------------------------
# legacy whole-module pickles
# ---------------------------------------------------------------------------
# The reference writes ``torch.save(self.actor, path)`` (models/agent.py:161): a pickle of the whole
# ``models.d{3,4}pg.networks.PolicyNetwork`` object.  It is read with ``torch.load(weights_only=True)``
# — torch's own no-code unpickler, which builds tensors and containers and refuses every global that is
# not on its allowlist — with exactly these additions to that allowlist, by exact (module, name):
#
# * stand-ins for the reference's two network classes (plain ``nn.Module`` subclasses with no code of
#   their own, registered under the reference's module paths so that nothing of the reference is imported);
# * ``torch.nn.Linear`` and ``torch.nn.BatchNorm1d``, the only submodules those networks hold
#   (models/d4pg/networks.py:23-25,59-61; models/d3pg/networks.py:19).
#
# Nothing is matched by prefix, no ``getattr``, no numpy / copyreg / builtins callables: a file that names
# ``os.system``, ``torch.hub.load`` or anything else is refused by the unpickler before anything runs.

_LEGACY_CLASSES = tuple((f"models.{algo}.networks", name)
                        for algo in ("d4pg", "d3pg") for name in ("PolicyNetwork", "ValueNetwork"))


def _legacy_stub(module: str, name: str) -> type:
    cls = type(name, (torch.nn.Module,), {"__module__": module, "__qualname__": name,
                                           "__doc__": f"stand-in for the reference's {module}.{name}"})
    return cls


def _legacy_allowlist():
    stubs = [_legacy_stub(m, n) for m, n in _LEGACY_CLASSES]
    return stubs + [torch.nn.modules.linear.Linear, torch.nn.modules.batchnorm.BatchNorm1d]


def import_legacy_actor(path: str) -> Dict[str, torch.Tensor]:
    """Extract the ``state_dict`` from a reference-style ``torch.save(module)`` file without executing
    anything the file names (see the allowlist above)."""
    with torch.serialization.safe_globals(_legacy_allowlist()):
        obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, torch.nn.Module):
        sd = obj.state_dict()
    elif isinstance(obj, dict):
        sd = obj
    else:
        raise ValueError(f"{path}: unsupported legacy checkpoint object {type(obj).__name__}")
    keep = OrderedDict((k, v.detach().cpu()) for k, v in sd.items() if k.startswith("linear"))
    if "linear1.weight" not in keep:
        raise ValueError(f"{path}: no linear1/2/3 parameters found")
    return keep
