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
  modules (a restricted unpickler maps its classes onto stubs).
"""
from __future__ import annotations

import io
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


def load_actor(path: str) -> Dict[str, torch.Tensor]:
    """Returns a reference-layout ``state_dict`` from any supported actor file."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
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

class _LegacyModule(torch.nn.Module):
    """Stand-in for ``models.d{3,4}pg.networks.{PolicyNetwork,ValueNetwork}``."""

    def __init__(self, *a, **k):   # never called by unpickling (copyreg.__newobj__ bypasses it)
        super().__init__()


_SAFE_PREFIXES = ("torch.", "collections", "numpy", "_codecs", "copyreg", "builtins")
_SAFE_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes",
                  "complex", "slice", "range", "object", "getattr"}


class _LegacyUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if module.startswith("models.") and name in ("PolicyNetwork", "ValueNetwork"):
            return _LegacyModule
        if module in ("builtins", "__builtin__"):      # protocol-2 pickles spell it the python-2 way
            if name not in _SAFE_BUILTINS:
                raise pickle.UnpicklingError(f"refusing builtins.{name}")
            return super().find_class("builtins", name)
        if module == "torch" or module.startswith(_SAFE_PREFIXES):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to import {module}.{name} from a legacy checkpoint")


class _LegacyPickleModule:
    """``pickle_module`` shim for ``torch.load``."""
    __name__ = "legacy_pickle"
    Unpickler = _LegacyUnpickler
    load = staticmethod(lambda f, **kw: _LegacyUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _LegacyUnpickler(io.BytesIO(b), **kw).load())
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    PicklingError = pickle.PicklingError
    UnpicklingError = pickle.UnpicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def import_legacy_actor(path: str) -> Dict[str, torch.Tensor]:
    """Extract ``state_dict`` from a reference-style ``torch.save(module)`` file."""
    obj = torch.load(path, map_location="cpu", pickle_module=_LegacyPickleModule, weights_only=False)
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
