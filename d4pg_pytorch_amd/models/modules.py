"""``torch.nn`` front-ends of the actor and the critics, for code written against the reference's classes.

The learner does not use these: its parameters live in one flat fp32 buffer (``models/nets.py``) that the HIP
kernels read.  They exist so that a user of ``PolicyNetwork`` / ``ValueNetwork`` (``models/d4pg/networks.py:6-81``,
``models/d3pg/networks.py:6-75``) finds the same constructors, attribute names (``linear1..3``, ``z_atoms``),
``forward``, ``get_action`` and ``get_probs`` — and can move weights both ways:

* ``PolicyNetwork.from_checkpoint(path)`` loads an actor checkpoint written by the exploiter
  (``utils/checkpoint.py``) or a learner bundle's ``actor`` entry;
* ``module.load_view(view)`` / ``module.store_view(view)`` copy from / into a ``NetView`` of the flat buffer.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn


def _default_device(device):
    if device is None:
        return "cuda" if torch.cuda.is_available() else "cpu"       # the reference defaults to 'cuda'
    return device


class _ThreeLayer(nn.Module):
    """input -> hidden (ReLU) -> hidden (ReLU) -> output; the last layer starts in U(-init_w, init_w)."""

    def __init__(self, in_dim: int, hidden: int, out_dim: int, init_w: float, device):
        super().__init__()
        self.linear1 = nn.Linear(in_dim, hidden)
        self.linear2 = nn.Linear(hidden, hidden)
        self.linear3 = nn.Linear(hidden, out_dim)
        with torch.no_grad():
            self.linear3.weight.uniform_(-init_w, init_w)
            self.linear3.bias.uniform_(-init_w, init_w)
        self.device = torch.device(_default_device(device))
        nn.Module.to(self, self.device)

    def _trunk(self, x: torch.Tensor) -> torch.Tensor:
        return self.linear3(torch.relu(self.linear2(torch.relu(self.linear1(x)))))

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        self.device = self.linear1.weight.device
        return out

    # ---- exchange with the flat parameter buffer ----
    def load_view(self, view) -> "_ThreeLayer":
        """Copy the parameters of a ``NetView`` (``models/nets.py``) into this module."""
        sd = {k: v.detach().to(self.device) for k, v in view.state_dict().items()}
        self.load_state_dict(sd, strict=False)
        return self

    def store_view(self, view) -> None:
        """Copy this module's parameters into a ``NetView`` of the learner's / an agent's flat buffer."""
        own = {k: v.detach() for k, v in self.state_dict().items() if k.startswith("linear")}
        view.load_state_dict(own)


class PolicyNetwork(_ThreeLayer):
    """Actor: ``tanh`` action in ``[-1, 1]^A`` for a state (``models/d4pg/networks.py:46-81``)."""

    def __init__(self, num_states: int, num_actions: int, hidden_size: int, init_w: float = 3e-3, device=None):
        super().__init__(num_states, hidden_size, num_actions, init_w, device)

    def forward(self, state: torch.Tensor) -> torch.Tensor:
        return torch.tanh(self._trunk(state))

    def get_action(self, state) -> torch.Tensor:
        """One state (numpy / list / tensor, shape ``[S]``) -> action tensor ``[1, A]`` on the module's device."""
        s = torch.as_tensor(np.asarray(state) if not torch.is_tensor(state) else state, dtype=torch.float32)
        return self.forward(s.reshape(1, -1).to(self.device))

    @classmethod
    def from_checkpoint(cls, path: str, device="cpu", allow_legacy: bool = False) -> "PolicyNetwork":
        """Actor checkpoint (or learner bundle) written by this framework; ``allow_legacy`` additionally accepts
        the reference's whole-module pickles through the allow-listed loader (``utils/checkpoint.py``)."""
        from ..utils.checkpoint import load_actor
        sd: Dict[str, torch.Tensor] = load_actor(path, allow_legacy=allow_legacy)
        hidden, n_in = sd["linear1.weight"].shape
        net = cls(n_in, sd["linear3.weight"].shape[0], hidden, device=device)
        net.load_state_dict({k: v for k, v in sd.items() if k.startswith("linear")})
        return net


class ValueNetwork(_ThreeLayer):
    """Critic.  With ``num_atoms`` > 1 the distributional critic of D4PG (logits over ``z_atoms``,
    ``models/d4pg/networks.py:6-43``); with the default single output the scalar Q of D3PG / DDPG
    (``models/d3pg/networks.py:6-35``; ``batch_norm=True`` adds that file's unused ``bn1`` so its
    ``state_dict`` keys line up)."""

    def __init__(self, num_states: int, num_actions: int, hidden_size: int, v_min: Optional[float] = None,
                 v_max: Optional[float] = None, num_atoms: int = 1, init_w: float = 3e-3, device=None,
                 batch_norm: bool = False):
        super().__init__(num_states + num_actions, hidden_size, num_atoms, init_w, device)
        self.num_atoms = int(num_atoms)
        if self.num_atoms > 1:
            if v_min is None or v_max is None:
                raise ValueError("a distributional critic needs v_min and v_max")
            self.z_atoms = np.linspace(v_min, v_max, num_atoms)
        if batch_norm:
            self.bn1 = nn.BatchNorm1d(num_states + num_actions).to(self.device)     # never applied (as in the reference)

    def forward(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        return self._trunk(torch.cat([state, action], dim=1))

    def get_probs(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        if self.num_atoms < 2:
            raise ValueError("get_probs is defined for the distributional critic (num_atoms > 1)")
        return torch.softmax(self.forward(state, action), dim=1)

    def expected_q(self, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
        """E[Z] under the predicted distribution (the actor's objective, ``models/d4pg/d4pg.py:117-120``), or
        the scalar Q itself."""
        if self.num_atoms < 2:
            return self.forward(state, action).squeeze(1)
        z = torch.as_tensor(self.z_atoms, dtype=torch.float32, device=self.device)
        return (self.get_probs(state, action) * z).sum(dim=1)
