"""Learner cores, parameter layouts and the ``torch.nn`` front-ends of the networks."""
from .modules import PolicyNetwork, ValueNetwork  # noqa: F401
