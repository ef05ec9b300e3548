"""Learner ops: torch oracles (``ref``), kernel bindings (``hip``) and tensor-level
front-ends (``fused``)."""
from . import ref  # noqa: F401
from .fused import (DeviceOUNoise, d4pg_critic_loss, expected_q, l2_project, linear, mlp3_forward,  # noqa: F401
                    nstep_assemble, polyak_)
