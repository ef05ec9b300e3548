"""MI355X-native distributed D4PG / D3PG / DDPG actor-learner framework."""
import os as _os

# Learner ranks, the GPU exploiter and RCCL share device buffers across processes through dmabuf IPC
# handles; the legacy IPC mode fails with "hipIpcGetMemHandle: invalid argument" on hosts whose driver
# only offers dmabuf.  Has to be in the environment before the HIP runtime starts, i.e. before the first
# GPU call of the process (importing this package first is enough).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
