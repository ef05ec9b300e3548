from .logger import Logger, read_events  # noqa: F401
from .misc import empty_torch_queue, frames_to_gif, make_gif, read_config  # noqa: F401
