"""Small utilities with reference counterparts in ``utils/utils.py``."""
from __future__ import annotations

import os
import queue as _queue
from glob import glob
from typing import List, Sequence

import numpy as np

from ..config import read_config  # noqa: F401  (reference exposes read_config from utils.utils)


def frames_to_gif(frames: Sequence[np.ndarray], output: str, duration_ms: int = 40) -> str:
    """Write RGB frames as an animated GIF (Pillow; imageio is not required)."""
    from PIL import Image
    if not frames:
        raise ValueError("no frames to write")
    os.makedirs(os.path.dirname(output) or ".", exist_ok=True)
    imgs = [Image.fromarray(np.asarray(f, dtype=np.uint8)) for f in frames]
    imgs[0].save(output, save_all=True, append_images=imgs[1:], duration=duration_ms, loop=0)
    return output


def make_gif(source_dir: str, output: str) -> str:
    """GIF from ``<int>.png`` frames in a directory, ordered by the integer stem
    (``utils/utils.py:37-52``)."""
    from PIL import Image
    paths = sorted(glob(os.path.join(source_dir, "*.png")),
                   key=lambda s: int(os.path.splitext(os.path.basename(s))[0]))
    return frames_to_gif([np.asarray(Image.open(p).convert("RGB")) for p in paths], output)


def empty_torch_queue(q, drain: bool = False) -> int:
    """Release a multiprocessing queue at shutdown so that no process blocks on exit
    (the job the reference's helper of the same name does, ``utils/utils.py:69-76``).

    The reference drains with ``get_nowait`` and then closes.  Draining is only safe
    while the producer is alive: a producer that exits mid-message leaves a truncated
    pickle in the pipe and the next ``get`` blocks forever.  So by default this only
    detaches: ``cancel_join_thread`` (a producer must not wait at exit for its feeder
    thread to flush into a pipe nobody reads any more; one 256-row batch already
    exceeds the pipe buffer) and ``close``.  ``drain=True`` additionally discards what
    is readable first.  Returns the number of items dropped."""
    n = 0
    while drain:
        try:
            q.get_nowait()
            n += 1
        except (_queue.Empty, OSError, ValueError, EOFError):
            break
    try:
        q.cancel_join_thread()
        q.close()
    except Exception:
        pass
    return n
