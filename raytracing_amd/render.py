"""renderFrame for the Python harness: same arguments and picture layout as the
reference's renderFrame (jvm/src/main/scala/main.scala:73-99), computed by
libhiptrace on the GPU.
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Sequence, Union

import numpy as np

from . import _abi as A
from . import lib as ht
from .scene import Camera, FlatScene, Hittable, flatten


def render_frame(camera: Camera, world: Union[Sequence[Hittable], FlatScene, "ht.Scene"], img: np.ndarray,
                 samples: int, update: Optional[Callable[[int], None]] = None,
                 finish: Optional[Callable[[float], None]] = None,
                 should_break: Optional[Callable[[], bool]] = None, seed: int = 0,
                 precision: int = A.HT_PRECISION_FP32, batch: int = 64) -> np.ndarray:
    """Fill `img` ([H, W] uint32, 0x00RRGGBB, row 0 = top: the layout of the
    reference's TYPE_INT_RGB BufferedImage) with `samples` paths per pixel.

    The device renders every row at once, in batches of `batch` samples (an
    ht_frame): after each batch `img` holds the picture so far and update() gets the
    index of the row the reference would have finished after the same share of the
    work (H-1 at the end); should_break() is polled between batches, as the
    reference polls it between rows (main.scala:80), and a frame that is stopped
    keeps what it has; finish(seconds) as in the reference.
    """
    H, W = img.shape
    own = not isinstance(world, ht.Scene)
    scene = world if not own else ht.Scene(world if isinstance(world, FlatScene) else flatten(world))
    start = time.time()
    try:
        with ht.Frame(scene, W, H) as frame:
            done = 0
            while done < samples:
                if should_break is not None and should_break():
                    break
                n = min(batch, samples - done)
                frame.accumulate(camera, n, sample_offset=done, seed=seed, precision=precision)
                done += n
                img[...] = frame.read()["packed"]
                if update is not None:
                    update(max(0, H * done // samples - 1))
    finally:
        if own:
            scene.close()
    if finish is not None:
        finish(int((time.time() - start) * 1000) / 1000.0)  # whole milliseconds, main.scala:77, 97-98
    return img
