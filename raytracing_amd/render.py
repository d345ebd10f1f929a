"""renderFrame for the Python harness: same arguments and picture layout as the
reference's renderFrame (jvm/src/main/scala/main.scala:73-99), computed by
libhiptrace on the GPU.
"""
from __future__ import annotations

import threading
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
                 precision: int = A.HT_PRECISION_FP32) -> np.ndarray:
    """Fill `img` ([H, W] uint32, 0x00RRGGBB, row 0 = top: the layout of the
    reference's TYPE_INT_RGB BufferedImage) with `samples` paths per pixel.

    update(j) is called once with the last row index when the frame is done (a
    GPU frame has no per-row progress); finish(seconds) as in the reference;
    should_break() is polled from a watcher thread and forwarded to ht_cancel,
    which takes effect at the next sample batch.
    """
    H, W = img.shape
    own = not isinstance(world, ht.Scene)
    scene = world if not own else ht.Scene(world if isinstance(world, FlatScene) else flatten(world))
    start = time.time()
    stop_watch = threading.Event()
    watcher = None
    if should_break is not None:
        def watch():
            while not stop_watch.wait(0.01):
                if should_break():
                    scene.cancel()
                    return
        watcher = threading.Thread(target=watch, daemon=True)
        watcher.start()
    try:
        out = scene.render(camera, W, H, samples, seed=seed, precision=precision)
        img[...] = out["packed"]
        if update is not None:
            update(H - 1)
    except ht.HipTraceError as e:
        if e.code != A.HT_E_CANCELLED:
            raise
    finally:
        stop_watch.set()
        if watcher is not None:
            watcher.join()
        if own:
            scene.close()
    if finish is not None:
        finish(time.time() - start)
    return img
