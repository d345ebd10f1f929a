"""ctypes binding of libhiptrace.so (include/hiptrace.h).

The shared library is built in-tree by raytracing_amd/csrc/Makefile
(`__graft_entry__.build()` runs it).  There is no fallback: if the library is
missing or no HIP device is visible, calls fail with an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPTRACE_LIB: another build of the same library (A/B measurements); default: the in-tree build
LIB_PATH = os.environ.get("HIPTRACE_LIB") or os.path.join(_HERE, "libhiptrace.so")


class HipTraceError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libhiptrace error {code}: {message}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libhiptrace.so and declare every entry point of the header."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.ht_version.restype = C.c_int
    L.ht_device_count.restype = C.c_int
    L.ht_last_error.restype = C.c_char_p
    L.ht_scene_create.argtypes = [C.POINTER(A.ht_scene_desc), C.c_int, C.POINTER(vp)]
    L.ht_scene_destroy.argtypes = [vp]
    L.ht_scene_destroy.restype = None
    L.ht_render.argtypes = [vp, C.POINTER(A.ht_camera), C.POINTER(A.ht_render_params), A._pf32, A._pu32,
                            C.POINTER(A.ht_stats)]
    L.ht_render_noalloc.argtypes = [vp, C.POINTER(A.ht_camera), C.POINTER(A.ht_render_params), vp, vp, vp]
    L.ht_render_accumulate.argtypes = [vp, C.POINTER(A.ht_camera), C.POINTER(A.ht_render_params), vp, vp,
                                       C.POINTER(A.ht_stats)]
    L.ht_render_sharded.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(A.ht_camera), C.POINTER(A.ht_render_params),
                                    A._pf32, A._pu32, C.POINTER(A.ht_stats)]
    L.ht_resolve.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, A._pf32, A._pu32, vp]
    L.ht_resolve_device.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, vp, vp]
    L.ht_shard_tile_pixels.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.ht_shard_tile_pixels.restype = C.c_int64
    L.ht_resolve_tiles_device.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp]
    L.ht_assemble_tiles_device.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
    L.ht_first_hit.argtypes = [vp, C.POINTER(A.ht_camera), C.c_int32, C.c_int32, C.c_int32, A._pf32,
                               A._pi32, A._pf32, A._pf64, A._pf64]
    L.ht_cancel.argtypes = [vp]
    L.ht_scene_get_info.argtypes = [vp, C.POINTER(A.ht_scene_info)]
    L.ht_set_option.argtypes = [C.c_char_p, C.c_int]
    L.ht_debug_timeline.argtypes = [vp, A._pu32, C.c_int32]
    L.ht_frame_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.ht_frame_destroy.argtypes = [vp]
    L.ht_frame_destroy.restype = None
    L.ht_frame_clear.argtypes = [vp]
    L.ht_frame_accumulate.argtypes = [vp, C.POINTER(A.ht_camera), C.POINTER(A.ht_render_params),
                                      C.POINTER(A.ht_stats)]
    L.ht_frame_read.argtypes = [vp, A._pf32, A._pu32]
    L.ht_frame_samples.argtypes = [vp]
    L.ht_frame_samples.restype = C.c_int64
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise HipTraceError(rc, load().ht_last_error().decode("utf-8", "replace"))


def version() -> int:
    return load().ht_version()


def device_count() -> int:
    return load().ht_device_count()


def set_option(name: str, value: int):
    _check(load().ht_set_option(name.encode(), int(value)))


def make_params(width, height, samples, sample_offset=0, max_depth=50, jitter=1,
                precision=A.HT_PRECISION_FP32, batch=0, seed=0, device=-1,
                collect_counters=0, tile_first=0, tile_stride=0,
                shard_mode=A.HT_SHARD_SAMPLES) -> A.ht_render_params:
    p = A.ht_render_params()
    p.width, p.height, p.samples, p.sample_offset = width, height, samples, sample_offset
    p.max_depth, p.jitter, p.precision, p.batch = max_depth, jitter, precision, batch
    p.seed, p.device, p.collect_counters = seed, device, collect_counters
    p.tile_first, p.tile_stride, p.shard_mode = tile_first, tile_stride, shard_mode
    return p


def shard_tile_pixels(width: int, height: int, n_shards: int) -> int:
    """Pixels in one shard's tile buffer when a frame's 8x8 tiles are dealt out to `n_shards` shards."""
    return int(load().ht_shard_tile_pixels(width, height, n_shards))


class Scene:
    """A device-resident scene (ht_scene) made from a scene.FlatScene."""

    def __init__(self, flat, device: int = -1):
        self.flat = flat
        self._h = C.c_void_p()
        _check(load().ht_scene_create(C.byref(flat.desc), device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            load().ht_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def info(self) -> dict:
        i = A.ht_scene_info()
        _check(load().ht_scene_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in A.ht_scene_info._fields_}

    def debug_timeline(self) -> np.ndarray:
        """ht_debug_timeline: [3, 4096] counts per 10.24 us bin (units fetched, waves that found the
        queue empty, waves that ended) of the launches made while the option "timeline" was on."""
        out = np.zeros((3, 4096), dtype=np.uint32)
        _check(load().ht_debug_timeline(self._h, out.ctypes.data_as(A._pu32), out.size))
        return out

    def cancel(self):
        _check(load().ht_cancel(self._h))

    def render(self, camera, width, height, samples, want_stats=True, **kw):
        """ht_render: returns dict(rgb=[H,W,3] float32 (row 0 = bottom),
        packed=[H,W] uint32 (row 0 = top), stats=dict)."""
        p = make_params(width, height, samples, **kw)
        rgb = np.zeros((height, width, 3), dtype=np.float32)
        packed = np.zeros((height, width), dtype=np.uint32)
        st = A.ht_stats()
        cc = camera.to_c()
        _check(load().ht_render(self._h, C.byref(cc), C.byref(p), rgb.ctypes.data_as(A._pf32),
                                packed.ctypes.data_as(A._pu32), C.byref(st)))
        return {"rgb": rgb, "packed": packed, "stats": st.as_dict()}

    def render_noalloc(self, camera, params: A.ht_render_params, rgb_ptr: int, packed_ptr: int, stream: int = 0):
        """ht_render_noalloc: the frame of ht_render into DEVICE buffers (addresses as int,
        0 = skip), enqueued on `stream`; the sums live in the library's static array."""
        cc = camera.to_c()
        _check(load().ht_render_noalloc(self._h, C.byref(cc), C.byref(params), C.c_void_p(rgb_ptr),
                                        C.c_void_p(packed_ptr), C.c_void_p(stream)))

    def render_accumulate(self, camera, params: A.ht_render_params, accum_ptr: int, stream: int = 0,
                          want_stats: bool = True) -> Optional[dict]:
        """ht_render_accumulate into a DEVICE buffer (address as int)."""
        st = A.ht_stats()
        cc = camera.to_c()
        _check(load().ht_render_accumulate(self._h, C.byref(cc), C.byref(params), C.c_void_p(accum_ptr),
                                           C.c_void_p(stream), C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def resolve(self, accum_ptr: int, width: int, height: int, total_samples: int, stream: int = 0):
        rgb = np.zeros((height, width, 3), dtype=np.float32)
        packed = np.zeros((height, width), dtype=np.uint32)
        _check(load().ht_resolve(self._h, C.c_void_p(accum_ptr), width, height, total_samples,
                                 rgb.ctypes.data_as(A._pf32), packed.ctypes.data_as(A._pu32),
                                 C.c_void_p(stream)))
        return {"rgb": rgb, "packed": packed}

    def resolve_device(self, accum_ptr: int, width: int, height: int, total_samples: int,
                       rgb_ptr: int, packed_ptr: int, stream: int = 0):
        """ht_resolve_device: both outputs are DEVICE addresses (0 = skip)."""
        _check(load().ht_resolve_device(self._h, C.c_void_p(accum_ptr), width, height, total_samples,
                                        C.c_void_p(rgb_ptr), C.c_void_p(packed_ptr), C.c_void_p(stream)))

    def resolve_tiles_device(self, accum_ptr: int, width: int, height: int, total_samples: int, tile_first: int,
                             tile_stride: int, mean_tiles_ptr: int, packed_tiles_ptr: int, stream: int = 0):
        """ht_resolve_tiles_device: the tiles (tile_first, tile_stride) of the frame into compact tile
        buffers on the device (addresses as int, 0 = skip)."""
        _check(load().ht_resolve_tiles_device(self._h, C.c_void_p(accum_ptr), width, height, total_samples, tile_first,
                                              tile_stride, C.c_void_p(mean_tiles_ptr), C.c_void_p(packed_tiles_ptr),
                                              C.c_void_p(stream)))

    def assemble_tiles_device(self, width: int, height: int, n_shards: int, mean_tiles_all_ptr: int,
                              packed_tiles_all_ptr: int, rgb_ptr: int, packed_ptr: int, stream: int = 0):
        """ht_assemble_tiles_device: the frame from the gathered tile buffers of all shards."""
        _check(load().ht_assemble_tiles_device(self._h, width, height, n_shards, C.c_void_p(mean_tiles_all_ptr),
                                               C.c_void_p(packed_tiles_all_ptr), C.c_void_p(rgb_ptr),
                                               C.c_void_p(packed_ptr), C.c_void_p(stream)))

    def first_hit(self, camera, width, height, precision=A.HT_PRECISION_FP32):
        t = np.zeros((height, width), dtype=np.float32)
        pid = np.zeros((height, width), dtype=np.int32)
        n = np.zeros((height, width, 3), dtype=np.float32)
        t64 = np.zeros((height, width), dtype=np.float64)
        n64 = np.zeros((height, width, 3), dtype=np.float64)
        cc = camera.to_c()
        _check(load().ht_first_hit(self._h, C.byref(cc), width, height, precision,
                                   t.ctypes.data_as(A._pf32), pid.ctypes.data_as(A._pi32),
                                   n.ctypes.data_as(A._pf32), t64.ctypes.data_as(A._pf64),
                                   n64.ctypes.data_as(A._pf64)))
        return {"t": t, "prim_id": pid, "normal": n, "t64": t64, "normal64": n64}


def render_sharded(scenes, camera, width, height, samples, **kw) -> dict:
    """ht_render_sharded: one frame by several Scene handles (one per GPU of the node, or
    several on one GPU), each rendering its share of the samples per pixel."""
    p = make_params(width, height, samples, **kw)
    rgb = np.zeros((height, width, 3), dtype=np.float32)
    packed = np.zeros((height, width), dtype=np.uint32)
    st = A.ht_stats()
    cc = camera.to_c()
    handles = (C.c_void_p * len(scenes))(*[s.handle for s in scenes])
    _check(load().ht_render_sharded(handles, len(scenes), C.byref(cc), C.byref(p), rgb.ctypes.data_as(A._pf32),
                                    packed.ctypes.data_as(A._pu32), C.byref(st)))
    return {"rgb": rgb, "packed": packed, "stats": st.as_dict()}


class Frame:
    """A progressive frame (ht_frame): device-resident radiance sums that sample
    batches are added to and that can be read back after any of them."""

    def __init__(self, scene: Scene, width: int, height: int):
        self.scene, self.width, self.height = scene, width, height
        self._h = C.c_void_p()
        _check(load().ht_frame_create(scene.handle, width, height, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            load().ht_frame_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def clear(self):
        _check(load().ht_frame_clear(self._h))

    @property
    def samples(self) -> int:
        return int(load().ht_frame_samples(self._h))

    def accumulate(self, camera, samples: int, sample_offset: Optional[int] = None, **kw) -> dict:
        """Add `samples` paths per pixel; by default they continue where the frame stands."""
        if sample_offset is None:
            sample_offset = self.samples
        p = make_params(self.width, self.height, samples, sample_offset=sample_offset, **kw)
        st = A.ht_stats()
        cc = camera.to_c()
        _check(load().ht_frame_accumulate(self._h, C.byref(cc), C.byref(p), C.byref(st)))
        return st.as_dict()

    def read(self) -> dict:
        rgb = np.zeros((self.height, self.width, 3), dtype=np.float32)
        packed = np.zeros((self.height, self.width), dtype=np.uint32)
        _check(load().ht_frame_read(self._h, rgb.ctypes.data_as(A._pf32), packed.ctypes.data_as(A._pu32)))
        return {"rgb": rgb, "packed": packed}
