"""ctypes binding of oracle/liboracle.so — the fp64 CPU restatement of the
reference's hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (raytracing_amd) never
imports this module.  PARITY UNPINNED (see oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

OR_SAMPLER_REFERENCE, OR_SAMPLER_MATCHED = 0, 1
OR_WORLD_REFERENCE_BVH, OR_WORLD_LINEAR = 0, 1


class or_counters(C.Structure):
    _fields_ = [("n_rays", C.c_int64), ("n_box", C.c_int64), ("n_sphere", C.c_int64),
                ("n_rect", C.c_int64), ("n_tri", C.c_int64), ("n_shade", C.c_int64 * 4)]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("oracle.c", "oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        pd, pi, pu = A._pf64, A._pi32, A._pu32
        L.or_scene_create.argtypes = [C.POINTER(A.ht_scene_desc), C.c_uint64, C.POINTER(C.c_void_p)]
        L.or_scene_destroy.argtypes = [C.c_void_p]
        L.or_scene_destroy.restype = None
        L.or_first_hit.argtypes = [C.c_void_p, C.POINTER(A.ht_camera), C.c_int32, C.c_int32, C.c_int,
                                   C.c_double, C.c_double, pd, pi, pd, pd]
        L.or_first_hit_fp32_rays.argtypes = L.or_first_hit.argtypes
        L.or_render.argtypes = [C.c_void_p, C.POINTER(A.ht_camera), C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                pd, pu, C.POINTER(or_counters)]
        L.or_path_radiance.argtypes = [C.c_void_p, C.POINTER(A.ht_camera), C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int,
                                       C.c_int, C.c_uint64, pd]
        L.or_philox4x32_10.argtypes = [pu, pu, pu]
        L.or_philox4x32_10.restype = None
        L.or_prim_hit.argtypes = [C.c_void_p, C.c_int32, pd, pd, C.c_double, C.c_double, pd, pd, pd,
                                  C.POINTER(C.c_int)]
        L.or_texture.argtypes = [C.c_void_p, C.c_int32, pd, pd, pd]
        L.or_texture.restype = None
        L.or_perlin_noise.argtypes = [C.c_void_p, C.c_int32, pd]
        L.or_perlin_noise.restype = C.c_double
        L.or_perlin_turb.argtypes = [C.c_void_p, C.c_int32, pd]
        L.or_perlin_turb.restype = C.c_double
        L.or_scatter_matched.argtypes = [C.c_void_p, C.c_int32, pd, pd, C.c_double, pd, pd, pd, pd, pd, pd]
        L.or_to_rgb.argtypes = [pd]
        L.or_to_rgb.restype = C.c_uint32
        L.or_camera_basis.argtypes = [C.POINTER(A.ht_camera), pd]
        L.or_camera_basis.restype = None
        L.or_bvh_stats.argtypes = [C.c_void_p, pi, pi]
        L.or_bvh_stats.restype = None
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pd(a):
    return a.ctypes.data_as(A._pf64)


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().or_philox4x32_10(c.ctypes.data_as(A._pu32), k.ctypes.data_as(A._pu32), o.ctypes.data_as(A._pu32))
    return o


def to_rgb(rgb):
    a = _d(rgb)
    return int(lib().or_to_rgb(_pd(a)))


def camera_basis(camera):
    out = np.zeros(22)
    cc = camera.to_c()
    lib().or_camera_basis(C.byref(cc), _pd(out))
    names = ["origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"]
    d = {n: out[i * 3:i * 3 + 3].copy() for i, n in enumerate(names)}
    d["lens_radius"] = float(out[21])
    return d


class OracleScene:
    """An or_scene built from a raytracing_amd.scene.FlatScene."""

    def __init__(self, flat, bvh_seed: int = 0):
        self.flat = flat
        self._h = C.c_void_p()
        rc = lib().or_scene_create(C.byref(flat.desc), bvh_seed, C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"or_scene_create failed: {rc}")

    def close(self):
        if self._h:
            lib().or_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh_stats(self):
        n, d = C.c_int32(), C.c_int32()
        lib().or_bvh_stats(self._h, C.byref(n), C.byref(d))
        return n.value, d.value

    def first_hit(self, camera, W, H, world_mode=OR_WORLD_REFERENCE_BVH, off=(0.5, 0.5), fp32_rays=False):
        """fp32_rays: round every camera ray (origin, direction) to fp32 before the fp64 trace."""
        t = np.zeros((H, W))
        pid = np.zeros((H, W), dtype=np.int32)
        n = np.zeros((H, W, 3))
        uv = np.zeros((H, W, 2))
        cc = camera.to_c()
        fn = lib().or_first_hit_fp32_rays if fp32_rays else lib().or_first_hit
        rc = fn(self._h, C.byref(cc), W, H, world_mode, off[0], off[1], _pd(t),
                pid.ctypes.data_as(A._pi32), _pd(n), _pd(uv))
        if rc != 0:
            raise RuntimeError(f"or_first_hit failed: {rc}")
        return {"t": t, "prim_id": pid, "normal": n, "uv": uv}

    def render(self, camera, W, H, samples, sample_offset=0, max_depth=50, jitter=1,
               sampler=OR_SAMPLER_MATCHED, world_mode=OR_WORLD_REFERENCE_BVH, seed=0, nthreads=0,
               counters=False):
        rgb = np.zeros((H, W, 3))
        packed = np.zeros((H, W), dtype=np.uint32)
        cc = camera.to_c()
        ctr = or_counters()
        rc = lib().or_render(self._h, C.byref(cc), W, H, samples, sample_offset, max_depth, jitter,
                             sampler, world_mode, seed, nthreads, _pd(rgb),
                             packed.ctypes.data_as(A._pu32), C.byref(ctr) if counters else None)
        if rc != 0:
            raise RuntimeError(f"or_render failed: {rc}")
        out = {"rgb": rgb, "packed": packed}
        if counters:
            out["counters"] = {k: (list(getattr(ctr, k)) if k == "n_shade" else getattr(ctr, k))
                               for k, _ in or_counters._fields_}
        return out

    def path_radiance(self, camera, W, H, i, j, sample, max_depth=50, jitter=1,
                      sampler=OR_SAMPLER_MATCHED, world_mode=OR_WORLD_REFERENCE_BVH, seed=0):
        out = np.zeros(3)
        cc = camera.to_c()
        rc = lib().or_path_radiance(self._h, C.byref(cc), W, H, i, j, sample, max_depth, jitter,
                                    sampler, world_mode, seed, _pd(out))
        if rc != 0:
            raise RuntimeError(f"or_path_radiance failed: {rc}")
        return out

    def prim_hit(self, prim, origin, direction, t_min=0.001, t_max=float("inf")):
        o, d = _d(origin), _d(direction)
        t = C.c_double()
        n, uv = np.zeros(3), np.zeros(2)
        ff = C.c_int()
        hit = lib().or_prim_hit(self._h, prim, _pd(o), _pd(d), t_min, t_max, C.byref(t), _pd(n), _pd(uv),
                                C.byref(ff))
        if not hit:
            return None
        return {"t": t.value, "normal": n, "uv": uv, "front_face": bool(ff.value)}

    def texture(self, tex, uv, p):
        u, pp, out = _d(uv), _d(p), np.zeros(3)
        lib().or_texture(self._h, tex, _pd(u), _pd(pp), _pd(out))
        return out

    def perlin_noise(self, k, p):
        pp = _d(p)
        return float(lib().or_perlin_noise(self._h, k, _pd(pp)))

    def perlin_turb(self, k, p):
        pp = _d(p)
        return float(lib().or_perlin_turb(self._h, k, _pd(pp)))

    def scatter_matched(self, material, ray_o, ray_d, t, outward_normal, uv, u):
        o, d, n, uvv, uu = _d(ray_o), _d(ray_d), _d(outward_normal), _d(uv), _d(u)
        do, at, em = np.zeros(3), np.zeros(3), np.zeros(3)
        did = lib().or_scatter_matched(self._h, material, _pd(o), _pd(d), t, _pd(n), _pd(uvv), _pd(uu),
                                       _pd(do), _pd(at), _pd(em))
        return bool(did), do, at, em
