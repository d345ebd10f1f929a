"""Minimal device-buffer helper for the GPU tests: hipMalloc / hipMemcpy through
ctypes on the HIP runtime that libhiptrace.so already loaded (no torch needed)."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        from raytracing_amd import lib

        lib.load()  # pulls in libamdhip64
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip


class RawBuffer:
    """A device allocation of `nbytes` bytes filled with `fill`, read back as any dtype."""

    def __init__(self, nbytes: int, fill: int = 0):
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        assert hip().hipMalloc(C.byref(self.ptr), nbytes) == 0
        assert hip().hipMemset(self.ptr, fill, nbytes) == 0
        assert hip().hipDeviceSynchronize() == 0

    @property
    def address(self) -> int:
        return self.ptr.value

    def to_host(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        assert hip().hipDeviceSynchronize() == 0
        assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2) == 0  # D2H
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """A zero-initialised int64 device array (radiance sums, 32.32 fixed point)."""

    def __init__(self, n: int):
        self.n = n
        self.ptr = C.c_void_p()
        assert hip().hipMalloc(C.byref(self.ptr), n * 8) == 0
        assert hip().hipMemset(self.ptr, 0, n * 8) == 0
        assert hip().hipDeviceSynchronize() == 0

    @property
    def address(self) -> int:
        return self.ptr.value

    def to_host(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.int64)
        assert hip().hipDeviceSynchronize() == 0
        assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.n * 8, 2) == 0  # D2H
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def free_device_bytes() -> int:
    """Free device memory as hipMemGetInfo reports it (after a device synchronisation)."""
    h = hip()
    h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    assert h.hipDeviceSynchronize() == 0
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert h.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return int(free.value)
