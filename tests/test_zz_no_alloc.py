"""ht_render_noalloc (the entry point that allocates nothing) against ht_render, through
the C ABI.  Both run the same kernels on the same Philox-keyed paths and resolve the same
integer sums, so the pictures must be equal bit for bit.  Two calls in a row on the same
GPU share the library's static scratch array; the second, smaller frame must not see what
the first one left there."""
import numpy as np
import pytest

from hipmem import RawBuffer
from raytracing_amd import _abi as A

pytestmark = pytest.mark.gpu


def _frame(ht, scene, cam, W, H, spp, seed, **kw):
    rgb = RawBuffer(W * H * 3 * 4, fill=0xEE)
    packed = RawBuffer(W * H * 4, fill=0xEE)
    p = ht.make_params(W, H, spp, seed=seed, **kw)
    scene.render_noalloc(cam, p, rgb.address, packed.address, 0)
    return rgb.to_host(np.float32, (H, W, 3)), packed.to_host(np.uint32, (H, W))


@pytest.mark.parametrize("name", ["dragon", "scene"])
def test_noalloc_equals_render_twice_in_a_row(ht, flat_scene, name):
    assert ht.device_count() >= 1, "GPU tests need a HIP device; libhiptrace has no CPU path"
    cam, _, flat = flat_scene(name)
    with ht.Scene(flat, device=0) as s:
        # first call: several launches (batch 3 of 7 samples), ragged picture size
        rgb, packed = _frame(ht, s, cam, 203, 117, 7, seed=11, batch=3)
        ref = s.render(cam, 203, 117, 7, seed=11)
        assert np.array_equal(packed, ref["packed"])
        assert np.array_equal(rgb, ref["rgb"])
        # second call, smaller: the sums the first call left behind must not leak in
        rgb2, packed2 = _frame(ht, s, cam, 64, 36, 3, seed=12)
        ref2 = s.render(cam, 64, 36, 3, seed=12)
        assert np.array_equal(packed2, ref2["packed"])
        assert np.array_equal(rgb2, ref2["rgb"])
        # and the first frame again, fp64 arithmetic this time
        rgb3, packed3 = _frame(ht, s, cam, 96, 54, 2, seed=13, precision=A.HT_PRECISION_FP64)
        ref3 = s.render(cam, 96, 54, 2, seed=13, precision=A.HT_PRECISION_FP64)
        assert np.array_equal(packed3, ref3["packed"]) and np.array_equal(rgb3, ref3["rgb"])


def test_noalloc_limits(ht, flat_scene):
    cam, _, flat = flat_scene("plane")
    with ht.Scene(flat, device=0) as s:
        # one pixel more than the static array holds
        W = 2048
        H = A.HT_NOALLOC_MAX_PIXELS // W + 1
        with pytest.raises(ht.HipTraceError) as e:
            s.render_noalloc(cam, ht.make_params(W, H, 1), 0, 0, 0)
        assert e.value.code == A.HT_E_TOO_LARGE and "static scratch" in str(e.value)
        with pytest.raises(ht.HipTraceError) as e:
            s.render_noalloc(cam, ht.make_params(64, 36, 1, collect_counters=1), 0, 0, 0)
        assert e.value.code == A.HT_E_INVALID
        # the largest frame it accepts: 2048 x 1024 = HT_NOALLOC_MAX_PIXELS exactly
        H = A.HT_NOALLOC_MAX_PIXELS // W
        packed = RawBuffer(W * H * 4)
        s.render_noalloc(cam, ht.make_params(W, H, 1, seed=4), 0, packed.address, 0)
        ref = s.render(cam, W, H, 1, seed=4)
        assert np.array_equal(packed.to_host(np.uint32, (H, W)), ref["packed"])


def test_noalloc_allocates_nothing_and_can_be_captured(ht, flat_scene):
    """What the entry point exists for: it takes no device memory (free bytes of the device are
    the same before and after, also right after an option changed the launch shape it had
    cached) and it only enqueues work, so a HIP graph can capture it (global capture mode: a
    hipMalloc or a wait on the device inside the call would fail the capture) and replaying
    the graph gives ht_render's picture."""
    import ctypes as C

    from hipmem import hip

    h = hip()
    h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    h.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    h.hipStreamDestroy.argtypes = [C.c_void_p]
    h.hipStreamSynchronize.argtypes = [C.c_void_p]
    h.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    h.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    h.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    h.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    h.hipGraphExecDestroy.argtypes = [C.c_void_p]
    h.hipGraphDestroy.argtypes = [C.c_void_p]

    def free_bytes():
        assert h.hipDeviceSynchronize() == 0
        free, total = C.c_size_t(), C.c_size_t()
        assert h.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    cam, _, flat = flat_scene("dragon")
    W, H, spp = 160, 90, 5
    with ht.Scene(flat, device=0) as s:
        ref = s.render(cam, W, H, spp, seed=31)
        rgb = RawBuffer(W * H * 12, fill=0xEE)
        packed = RawBuffer(W * H * 4, fill=0xEE)
        stream = C.c_void_p()
        assert h.hipStreamCreate(C.byref(stream)) == 0
        p = ht.make_params(W, H, spp, seed=31, batch=2)
        try:
            s.render_noalloc(cam, p, rgb.address, packed.address, stream.value)  # warm-up: code objects loaded
            assert h.hipStreamSynchronize(stream) == 0
            before = free_bytes()
            s.render_noalloc(cam, p, rgb.address, packed.address, stream.value)
            ht.set_option("max_blocks_per_cu", 7)  # the cached launch shape is worked out again
            try:
                s.render_noalloc(cam, p, rgb.address, packed.address, stream.value)
            finally:
                ht.set_option("max_blocks_per_cu", 8)
            s.render_noalloc(cam, p, rgb.address, packed.address, stream.value)
            assert free_bytes() == before, "ht_render_noalloc changed the device's free memory"
            assert np.array_equal(packed.to_host(np.uint32, (H, W)), ref["packed"])

            # capture one call into a graph, replay it twice into cleared buffers
            assert h.hipStreamBeginCapture(stream, 0) == 0  # hipStreamCaptureModeGlobal
            try:
                s.render_noalloc(cam, p, rgb.address, packed.address, stream.value)
            finally:
                graph = C.c_void_p()
                rc = h.hipStreamEndCapture(stream, C.byref(graph))
            assert rc == 0 and graph.value, "the call could not be captured"
            exe = C.c_void_p()
            assert h.hipGraphInstantiate(C.byref(exe), graph, None, None, 0) == 0
            for _ in range(2):
                h.hipMemset(packed.ptr, 0xEE, W * H * 4)
                h.hipMemset(rgb.ptr, 0xEE, W * H * 12)
                assert h.hipGraphLaunch(exe, stream) == 0
                assert h.hipStreamSynchronize(stream) == 0
                assert np.array_equal(packed.to_host(np.uint32, (H, W)), ref["packed"])
                assert np.array_equal(rgb.to_host(np.float32, (H, W, 3)), ref["rgb"])
            assert h.hipGraphExecDestroy(exe) == 0
            assert h.hipGraphDestroy(graph) == 0
        finally:
            h.hipStreamDestroy(stream)
