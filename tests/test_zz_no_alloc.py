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
