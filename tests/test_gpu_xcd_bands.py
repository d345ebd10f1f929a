"""The banded work queue of the HBM store (option "xcd_bands"): a launch's tiles in eight bands with
a queue head each, a wave serving the band of the XCD it runs on first.  It orders the work only:
every (tile, sample) must still be rendered exactly once, so the integer sums (the reference's
pixels are independent sums, jvm/src/main/scala/main.scala:79-91) must equal those of the single
queue bit for bit, for whole frames, tile shards, ragged edges and frames with fewer tiles than the
bands need."""
import numpy as np
import pytest

from hipmem import DeviceBuffer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(ht):
    assert ht.device_count() >= 1, "GPU tests need a HIP device (libhiptrace has no CPU path)"
    ht.set_option("force_hbm_scene", 1)
    yield ht
    ht.set_option("force_hbm_scene", 0)
    ht.set_option("xcd_bands", 1)
    ht.set_option("timeline", 0)
    ht.set_option("chunk", 0)


def sums(dev, s, cam, W, H, spp, **kw):
    buf = DeviceBuffer(W * H * 3)
    s.render_accumulate(cam, dev.make_params(W, H, spp, seed=5, **kw), buf.address)
    out = buf.to_host().reshape(H, W, 3)
    buf.free()
    return out


@pytest.mark.parametrize("W,H,spp,first,stride", [
    (320, 180, 24, 0, 0),    # 920 tiles: eight bands of 115
    (323, 181, 9, 0, 0),     # ragged edge tiles, a tile count that eight does not divide
    (320, 180, 16, 1, 2),    # tile shards: the bands are cut from the shard's own tiles
    (320, 180, 16, 2, 3),
    (64, 56, 40, 0, 0),      # 56 tiles: fewer than the bands need, one queue
    (128, 64, 300, 0, 0),    # 128 tiles, many steps: the waves' first units alone exhaust no band
])
def test_bands_change_no_sum(dev, flat_scene, W, H, spp, first, stride):
    cam, _, flat = flat_scene("dragon")
    with dev.Scene(flat, device=0) as s:
        assert s.info()["hbm_store_fp32"] == 1
        dev.set_option("xcd_bands", 0)
        want = sums(dev, s, cam, W, H, spp, tile_first=first, tile_stride=stride)
        dev.set_option("xcd_bands", 1)
        got = sums(dev, s, cam, W, H, spp, tile_first=first, tile_stride=stride)
        np.testing.assert_array_equal(got, want)
        assert want.any()


def test_every_unit_is_fetched_once(dev, flat_scene):
    """Units fetched by the waves (profiling option "timeline"): with the bands exactly the units of
    the launch, as with one queue."""
    cam, _, flat = flat_scene("dragon")
    W, H, spp = 320, 180, 24
    with dev.Scene(flat, device=0) as s:
        n = {}
        dev.set_option("chunk", 8)  # (left to itself a launch this small takes smaller units)
        for bands in (0, 1):
            dev.set_option("xcd_bands", bands)
            dev.set_option("timeline", 1)
            sums(dev, s, cam, W, H, spp)
            n[bands] = int(s.debug_timeline()[0].sum())
            dev.set_option("timeline", 0)
        dev.set_option("chunk", 0)
        assert n[0] == n[1] == 40 * 23 * (2 + 2)  # 16 samples in units of 8, the last 8 in units of 4
