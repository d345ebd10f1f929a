"""Tile sharding on the device: the frame's 8x8 tiles dealt out round-robin to G shards,
every sample of a tile on its owner (ht_render_params.tile_first / tile_stride), the shards'
own tiles resolved into compact tile buffers and put together (ht_resolve_tiles_device,
ht_assemble_tiles_device), and ht_render_sharded's HT_SHARD_TILES mode.  The reference's pixels
are independent sums (jvm/src/main/scala/main.scala:79-91) and the device sums are integers,
so every comparison here is `==`."""
import numpy as np
import pytest

from hipmem import DeviceBuffer, RawBuffer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(ht):
    assert ht.device_count() >= 1, "GPU tests need a HIP device (libhiptrace has no CPU path)"
    return ht


def tile_owner_mask(W, H, first, stride):
    """[H, W] bool (row 0 = bottom): pixels inside the tiles first, first + stride, ..."""
    j, i = np.mgrid[0:H, 0:W]
    tile = (j // 8) * ((W + 7) // 8) + i // 8
    return tile % stride == first


@pytest.mark.parametrize("name,W,H,G", [("dragon", 100, 52, 3), ("cornell", 64, 64, 8), ("scene", 61, 35, 2)])
def test_tile_shards_partition_the_frame(dev, flat_scene, name, W, H, G):
    """The G tile shards' sum buffers are disjoint and add up to the unsharded buffer, bit for
    bit; the statistics count the paths of the shard's own pixels."""
    cam, _, flat = flat_scene(name)
    spp = 5
    with dev.Scene(flat, device=0) as s:
        whole = DeviceBuffer(W * H * 3)
        s.render_accumulate(cam, dev.make_params(W, H, spp, seed=9), whole.address)
        want = whole.to_host().reshape(H, W, 3)
        total = np.zeros_like(want)
        for g in range(G):
            buf = DeviceBuffer(W * H * 3)
            st = s.render_accumulate(cam, dev.make_params(W, H, spp, seed=9, tile_first=g, tile_stride=G), buf.address)
            got = buf.to_host().reshape(H, W, 3)
            mine = tile_owner_mask(W, H, g, G)
            assert not got[~mine].any(), "a shard wrote outside its own tiles"
            np.testing.assert_array_equal(got[mine], want[mine])
            assert st["samples"] == int(mine.sum()) * spp and st["samples_per_pixel"] == spp
            total += got
            buf.free()
        np.testing.assert_array_equal(total, want)
        whole.free()


@pytest.mark.parametrize("W,H,G", [(100, 52, 3), (64, 40, 8), (33, 17, 2), (16, 8, 5)])
def test_resolved_tiles_assemble_to_the_resolved_frame(dev, flat_scene, W, H, G):
    """Every shard resolves its own tiles into a tile buffer; the buffers laid end to end (what a
    gather leaves on the root) assemble into exactly the pictures ht_resolve_device makes of the
    whole sum buffer, also with ragged edge tiles and more shards than tiles."""
    cam, _, flat = flat_scene("dragon")
    spp = 4
    px = dev.shard_tile_pixels(W, H, G)
    n_tiles = ((W + 7) // 8) * ((H + 7) // 8)
    assert px == -(-n_tiles // G) * 64
    with dev.Scene(flat, device=0) as s:
        whole = DeviceBuffer(W * H * 3)
        s.render_accumulate(cam, dev.make_params(W, H, spp, seed=4), whole.address)
        rgb = RawBuffer(W * H * 12)
        packed = RawBuffer(W * H * 4)
        s.resolve_device(whole.address, W, H, spp, rgb.address, packed.address)
        want_rgb, want_packed = rgb.to_host(np.float32, (H, W, 3)), packed.to_host(np.uint32, (H, W))
        mean_all, packed_all = RawBuffer(px * G * 12, fill=0xAB), RawBuffer(px * G * 4, fill=0xAB)
        for g in range(G):
            buf = DeviceBuffer(W * H * 3)
            s.render_accumulate(cam, dev.make_params(W, H, spp, seed=4, tile_first=g, tile_stride=G), buf.address)
            s.resolve_tiles_device(buf.address, W, H, spp, g, G, mean_all.address + g * px * 12,
                                   packed_all.address + g * px * 4)
            buf.free()
        rgb2, packed2 = RawBuffer(W * H * 12, fill=0xCD), RawBuffer(W * H * 4, fill=0xCD)
        s.assemble_tiles_device(W, H, G, mean_all.address, packed_all.address, rgb2.address, packed2.address)
        np.testing.assert_array_equal(packed2.to_host(np.uint32, (H, W)), want_packed)
        np.testing.assert_array_equal(rgb2.to_host(np.float32, (H, W, 3)), want_rgb)
        # a NULL input skips its output
        packed3 = RawBuffer(W * H * 4, fill=0xCD)
        s.assemble_tiles_device(W, H, G, 0, packed_all.address, 0, packed3.address)
        np.testing.assert_array_equal(packed3.to_host(np.uint32, (H, W)), want_packed)
        for b in (whole, rgb, packed, mean_all, packed_all, rgb2, packed2, packed3):
            b.free()


@pytest.mark.parametrize("n", [2, 3, 5])
def test_sharded_render_by_tiles_equals_one_render(dev, flat_scene, n):
    """ht_render_sharded with HT_SHARD_TILES: n handles (here on one device; one per GPU on a
    node) render disjoint tiles; the frame is ht_render's, bit for bit, in both host pictures."""
    from raytracing_amd import _abi as A

    cam, _, flat = flat_scene("dragon")
    W, H = 128, 72
    scenes = [dev.Scene(flat, device=0) for _ in range(n)]
    try:
        for spp in (7, 2):
            whole = scenes[0].render(cam, W, H, spp, seed=21)
            got = dev.render_sharded(scenes, cam, W, H, spp, seed=21, shard_mode=A.HT_SHARD_TILES)
            np.testing.assert_array_equal(got["packed"], whole["packed"])
            np.testing.assert_array_equal(got["rgb"], whole["rgb"])
            assert got["stats"]["samples"] == W * H * spp
    finally:
        for s in scenes:
            s.close()


def test_tile_parameter_errors(dev, flat_scene):
    cam, _, flat = flat_scene("scene")
    with dev.Scene(flat, device=0) as a, dev.Scene(flat, device=0) as b:
        buf = DeviceBuffer(32 * 18 * 3)
        for first, stride in ((2, 2), (-1, 4), (1, 0), (0, -3)):
            with pytest.raises(dev.HipTraceError) as e:
                a.render_accumulate(cam, dev.make_params(32, 18, 1, tile_first=first, tile_stride=stride), buf.address)
            assert e.value.code == -1
        with pytest.raises(dev.HipTraceError) as e:
            dev.render_sharded([a, b], cam, 32, 18, 2, tile_first=1, tile_stride=2)
        assert "tile_stride" in str(e.value)
        with pytest.raises(dev.HipTraceError):
            dev.render_sharded([a, b], cam, 32, 18, 2, shard_mode=7)
        buf.free()
