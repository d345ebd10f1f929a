"""The wavefront variant (generate / trace / shade + compaction / reduce kernels, path
state in device memory) against the persistent kernel: both follow the same Philox-keyed
paths with the same arithmetic, so the integer radiance sums must be the same numbers.
(The two are separate instantiations of the shading code; where the compiler fuses a
multiply-add differently a path that grazes an edge may go the other way, as between the
two scene stores: a few sums in 10^5 at most.)"""
import numpy as np
import pytest

from hipmem import DeviceBuffer
from raytracing_amd import _abi as A

pytestmark = pytest.mark.gpu


def sums(ht, s, cam, W, H, spp, wavefront, **kw):
    acc = DeviceBuffer(H * W * 3)
    ht.set_option("wavefront", 1 if wavefront else 0)
    try:
        st = s.render_accumulate(cam, ht.make_params(W, H, spp, seed=31, device=0, **kw), acc.address, 0)
    finally:
        ht.set_option("wavefront", 0)
    return acc.to_host(), st


@pytest.mark.parametrize("hbm", [0, 1])
@pytest.mark.parametrize("name", ["dragon", "scene", "random", "cornell", "perlin", "checker"])
def test_wavefront_sums_equal_persistent_kernel(ht, flat_scene, name, hbm):
    cam, _, flat = flat_scene(name)
    W, H, spp = 200, 112, 5
    ht.set_option("force_hbm_scene", hbm)
    try:
        with ht.Scene(flat, device=0) as s:
            assert s.info()["hbm_store_fp32"] == hbm
            ref, _ = sums(ht, s, cam, W, H, spp, False)
            got, st = sums(ht, s, cam, W, H, spp, True)
            # few paths in flight: many rounds of generate / trace / shade with regeneration
            ht.set_option("wavefront_paths", 4096)
            try:
                small, _ = sums(ht, s, cam, W, H, spp, True)
            finally:
                ht.set_option("wavefront_paths", 1 << 24)
    finally:
        ht.set_option("force_hbm_scene", 0)
    assert st["samples"] == W * H * spp
    # The two are separate kernels with the same device functions inlined.  Where the compiler
    # fuses a multiply-add in one and not in the other, a hit point moves by an ulp: invisible
    # on a solid colour, 1e-7..1e-5 of the mean radiance through a Perlin texture (a fifth of
    # perlin.json's sums differ that way, median 1e-7), and now and then a path crosses a
    # checker boundary or a silhouette and goes another way altogether (90 of 67 200 sums on
    # checker.json, 2-8 on dragon, scene, random).  Built with -ffp-contract=on every sum of
    # every scene is equal.
    d = np.abs(got - ref).astype(np.float64) / 2 ** 32 / spp  # in units of the pixel's mean radiance
    print(f"{name} (hbm store {hbm}): {int((got != ref).sum())} of {ref.size} sums differ, "
          f"{int((d > 1e-4).sum())} by more than 1e-4 of mean radiance, max {d.max():.3g}")
    assert (d > 1e-4).mean() <= 2e-3
    assert np.abs(got - ref).max() <= 16 * 2 ** 32  # at most one path's weight (the Cornell light: 15)
    # the number of paths in flight changes nothing at all
    assert np.array_equal(small, got)


def test_wavefront_sample_ranges_add_up(ht, flat_scene):
    cam, _, flat = flat_scene("dragon")
    W, H = 160, 90
    with ht.Scene(flat, device=0) as s:
        whole, _ = sums(ht, s, cam, W, H, 6, True)
        a, _ = sums(ht, s, cam, W, H, 2, True)
        b, _ = sums(ht, s, cam, W, H, 4, True, sample_offset=2)
        c, _ = sums(ht, s, cam, W, H, 6, True, batch=1)
    assert np.array_equal(whole, a + b) and np.array_equal(whole, c)
