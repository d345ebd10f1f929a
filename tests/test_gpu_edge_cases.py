"""Edge cases of the device path against the oracle: ragged picture sizes, the
depth limit, sample offsets near the top of the range, degenerate geometry, a
scene built to mix every primitive, material and texture kind in one hierarchy."""
import numpy as np
import pytest

from raytracing_amd import _abi as A
from raytracing_amd.scene import (Camera, Checker, Diffuse, Emission, Glass, Glossy, Perlin, SolidColor, Sphere,
                                  Triangle, XYRect, XZRect, YZRect, flatten)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(ht):
    assert ht.device_count() >= 1, "GPU tests need a HIP device; libhiptrace has no CPU path"
    return ht


def close(dev_rgb, ref_rgb, frac=0.99, tol=2e-3):
    d = np.abs(dev_rgb.astype(np.float64) - ref_rgb).max(axis=-1)
    return (d < tol).mean() >= frac


@pytest.mark.parametrize("W,H", [(2, 2), (3, 5), (9, 7), (8, 8), (17, 64), (65, 3)])
def test_ragged_picture_sizes(dev, oracle_mod, flat_scene, W, H):
    """Tiles are 8x8: pictures that do not fill their last tiles, down to the
    smallest the reference's W-1 / H-1 divisors allow."""
    cam, _, flat = flat_scene("scene")
    o = oracle_mod.OracleScene(flat).render(cam, W, H, 3, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=4)
    with dev.Scene(flat, device=0) as s:
        g64 = s.render(cam, W, H, 3, seed=4, precision=A.HT_PRECISION_FP64)
        g32 = s.render(cam, W, H, 3, seed=4)
        fh = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
    assert np.abs(g64["rgb"] - o["rgb"]).max() <= 3e-6
    assert close(g32["rgb"], o["rgb"], frac=0.95)
    ofh = oracle_mod.OracleScene(flat).first_hit(cam, W, H)
    assert np.array_equal(fh["prim_id"], ofh["prim_id"])
    assert g32["stats"]["samples"] == W * H * 3


@pytest.mark.parametrize("depth", [1, 2, 3, 50, 200])
def test_depth_limit(dev, oracle_mod, flat_scene, depth):
    """Camera.ray_color's depth argument (Camera.scala:25-27): a path contributes
    nothing once `depth` rays have been traced; the closed Cornell box has the
    long paths that show it."""
    cam, _, flat = flat_scene("cornell")
    W, H, spp = 48, 48, 4
    o = oracle_mod.OracleScene(flat).render(cam, W, H, spp, max_depth=depth, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=8)
    with dev.Scene(flat, device=0) as s:
        g = s.render(cam, W, H, spp, seed=8, max_depth=depth, precision=A.HT_PRECISION_FP64)
        g32 = s.render(cam, W, H, spp, seed=8, max_depth=depth)
    assert np.abs(g["rgb"] - o["rgb"]).max() <= 3e-6 * max(1.0, float(o["rgb"].max()))
    assert close(g32["rgb"], o["rgb"], frac=0.97, tol=2e-3 * max(1.0, float(o["rgb"].max())))
    if depth == 1:
        # only what the camera ray itself hits emits: everything but the light is black
        assert (g["rgb"].max(axis=-1) > 0).mean() < 0.1


def test_sample_offsets_up_to_the_top_of_the_range(dev, oracle_mod, flat_scene):
    """Global sample indices are 31-bit: a shard that starts near the top renders the
    same paths the oracle renders for those indices."""
    cam, _, flat = flat_scene("scene")
    W, H = 40, 24
    first = 2 ** 31 - 1 - 5
    o = oracle_mod.OracleScene(flat).render(cam, W, H, 5, sample_offset=first, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=2)
    with dev.Scene(flat, device=0) as s:
        g = s.render(cam, W, H, 5, sample_offset=first, seed=2, precision=A.HT_PRECISION_FP64)
        h = s.render(cam, W, H, 5, sample_offset=0, seed=2, precision=A.HT_PRECISION_FP64)
    assert np.abs(g["rgb"] - o["rgb"]).max() <= 3e-6
    assert np.abs(g["rgb"] - h["rgb"]).max() > 1e-3  # other samples, other picture


def mixed_world():
    """Every primitive kind inside one hierarchy (more than 8 primitives, so nothing but
    the floor is tested linearly), every material, Checker over Perlin and solid colour."""
    grey = Diffuse(SolidColor((0.7, 0.7, 0.7)))
    ranvec, perm = Perlin.tables(3)
    marble = Diffuse(Checker(3.0, Perlin(2.0, ranvec, perm), SolidColor((0.2, 0.3, 0.9))))
    lamp = Emission(SolidColor((1.0, 0.9, 0.8)), 6.0)
    world = [Sphere((0.0, -1000.0, 0.0), 1000.0, marble)]
    for i in range(5):
        world.append(Sphere((-4.0 + 2.0 * i, 0.6, 0.5 * i - 1.0), 0.6,
                            [grey, Glass(SolidColor((1, 1, 1)), 1.5), Glossy(SolidColor((0.9, 0.6, 0.3)), 0.2),
                             Glossy(SolidColor((0.8, 0.8, 0.8)), 0.0), marble][i]))
    world += [
        XYRect(-3.0, 3.0, 0.0, 2.5, -3.0, Glossy(SolidColor((0.9, 0.9, 0.9)), 0.05)),
        XZRect(-1.0, 1.0, -1.0, 1.0, 4.0, lamp),
        YZRect(0.0, 2.0, -2.0, 2.0, 5.0, grey),
        Triangle(((-2.0, 0.0, 2.0), (0.0, 0.0, 2.5), (-1.0, 1.8, 2.2)), grey),
        Triangle(((1.0, 0.0, 2.0), (3.0, 0.0, 2.0), (2.0, 1.5, 1.5)), Glass(SolidColor((0.9, 1.0, 0.9)), 1.3)),
        # degenerate input the reference accepts: a triangle of zero area (never hit: det = 0)
        Triangle(((0.0, 1.0, 0.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)), grey),
    ]
    cam = Camera((0.0, 2.5, 9.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 40.0, 16 / 9, 0.05, -1.0, (0.05, 0.06, 0.1))
    return cam, world


def test_mixed_scene_first_hit_and_render(dev, oracle_mod):
    cam, world = mixed_world()
    flat = flatten(world)
    W, H = 192, 108
    osc = oracle_mod.OracleScene(flat, bvh_seed=3)
    ofh = osc.first_hit(cam, W, H)
    o = osc.render(cam, 96, 54, 6, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=12)
    for hbm in (0, 1):
        dev.set_option("force_hbm_scene", hbm)
        try:
            with dev.Scene(flat, device=0) as s:
                assert s.info()["n_nodes"] > 0 and s.info()["hbm_store_fp32"] == hbm
                fh = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
                g64 = s.render(cam, 96, 54, 6, seed=12, precision=A.HT_PRECISION_FP64)
                g32 = s.render(cam, 96, 54, 6, seed=12)
        finally:
            dev.set_option("force_hbm_scene", 0)
        assert np.array_equal(fh["prim_id"], ofh["prim_id"])
        assert len(np.unique(ofh["prim_id"])) >= 7             # most of the objects are in view
        assert 11 not in np.unique(fh["prim_id"])                # the zero-area triangle is never hit
        hit = ofh["prim_id"] >= 0
        np.testing.assert_allclose(fh["t64"][hit], ofh["t"][hit], rtol=1e-9)
        scale = max(1.0, float(o["rgb"].max()))
        assert np.abs(g64["rgb"] - o["rgb"]).max() <= 3e-6 * scale
        assert close(g32["rgb"], o["rgb"], frac=0.98, tol=2e-3 * scale)


def test_single_primitive_worlds(dev, oracle_mod):
    """A world of one object: the reference builds BVH(only, only) (BVH.scala:30)."""
    cam = Camera((0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.5, 0.0, -1.0, (0.5, 0.7, 1.0))
    for prim in (Sphere((0.0, 0.0, 0.0), 1.0, Diffuse(SolidColor((0.8, 0.3, 0.3)))),
                 Triangle(((-1.0, -1.0, 0.0), (1.0, -1.0, 0.0), (0.0, 1.0, 0.0)), Diffuse(SolidColor((0.3, 0.8, 0.3)))),
                 XYRect(1.0, -1.0, -1.0, 1.0, 0.0, Emission(SolidColor((1.0, 1.0, 1.0)), 2.0))):  # sides swapped
        flat = flatten([prim])
        o = oracle_mod.OracleScene(flat).render(cam, 60, 40, 4, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=1)
        ofh = oracle_mod.OracleScene(flat).first_hit(cam, 60, 40)
        with dev.Scene(flat, device=0) as s:
            g = s.render(cam, 60, 40, 4, seed=1, precision=A.HT_PRECISION_FP64)
            fh = s.first_hit(cam, 60, 40, precision=A.HT_PRECISION_FP64)
        assert np.array_equal(fh["prim_id"], ofh["prim_id"])
        assert np.abs(g["rgb"] - o["rgb"]).max() <= 3e-6 * max(1.0, float(o["rgb"].max()))


def test_axis_parallel_rays(dev, oracle_mod):
    """Rays with direction components that are exactly zero (1/d = inf in the box test):
    a camera looking straight down an axis, odd picture sizes so that the middle column
    and row have d.x = 0 and d.y = 0.  First hits equal the oracle's in both scene stores."""
    from raytracing_amd.meshgen import bumpy_ellipsoid
    from raytracing_amd.scene import Mesh
    grey = Diffuse(SolidColor((0.7, 0.7, 0.7)))
    world = [Sphere((0.0, -1001.5, 0.0), 1000.0, grey), Mesh(bumpy_ellipsoid(12, 24), grey)]
    world += [Sphere((0.5 * i - 2.0, 0.0, 1.5), 0.2, grey) for i in range(9)]  # centres on the plane y = 0
    flat = flatten(world)
    for look_from, vup in (((0.0, 0.0, 6.0), (0.0, 1.0, 0.0)), ((6.0, 0.0, 0.0), (0.0, 1.0, 0.0)),
                           ((0.0, 6.0, 0.0), (0.0, 0.0, 1.0))):
        cam = Camera(look_from, (0.0, 0.0, 0.0), vup, 40.0, 1.0, 0.0, -1.0, (0.5, 0.7, 1.0))
        W = H = 65
        ofh = oracle_mod.OracleScene(flat).first_hit(cam, W, H)
        o = oracle_mod.OracleScene(flat).render(cam, W, H, 2, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=6)
        for hbm in (0, 1):
            dev.set_option("force_hbm_scene", hbm)
            try:
                with dev.Scene(flat, device=0) as s:
                    fh = s.first_hit(cam, W, H)
                    fh64 = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
                    g64 = s.render(cam, W, H, 2, seed=6, precision=A.HT_PRECISION_FP64)
            finally:
                dev.set_option("force_hbm_scene", 0)
            assert np.array_equal(fh64["prim_id"], ofh["prim_id"])
            # fp32 may differ from fp64 on silhouettes, never on the axis-parallel middle row and column
            assert np.array_equal(fh["prim_id"][H // 2, :], ofh["prim_id"][H // 2, :])
            assert np.array_equal(fh["prim_id"][:, W // 2], ofh["prim_id"][:, W // 2])
            assert (fh["prim_id"] != ofh["prim_id"]).mean() < 0.01
            assert np.abs(g64["rgb"] - o["rgb"]).max() <= 3e-6
        assert len(np.unique(ofh["prim_id"][H // 2, :])) >= 3  # the middle row sees several objects


def test_scene_far_from_the_origin(dev, oracle_mod):
    """Boxes whose centre is large against their half-extent (the fp32 nodes store both):
    the whole scene and the camera moved 6000 units away.  The fp64 path still equals the
    oracle; the fp32 path sees what fp32 coordinates allow, the same in both scene stores."""
    from raytracing_amd.meshgen import bumpy_ellipsoid
    from raytracing_amd.scene import Mesh
    off = np.array([3000.0, -2000.0, 5000.0])
    grey = Diffuse(SolidColor((0.7, 0.7, 0.7)))
    world = [Sphere(tuple(np.array([0.0, -1001.5, 0.0]) + off), 1000.0, grey), Mesh(bumpy_ellipsoid(16, 32) + off, grey)]
    world += [Sphere(tuple(np.array([0.6 * i - 2.4, 0.3, 1.8]) + off), 0.25, grey) for i in range(9)]
    flat = flatten(world)
    cam = Camera(tuple(np.array([2.0, 2.0, 7.0]) + off), tuple(off), (0.0, 1.0, 0.0), 35.0, 1.5, 0.0, -1.0, (0.5, 0.7, 1.0))
    W, H = 120, 80
    ofh = oracle_mod.OracleScene(flat).first_hit(cam, W, H)
    got = []
    for hbm in (0, 1):
        dev.set_option("force_hbm_scene", hbm)
        try:
            with dev.Scene(flat, device=0) as s:
                fh64 = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
                got.append(s.first_hit(cam, W, H)["prim_id"])
        finally:
            dev.set_option("force_hbm_scene", 0)
        assert np.array_equal(fh64["prim_id"], ofh["prim_id"])
    assert np.array_equal(got[0], got[1])                      # the two stores agree in fp32
    assert (got[0] != ofh["prim_id"]).mean() < 0.03            # fp32 coordinates at 6000: silhouettes only
    assert len(np.unique(ofh["prim_id"])) > 50


@pytest.mark.parametrize("n_sph,n_tri,n_rect", [(9, 0, 0), (0, 9, 0), (3, 7, 0), (17, 23, 2), (2, 2, 6), (40, 41, 0)])
def test_leaves_of_two_primitives(dev, oracle_mod, n_sph, n_tri, n_rect):
    """The hierarchy puts two triangles, or two ordinary spheres, that end up alone under a
    node into one leaf and renumbers the primitives so that they are neighbours; odd counts,
    mixed kinds and a big sphere among the small ones must all keep their ids: the closest
    hit of every pixel is the oracle's (which knows no such leaves), in both stores."""
    rng = np.random.default_rng(n_sph * 1000 + n_tri * 10 + n_rect)
    grey = Diffuse(SolidColor((0.7, 0.7, 0.7)))
    world = [Sphere((0.0, -1000.5, 0.0), 1000.0, grey)]  # hoisted, and first in the sphere array
    eye = np.array((0.0, 1.0, 12.0))
    for _ in range(n_sph):
        world.append(Sphere(tuple(rng.uniform(-3, 3, 3) * (1, 0.5, 1) + (0, 1.5, 0)), float(rng.uniform(0.1, 0.5)), grey))
    for _ in range(n_tri):
        a = rng.uniform(-3, 3, 3) * (1, 0.5, 1) + (0, 1.5, 0)
        b, c = a + rng.uniform(-0.8, 0.8, 3), a + rng.uniform(-0.8, 0.8, 3)
        if np.dot(np.cross(b - a, c - a), eye - a) < 0:  # triangles are one-sided (Hittable.scala:102): face the camera
            b, c = c, b
        world.append(Triangle((tuple(a), tuple(b), tuple(c)), grey))
    for k in range(n_rect):
        c = rng.uniform(-2, 2, 3) * (1, 0.5, 1) + (0, 1.5, 0)
        R = (XYRect, XZRect, YZRect)[k % 3]
        world.append(R(c[0] - 0.4, c[0] + 0.4, c[1] - 0.3, c[1] + 0.3, float(c[2]), grey))
    rng.shuffle(world)
    flat = flatten(world)
    cam = Camera(tuple(eye), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 35.0, 1.5, 0.0, -1.0, (0.5, 0.7, 1.0))
    W, H = 150, 100
    ofh = oracle_mod.OracleScene(flat, bvh_seed=2).first_hit(cam, W, H)
    assert len(np.unique(ofh["prim_id"])) > (n_sph + n_tri + n_rect) // 2
    for hbm in (0, 1):
        dev.set_option("force_hbm_scene", hbm)
        try:
            with dev.Scene(flat, device=0) as s:
                info = s.info()
                assert 0 < info["n_nodes"] < flat.n_prims - 1  # fewer nodes than one leaf per primitive takes
                fh = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
                g32 = s.first_hit(cam, W, H)
        finally:
            dev.set_option("force_hbm_scene", 0)
        assert np.array_equal(fh["prim_id"], ofh["prim_id"])
        hit = ofh["prim_id"] >= 0
        np.testing.assert_allclose(fh["t64"][hit], ofh["t"][hit], rtol=1e-9)
        assert (g32["prim_id"] == ofh["prim_id"]).mean() > 0.995  # fp32: silhouettes may go either way


@pytest.mark.parametrize("name", ["dragon", "cornell", "random"])
def test_kernel_held_to_80_scalar_registers_gives_the_same_sums(dev, flat_scene, name):
    """Scenes that have room for 32 waves per CU run a copy of the kernel compiled with a cap
    on its scalar registers; with the cap on resident waves at 28 the library keeps the plain
    one.  Same source, same arithmetic: the frames are equal, and the launch shapes are not."""
    cam, _, flat = flat_scene(name)
    W, H, spp = 203, 117, 6
    with dev.Scene(flat, device=0) as s:
        a = s.render(cam, W, H, spp, seed=9)
        dev.set_option("max_waves_per_cu", 28)
        try:
            b = s.render(cam, W, H, spp, seed=9)
        finally:
            dev.set_option("max_waves_per_cu", 0)
    wa = a["stats"]["block_threads"] * a["stats"]["blocks_per_cu"]
    wb = b["stats"]["block_threads"] * b["stats"]["blocks_per_cu"]
    print(f"{name}: {wa // 64} against {wb // 64} waves per CU")
    assert wb <= 28 * 64 and (wa == 32 * 64) == (name != "random")  # random.json's image leaves no room
    np.testing.assert_array_equal(a["rgb"], b["rgb"])


@pytest.mark.parametrize("name", ["cornell", "scene", "perlin", "checker"])
def test_kernels_without_a_traversal_phase_give_the_same_sums(dev, flat_scene, name):
    """A scene without a hierarchy (all its primitives are tested when a ray starts) runs a kernel
    variant compiled without the traversal loop, the stacks and the box-test terms of a ray.  The
    general variant renders the same paths; the sums may differ only where the compiler placed a
    fused multiply-add differently in the two (a path that grazes an edge)."""
    cam, _, flat = flat_scene(name)
    W, H, spp = 203, 117, 6
    with dev.Scene(flat, device=0) as s:
        assert s.info()["n_nodes"] == 0
        a = s.render(cam, W, H, spp, seed=9)
        dev.set_option("flat_kernels", 0)
        try:
            b = s.render(cam, W, H, spp, seed=9)
        finally:
            dev.set_option("flat_kernels", 1)
    ka, kb = a["stats"]["kernel"], b["stats"]["kernel"]
    print(f"{name}: {ka} against {kb}")
    assert ka != kb and any(tag in ka for tag in ("<16u", "<17u", "<18u", ", 19u"))
    differ = (a["rgb"] != b["rgb"]).any(axis=-1).mean()
    print(f"{name}: {differ:.2e} of the pixels differ")
    assert differ < (0.25 if name == "perlin" else 2e-3)  # Perlin: a hit point one ulp away reads the noise elsewhere
    np.testing.assert_allclose(a["rgb"], b["rgb"], rtol=0, atol=2e-2 if name != "perlin" else 5e-2)
