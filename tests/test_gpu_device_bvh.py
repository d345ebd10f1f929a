"""The hierarchy built on the device (option "gpu_bvh": Morton order, radix sort, Karras'
hierarchy, bottom-up boxes) against the host's SAH hierarchy: any correct hierarchy gives the
same closest hit, so the first-hit buffers of the two must agree with each other and with
the oracle, and the radiance sums must agree except where a path grazes an edge."""
import numpy as np
import pytest

from raytracing_amd import _abi as A
from raytracing_amd.meshgen import bumpy_ellipsoid
from raytracing_amd.scene import Diffuse, Glass, Mesh, SolidColor, Sphere, flatten

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev(ht):
    assert ht.device_count() >= 1
    yield ht
    ht.set_option("gpu_bvh", 0)
    ht.set_option("force_hbm_scene", 0)


@pytest.mark.parametrize("name", ["dragon", "random", "scene", "triangles", "cornell"])
def test_small_scenes_first_hit_equal_host_hierarchy_and_oracle(dev, oracle_mod, flat_scene, name):
    cam, _, flat = flat_scene(name)
    W, H = 200, 112
    o = oracle_mod.OracleScene(flat, bvh_seed=1).first_hit(cam, W, H)
    with dev.Scene(flat, device=0) as s:
        host_info = s.info()
        h64 = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
        h32 = s.first_hit(cam, W, H)
        him = s.render(cam, W, H, 4, seed=3)
    dev.set_option("gpu_bvh", 2)
    with dev.Scene(flat, device=0) as s:
        info = s.info()
        g64 = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
        g32 = s.first_hit(cam, W, H)
        gim = s.render(cam, W, H, 4, seed=3)
    if host_info["n_nodes"] > 1:  # a hierarchy exists (cornell: all primitives are tested linearly)
        assert (info["n_nodes"], info["bvh_depth"]) != (host_info["n_nodes"], host_info["bvh_depth"]), "the device hierarchy was not used"
    assert np.array_equal(g64["prim_id"], o["prim_id"])
    hit = o["prim_id"] >= 0
    np.testing.assert_allclose(g64["t64"][hit], o["t"][hit], rtol=1e-9)
    assert np.array_equal(g64["prim_id"], h64["prim_id"]) and np.array_equal(g64["t64"], h64["t64"])
    # fp32: the same primitive gives the same t whatever the hierarchy; ties between
    # neighbouring primitives may fall the other way
    same = g32["prim_id"] == h32["prim_id"]
    assert same.mean() > 0.999 and np.array_equal(g32["t"][same], h32["t"][same])
    d = np.abs(gim["rgb"] - him["rgb"]).max(axis=-1)
    assert (d > 1e-6).mean() < 2e-3  # paths are keyed, not scheduled: equal but for grazing rays


def test_large_mesh(dev, oracle_mod, flat_scene):
    cam, _, _ = flat_scene("dragon")
    red = Diffuse(SolidColor((0.8, 0.1, 0.1)))
    world = [Sphere((0.0, -1000.0, 0.0), 1000.0, Diffuse(SolidColor((0.0, 0.8, 0.0)))), Mesh(bumpy_ellipsoid(72, 144), red),
             Sphere((1.6, 0.7, 1.2), 0.7, Glass(SolidColor((1.0, 1.0, 1.0)), 1.5))]
    flat = flatten(world)
    W, H = 160, 90
    o = oracle_mod.OracleScene(flat, bvh_seed=1).first_hit(cam, W, H)
    with dev.Scene(flat, device=0) as s:
        host_info = s.info()
    dev.set_option("gpu_bvh", 1000)
    with dev.Scene(flat, device=0) as s:
        info = s.info()
        g = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
        a = s.render(cam, 320, 180, 8, seed=2, batch=8)
        b = s.render(cam, 320, 180, 8, seed=2, batch=3)
    assert info["hbm_store_fp32"] == 1 and info["n_prims"] == flat.n_prims
    assert (info["n_nodes"], info["bvh_depth"]) != (host_info["n_nodes"], host_info["bvh_depth"])
    assert np.array_equal(g["prim_id"], o["prim_id"])
    hit = o["prim_id"] >= 0
    np.testing.assert_allclose(g["t64"][hit], o["t"][hit], rtol=1e-9)
    np.testing.assert_array_equal(a["rgb"], b["rgb"])
    print(f"20k-triangle mesh: host {host_info['n_nodes']} nodes, depth {host_info['bvh_depth']}; "
          f"device {info['n_nodes']} nodes, depth {info['bvh_depth']}")


def test_scene_below_the_threshold_is_built_on_the_host(dev, flat_scene):
    cam, _, flat = flat_scene("dragon")
    with dev.Scene(flat, device=0) as s:
        host_info = s.info()
    dev.set_option("gpu_bvh", 100000)
    with dev.Scene(flat, device=0) as s:
        info = s.info()
    assert (info["n_nodes"], info["bvh_depth"]) == (host_info["n_nodes"], host_info["bvh_depth"])
