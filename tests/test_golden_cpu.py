"""The committed golden fixtures (tests/golden/*.npz, made by make_golden.py from the
oracle) are reproduced by the oracle: guards the checker itself against drift."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

SCENES = ["scene", "dragon", "cornell", "random", "perlin", "checker", "light", "plane", "triangles"]


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_golden(oracle_mod, flat_scene, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    fw, fh_, iw, ih, spp, seed = (int(x) for x in g["meta"])
    cam, _, flat = flat_scene(name)
    s = oracle_mod.OracleScene(flat, bvh_seed=1)
    fh = s.first_hit(cam, fw, fh_)
    assert np.array_equal(fh["prim_id"], g["fh_prim"].astype(np.int32))
    np.testing.assert_allclose(fh["t"], g["fh_t"], rtol=1e-12)
    np.testing.assert_allclose(fh["normal"], g["fh_normal"], rtol=1e-12, atol=1e-14)
    im = s.render(cam, iw, ih, spp, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=seed)
    np.testing.assert_allclose(im["rgb"], g["im_rgb"], rtol=2e-6, atol=1e-7)
    assert np.array_equal(im["packed"], g["im_packed"])


def test_golden_independent_of_reference_bvh_seed(oracle_mod, flat_scene):
    """The closest hit does not depend on which random hierarchy BVH.apply drew."""
    g = np.load(os.path.join(GOLDEN, "dragon.npz"))
    fw, fh_ = int(g["meta"][0]), int(g["meta"][1])
    cam, _, flat = flat_scene("dragon")
    for seed in (2, 99):
        fh = oracle_mod.OracleScene(flat, bvh_seed=seed).first_hit(cam, fw, fh_)
        assert np.array_equal(fh["prim_id"], g["fh_prim"].astype(np.int32))
        np.testing.assert_allclose(fh["t"], g["fh_t"], rtol=1e-12)
