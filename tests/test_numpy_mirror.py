"""The C oracle against the independent numpy restatement (tests/numpy_first_hit.py) on the
reference's nine scenes: two restatements written separately from the Scala must agree."""
import numpy as np
import pytest

from numpy_first_hit import first_hit

SCENES = ["scene", "dragon", "cornell", "random", "perlin", "checker", "light", "plane", "triangles"]


@pytest.mark.parametrize("name", SCENES)
def test_oracle_first_hit_equals_numpy_restatement(oracle_mod, flat_scene, name):
    cam, _, flat = flat_scene(name)
    W, H = (96, 54) if name in ("random", "dragon") else (160, 90)
    t, pid, n, tie = first_hit(cam, flat.prim_type, flat.prim_data, W, H)
    o = oracle_mod.OracleScene(flat, bvh_seed=1).first_hit(cam, W, H)
    assert tie.mean() < 0.01
    ok = ~tie
    assert np.array_equal(pid[ok], o["prim_id"][ok])
    hit = ok & (pid >= 0)
    assert hit.sum() > (100 if name == "triangles" else 0.2 * W * H)  # triangles.json is mostly sky
    assert np.all(np.isinf(o["t"][ok & (pid < 0)]))
    np.testing.assert_allclose(o["t"][hit], t[hit], rtol=1e-12)
    np.testing.assert_allclose(o["normal"][hit], n[hit], rtol=1e-9, atol=1e-12)
