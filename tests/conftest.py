import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scenes_dir():
    return SCENES


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle


_cache = {}


def load_flat(name, perlin_seed=1):
    """(camera, world, FlatScene) of a scene fixture, cached per session."""
    from raytracing_amd.scene import flatten, load_scene

    key = (name, perlin_seed)
    if key not in _cache:
        cam, world = load_scene(os.path.join(SCENES, name + ".json"), perlin_seed=perlin_seed)
        _cache[key] = (cam, world, flatten(world))
    return _cache[key]


@pytest.fixture(scope="session")
def flat_scene():
    return load_flat


@pytest.fixture(scope="session")
def ht():
    """The C-ABI binding; GPU tests fail loudly if the library or device is missing."""
    from raytracing_amd import lib

    lib.load()
    return lib
