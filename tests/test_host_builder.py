"""The scene builder (raytracing_amd/csrc/host_scene.cpp) on the CPU, under AddressSanitizer
and UBSan: random worlds of every mix of kinds go through build_host_scene, and a checker
walks the result: every primitive reached exactly once through the hoist list or the
hierarchy, leaves of two hold neighbours of one kind (never a big sphere), prim_ref finds
every primitive again after the renumbering, child boxes contain what lies below them, the
recorded depth is the walked depth, big spheres come first, and the on-chip image holds the
narrowed child references."""
import os
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")


@pytest.fixture(scope="module")
def checker():
    subprocess.check_call(["make", "-C", CSRC, "build_check"], stdout=subprocess.DEVNULL)
    return os.path.join(CSRC, "build_check")


CASES = [(0, 9, 0), (9, 0, 0), (0, 0, 9), (3, 7, 2), (1, 0, 0), (1, 1, 0), (3, 3, 2), (2, 2, 6), (17, 23, 2), (40, 41, 5),
         (485, 0, 0), (2, 871, 0), (0, 1300, 0), (0, 2500, 0), (300, 6000, 20), (5000, 100, 0), (4097, 0, 0)]


@pytest.mark.parametrize("n_sph,n_tri,n_rect", CASES)
@pytest.mark.parametrize("force_hbm", [0, 1])
def test_builder_invariants(checker, n_sph, n_tri, n_rect, force_hbm):
    for seed in (1, 2):
        r = subprocess.run([checker, str(n_sph), str(n_tri), str(n_rect), str(seed), str(force_hbm)],
                           capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr[-3000:]
        fields = dict(kv.split("=") for kv in r.stdout.split()[1:])
        n = n_sph + n_tri + n_rect
        assert int(fields["prims"]) == n
        if force_hbm:
            assert fields["wide32"] == "1" and fields["wide64"] == "1"
        if n > 8:
            assert int(fields["nodes"]) > 0 and int(fields["hoisted"]) <= 8
            if n_rect == 0 and n >= 40:  # leaves of two: clearly fewer nodes than one leaf per primitive needs
                assert int(fields["nodes"]) < 0.75 * n and int(fields["pairs"]) > 0.3 * n
        else:
            assert int(fields["nodes"]) == 0 and int(fields["hoisted"]) == n


def test_store_follows_the_size(checker):
    """Up to 4096 primitives of a kind and an image that fits LDS: on chip; beyond: HBM."""
    def run(*a):
        r = subprocess.run([checker, *map(str, a)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        return dict(kv.split("=") for kv in r.stdout.split()[1:])
    assert run(0, 1300, 0, 3)["wide32"] == "0"
    assert run(0, 2500, 0, 3)["wide32"] == "1"   # ~84 bytes per triangle with its share of the nodes
    assert run(4097, 0, 0, 3)["wide32"] == "1"   # more than 4096 of a kind
    f = run(1500, 0, 0, 3)
    assert f["wide32"] == "0" and f["wide64"] == "1"  # the fp64 image is twice the size


@pytest.mark.parametrize("n_sph,n_tri,n_rect", [(40, 41, 5), (2, 871, 0), (3000, 70000, 10), (1, 1, 9)])
def test_hierarchy_from_another_builder_is_checked_and_used(checker, n_sph, n_tri, n_rect):
    """BuildOptions::external_bvh (the device builder's hook) with CPU stand-ins: a sound
    hierarchy is converted (leaves of two included) and passes every check of the walk; one
    that loses a primitive, or is deeper than the traversal stack, is refused and the host
    builds its own."""
    def run(mode):
        r = subprocess.run([checker, str(n_sph), str(n_tri), str(n_rect), "5", "0", str(mode)], capture_output=True,
                           text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr[-3000:]
        return dict(kv.split("=") for kv in r.stdout.split()[1:])
    own, good, lossy = run(0), run(1), run(2)
    assert own["external"] == "0" and good["external"] == "1"
    assert lossy["external"] == "0" and lossy["nodes"] == own["nodes"] and lossy["depth"] == own["depth"]
    if 60 < n_sph + n_tri + n_rect < 2000:  # a chain over that many primitives is deeper than any stack
        chain = run(3)                      # (the stand-in recurses once per level: small scenes only)
        assert chain["external"] == "0" and chain["nodes"] == own["nodes"]


def test_work_queue_hands_every_sample_out_once(checker):
    """work_queue.h on the CPU: the division by multiplication against `/` (small, large and random
    divisors, dividends at the multiples and the ends of the range), and for 600 launches of random
    size, tile shard, sample count and unit sizes every (own tile, sample) numbered exactly once by
    the one queue of the on-chip store and by the bands' queues of the HBM store."""
    for seed in (1, 2):
        r = subprocess.run([checker, "queue", str(seed)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and r.stdout.startswith("ok queue launches=600"), r.stdout + r.stderr[-3000:]
