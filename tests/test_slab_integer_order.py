"""The fp32 node step orders a box's entry distances as the signed integers of their bit patterns and
leaves out the clamp to t_min (raytracing_amd/csrc/kernels.hip, slab2).  Checked here on the CPU, in
numpy's fp32, against the test with the clamp that AABB.hit amounts to (AABB.scala:5-19: the ray
interval [t_min, t_max] is narrowed axis by axis and the box is hit while it stays non-empty):
the integer form takes every box the clamped form takes, and no box that lies behind the ray."""
import numpy as np


def _both_forms(near, far, t_min, t_far):
    """near, far: [n, 3] float32 slab distances per axis (near <= far); returns (clamped, integer)."""
    tn = np.maximum(near.max(axis=1), np.float32(t_min))
    tf = np.minimum(far.min(axis=1), np.float32(t_far))
    clamped = tn <= tf
    n_int = near.view(np.int32).max(axis=1)
    integer = n_int <= tf.view(np.int32)
    return clamped, integer


def test_integer_order_takes_what_the_clamped_test_takes():
    rng = np.random.default_rng(7)
    n = 400_000
    # slab distances of every sign and size, some exactly zero or minus zero
    a = (rng.standard_normal((n, 3)) * 10.0 ** rng.integers(-3, 4, (n, 3))).astype(np.float32)
    b = (rng.standard_normal((n, 3)) * 10.0 ** rng.integers(-3, 4, (n, 3))).astype(np.float32)
    z = rng.random((n, 3))
    a[z < 0.01] = 0.0
    a[(z >= 0.01) & (z < 0.02)] = -0.0
    near, far = np.minimum(a, b), np.maximum(a, b)
    t_far = np.where(rng.random(n) < 0.3, np.float32(np.inf), np.abs(rng.standard_normal(n)).astype(np.float32) * 20)
    clamped, integer = _both_forms(near, far, 0.001, t_far.astype(np.float32))
    assert clamped.sum() > n // 20 and (~clamped).sum() > n // 20
    assert not (clamped & ~integer).any()  # never drops a box the reference's test enters
    # what it takes in addition ends before t_min or has no thickness: a visit that finds no hit
    extra = integer & ~clamped
    tf = np.minimum(far.min(axis=1), t_far.astype(np.float32))
    assert extra.any() and (tf[extra] < np.float32(0.001)).all()


def test_integer_order_leaves_out_boxes_behind_the_ray():
    rng = np.random.default_rng(8)
    n = 200_000
    near = -(np.abs(rng.standard_normal((n, 3))).astype(np.float32) * 5 + 1)
    far = near + np.abs(rng.standard_normal((n, 3))).astype(np.float32)
    # at least one axis whose slab ends behind the origin, the others anywhere
    k = rng.integers(0, 3, n)
    far[np.arange(n), k] = np.minimum(far[np.arange(n), k], np.float32(-1e-3))
    other = rng.random((n, 3)) < 0.5
    other[np.arange(n), k] = False
    far[other] = np.abs(far[other]) + 1  # those slabs contain the origin
    assert (near <= far).all()
    _, integer = _both_forms(near, far, 0.001, np.full(n, np.inf, dtype=np.float32))
    thick = (far - near).min(axis=1) > 0
    assert not integer[thick].any()
