"""Parity of the HIP path (through the C ABI) with the CPU oracle.  All tests here
need a real MI355X: run with -m gpu.

Tolerances
  * fp64 device path: same arithmetic as the oracle up to fused multiply-adds and
    libm differences -> 1e-9 relative on t, 1e-8 on normals (as vectors), 3e-6 on radiance.
  * fp32 device path (the production path): primitive ids exact wherever the oracle's
    own answer is stable under a ~1 ulp perturbation of the ray; t and normal within
    1e-5 relative, widened per pixel by 8x the oracle's sensitivity to that same
    perturbation (a ray stored in fp32 is only defined to ~1e-7 relative, and grazing
    hits amplify that).  The share of hit pixels inside the PLAIN 1e-5 is printed and
    asserted per scene: >= 99.9 % for t on every scene, >= 99.9 % for normals on every
    scene but random.json (>= 99.55 %, measured 99.61 %).  random.json is the one scene where
    fp32 rays cannot do better: its 0.2-radius spheres stand 10-25 units from the camera, so the
    2^-24 rounding of a ray direction moves the hit point by ~1.5e-6 and the normal by ~1e-5.
    Two floors are pinned on the CPU: the oracle fed exact camera rays rounded ONCE to fp32 and
    tracing them in fp64 leaves 0.11 % of that scene's normals outside 1e-5
    (tests/test_oracle_kat.py); rays MADE in fp32 as the device makes them (fp32 basis, two
    fused multiply-adds per component) and then traced exactly leave 0.35 %, p99.9 1.74e-5
    (tests/test_numpy_mirror.py).  The device leaves 0.39 %, p99.9 1.75e-5: its intersection
    and hit-point arithmetic add next to nothing to the error of its ray.
  * matched-stream renders (same Philox-keyed paths on both sides): >= 99 % of the
    pixels within 2e-3 absolute and PSNR >= 45 dB on the 8-bit toRGB codes; the rest
    are paths where fp32 rounding flipped a discrete decision.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from raytracing_amd import _abi as A

pytestmark = pytest.mark.gpu

SCENES = ["scene", "dragon", "cornell", "random", "perlin", "checker", "light", "plane", "triangles"]


@pytest.fixture(scope="module")
def dev(ht):
    assert ht.device_count() >= 1, "GPU tests need a HIP device; libhiptrace has no CPU path"
    return ht


_scene_cache = {}


def gpu_scene(dev, flat_scene, name):
    if name not in _scene_cache:
        cam, _, flat = flat_scene(name)
        _scene_cache[name] = (cam, flat, dev.Scene(flat, device=0))
    return _scene_cache[name]


def psnr_codes(a, b):
    ca = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.float64)
    cb = np.stack([(b >> 16) & 255, (b >> 8) & 255, b & 255], -1).astype(np.float64)
    mse = ((ca - cb) ** 2).mean()
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


# the frames of BASELINE.json's configurations, at their full sizes
FULL_SIZE = [("dragon", 1920, 1080), ("random", 1920, 1080), ("cornell", 1024, 1024), ("perlin", 1920, 1080),
             ("checker", 1920, 1080), ("scene", 400, 225)]


@pytest.mark.parametrize("name,W,H", [(n, 160, 90) for n in SCENES] + FULL_SIZE)
def test_first_hit_fp64_matches_oracle(dev, oracle_mod, flat_scene, name, W, H):
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    o = oracle_mod.OracleScene(flat, bvh_seed=1).first_hit(cam, W, H)
    g = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
    hit = o["prim_id"] >= 0
    # Same primitive everywhere, except on rays that graze an edge: a ray through the line where
    # two walls of the Cornell box meet lies on the inclusive edge of both rects (Hittable.scala:57-62),
    # which of the two the reference reports depends on its randomly built tree (BVH.scala:12-13, 19;
    # the oracle's answer changes with its tree seed there), and an arithmetic that rounds
    # differently from the JVM's (the device fuses multiply-adds) may find the point an ulp outside
    # of both.  Such a pixel is told by the oracle itself: its answer changes when the ray moves by
    # an ulp of its direction.
    differ = g["prim_id"] != o["prim_id"]
    if differ.any():
        osc = oracle_mod.OracleScene(flat, bvh_seed=1)
        d = 4e-16 * W
        unstable = np.zeros((H, W), bool)
        for dx, dy in ((d, 0), (-d, 0), (0, d), (0, -d), (d, d), (-d, -d), (d, -d), (-d, d)):
            unstable |= osc.first_hit(cam, W, H, off=(0.5 + dx, 0.5 + dy))["prim_id"] != o["prim_id"]
        unstable |= oracle_mod.OracleScene(flat, bvh_seed=2).first_hit(cam, W, H)["prim_id"] != o["prim_id"]
        assert not np.any(differ & ~unstable), "primitive id differs on a pixel whose oracle answer is stable"
        assert differ.mean() < 5e-4
        print(f"{name} {W}x{H}: {int(differ.sum())} pixels on rays that graze an edge (oracle unstable on {int(unstable.sum())})")
        hit = hit & ~differ
    assert np.all(np.isinf(g["t64"][(o["prim_id"] < 0) & ~differ]))
    np.testing.assert_allclose(g["t64"][hit], o["t"][hit], rtol=1e-9)
    # the oracle's own (ray.at(t) - center)/radius carries ~1e-10 of cancellation on small
    # far spheres, so normals are compared as vectors
    nerr = np.linalg.norm(g["normal64"][hit] - o["normal"][hit], axis=-1) / np.linalg.norm(o["normal"][hit], axis=-1)
    assert nerr.size == 0 or nerr.max() < 1e-8


@pytest.mark.parametrize("name,W,H", [(n, 320, 180) for n in SCENES] + FULL_SIZE)
def test_first_hit_fp32_matches_oracle(dev, oracle_mod, flat_scene, name, W, H):
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    osc = oracle_mod.OracleScene(flat, bvh_seed=1)
    o = osc.first_hit(cam, W, H)
    d = 2.5e-7 * W  # about one ulp of an fp32 ray direction, in pixels
    stable = np.ones((H, W), bool)
    sens_t = np.zeros((H, W))
    sens_n = np.zeros((H, W))
    hit = o["prim_id"] >= 0
    nlen = np.where(hit, np.linalg.norm(o["normal"], axis=-1), 1.0)
    for dx, dy in ((d, 0), (-d, 0), (0, d), (0, -d)):
        p = osc.first_hit(cam, W, H, off=(0.5 + dx, 0.5 + dy))
        stable &= p["prim_id"] == o["prim_id"]
        both = hit & (p["prim_id"] == o["prim_id"])
        sens_t[both] = np.maximum(sens_t[both], np.abs(p["t"][both] - o["t"][both]) / o["t"][both])
        sens_n[both] = np.maximum(sens_n[both],
                                  np.linalg.norm(p["normal"][both] - o["normal"][both], axis=-1) / nlen[both])
    g = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP32)
    mism = g["prim_id"] != o["prim_id"]
    assert not np.any(mism & stable), "primitive id differs on a pixel whose oracle answer is stable"
    assert mism.mean() < 1e-3
    ok = hit & ~mism
    terr = np.abs(g["t64"][ok] - o["t"][ok]) / o["t"][ok]
    nerr = np.linalg.norm(g["normal64"][ok] - o["normal"][ok], axis=-1) / nlen[ok]
    assert np.all(terr <= 1e-5 + 8 * sens_t[ok]), f"t error {terr.max():.3e}"
    assert np.all(nerr <= 1e-5 + 8 * sens_n[ok]), f"normal error {nerr.max():.3e}"
    if ok.any():
        t_in, n_in = float((terr <= 1e-5).mean()), float((nerr <= 1e-5).mean())
        print(f"{name} {W}x{H}: inside the plain 1e-5: t {t_in:.5f} (max {terr.max():.2e}), "
              f"normal {n_in:.5f} (max {nerr.max():.2e}, p99.9 {np.percentile(nerr, 99.9):.2e}); "
              f"ids differ on {int(mism.sum())} unstable pixels")
        assert t_in >= 0.999, f"{name}: {1 - t_in:.2e} of the hit pixels have t outside 1e-5"
        floor = 0.9955 if name == "random" else 0.999
        assert n_in >= floor, f"{name}: {1 - n_in:.2e} of the hit pixels have the normal outside 1e-5"
    # misses are +inf / -1 / zero normal
    miss = ~hit & ~mism
    assert np.all(np.isinf(g["t"][miss])) and np.all(g["normal"][miss] == 0)


@pytest.mark.parametrize("name", SCENES)
def test_golden_fixture_first_hit(dev, flat_scene, name):
    """Device vs the committed fixture (no oracle in the loop)."""
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    W, H = int(gold["meta"][0]), int(gold["meta"][1])
    g = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP64)
    assert np.array_equal(g["prim_id"], gold["fh_prim"].astype(np.int32))
    hit = gold["fh_prim"] >= 0
    np.testing.assert_allclose(g["t64"][hit], gold["fh_t"][hit], rtol=1e-9)
    g32 = s.first_hit(cam, W, H, precision=A.HT_PRECISION_FP32)
    assert (g32["prim_id"] == gold["fh_prim"]).mean() > 0.999
    same = hit & (g32["prim_id"] == gold["fh_prim"])
    terr = np.abs(g32["t64"][same] - gold["fh_t"][same]) / gold["fh_t"][same]
    assert np.percentile(terr, 99) < 1e-5


@pytest.mark.parametrize("name", SCENES)
def test_matched_render_fp64(dev, oracle_mod, flat_scene, name):
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    W, H, spp = 80, 45, 6
    o = oracle_mod.OracleScene(flat, bvh_seed=1).render(cam, W, H, spp, sampler=oracle_mod.OR_SAMPLER_MATCHED,
                                                        seed=77)
    g = s.render(cam, W, H, spp, precision=A.HT_PRECISION_FP64, seed=77)
    scale = max(1.0, float(o["rgb"].max()))
    assert np.abs(g["rgb"] - o["rgb"]).max() <= 3e-6 * scale
    # 8-bit codes: equal except where the fp32 accumulator lands on a truncation boundary
    assert psnr_codes(g["packed"], o["packed"]) > 48


@pytest.mark.parametrize("name,W,H", FULL_SIZE)
def test_matched_render_fp64_full_size(dev, oracle_mod, flat_scene, name, W, H):
    """The same paths at the full frame, two samples per pixel (4 million paths at 1920x1080): a path
    that grazes an edge may go another way on the device, which fuses multiply-adds where the JVM
    does not, so the bound is on the share of pixels: at least 99.99 % within 3e-6."""
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    spp = 2
    o = oracle_mod.OracleScene(flat, bvh_seed=1).render(cam, W, H, spp, sampler=oracle_mod.OR_SAMPLER_MATCHED, seed=77)
    g = s.render(cam, W, H, spp, precision=A.HT_PRECISION_FP64, seed=77)
    scale = max(1.0, float(o["rgb"].max()))
    d = np.abs(g["rgb"] - o["rgb"]).max(axis=-1)
    share = float((d <= 3e-6 * scale).mean())
    print(f"{name} {W}x{H}x{spp} fp64: {share:.6f} of the pixels within 3e-6, {int((d > 3e-6 * scale).sum())} outside")
    assert share >= 0.9999
    assert psnr_codes(g["packed"], o["packed"]) > 48


# matched paths at the full frames of BASELINE.json's configurations, eight samples per pixel each
# (the oracle follows about ten million paths per second on the GPU box's sixteen cores)
@pytest.mark.parametrize("name,W,H,spp", [(n, 160, 90, 8) for n in SCENES] + [(n, w, h, 8) for n, w, h in FULL_SIZE])
def test_matched_render_fp32(dev, oracle_mod, flat_scene, name, W, H, spp):
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    o = oracle_mod.OracleScene(flat, bvh_seed=1).render(cam, W, H, spp, sampler=oracle_mod.OR_SAMPLER_MATCHED,
                                                        seed=5)
    g = s.render(cam, W, H, spp, precision=A.HT_PRECISION_FP32, seed=5)
    d = np.abs(g["rgb"].astype(np.float64) - o["rgb"]).max(axis=-1)
    assert (d < 2e-3).mean() >= 0.99, f"only {(d < 2e-3).mean():.4f} of the pixels follow the oracle's paths"
    assert abs(float(g["rgb"].mean()) - float(o["rgb"].mean())) < 2e-3 * max(1.0, float(o["rgb"].mean()))
    assert psnr_codes(g["packed"], o["packed"]) >= 45.0


@pytest.mark.parametrize("name", ["scene", "cornell", "dragon"])
def test_golden_fixture_image(dev, flat_scene, name):
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    _, _, W, H, spp, seed = (int(x) for x in gold["meta"])
    g = s.render(cam, W, H, spp, precision=A.HT_PRECISION_FP64, seed=seed)
    scale = max(1.0, float(gold["im_rgb"].max()))
    assert np.abs(g["rgb"] - gold["im_rgb"]).max() <= 3e-6 * scale
    g32 = s.render(cam, W, H, spp, precision=A.HT_PRECISION_FP32, seed=seed)
    assert psnr_codes(g32["packed"], gold["im_packed"]) >= 40.0


def test_converged_image_vs_reference_sampler(dev, oracle_mod, flat_scene):
    """Independent noise: device (fp32, Philox, direct sampling) against the oracle
    using the reference's own rejection samplers.  The two 256-spp renders differ by
    Monte-Carlo noise only; the device must be no further from one oracle render than
    a second oracle render with another seed is."""
    cam, flat, s = gpu_scene(dev, flat_scene, "scene")
    W, H, spp = 64, 36, 256
    osc = oracle_mod.OracleScene(flat, bvh_seed=1)
    ref_a = osc.render(cam, W, H, spp, sampler=oracle_mod.OR_SAMPLER_REFERENCE, seed=1)["rgb"]
    ref_b = osc.render(cam, W, H, spp, sampler=oracle_mod.OR_SAMPLER_REFERENCE, seed=2)["rgb"]
    g = s.render(cam, W, H, spp, seed=3)["rgb"].astype(np.float64)
    noise = np.sqrt(((ref_a - ref_b) ** 2).mean())
    diff = np.sqrt(((ref_a - g) ** 2).mean())
    assert 0.75 < diff / noise < 1.33


def _codes_of(rgb):
    """Vec3.toRGB per channel (Vec3.scala:82-84) of linear means, as floats 0..254."""
    return np.floor(255.0 * np.clip(np.sqrt(np.maximum(rgb, 0.0)), 0.0, 0.999))


def _psnr(a, b):
    mse = ((a - b) ** 2).mean()
    return 99.0 if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


@pytest.mark.parametrize("name,W,H,spp,bin_", [("dragon", 160, 90, 320, 1), ("random", 160, 90, 640, 1),
                                                ("cornell", 128, 128, 1024, 8),
                                                # a ninth of the headline frames (the oracle follows ten million
                                                # paths a second on sixteen cores: about a minute for these three)
                                                ("dragon", 640, 360, 320, 1), ("random", 640, 360, 640, 1),
                                                ("cornell", 512, 512, 1024, 8)])
def test_converged_image_psnr_vs_reference_sampler(dev, oracle_mod, flat_scene, name, W, H, spp, bin_):
    """The converged-image gate on the headline scenes: device (fp32, Philox, direct
    sampling) against the oracle running the reference's own rejection samplers
    (main.scala:87-93, Vec3.scala:17-34) at matched spp, PSNR on the 8-bit toRGB codes.
    Two independent Monte-Carlo renders differ by noise, so the gate has two parts: the
    spp is high enough that two oracle renders with different seeds agree to >= 40 dB
    (the noise floor, asserted and printed), and the device is >= 40 dB from the oracle
    and no further from it than that floor minus 1 dB; a bias of about one 8-bit code
    anywhere in the picture would cost more than that.  The Cornell box (small light, no next-event
    estimation in the reference) has per-pixel noise far above 40 dB at any spp a CPU can
    render, so its pictures are compared after averaging 8x8 pixel blocks of linear
    radiance."""
    cam, flat, s = gpu_scene(dev, flat_scene, name)
    osc = oracle_mod.OracleScene(flat, bvh_seed=1)
    ref_a = osc.render(cam, W, H, spp, sampler=oracle_mod.OR_SAMPLER_REFERENCE, seed=1)["rgb"]
    ref_b = osc.render(cam, W, H, spp, sampler=oracle_mod.OR_SAMPLER_REFERENCE, seed=2)["rgb"]
    g = s.render(cam, W, H, spp, seed=3)["rgb"].astype(np.float64)

    def picture(rgb):
        if bin_ > 1:
            rgb = rgb.reshape(H // bin_, bin_, W // bin_, bin_, 3).mean(axis=(1, 3))
        return _codes_of(rgb)

    floor = _psnr(picture(ref_a), picture(ref_b))
    got = min(_psnr(picture(ref_a), picture(g)), _psnr(picture(ref_b), picture(g)))
    print(f"{name} {W}x{H} {spp} spp (bin {bin_}): oracle two-seed PSNR {floor:.2f} dB, device vs oracle {got:.2f} dB")
    assert floor >= 40.0, f"{name}: the oracle's own two-seed PSNR is {floor:.2f} dB; raise spp"
    assert got >= 40.0, f"{name}: device vs oracle {got:.2f} dB (noise floor {floor:.2f} dB)"
    assert got >= floor - 1.0, f"{name}: device vs oracle {got:.2f} dB, noise floor {floor:.2f} dB"


# ---- properties at the benchmark's full sizes ---------------------------------------------------
@pytest.mark.parametrize("name,W,H", [("dragon", 1920, 1080), ("random", 1920, 1080), ("cornell", 1024, 1024)])
def test_full_size_properties(dev, flat_scene, name, W, H):
    from hipmem import DeviceBuffer

    cam, flat, s = gpu_scene(dev, flat_scene, name)

    def sums(first, count, batch=0):
        acc = DeviceBuffer(H * W * 3)
        p = dev.make_params(W, H, count, sample_offset=first, seed=9, batch=batch, device=0)
        s.render_accumulate(cam, p, acc.address, 0)
        return acc

    whole_buf = sums(0, 6)
    whole = whole_buf.to_host()
    # radiance sums are integers (32.32 fixed point), so all of these are EXACT:
    # the same call twice
    assert np.array_equal(whole, sums(0, 6).to_host())
    # sample ranges are additive (what spp sharding over GPUs relies on)
    assert np.array_equal(whole, sums(0, 2).to_host() + sums(2, 4).to_host())
    # accumulating two calls into ONE buffer
    acc = sums(0, 2)
    p = dev.make_params(W, H, 4, sample_offset=2, seed=9, device=0)
    s.render_accumulate(cam, p, acc.address, 0)
    assert np.array_equal(whole, acc.to_host())
    # the launch batch size and the work-unit size do not change anything
    assert np.array_equal(whole, sums(0, 6, batch=1).to_host())
    dev.set_option("chunk", 3)
    try:
        assert np.array_equal(whole, sums(0, 6).to_host())
    finally:
        dev.set_option("chunk", 0)
    # radiance is bounded by the brightest source times the largest albedo chain (<= 1)
    one = 2.0 ** 32
    bright = 15.0 if name == "cornell" else 1.0
    assert int(whole.max()) <= 6 * bright * one * (1 + 1e-5) and int(whole.min()) >= 0
    # resolve: codes <= 254 per channel (Vec3.scala:82-84) and the picture is the row flip
    out = s.resolve(whole_buf.address, W, H, 6, 0)
    codes = out["packed"]
    assert int(((codes >> 16) & 255).max()) <= 254 and int((codes & 255).max()) <= 254
    mean = whole.reshape(H, W, 3).astype(np.float64) / one / 6
    np.testing.assert_allclose(out["rgb"], mean, rtol=1e-6, atol=1e-7)
    top_row = np.floor(255 * np.clip(np.sqrt(mean[H - 1]), 0, 0.999)).astype(np.uint32)
    got = np.stack([(codes[0] >> 16) & 255, (codes[0] >> 8) & 255, codes[0] & 255], -1)
    assert np.array_equal(got, top_row)


def test_jitter_off_render_equals_first_hit_shading(dev, oracle_mod, flat_scene):
    """plane.json, jitter off: every pixel is the texel under the pixel-centre ray."""
    cam, flat, s = gpu_scene(dev, flat_scene, "plane")
    o = oracle_mod.OracleScene(flat).render(cam, 64, 36, 1, jitter=0)
    g = s.render(cam, 64, 36, 1, jitter=0)
    np.testing.assert_allclose(g["rgb"], o["rgb"], atol=1e-6)
    assert np.array_equal(g["packed"], o["packed"])


def test_error_handling(dev, flat_scene):
    import ctypes as C

    cam, flat, s = gpu_scene(dev, flat_scene, "scene")
    with pytest.raises(dev.HipTraceError) as e:
        s.render(cam, 1, 10, 1)
    assert e.value.code == A.HT_E_INVALID and "W-1" in str(e.value)
    with pytest.raises(dev.HipTraceError):
        s.render(cam, 16, 16, 0)
    with pytest.raises(dev.HipTraceError):
        s.render(cam, 16, 16, 1, precision=7)
    # a material index out of range is rejected at scene creation
    bad = flat.prim_material.copy()
    bad[0] = 999
    flat.desc.prim_material = bad.ctypes.data_as(A._pi32)
    try:
        with pytest.raises(dev.HipTraceError) as e2:
            dev.Scene(flat)
        assert e2.value.code == A.HT_E_INVALID
    finally:  # (a pointer read from the struct is a view of the field, not a saved copy)
        flat.desc.prim_material = flat.prim_material.ctypes.data_as(A._pi32)
    h = C.c_void_p()
    assert dev.load().ht_scene_create(None, 0, C.byref(h)) == A.HT_E_INVALID
    assert dev.load().ht_scene_create(C.byref(flat.desc), 4096, C.byref(h)) == A.HT_E_NO_DEVICE


def test_cancel(dev, flat_scene):
    """shouldBreak of the reference stops the frame in flight only (main.scala:80)."""
    import threading

    cam, flat, s = gpu_scene(dev, flat_scene, "scene")
    # a request that arrives while nothing renders belongs to no frame: the next frame runs
    s.cancel()
    out = s.render(cam, 64, 36, 2)
    assert out["stats"]["samples"] == 64 * 36 * 2
    # a request from another thread stops a running frame at the next batch boundary, and
    # the frame holds exactly the samples it reports
    W, H, asked = 256, 144, 200000  # tens of seconds of work if nothing stopped it
    with dev.Frame(s, W, H) as f:
        timer = threading.Timer(0.05, s.cancel)
        timer.start()
        with pytest.raises(dev.HipTraceError) as e:
            f.accumulate(cam, asked, batch=1, seed=1)
        timer.join()
        assert e.value.code == A.HT_E_CANCELLED
        got = f.samples
        assert 0 < got < asked
        part = f.read()["packed"]
    np.testing.assert_array_equal(part, s.render(cam, W, H, got, seed=1)["packed"])
    # offsets past the 31-bit sample index are refused, not wrapped
    with pytest.raises(dev.HipTraceError) as e:
        s.render(cam, 16, 16, 2, sample_offset=2 ** 31 - 2)
    assert e.value.code == A.HT_E_INVALID


def test_render_frame_mirror(dev, flat_scene):
    """renderFrame(camera, world, img, samples, update, finish, shouldBreak) with the
    reference's argument meaning (main.scala:73-99)."""
    from raytracing_amd.render import render_frame

    cam, world, flat = flat_scene("scene")
    img = np.zeros((45, 80), dtype=np.uint32)
    seen = {"rows": []}
    render_frame(cam, world, img, 20, update=seen["rows"].append,
                 finish=lambda t: seen.setdefault("time", t), seed=1, batch=8)
    assert seen["rows"] == [17, 35, 44] and seen["time"] >= 0  # 8, 16, 20 of 20 samples
    # sky at the top of the picture (row 0 = top): background (0.7, 0.8, 1.0) -> codes 213, 228, 254
    assert img[0, 0] == (213 << 16) | (228 << 8) | 254
    with dev.Scene(flat, device=0) as s:
        np.testing.assert_array_equal(img, s.render(cam, 80, 45, 20, seed=1)["packed"])
        # a render that is asked to stop immediately leaves the image untouched
        img2 = np.zeros((45, 80), dtype=np.uint32)
        render_frame(cam, s, img2, 64, should_break=lambda: True, seed=1)
        assert not img2.any()
        # stopped after two batches: the picture of the samples rendered so far stays
        calls = []
        img3 = np.zeros((45, 80), dtype=np.uint32)
        render_frame(cam, s, img3, 64, should_break=lambda: len(calls) >= 2, update=calls.append, seed=1, batch=8)
        assert len(calls) == 2
        np.testing.assert_array_equal(img3, s.render(cam, 80, 45, 16, seed=1)["packed"])
