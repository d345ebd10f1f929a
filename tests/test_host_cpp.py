"""The C++ front-end (raytracing_amd/host): its scene reader, image codecs and
command line, checked on the CPU against the Python host and PIL.  Nothing here
renders; the rendering path of the same binary is covered by the GPU tests."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd.scene import flatten, load_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "raytracing_amd", "host", "hiptrace-render")
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="hiptrace-render not built (make -C raytracing_amd/host)")


def run(*args, **kw):
    return subprocess.run([BIN, *args], capture_output=True, text=True, timeout=120, **kw)


def decode(path, tmp_path):
    out = str(tmp_path / "decoded.bin")
    r = run("--tool-decode", path, out)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, dtype=np.uint32)
    return raw[2:].reshape(int(raw[1]), int(raw[0]))


def pil_pixels(path):
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB")).astype(np.uint32)
    return im[..., 0] << 16 | im[..., 1] << 8 | im[..., 2]


@pytest.mark.parametrize("name", ["earthmap.jpg", "quad.png"])
def test_decoders_match_the_texels_of_the_scene_images(name, tmp_path):
    """FileLoader.scala:14 reads textures with ImageIO, which wraps the IJG decoder;
    PIL wraps the same arithmetic, so equality here pins the texels the reference sees."""
    path = os.path.join(SCENES, name)
    np.testing.assert_array_equal(decode(path, tmp_path), pil_pixels(path))


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("quality", [35, 90])
def test_jpeg_decoder_chroma_layouts(subsampling, quality, tmp_path):
    from PIL import Image
    base = np.asarray(Image.open(os.path.join(SCENES, "earthmap.jpg")).convert("RGB"))[60:187, 400:601]
    p = str(tmp_path / "t.jpg")
    Image.fromarray(base).save(p, quality=quality, subsampling=subsampling)
    np.testing.assert_array_equal(decode(p, tmp_path), pil_pixels(p))


def test_jpeg_decoder_restart_intervals_and_grey(tmp_path):
    from PIL import Image
    base = np.asarray(Image.open(os.path.join(SCENES, "earthmap.jpg")).convert("RGB"))[200:263, 100:197]
    p = str(tmp_path / "r.jpg")
    Image.fromarray(base).save(p, quality=80, subsampling=2, restart_marker_blocks=3)
    np.testing.assert_array_equal(decode(p, tmp_path), pil_pixels(p))
    p = str(tmp_path / "g.jpg")
    Image.fromarray(base).convert("L").save(p)
    np.testing.assert_array_equal(decode(p, tmp_path), pil_pixels(p))


def test_jpeg_decoder_refuses_progressive_files(tmp_path):
    from PIL import Image
    p = str(tmp_path / "p.jpg")
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(p, progressive=True)
    r = run("--tool-decode", p, str(tmp_path / "o.bin"))
    assert r.returncode != 0 and "baseline" in r.stderr


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1"])
def test_png_decoder_colour_types(mode, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "t.png")
    Image.fromarray(base).convert(mode).save(p)
    np.testing.assert_array_equal(decode(p, tmp_path), pil_pixels(p))


def test_png_encoder_round_trip(tmp_path):
    src = os.path.join(SCENES, "earthmap.jpg")
    out = str(tmp_path / "out.png")
    assert run("--tool-recode", src, out).returncode == 0
    from PIL import Image
    with Image.open(out) as im:
        assert im.mode == "RGB" and im.size == (1024, 512)  # 8-bit RGB, as ImageIO writes TYPE_INT_RGB
    np.testing.assert_array_equal(pil_pixels(out), pil_pixels(src))


@pytest.mark.parametrize("scene", sorted(os.path.basename(p) for p in glob.glob(os.path.join(SCENES, "*.json"))
                                         if not p.endswith("miata.json")))
def test_cpp_reader_flattens_like_the_python_reader(scene, tmp_path):
    """Both hosts hand ht_scene_create the same arrays, bit for bit (fromSTL's
    pattern, the materials map, shared textures, Perlin tables, image texels)."""
    path = os.path.join(SCENES, scene)
    out = str(tmp_path / "flat.json")
    r = run("--tool-flatten", path, out)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    cam, world = load_scene(path, perlin_seed=1)
    fl = flatten(world)
    pairs = {"prim_type": fl.prim_type, "prim_material": fl.prim_material, "prim_group": fl.prim_group,
             "prim_data": fl.prim_data, "mat_type": fl.a_mat_type, "mat_texture": fl.a_mat_texture,
             "mat_param": fl.a_mat_param, "tex_type": fl.a_tex_type, "tex_data": fl.a_tex_data,
             "tex_child": fl.a_tex_child}
    for key, arr in pairs.items():
        np.testing.assert_array_equal(np.asarray(d[key]), np.asarray(arr).ravel(), err_msg=key)
    n_img, n_per = len(fl.images), len(fl.perlins)
    np.testing.assert_array_equal(d["image_width"], fl.a_img_w[:n_img])
    np.testing.assert_array_equal(d["image_height"], fl.a_img_h[:n_img])
    np.testing.assert_array_equal(d["image_offset"], fl.a_img_off[:n_img])
    if n_per:
        np.testing.assert_array_equal(np.asarray(d["perlin_ranvec"]), fl.a_ranvec.ravel())
        np.testing.assert_array_equal(np.asarray(d["perlin_perm"]), fl.a_perm.ravel())
    else:
        assert d["perlin_ranvec"] == [] and d["perlin_perm"] == []
    np.testing.assert_array_equal(np.asarray(d["camera"]), np.frombuffer(bytes(cam.to_c()), dtype=np.float64))
    h = 0
    for t in np.asarray(fl.a_texels).ravel()[:d["n_texels"]].tolist():
        h = (h * 1000003 + t) & ((1 << 64) - 1)
    assert h == d["texel_hash"]


def test_missing_stl_is_a_load_error_not_a_crash(tmp_path):
    """miata.json names an STL the reference does not ship: loadScene prints the
    error and cmd exits with status 1 (main.scala:66-70, 115-117)."""
    r = run("-o", str(tmp_path / "x.png"), os.path.join(SCENES, "miata.json"))
    assert r.returncode == 1
    assert "Error loading scene" in r.stdout and "miata.stl" in r.stderr


def test_command_line_follows_the_reference_options(tmp_path):
    # no -o: the reference opens its window; this program has none
    assert run(os.path.join(SCENES, "scene.json")).returncode == 2
    # dependsOnAll(filename, List(scene)), main.scala:28
    r = run("-o", str(tmp_path / "x.png"))
    assert r.returncode == 1 and "scene" in r.stderr
    r = run("--width", "abc", "-o", "x.png", "scene.json")
    assert r.returncode == 1 and "bad Int value" in r.stderr
    assert run("--bogus").returncode == 1
    assert run("--help").returncode == 0


@pytest.mark.parametrize("seconds,text", [
    ("0.5", "00m:00.50s"), ("12.345", "00m:12.35s"), ("90", "02m:30.00s"),  # minutes are ROUNDED (main.scala:103)
    ("59.995", "01m:60.00s"), ("3725.2", "01h:02m:05.20s"), ("0.125", "00m:00.13s"),  # HALF_UP like java.util.Formatter
])
def test_format_duration_quirks(seconds, text):
    assert run("--tool-duration", seconds).stdout.strip() == text


def test_generated_mesh_scene_reads_the_same_in_both_hosts(tmp_path):
    """A procedural mesh written as ASCII STL (the stand-in for the miata.stl the
    reference does not ship) next to a copy of miata.json: BVH.fromSTL's pattern
    (BVH.scala:36-43) gets every facet back, in order, in both readers."""
    import shutil
    from raytracing_amd.meshgen import bumpy_ellipsoid, rounded, write_ascii_stl
    from raytracing_amd.scene import Mesh
    tris = bumpy_ellipsoid(20, 40)
    write_ascii_stl(str(tmp_path / "miata.stl"), tris)
    shutil.copy(os.path.join(SCENES, "miata.json"), tmp_path / "miata.json")
    cam, world = load_scene(str(tmp_path / "miata.json"))
    got = np.array([t.vertices for t in world[1].triangles])
    np.testing.assert_array_equal(got, rounded(tris))
    fl = flatten(world)
    # the array form of the same entry flattens to the same primitives
    fm = flatten([world[0], Mesh(rounded(tris), world[1].triangles[0].material)])
    np.testing.assert_array_equal(fm.prim_data, fl.prim_data)
    np.testing.assert_array_equal(fm.prim_group, fl.prim_group)
    np.testing.assert_array_equal(fm.prim_material, fl.prim_material)
    out = str(tmp_path / "flat.json")
    r = run("--tool-flatten", str(tmp_path / "miata.json"), out)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    np.testing.assert_array_equal(np.asarray(d["prim_data"]), fl.prim_data.ravel())
    np.testing.assert_array_equal(np.asarray(d["prim_group"]), fl.prim_group)
