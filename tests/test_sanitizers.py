"""Host code under AddressSanitizer / UBSan (CPU builds only) and the decoders against
malformed texture files.  The reference leaves image decoding to javax.imageio, which is
memory-safe; the C++ host's own PNG / JPEG decoders must fail with a clean error on a
crafted file, never read or write out of bounds."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, SCENES

HOST = os.path.join(ROOT, "raytracing_amd", "host")
ASAN_BIN = os.path.join(HOST, "hiptrace-render-asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def asan_bin():
    subprocess.check_call(["make", "-C", HOST, "asan", "-j4"], stdout=subprocess.DEVNULL)
    return ASAN_BIN


def decode(asan_bin, path, tmp_path):
    out = str(tmp_path / "o.bin")
    r = subprocess.run([asan_bin, "--tool-decode", path, out], capture_output=True, text=True, timeout=120, env=ENV)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    return r, out


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "oracle", "oracle_asan")], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr[-2000:]


@pytest.mark.parametrize("name", ["earthmap.jpg", "quad.png"])
def test_good_files_decode_under_asan(asan_bin, name, tmp_path):
    r, out = decode(asan_bin, os.path.join(SCENES, name), tmp_path)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, dtype=np.uint32)
    assert raw[0] * raw[1] == raw.size - 2


def _segments(jpeg):
    """(marker, start, end) of every marker segment before the scan data."""
    out, p = [], 2
    while p + 4 <= len(jpeg) and jpeg[p] == 0xFF:
        marker = jpeg[p + 1]
        ln = jpeg[p + 2] << 8 | jpeg[p + 3]
        out.append((marker, p, p + 2 + ln))
        if marker == 0xDA:
            break
        p += 2 + ln
    return out


@pytest.fixture(scope="module")
def small_jpeg(tmp_path_factory):
    from PIL import Image
    p = str(tmp_path_factory.mktemp("j") / "s.jpg")
    base = np.asarray(Image.open(os.path.join(SCENES, "earthmap.jpg")).convert("RGB"))[100:148, 300:364]
    Image.fromarray(base).save(p, quality=85, subsampling=2)
    return bytearray(open(p, "rb").read())


def test_dht_whose_lengths_are_no_prefix_code(asan_bin, small_jpeg, tmp_path):
    """bits[1] = 3 asks for three codes of length 1: the lookup table would be written past
    its end."""
    j = bytearray(small_jpeg)
    marker, a, b = next(s for s in _segments(j) if s[0] == 0xC4)
    j[a + 5] = 3  # count of 1-bit codes of the first table
    p = str(tmp_path / "dht.jpg")
    open(p, "wb").write(j)
    r, _ = decode(asan_bin, p, tmp_path)
    assert r.returncode != 0 and "DHT" in r.stderr


def test_dht_with_more_codes_than_values(asan_bin, small_jpeg, tmp_path):
    j = bytearray(small_jpeg)
    marker, a, b = next(s for s in _segments(j) if s[0] == 0xC4)
    j[a + 5 + 15] = 255  # 255 codes of length 16, far beyond the values that follow
    p = str(tmp_path / "dht2.jpg")
    open(p, "wb").write(j)
    r, _ = decode(asan_bin, p, tmp_path)
    assert r.returncode != 0 and "DHT" in r.stderr


def test_sos_header_cut_short(asan_bin, small_jpeg, tmp_path):
    j = bytearray(small_jpeg)
    marker, a, b = next(s for s in _segments(j) if s[0] == 0xDA)
    cut = j[:a + 2] + bytes([0, 3, 3])  # length 3: the component count and nothing else, then end of file
    p = str(tmp_path / "sos.jpg")
    open(p, "wb").write(cut)
    r, _ = decode(asan_bin, p, tmp_path)
    assert r.returncode != 0 and "SOS" in r.stderr


def test_sof_with_fractional_sampling_ratio(asan_bin, small_jpeg, tmp_path):
    """h = 3 beside h = 2: 3 / 2 truncates to 1 and the upsampler would copy rows that the
    smaller plane does not have."""
    j = bytearray(small_jpeg)
    marker, a, b = next(s for s in _segments(j) if s[0] == 0xC0)
    j[a + 11] = 0x32  # first component: h = 3, v = 2
    j[a + 14] = 0x21  # second: h = 2, v = 1
    p = str(tmp_path / "sof.jpg")
    open(p, "wb").write(j)
    r, _ = decode(asan_bin, p, tmp_path)
    assert r.returncode != 0 and ("sampling" in r.stderr or "JPEG" in r.stderr)


def test_truncated_jpeg_everywhere(asan_bin, small_jpeg, tmp_path):
    """Every prefix either decodes (missing scan data reads as zeros) or fails cleanly."""
    for cut in list(range(2, 700, 37)) + [len(small_jpeg) // 2, len(small_jpeg) - 3]:
        p = str(tmp_path / "cut.jpg")
        open(p, "wb").write(small_jpeg[:cut])
        decode(asan_bin, p, tmp_path)


def _png(width, height, ctype=2, depth=8, idat=b""):
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0)) +
            chunk(b"IDAT", idat) + chunk(b"IEND", b""))


def test_png_with_absurd_dimensions(asan_bin, tmp_path):
    for w, h in ((0xFFFFFFF0, 0xFFFFFFF0), (70000, 10), (60000, 60000)):
        p = str(tmp_path / "big.png")
        open(p, "wb").write(_png(w, h, idat=zlib.compress(b"\0" * 16)))
        r, _ = decode(asan_bin, p, tmp_path)
        assert r.returncode != 0 and "too large" in r.stderr


def test_png_with_short_pixel_data_and_bad_palette_index(asan_bin, tmp_path):
    p = str(tmp_path / "short.png")
    open(p, "wb").write(_png(8, 8, idat=zlib.compress(b"\0" * 20)))
    r, _ = decode(asan_bin, p, tmp_path)
    assert r.returncode != 0 and "PNG" in r.stderr
    # palette picture whose indices point past a one-entry palette
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    raw = b"".join(b"\0" + bytes([200] * 4) for _ in range(4))
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 3, 0, 0, 0)) +
           chunk(b"PLTE", b"\x01\x02\x03") + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    open(p, "wb").write(png)
    r, _ = decode(asan_bin, p, tmp_path)
    assert r.returncode != 0 and "palette" in r.stderr


def test_scene_that_names_a_malformed_texture_is_a_load_error(asan_bin, small_jpeg, tmp_path):
    """loadScene catches everything and reports it (main.scala:66-70)."""
    j = bytearray(small_jpeg)
    marker, a, b = next(s for s in _segments(j) if s[0] == 0xC4)
    j[a + 5] = 3
    open(tmp_path / "bad.jpg", "wb").write(j)
    scene = open(os.path.join(SCENES, "scene.json")).read().replace("earthmap.jpg", "bad.jpg")
    open(tmp_path / "scene.json", "w").write(scene)
    r = subprocess.run([asan_bin, "--tool-flatten", str(tmp_path / "scene.json"), str(tmp_path / "flat.json")],
                       capture_output=True, text=True, timeout=120, env=ENV)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
    assert r.returncode != 0 and "DHT" in r.stderr
