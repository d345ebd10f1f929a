"""The JVM binding (bindings/jvm/HipTrace.scala) cannot be compiled here (no JDK), so its
byte-layout assumptions are checked from C and Python instead: every offset its `Layout`
object states must be offsetof() of include/hiptrace.h, and a frame rendered through
structs filled by exactly those offsets, with the entry points the binding calls, must be
the frame the ordinary host renders."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from raytracing_amd import _abi as A

SCALA = os.path.join(ROOT, "bindings", "jvm", "HipTrace.scala")
STRUCTS = {"DESC": "ht_scene_desc", "CAMERA": "ht_camera", "PARAMS": "ht_render_params"}


def scala_layout():
    text = open(SCALA).read()
    body = text[text.index("object Layout {"):text.index("import Layout.*")]
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"val (\w+) = (-?\d+)L?\b", body)}
    assert len(vals) > 50
    return vals


def test_layout_constants_equal_offsetof(tmp_path):
    vals = scala_layout()
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{ROOT}/include/hiptrace.h"', "int main(void) {"]
    expected = {}
    for name, v in vals.items():
        prefix, _, field = name.partition("_")
        if prefix in STRUCTS and field == "SIZE":
            lines.append(f'  printf("{name} %zu\\n", sizeof({STRUCTS[prefix]}));')
        elif prefix in STRUCTS:
            lines.append(f'  printf("{name} %zu\\n", offsetof({STRUCTS[prefix]}, {field.lower()}));')
        elif prefix in ("PRIM", "TEX") and field == "STRIDE":
            lines.append(f'  printf("{name} %d\\n", HT_{name});')
        elif prefix in ("PRIM", "MAT", "TEX", "E"):
            lines.append(f'  printf("{name} %d\\n", HT_{name});')
        else:
            raise AssertionError(f"unknown Layout constant {name}")
        expected[name] = v
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    got = dict((k, int(v)) for k, v in (ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines()))
    assert got == expected
    # every field of the three structs has a constant (no field forgotten)
    for prefix, cls in (("DESC", A.ht_scene_desc), ("CAMERA", A.ht_camera), ("PARAMS", A.ht_render_params)):
        for field, _ in cls._fields_:
            assert f"{prefix}_{field.upper()}" in vals, field


def test_binding_names_only_exported_symbols():
    text = open(SCALA).read()
    used = set(re.findall(r'fn(?:Void)?\("(ht_\w+)"', text))
    assert used and used <= set(A.EXPORTS)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["scene", "perlin", "cornell"])
def test_render_through_the_bindings_offsets(ht, flat_scene, name):
    """ht_scene_create / ht_frame_create / ht_frame_accumulate / ht_frame_read with the
    structs written at the Scala file's offsets, as FFM's MemorySegment.set would."""
    L = scala_layout()
    cam, _, flat = flat_scene(name)
    raw = C.CDLL(ht.LIB_PATH)  # no argtypes: plain pointers and ints, like a downcall handle
    keep = []

    desc = bytearray(L["DESC_SIZE"])
    def put_i32(off, v): struct.pack_into("<i", desc, off, int(v))
    def put_ptr(off, v): struct.pack_into("<Q", desc, off, int(v))
    d = flat.desc
    for field, ctype in A.ht_scene_desc._fields_:
        off = L["DESC_" + field.upper()]
        v = getattr(d, field)
        if ctype is C.c_int32:
            put_i32(off, v)
        else:  # a pointer: the address the ordinary host passes, written at the binding's offset
            put_ptr(off, C.cast(v, C.c_void_p).value or 0)
    put_ptr(L["DESC_PRIM_GROUP"], 0)  # the binding passes NULL there

    camb = bytearray(L["CAMERA_SIZE"])
    cc = cam.to_c()
    for key, vals in (("ORIGIN", cc.origin), ("TARGET", cc.target), ("VUP", cc.vup), ("BACKGROUND", cc.background)):
        struct.pack_into("<3d", camb, L["CAMERA_" + key], *vals)
    for key, v in (("VFOV", cc.vfov), ("ASPECT_RATIO", cc.aspect_ratio), ("APERTURE", cc.aperture),
                   ("FOCUS_DISTANCE", cc.focus_distance)):
        struct.pack_into("<d", camb, L["CAMERA_" + key], v)

    W, H, spp, seed = 96, 54, 5, 21
    par = bytearray(L["PARAMS_SIZE"])
    for key, v in (("WIDTH", W), ("HEIGHT", H), ("SAMPLES", spp), ("SAMPLE_OFFSET", 0), ("MAX_DEPTH", 50), ("JITTER", 1),
                   ("PRECISION", 0), ("BATCH", 0), ("DEVICE", -1), ("COLLECT_COUNTERS", 0)):
        struct.pack_into("<i", par, L["PARAMS_" + key], v)
    struct.pack_into("<q", par, L["PARAMS_SEED"], seed)

    def buf(b):
        a = (C.c_char * len(b)).from_buffer(b)
        keep.append(a)
        return C.c_void_p(C.addressof(a))

    scene, frame = C.c_void_p(), C.c_void_p()
    raw.ht_last_error.restype = C.c_char_p
    assert raw.ht_scene_create(buf(desc), C.c_int(-1), C.byref(scene)) == 0, raw.ht_last_error()
    try:
        assert raw.ht_frame_create(scene, C.c_int(W), C.c_int(H), C.byref(frame)) == 0, raw.ht_last_error()
        assert raw.ht_frame_accumulate(frame, buf(camb), buf(par), C.c_void_p(0)) == 0, raw.ht_last_error()
        packed = np.zeros((H, W), dtype=np.uint32)
        assert raw.ht_frame_read(frame, C.c_void_p(0), C.c_void_p(packed.ctypes.data)) == 0, raw.ht_last_error()
        raw.ht_frame_destroy.restype = None
        raw.ht_frame_destroy(frame)
    finally:
        raw.ht_scene_destroy.restype = None
        raw.ht_scene_destroy(scene)
    with ht.Scene(flat, device=0) as s:
        ref = s.render(cam, W, H, spp, seed=seed)
    assert np.array_equal(packed, ref["packed"])
