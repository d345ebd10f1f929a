"""The C-ABI shared library loads and exports every symbol include/hiptrace.h declares.
No compute is attempted here (these tests run without a GPU)."""
import ctypes as C
import os
import re

import pytest

from raytracing_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hiptrace.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ht_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert header_symbols() == sorted(A.EXPORTS)


def test_library_exports_every_declared_symbol(ht):
    L = ht.load()
    for name in header_symbols():
        assert hasattr(L, name), f"libhiptrace.so does not export {name}"


def test_version(ht):
    assert ht.version() == A.HT_VERSION


def test_struct_sizes_match_the_header(tmp_path):
    """Compile the header with gcc and compare sizeof / offsetof with the ctypes mirror."""
    import subprocess

    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "hiptrace.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ht_scene_desc), sizeof(ht_camera),'
        ' sizeof(ht_render_params), sizeof(ht_stats), sizeof(ht_scene_info),'
        ' offsetof(ht_render_params, seed), offsetof(ht_stats, n_rays));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(A.ht_scene_desc), C.sizeof(A.ht_camera), C.sizeof(A.ht_render_params),
            C.sizeof(A.ht_stats), C.sizeof(A.ht_scene_info), A.ht_render_params.seed.offset,
            A.ht_stats.n_rays.offset]
    assert got == want


def test_options_need_no_device(ht):
    """Every option that include/hiptrace.h documents for ht_set_option is accepted without a
    device, and an unknown one is refused."""
    import re

    text = open(os.path.join(ROOT, "include", "hiptrace.h")).read()
    doc = text[text.index("Process-wide tuning knobs"):text.index("int ht_set_option")]
    names = sorted(set(re.findall(r'"([a-z_0-9]+)"', doc)))
    assert len(names) >= 18 and {"xcd_bands", "chunk", "wavefront", "bvh_optimise"} <= set(names), names
    # a value every option takes; then back to the defaults the library starts with
    defaults = {"wait_threshold": 0, "max_blocks_per_cu": 8, "max_waves_per_cu": 0, "flat_kernels": 1,
                "bvh_optimise_max_nodes": 1 << 13, "block_threads": 0, "node_den": 2,
                "default_batch": 1024, "chunk": 0, "chunk_tail": 4, "gpu_bvh": 0,
                "force_hbm_scene": 0, "wavefront": 0, "wavefront_paths": 1 << 24, "wavefront_refill": 32,
                "bvh_optimise": 4, "timeline": 0, "xcd_bands": 1}
    assert set(names) == set(defaults), set(names) ^ set(defaults)
    for name in names:
        rc = ht.load().ht_set_option(name.encode(), 0 if name in ("block_threads", "force_hbm_scene", "gpu_bvh", "wavefront", "timeline") else 8)
        assert rc == 0, name
    for name, v in defaults.items():
        ht.set_option(name, v)
    with pytest.raises(ht.HipTraceError) as e:
        ht.set_option("no_such_option", 1)
    assert e.value.code == A.HT_E_INVALID and "no_such_option" in str(e.value)


def test_frame_calls_reject_null_handles(ht):
    L = ht.load()
    assert L.ht_frame_samples(None) == 0
    assert L.ht_frame_clear(None) == A.HT_E_INVALID
    assert L.ht_frame_read(None, None, None) == A.HT_E_INVALID
    L.ht_frame_destroy(None)  # a no-op, like free(NULL)
    L.ht_scene_destroy(None)


def test_no_device_fails_loudly(ht, flat_scene):
    """Without a HIP device the product path must refuse, not fall back."""
    if ht.device_count() > 0:
        pytest.skip("a HIP device is visible")
    _, _, flat = flat_scene("scene")
    with pytest.raises(ht.HipTraceError) as e:
        ht.Scene(flat)
    assert e.value.code == A.HT_E_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "oracle/" not in text and "liboracle" not in text, f
