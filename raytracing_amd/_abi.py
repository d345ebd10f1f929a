"""ctypes mirror of include/hiptrace.h (structs and constants only)."""
import ctypes as C

HT_VERSION = 100

HT_OK, HT_E_INVALID, HT_E_NO_DEVICE, HT_E_HIP, HT_E_TOO_LARGE, HT_E_CANCELLED = 0, -1, -2, -3, -4, -5

HT_PRIM_SPHERE, HT_PRIM_XYRECT, HT_PRIM_XZRECT, HT_PRIM_YZRECT, HT_PRIM_TRIANGLE = range(5)
HT_PRIM_STRIDE = 9
HT_MAT_DIFFUSE, HT_MAT_GLOSSY, HT_MAT_GLASS, HT_MAT_EMISSION = range(4)
HT_TEX_SOLID, HT_TEX_CHECKER, HT_TEX_PERLIN, HT_TEX_IMAGE = range(4)
HT_TEX_STRIDE = 4
HT_PERLIN_POINTS = 256
HT_PRECISION_FP32, HT_PRECISION_FP64 = 0, 1
HT_NOALLOC_MAX_PIXELS = 1 << 21
HT_SHARD_SAMPLES, HT_SHARD_TILES = 0, 1

_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)
_pu32 = C.POINTER(C.c_uint32)
_pf64 = C.POINTER(C.c_double)
_pf32 = C.POINTER(C.c_float)


class ht_scene_desc(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32),
        ("prim_type", _pi32),
        ("prim_material", _pi32),
        ("prim_data", _pf64),
        ("prim_group", _pi32),
        ("n_materials", C.c_int32),
        ("mat_type", _pi32),
        ("mat_texture", _pi32),
        ("mat_param", _pf64),
        ("n_textures", C.c_int32),
        ("tex_type", _pi32),
        ("tex_data", _pf64),
        ("tex_child", _pi32),
        ("n_images", C.c_int32),
        ("image_width", _pi32),
        ("image_height", _pi32),
        ("image_offset", _pi64),
        ("image_texels", _pu32),
        ("n_perlin", C.c_int32),
        ("perlin_ranvec", _pf64),
        ("perlin_perm", _pi32),
    ]


class ht_camera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("target", C.c_double * 3),
        ("vup", C.c_double * 3),
        ("vfov", C.c_double),
        ("aspect_ratio", C.c_double),
        ("aperture", C.c_double),
        ("focus_distance", C.c_double),
        ("background", C.c_double * 3),
    ]


class ht_render_params(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("samples", C.c_int32),
        ("sample_offset", C.c_int32),
        ("max_depth", C.c_int32),
        ("jitter", C.c_int32),
        ("precision", C.c_int32),
        ("batch", C.c_int32),
        ("seed", C.c_uint64),
        ("device", C.c_int32),
        ("collect_counters", C.c_int32),
        ("tile_first", C.c_int32),
        ("tile_stride", C.c_int32),
        ("shard_mode", C.c_int32),
        ("reserved", C.c_int32),
    ]


class ht_stats(C.Structure):
    _fields_ = [
        ("seconds_total", C.c_double),
        ("seconds_kernel", C.c_double),
        ("samples", C.c_int64),
        ("launches", C.c_int32),
        ("samples_per_pixel", C.c_int32),
        ("n_rays", C.c_int64),
        ("n_box", C.c_int64),
        ("n_sphere", C.c_int64),
        ("n_rect", C.c_int64),
        ("n_tri", C.c_int64),
        ("n_shade", C.c_int64 * 4),
        ("n_tex", C.c_int64 * 4),
        ("n_steps", C.c_int64),
        ("n_active", C.c_int64),
        ("n_shade_rounds", C.c_int64),
        ("n_regen_rounds", C.c_int64),
        ("n_miss", C.c_int64),
        ("cyc_trav", C.c_int64),
        ("cyc_shade", C.c_int64),
        ("cyc_regen", C.c_int64),
        ("block_threads", C.c_int32),
        ("blocks_per_cu", C.c_int32),
        ("lds_bytes", C.c_int32),
        ("n_cu", C.c_int32),
        ("shader_clock_mhz", C.c_double),
        ("batch", C.c_int32),
        ("kernel_sgprs", C.c_int32),
        ("n_node_steps", C.c_int64),
        ("kernel", C.c_char * 64),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            if isinstance(v, bytes):
                v = v.decode("ascii", "replace")
            d[name] = list(v) if hasattr(v, "__len__") and not isinstance(v, str) else v
        return d


class ht_scene_info(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32),
        ("n_nodes", C.c_int32),
        ("bvh_depth", C.c_int32),
        ("n_materials", C.c_int32),
        ("n_textures", C.c_int32),
        ("lds_bytes_fp32", C.c_int32),
        ("lds_bytes_fp64", C.c_int32),
        ("device", C.c_int32),
        ("hbm_store_fp32", C.c_int32),
        ("hbm_store_fp64", C.c_int32),
        ("sah_before", C.c_double),
        ("sah_after", C.c_double),
    ]


# every symbol include/hiptrace.h declares
EXPORTS = [
    "ht_version",
    "ht_device_count",
    "ht_scene_create",
    "ht_scene_destroy",
    "ht_render",
    "ht_render_noalloc",
    "ht_render_accumulate", "ht_render_sharded",
    "ht_resolve",
    "ht_resolve_device",
    "ht_shard_tile_pixels",
    "ht_resolve_tiles_device",
    "ht_assemble_tiles_device",
    "ht_frame_create",
    "ht_frame_destroy",
    "ht_frame_clear",
    "ht_frame_accumulate",
    "ht_frame_read",
    "ht_frame_samples",
    "ht_first_hit",
    "ht_cancel",
    "ht_last_error",
    "ht_scene_get_info",
    "ht_set_option",
    "ht_debug_timeline",
]
