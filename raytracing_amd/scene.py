"""Host-side scene model: the reference's case classes and its JSON scene reader,
restated for the Python harness, plus the flattening into the arrays the C ABI
takes (include/hiptrace.h).

Reference (Ivoah/RayTracing): shared/src/main/scala/JsonFormats.scala:5-99 (schema),
jvm/src/main/scala/main.scala:32-71 (loadScene), jvm/src/main/scala/FileLoader.scala:8-21
(path and image rules), shared/src/main/scala/BVH.scala:35-55 (ASCII STL grammar).
Nothing in here traces rays.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _abi as A

Vec3 = Tuple[float, float, float]


# ---- textures (Texture.scala:12-107) -------------------------------------
@dataclass
class SolidColor:
    color: Vec3


@dataclass
class Checker:
    size: float
    t1: "Texture"
    t2: "Texture"


@dataclass
class Perlin:
    scale: float
    # Texture.scala:38-42: drawn at load time from an unseeded RNG in the reference;
    # here from `perlin_seed` so both the device and the oracle shade the same table.
    ranvec: np.ndarray = field(default=None, repr=False)  # [256, 3]
    perm: np.ndarray = field(default=None, repr=False)  # [3, 256]

    @staticmethod
    def tables(seed: int) -> Tuple[np.ndarray, np.ndarray]:
        """256 unit vectors and three permutations from a splitmix64 stream: the
        same tables, bit for bit, as the C++ front-end draws for the same seed."""
        mask = (1 << 64) - 1
        state = [seed & mask]

        def nxt() -> int:
            state[0] = (state[0] + 0x9E3779B97F4A7C15) & mask
            z = state[0]
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
            return z ^ (z >> 31)

        n = A.HT_PERLIN_POINTS
        v = np.empty((n, 3))
        for i in range(n):
            x, y, z = (-1.0 + 2.0 * ((nxt() >> 11) * 2.0 ** -53) for _ in range(3))
            length = math.sqrt(x * x + y * y + z * z)  # Vec3.random(-1,1).unit_vector
            v[i] = (x / length, y / length, z / length)
        perm = np.empty((3, n), dtype=np.int32)
        for k in range(3):
            row = list(range(n))
            for i in range(n - 1, 0, -1):
                j = nxt() % (i + 1)
                row[i], row[j] = row[j], row[i]
            perm[k] = row
        return v, perm


@dataclass
class Image:
    width: int
    height: int
    pixels: np.ndarray = field(repr=False)  # [height*width] uint32 0x00RRGGBB, row 0 = top


Texture = Union[SolidColor, Checker, Perlin, Image]


# ---- materials (Material.scala:9-55) --------------------------------------
@dataclass
class Diffuse:
    texture: Texture


@dataclass
class Glossy:
    texture: Texture
    roughness: float


@dataclass
class Glass:
    texture: Texture
    ior: float


@dataclass
class Emission:
    texture: Texture
    strength: float


Material = Union[Diffuse, Glossy, Glass, Emission]


# ---- primitives (Hittable.scala:25-115) -----------------------------------
@dataclass
class Sphere:
    center: Vec3
    radius: float
    material: Material


@dataclass
class XYRect:
    x0: float
    x1: float
    y0: float
    y1: float
    z: float
    material: Material


@dataclass
class XZRect:
    x0: float
    x1: float
    z0: float
    z1: float
    y: float
    material: Material


@dataclass
class YZRect:
    y0: float
    y1: float
    z0: float
    z1: float
    x: float
    material: Material


@dataclass
class Triangle:
    vertices: Tuple[Vec3, Vec3, Vec3]
    material: Material


@dataclass
class STL:
    """An STL world entry: BVH.fromSTL's triangles in file order (BVH.scala:44-53)."""

    triangles: List[Triangle]


@dataclass
class Mesh:
    """An STL world entry held as one array: the same meaning as STL (facets in
    order, one material), without a Python object per facet.  For large meshes."""

    vertices: np.ndarray = field(repr=False)  # [n, 3, 3]: facet, corner, xyz
    material: Material = None


Hittable = Union[Sphere, XYRect, XZRect, YZRect, Triangle, STL, Mesh]


# ---- camera (Camera.scala:3-4) ---------------------------------------------
@dataclass
class Camera:
    origin: Vec3
    target: Vec3
    vup: Vec3
    vfov: float
    aspect_ratio: float
    aperture: float
    focus_distance: float
    background: Vec3

    def to_c(self) -> A.ht_camera:
        c = A.ht_camera()
        c.origin[:] = self.origin
        c.target[:] = self.target
        c.vup[:] = self.vup
        c.vfov = self.vfov
        c.aspect_ratio = self.aspect_ratio
        c.aperture = self.aperture
        c.focus_distance = self.focus_distance
        c.background[:] = self.background
        return c


# ---- file rules (jvm FileLoader.scala:8-21) ---------------------------------
class FileLoader:
    """Relative paths resolve against the scene's directory, base name only."""

    def __init__(self, scene_dir: str):
        self.scene_dir = scene_dir

    def _file(self, path: str) -> str:
        if os.path.isabs(path):
            return path
        return os.path.join(self.scene_dir, os.path.basename(path))

    def load_image(self, path: str) -> Image:
        from PIL import Image as PILImage  # decoder of the harness; the JVM host uses ImageIO

        with PILImage.open(self._file(path)) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint32)
        h, w, _ = rgb.shape
        packed = (rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2]
        return Image(w, h, np.ascontiguousarray(packed.reshape(-1), dtype=np.uint32))

    def load_file(self, path: str) -> str:
        with open(self._file(path), "r") as f:
            return f.read()


_NUM = r"(-?\d+.\d+)"  # the reference's own number pattern, '.' unescaped (BVH.scala:37)
_TRIANGLE_RE = re.compile(
    "^facet normal " + " ".join([_NUM] * 3) + "\n"
    "outer loop\n"
    "vertex " + " ".join([_NUM] * 3) + "\n"
    "vertex " + " ".join([_NUM] * 3) + "\n"
    "vertex " + " ".join([_NUM] * 3) + "\n"
    "endloop\n"
    "endfacet",
    re.M,
)


def from_stl(source: str, material: Material) -> STL:
    """BVH.fromSTL (BVH.scala:35-55): facet normals are matched but ignored."""
    tris = []
    for m in _TRIANGLE_RE.finditer(source):
        g = [float(x) for x in m.groups()]
        tris.append(Triangle(((g[3], g[4], g[5]), (g[6], g[7], g[8]), (g[9], g[10], g[11])), material))
    return STL(tris)


def _vec3(v: Sequence[float]) -> Vec3:
    return (float(v[0]), float(v[1]), float(v[2]))


def _read_texture(j: dict, loader: FileLoader, perlin_seed: List[int]) -> Texture:
    t = j["type"]  # JsonFormats.scala:19-29
    if t == "SolidColor":
        return SolidColor(_vec3(j["color"]))
    if t == "Checker":
        return Checker(float(j["size"]), _read_texture(j["t1"], loader, perlin_seed),
                       _read_texture(j["t2"], loader, perlin_seed))
    if t == "Perlin":
        ranvec, perm = Perlin.tables(perlin_seed[0])
        perlin_seed[0] += 1
        return Perlin(float(j["scale"]), ranvec, perm)
    if t == "Image":
        return loader.load_image(j["file"])
    raise ValueError(f"unknown texture type {t!r}")


def _read_material(j: dict, loader: FileLoader, perlin_seed: List[int]) -> Material:
    t = j["type"]  # JsonFormats.scala:35-45
    tex = _read_texture(j["texture"], loader, perlin_seed)
    if t == "Diffuse":
        return Diffuse(tex)
    if t == "Glass":
        return Glass(tex, float(j["ior"]))
    if t == "Glossy":
        return Glossy(tex, float(j["roughness"]))
    if t == "Emission":
        return Emission(tex, float(j["strength"]))
    raise ValueError(f"unknown material type {t!r}")


def _read_hittable(j: dict, loader: FileLoader, materials: Dict[str, Material]) -> Hittable:
    t = j["type"]  # JsonFormats.scala:87-99
    mat = materials[j["material"]]
    if t == "STL":
        return from_stl(loader.load_file(j["file"]), mat)
    if t == "Triangle":
        v = j["vertices"]
        return Triangle((_vec3(v[0]), _vec3(v[1]), _vec3(v[2])), mat)
    if t == "Sphere":
        return Sphere(_vec3(j["center"]), float(j["radius"]), mat)
    s = j.get("sides")
    if t == "XYRect":
        return XYRect(float(s[0]), float(s[1]), float(s[2]), float(s[3]), float(j["z"]), mat)
    if t == "XZRect":
        return XZRect(float(s[0]), float(s[1]), float(s[2]), float(s[3]), float(j["y"]), mat)
    if t == "YZRect":
        return YZRect(float(s[0]), float(s[1]), float(s[2]), float(s[3]), float(j["x"]), mat)
    raise ValueError(f"unknown hittable type {t!r}")


def load_scene(path: str, perlin_seed: int = 1) -> Tuple[Camera, List[Hittable]]:
    """loadScene (main.scala:32-71) for the JSON format: returns (camera, world list).

    The reference wraps the world list in its own BVH; here the flat list is
    returned and the library builds its own hierarchy.
    """
    loader = FileLoader(os.path.dirname(os.path.abspath(path)))
    with open(path, "r") as f:
        j = json.load(f)
    seed_box = [perlin_seed]
    materials = {k: _read_material(v, loader, seed_box) for k, v in j["materials"].items()}
    c = j["camera"]
    camera = Camera(_vec3(c["origin"]), _vec3(c["target"]), _vec3(c["vup"]), float(c["vfov"]),
                    float(c["aspect_ratio"]), float(c["aperture"]), float(c["focus_distance"]),
                    _vec3(c["background"]))
    world = [_read_hittable(o, loader, materials) for o in j["world"]]
    return camera, world


# ---- flattening ---------------------------------------------------------------
class FlatScene:
    """The world as the SoA arrays of ht_scene_desc.  Keeps the numpy buffers alive."""

    def __init__(self, world: Sequence[Hittable]):
        prim_type: List[int] = []
        prim_mat: List[int] = []
        prim_data: List[List[float]] = []
        prim_group: List[int] = []
        self._mat_index: Dict[int, int] = {}
        self._tex_index: Dict[int, int] = {}
        self.mat_type: List[int] = []
        self.mat_texture: List[int] = []
        self.mat_param: List[float] = []
        self.tex_type: List[int] = []
        self.tex_data: List[List[float]] = []
        self.tex_child: List[List[int]] = []
        self.images: List[Image] = []
        self.perlins: List[Perlin] = []
        self._keep = []

        def add(kind: int, data: Sequence[float], material: Material, group: int):
            prim_type.append(kind)
            prim_mat.append(self._material(material))
            prim_data.append(list(data) + [0.0] * (A.HT_PRIM_STRIDE - len(data)))
            prim_group.append(group)

        for g, o in enumerate(world):
            if isinstance(o, Sphere):
                add(A.HT_PRIM_SPHERE, [*o.center, o.radius], o.material, g)
            elif isinstance(o, XYRect):
                add(A.HT_PRIM_XYRECT, [o.x0, o.x1, o.y0, o.y1, o.z], o.material, g)
            elif isinstance(o, XZRect):
                add(A.HT_PRIM_XZRECT, [o.x0, o.x1, o.z0, o.z1, o.y], o.material, g)
            elif isinstance(o, YZRect):
                add(A.HT_PRIM_YZRECT, [o.y0, o.y1, o.z0, o.z1, o.x], o.material, g)
            elif isinstance(o, Triangle):
                add(A.HT_PRIM_TRIANGLE, [*o.vertices[0], *o.vertices[1], *o.vertices[2]], o.material, g)
            elif isinstance(o, STL):
                for t in o.triangles:
                    add(A.HT_PRIM_TRIANGLE, [*t.vertices[0], *t.vertices[1], *t.vertices[2]], t.material, g)
            elif isinstance(o, Mesh):
                v = np.asarray(o.vertices, dtype=np.float64).reshape(-1, 9)
                m = self._material(o.material)
                prim_type.extend([A.HT_PRIM_TRIANGLE] * len(v))
                prim_mat.extend([m] * len(v))
                prim_data.extend(v.tolist())
                prim_group.extend([g] * len(v))
            else:
                raise TypeError(f"not a Hittable: {o!r}")
        if not prim_type:
            raise ValueError("empty world")

        self.prim_type = np.asarray(prim_type, dtype=np.int32)
        self.prim_material = np.asarray(prim_mat, dtype=np.int32)
        self.prim_data = np.asarray(prim_data, dtype=np.float64).reshape(-1, A.HT_PRIM_STRIDE)
        self.prim_group = np.asarray(prim_group, dtype=np.int32)
        self.n_prims = len(prim_type)

        self.a_mat_type = np.asarray(self.mat_type, dtype=np.int32)
        self.a_mat_texture = np.asarray(self.mat_texture, dtype=np.int32)
        self.a_mat_param = np.asarray(self.mat_param, dtype=np.float64)
        self.a_tex_type = np.asarray(self.tex_type, dtype=np.int32)
        self.a_tex_data = np.asarray(self.tex_data, dtype=np.float64).reshape(-1, A.HT_TEX_STRIDE)
        self.a_tex_child = np.asarray(self.tex_child, dtype=np.int32).reshape(-1, 2)

        n_img = len(self.images)
        self.a_img_w = np.asarray([im.width for im in self.images] or [0], dtype=np.int32)
        self.a_img_h = np.asarray([im.height for im in self.images] or [0], dtype=np.int32)
        offs, total = [], 0
        for im in self.images:
            offs.append(total)
            total += im.width * im.height
        self.a_img_off = np.asarray(offs or [0], dtype=np.int64)
        self.a_texels = (np.concatenate([im.pixels for im in self.images]).astype(np.uint32)
                         if n_img else np.zeros(1, dtype=np.uint32))
        n_per = len(self.perlins)
        self.a_ranvec = (np.stack([p.ranvec for p in self.perlins]).astype(np.float64)
                         if n_per else np.zeros((1, 256, 3)))
        self.a_perm = (np.stack([p.perm for p in self.perlins]).astype(np.int32)
                       if n_per else np.zeros((1, 3, 256), dtype=np.int32))
        for name in ("a_ranvec", "a_perm", "a_texels", "prim_data", "a_tex_data", "a_tex_child"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name)))

        d = A.ht_scene_desc()
        d.n_prims = self.n_prims
        d.prim_type = self.prim_type.ctypes.data_as(A._pi32)
        d.prim_material = self.prim_material.ctypes.data_as(A._pi32)
        d.prim_data = self.prim_data.ctypes.data_as(A._pf64)
        d.prim_group = self.prim_group.ctypes.data_as(A._pi32)
        d.n_materials = len(self.mat_type)
        d.mat_type = self.a_mat_type.ctypes.data_as(A._pi32)
        d.mat_texture = self.a_mat_texture.ctypes.data_as(A._pi32)
        d.mat_param = self.a_mat_param.ctypes.data_as(A._pf64)
        d.n_textures = len(self.tex_type)
        d.tex_type = self.a_tex_type.ctypes.data_as(A._pi32)
        d.tex_data = self.a_tex_data.ctypes.data_as(A._pf64)
        d.tex_child = self.a_tex_child.ctypes.data_as(A._pi32)
        d.n_images = n_img
        d.image_width = self.a_img_w.ctypes.data_as(A._pi32)
        d.image_height = self.a_img_h.ctypes.data_as(A._pi32)
        d.image_offset = self.a_img_off.ctypes.data_as(A._pi64)
        d.image_texels = self.a_texels.ctypes.data_as(A._pu32)
        d.n_perlin = n_per
        d.perlin_ranvec = self.a_ranvec.ctypes.data_as(A._pf64)
        d.perlin_perm = self.a_perm.ctypes.data_as(A._pi32)
        self.desc = d

    # materials / textures are de-duplicated by object identity, in first-use order
    def _material(self, m: Material) -> int:
        k = id(m)
        if k in self._mat_index:
            return self._mat_index[k]
        self._keep.append(m)
        tex = self._texture(m.texture)
        if isinstance(m, Diffuse):
            kind, param = A.HT_MAT_DIFFUSE, 0.0
        elif isinstance(m, Glossy):
            kind, param = A.HT_MAT_GLOSSY, m.roughness
        elif isinstance(m, Glass):
            kind, param = A.HT_MAT_GLASS, m.ior
        elif isinstance(m, Emission):
            kind, param = A.HT_MAT_EMISSION, m.strength
        else:
            raise TypeError(f"not a Material: {m!r}")
        idx = len(self.mat_type)
        self.mat_type.append(kind)
        self.mat_texture.append(tex)
        self.mat_param.append(float(param))
        self._mat_index[k] = idx
        return idx

    def _texture(self, t: Texture) -> int:
        k = id(t)
        if k in self._tex_index:
            return self._tex_index[k]
        self._keep.append(t)
        idx = len(self.tex_type)
        self.tex_type.append(0)
        self.tex_data.append([0.0] * A.HT_TEX_STRIDE)
        self.tex_child.append([-1, -1])
        self._tex_index[k] = idx
        if isinstance(t, SolidColor):
            self.tex_type[idx] = A.HT_TEX_SOLID
            self.tex_data[idx] = [*t.color, 0.0]
        elif isinstance(t, Checker):
            self.tex_type[idx] = A.HT_TEX_CHECKER
            self.tex_data[idx] = [t.size, 0.0, 0.0, 0.0]
            self.tex_child[idx] = [self._texture(t.t1), self._texture(t.t2)]
        elif isinstance(t, Perlin):
            self.tex_type[idx] = A.HT_TEX_PERLIN
            self.tex_data[idx] = [t.scale, 0.0, 0.0, 0.0]
            self.tex_child[idx] = [len(self.perlins), -1]
            self.perlins.append(t)
        elif isinstance(t, Image):
            self.tex_type[idx] = A.HT_TEX_IMAGE
            self.tex_child[idx] = [len(self.images), -1]
            self.images.append(t)
        else:
            raise TypeError(f"not a Texture: {t!r}")
        return idx


def flatten(world: Sequence[Hittable]) -> FlatScene:
    return FlatScene(world)
