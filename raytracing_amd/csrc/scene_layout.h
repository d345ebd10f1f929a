// scene_layout.h — the scene image shared by the host flattener and the device
// kernels.
//
// The whole scene (BVH nodes, primitives, materials, textures, Perlin tables)
// is one byte blob addressed by the byte offsets below.  Image texels are a
// separate global array.  Two blobs exist per scene: R = float (the production
// path) and R = double (the validation path); the layout is the same with
// sizeof(R) scaled.
//
// A blob comes in one of two stores:
//   on-chip (wide = 0): every workgroup copies the blob into LDS once and
//     traverses it there; references are 16 bits (<= 4096 primitives per kind,
//     <= 32768 nodes), and so are the traversal-stack entries;
//   HBM (wide = 1): for scenes beyond the LDS store (large meshes) the kernels
//     read the blob where it lies in global memory (L2 / Infinity Cache serve
//     the hot top of the tree); references and stack entries are 32 bits.
#pragma once
#include <stdint.h>

namespace ht {

// ---- child / primitive references ------------------------------------------
//   on-chip store (16 bits)                   HBM store (32 bits)
//   inner node : index (< 0x8000)             index (< 0x80000000)
//   leaf       : 0x8000 | kind << 12 | i      0x80000000 | kind << 28 | i
//   empty      : 0xFFFF                       0xF00FFFFF
// (the 32-bit form is the host builder's; the HBM store's image holds wide_ref24() of it)
// (a wide reference has 24 significant bits: 31-28 and 19-0; an inner index is below 2^20
// because a scene holds at most 2^20 primitives.  The traversal stack of the HBM store keeps
// them as three bytes, RefT<true>::STACK_ENTRY.)
// (i = index into the kind's array)
enum : uint32_t {
  REF_LEAF = 0x8000u,
  REF_EMPTY = 0xFFFFu,
  REF_KIND_SHIFT = 12,
  REF_INDEX_MASK = 0x0FFFu,
  WREF_LEAF = 0x80000000u,
  WREF_EMPTY = 0xF00FFFFFu,  // leaf bit, kind 7, index 0xFFFFF: 24 significant bits like every wide reference
  WREF_KIND_SHIFT = 28,
  WREF_INDEX_MASK = 0x0FFFFFFFu,
};
template <bool WIDE> struct RefT;
template <> struct RefT<false> {
  static constexpr uint32_t STACK_ENTRY = 2;  // bytes of a traversal-stack entry
  static constexpr uint32_t LEAF = REF_LEAF, EMPTY = REF_EMPTY, KIND_SHIFT = REF_KIND_SHIFT,
                            INDEX_MASK = REF_INDEX_MASK;
};
// HBM store: the image holds references in 24 bits, leaf bit 23, kind in 22-20, index in 19-0
// (wide_ref24 of the builder's 32-bit form), so that a traversal-stack entry of three bytes is
// the reference itself and needs no packing.
template <> struct RefT<true> {
  static constexpr uint32_t STACK_ENTRY = 3;
  static constexpr uint32_t LEAF = 0x800000u, EMPTY = 0xFFFFFFu, KIND_SHIFT = 20, INDEX_MASK = 0xFFFFFu;
};
// 32-bit reference of the builder -> the 24-bit form of the HBM store's image
inline uint32_t wide_ref24(uint32_t w) { return (w & 0xFFFFFu) | ((w >> 8) & 0xF00000u); }
// 32-bit reference -> its 16-bit form (the host builds with 32-bit references)
inline uint32_t narrow_ref(uint32_t w) {
  if (w == WREF_EMPTY) return REF_EMPTY;
  if (w & WREF_LEAF) return REF_LEAF | (((w >> WREF_KIND_SHIFT) & 7u) << REF_KIND_SHIFT) | (w & REF_INDEX_MASK);
  return w;
}
enum : uint32_t {
  KIND_SPHERE = 0,      // c = |o-c|^2 - r^2 evaluated directly
  KIND_XYRECT = 1,
  KIND_XZRECT = 2,
  KIND_YZRECT = 3,
  KIND_TRIANGLE = 4,
  KIND_BIG_SPHERE = 5,  // c = |o|^2 - 2 o.c + K, K = |c|^2 - r^2 from the host in fp64:
                        // well conditioned when r dwarfs the rest of the scene
  KIND_TRI_PAIR = 6,    // a leaf of two triangles, i and i + 1 of the triangle array: one node
                        // visit fewer than two single leaves under a node of their own.  Only a
                        // node's child is ever a pair; a closest hit is recorded as the single
                        // triangle's reference.
  KIND_SPH_PAIR = 7,    // the same for two ordinary spheres (index < 0xFFF: kind 7 with all index
                        // bits set is the empty reference)
};
// SceneLayout::tree_kinds.  TREE_SPHERES means ordinary spheres only: a KIND_BIG_SPHERE
// inside the hierarchy sets TREE_MIXED as well, which sends leaves through the full dispatch.
enum : uint32_t { TREE_SPHERES = 1u, TREE_RECTS = 2u, TREE_TRIANGLES = 4u, TREE_MIXED = 8u };
constexpr uint32_t MAX_NODES = 0x8000u;        // on-chip store
constexpr uint32_t MAX_PER_KIND = 0x1000u;     // on-chip store
constexpr uint32_t WIDE_MAX_PER_KIND = 1u << 20;

// per-primitive meta word: material << 20 | primitive id
constexpr uint32_t META_PRIM_MASK = 0xFFFFFu;
constexpr uint32_t META_MAT_SHIFT = 20;
constexpr uint32_t MAX_PRIMS = 1u << 20;
constexpr uint32_t MAX_MATERIALS = 1u << 12;
constexpr uint32_t MAX_TEXTURES = 1u << 15;

// material word 0: kind | texture << 4 ; word 1 (R) : parameter
// texture record R4: data[0..2], word 3 low 32 bits: kind | child0 << 2 | child1 << 17

// node: the bounds of the two children, per axis four numbers, one pair for each child
//   R = double: x: lo0 lo1 hi0 hi1, then y, then z
//   R = float : x: centre0 centre1 half0 half1, then y, then z (half-extents rounded up)
// then the child references.
//   on-chip store: 12 R, then one u32 c0 | c1 << 16            (float 64 B, double 112 B)
//   HBM store, R = double: 12 doubles, then two u32             (112 B)
//   HBM store, R = float : 12 IEEE half floats rounded OUTWARD, then two u32 (32 B).
//     A traversal only needs boxes that contain the children; half-precision bounds
//     halve the cache footprint and the L2 traffic of a node visit.  Coordinates beyond
//     the half range become infinite bounds (always hit).  (On chip the conversion
//     instructions cost more than the LDS bandwidth they save: measured.)
constexpr int NODE_REALS = 12;
constexpr int STACK_MAX_DEPTH = 30;       // on-chip store
constexpr int WIDE_STACK_MAX_DEPTH = 48;  // HBM store

template <typename R>
constexpr uint32_t node_stride(bool wide) {
  return sizeof(R) == 4 ? (wide ? 32u : 64u) : 112u;
}

struct SceneLayout {
  uint32_t nodes;       // n_nodes * node_stride
  uint32_t sph;         // n_sph   * 4R   {cx, cy, cz, r}
  uint32_t sph_k;       // n_big   * R    K of the KIND_BIG_SPHERE entries, which come first in sph
  uint32_t rect;        // n_rect  * 8R   {a0, a1, b0, b1, k, -, -, -}
  uint32_t tri;         // n_tri   * 12R  {A, e1 = B-A, e2 = C-A, n = e1 x e2}; fp32: {A, (e2 x n)/|n|^2, (n x e1)/|n|^2, n}
  uint32_t meta_sph;    // n_sph   * u32
  uint32_t meta_rect;   // n_rect  * u32
  uint32_t meta_tri;    // n_tri   * u32
  uint32_t mat;         // n_mat   * 2R   {type | texture << 4 (low 32 bits), parameter}
  uint32_t tex;         // n_tex   * 4R
  uint32_t perlin;      // n_perlin * perlin_stride: 256 * 4R gradients, then 3*256 bytes perm
  uint32_t perlin_stride;
  uint32_t images;      // n_images * 4 u32 {width, height, offset_lo, offset_hi}
  uint32_t hoist;       // n_hoist * u32 leaf references tested by every ray before the BVH
  uint32_t root_box;    // 6R: bounds of everything inside the BVH (lo.xyz, hi.xyz)
  uint32_t total_bytes; // multiple of 16
  uint32_t n_nodes, n_sph, n_rect, n_tri, n_mat, n_tex, n_perlin, n_images, n_hoist;
  uint32_t depth;       // levels of traversal stack needed
  uint32_t has_uv_textures; // any Image texture in the scene
  uint32_t solid_textures_only;   // every texture is a SolidColor
  uint32_t scatter_diffuse_only;  // every material is Diffuse or Emission (Emission never scatters)
  uint32_t tree_kinds;  // TREE_* bits: kinds of primitive inside the hierarchy (hoisted ones excluded)
  uint32_t wide;        // 0: on-chip store (LDS, 16-bit references), 1: HBM store (32-bit references)
};

}  // namespace ht
