// scene_layout.h — the on-chip scene image shared by the host flattener and the
// device kernels.
//
// The whole scene (BVH nodes, primitives, materials, textures, Perlin tables)
// is one byte blob that every workgroup copies into LDS once; the kernels
// address it by the byte offsets below.  Image texels stay in global memory.
// Two blobs exist per scene: R = float (the production path) and R = double
// (the validation path); the layout is the same with sizeof(R) scaled.
#pragma once
#include <stdint.h>

namespace ht {

// ---- child / primitive references (16 bits) -------------------------------
//   inner node : index                    (bit 15 clear, < 0x8000)
//   leaf       : 0x8000 | kind << 12 | i  (i = index into the kind's array)
//   empty      : 0xFFFF
enum : uint32_t {
  REF_LEAF = 0x8000u,
  REF_EMPTY = 0xFFFFu,
  REF_KIND_SHIFT = 12,
  REF_INDEX_MASK = 0x0FFFu,
};
enum : uint32_t {
  KIND_SPHERE = 0,      // c = |o-c|^2 - r^2 evaluated directly
  KIND_XYRECT = 1,
  KIND_XZRECT = 2,
  KIND_YZRECT = 3,
  KIND_TRIANGLE = 4,
  KIND_BIG_SPHERE = 5,  // c = |o|^2 - 2 o.c + K, K = |c|^2 - r^2 from the host in fp64:
                        // well conditioned when r dwarfs the rest of the scene
};
constexpr uint32_t MAX_NODES = 0x8000u;
constexpr uint32_t MAX_PER_KIND = 0x1000u;

// per-primitive meta word: material << 20 | primitive id
constexpr uint32_t META_PRIM_MASK = 0xFFFFFu;
constexpr uint32_t META_MAT_SHIFT = 20;
constexpr uint32_t MAX_PRIMS = 1u << 20;
constexpr uint32_t MAX_MATERIALS = 1u << 12;
constexpr uint32_t MAX_TEXTURES = 1u << 15;

// material word 0: kind | texture << 4 ; word 1 (R) : parameter
// texture record R4: data[0..2], word 3 low 32 bits: kind | child0 << 2 | child1 << 17

constexpr int NODE_REALS = 12;  // lo0.xyz hi0.xyz lo1.xyz hi1.xyz, then one u32: c0 | c1 << 16
constexpr int STACK_MAX_DEPTH = 30;

template <typename R>
constexpr uint32_t node_stride() {
  return sizeof(R) == 4 ? 64u : 112u;
}

struct SceneLayout {
  uint32_t nodes;       // n_nodes * node_stride
  uint32_t sph;         // n_sph   * 4R   {cx, cy, cz, r}
  uint32_t sph_k;       // n_sph   * R    K (KIND_BIG_SPHERE only)
  uint32_t rect;        // n_rect  * 8R   {a0, a1, b0, b1, k, -, -, -}
  uint32_t tri;         // n_tri   * 12R  {A, e1 = B-A, e2 = C-A, n = e1 x e2}
  uint32_t meta_sph;    // n_sph   * u32
  uint32_t meta_rect;   // n_rect  * u32
  uint32_t meta_tri;    // n_tri   * u32
  uint32_t mat;         // n_mat   * 4R
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
};

}  // namespace ht
