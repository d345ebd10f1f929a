// host_scene.h — host side of a scene handle: validated copy of the caller's
// description, the library's own BVH, and the two on-chip scene images.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/hiptrace.h"
#include "scene_layout.h"

namespace ht {

struct BvhNode {
  double lo[2][3], hi[2][3];  // boxes of the two children
  uint32_t child[2];          // 32-bit references (scene_layout.h); narrowed when a blob is written
};

struct HostScene {
  // validated copies (fp64, as handed over)
  std::vector<int32_t> prim_type, prim_material;
  std::vector<double> prim_data;
  std::vector<int32_t> mat_type, mat_texture;
  std::vector<double> mat_param;
  std::vector<int32_t> tex_type, tex_child;
  std::vector<double> tex_data;
  std::vector<int32_t> image_width, image_height;
  std::vector<int64_t> image_offset;
  std::vector<uint32_t> image_texels;
  std::vector<double> perlin_ranvec;
  std::vector<int32_t> perlin_perm;
  int32_t n_perlin = 0;

  // per-kind arrays: primitive ids in kind-local order
  std::vector<int32_t> sph, rect, tri;
  std::vector<uint8_t> sph_big;     // KIND_BIG_SPHERE flag per entry of sph
  std::vector<uint32_t> prim_ref;   // leaf reference (32-bit form) of every primitive

  std::vector<uint32_t> hoist;      // leaf references every ray tests linearly (not in the BVH)
  std::vector<BvhNode> nodes;       // hierarchy over the remaining primitives (may be empty)
  double root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
  int depth = 0;
  bool external_bvh_used = false;   // the hierarchy came from BuildOptions::external_bvh
  double sah_before = 0, sah_after = 0;  // sum of the inner nodes' box half-areas before / after the optimiser

  SceneLayout layout32{}, layout64{};
  std::vector<uint8_t> blob32, blob64;
};

// A hierarchy built elsewhere (on the device: bvh_gpu.h).  Inner node: the boxes of its two
// children and their references: an inner node's index, or EXTERNAL_LEAF | index of the
// primitive among the n boxes handed over.  Node 0 is the root, n - 1 nodes in all.
constexpr uint32_t EXTERNAL_LEAF = 0x80000000u;
struct ExternalNode {
  float lo[2][3], hi[2][3];
  uint32_t child[2];
};
// boxes: n x (lo.xyz, hi.xyz), rounded outward; cent_lo / cent_hi: bounds of the box centres.
// Returns false when it could not build (the host builds then).
typedef bool (*ExternalBvhBuilder)(void* ctx, uint32_t n, const float* boxes, const float cent_lo[3],
                                   const float cent_hi[3], ExternalNode* nodes);

struct BuildOptions {
  uint32_t lds_bytes = 160u * 1024u;  // LDS one workgroup may use on the target device
  bool force_hbm = false;             // put the scene image in the HBM store even if it fits on chip
  // hierarchies over at least external_min_prims primitives are built by external_bvh, if set
  // (its result is checked, and the host's own builder takes over when it does not hold up)
  ExternalBvhBuilder external_bvh = nullptr;
  void* external_ctx = nullptr;
  int external_min_prims = 0;
  // insertion-based optimisation of the finished hierarchy: passes over all nodes (0: off), for
  // hierarchies of at most optimise_max_nodes inner nodes
  int optimise_passes = 4;
  int optimise_max_nodes = 1 << 13;
};

// Validates `desc` and fills `out`; returns HT_OK or an error code with `err` set.
// Each precision's image goes to the on-chip store if it fits there (reference
// limits, LDS budget next to the stacks of a 256-thread workgroup), else to the
// HBM store (layoutNN.wide).
int build_host_scene(const ht_scene_desc* desc, HostScene& out, std::string& err,
                     const BuildOptions& opt = BuildOptions());

// Camera.scala:5-17 in fp64: origin, lower_left_corner, horizontal, vertical, u, v, w, lens_radius
struct CameraBasis {
  double origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3], background[3];
  double lens_radius;
};
void camera_basis(const ht_camera& cam, CameraBasis& out);

}  // namespace ht
