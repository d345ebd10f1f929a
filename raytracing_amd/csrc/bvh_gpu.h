// bvh_gpu.h — a linear BVH built on the device (Morton order + radix sort + the hierarchy
// of Karras 2012 + a bottom-up fit of the boxes): scene-load work for very large meshes,
// where the host's SAH build is what a first frame waits for.  The hierarchy it gives is
// looser than the host's (measured: see the option's description in hiptrace.h), so the
// host build stays the default.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ht {

// One inner node as the device writes it: the boxes of its two children and their
// references: an inner node's index, or LBVH_LEAF | index of the primitive in the input.
// Node 0 is the root.
constexpr uint32_t LBVH_LEAF = 0x80000000u;
struct LbvhNode {
  float lo[2][3], hi[2][3];
  uint32_t child[2];
};

// n >= 2 boxes (lo.xyz, hi.xyz per primitive, 6 floats each) in host memory; writes n - 1
// nodes to `nodes` (host memory).  `cent_lo` / `cent_hi`: bounds of the box centres.  The
// device must be current.  Synchronous.
hipError_t lbvh_build(uint32_t n, const float* boxes, const float cent_lo[3], const float cent_hi[3], LbvhNode* nodes);

}  // namespace ht
