// bvh_gpu.hip — see bvh_gpu.h.  HBM-bound integer work: keys, a radix sort (hipCUB), one
// thread per inner node for the hierarchy, one thread per leaf for the boxes.
#include "bvh_gpu.h"

#include <hipcub/hipcub.hpp>

namespace ht {
namespace {

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bit i of v -> bit 3 i
  uint64_t x = v & 0x1FFFFFu;
  x = (x | (x << 32)) & 0x001F00000000FFFFull;
  x = (x | (x << 16)) & 0x001F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

struct Grid {
  float lo[3], scale[3];  // cell = (centre - lo) * scale, 2^21 cells per axis
};

__global__ void __launch_bounds__(256) lbvh_keys_kernel(uint32_t n, const float* __restrict__ boxes, Grid g,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ order) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = boxes + (size_t)i * 6u;
  uint32_t cell[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float c = (0.5f * (b[a] + b[3 + a]) - g.lo[a]) * g.scale[a];
    cell[a] = (uint32_t)fminf(fmaxf(c, 0.0f), 2097151.0f);  // NaN -> 0
  }
  keys[i] = (spread21(cell[0]) << 2) | (spread21(cell[1]) << 1) | spread21(cell[2]);
  order[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; equal keys are told
// apart by their positions, so the order is total (Karras 2012, section 4)
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = keys[i], b = keys[j];
  if (a != b) return __clzll((long long)(a ^ b));
  return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// inner node i covers a range of the sorted leaves that has i at one end
__global__ void __launch_bounds__(256) lbvh_hierarchy_kernel(int n, const uint64_t* __restrict__ keys,
                                                             uint32_t* __restrict__ child, uint32_t* __restrict__ node_parent,
                                                             uint32_t* __restrict__ leaf_parent) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n - 1) return;
  const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = common_prefix(keys, n, i, i - d);
  int lmax = 2;
  while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;  // ends: out of range gives -1
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = common_prefix(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int split = i + s * d + (d < 0 ? -1 : 0);  // last leaf of the left child
  const int first = i < j ? i : j, last = i < j ? j : i;
  const bool left_leaf = first == split, right_leaf = last == split + 1;
  child[2 * i] = left_leaf ? (LBVH_LEAF | (uint32_t)split) : (uint32_t)split;
  child[2 * i + 1] = right_leaf ? (LBVH_LEAF | (uint32_t)(split + 1)) : (uint32_t)(split + 1);
  if (left_leaf) leaf_parent[split] = (uint32_t)i; else node_parent[split] = (uint32_t)i;
  if (right_leaf) leaf_parent[split + 1] = (uint32_t)i; else node_parent[split + 1] = (uint32_t)i;
}

struct Box6 {
  float v[6];
};
// a box another thread wrote: read past this CU's vector cache
__device__ __forceinline__ Box6 load_box_coherent(const float* p) {
  Box6 b;
#pragma unroll
  for (int a = 0; a < 6; ++a) b.v[a] = __hip_atomic_load(p + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return b;
}

// One thread per leaf climbs towards the root; at every inner node the first of the two
// threads to arrive stops, the second (both subtrees are complete then) writes the node.
// Every node is written exactly once and no thread waits for another.
__global__ void __launch_bounds__(256) lbvh_fit_kernel(int n, const float* __restrict__ boxes, const uint32_t* __restrict__ order,
                                                       const uint32_t* __restrict__ child, const uint32_t* __restrict__ node_parent,
                                                       const uint32_t* __restrict__ leaf_parent, uint32_t* __restrict__ arrived,
                                                       float* node_box, LbvhNode* __restrict__ nodes) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  uint32_t p = leaf_parent[k];
  for (int guard = 0; guard < 4096; ++guard) {  // a hierarchy is never this deep; a broken one must not spin
    __threadfence();  // what this thread wrote below is visible before its arrival is
    if (atomicAdd(&arrived[p], 1u) == 0u) return;
    __threadfence();
    LbvhNode nd;
    Box6 u;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint32_t r = child[2 * p + c];
      Box6 b;
      if (r & LBVH_LEAF) {
        const uint32_t item = order[r & ~LBVH_LEAF];
        const float* q = boxes + (size_t)item * 6u;
#pragma unroll
        for (int a = 0; a < 6; ++a) b.v[a] = q[a];
        nd.child[c] = LBVH_LEAF | item;
      } else {
        b = load_box_coherent(node_box + (size_t)r * 6u);
        nd.child[c] = r;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        nd.lo[c][a] = b.v[a];
        nd.hi[c][a] = b.v[3 + a];
        u.v[a] = c == 0 ? b.v[a] : fminf(u.v[a], b.v[a]);
        u.v[3 + a] = c == 0 ? b.v[3 + a] : fmaxf(u.v[3 + a], b.v[3 + a]);
      }
    }
    nodes[p] = nd;
#pragma unroll
    for (int a = 0; a < 6; ++a) __hip_atomic_store(node_box + (size_t)p * 6u + a, u.v[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == 0u) return;
    p = node_parent[p];
  }
}

struct DeviceArrays {
  void* p[12] = {};
  int used = 0;
  ~DeviceArrays() {
    for (int i = 0; i < used; ++i)
      if (p[i]) (void)hipFree(p[i]);
  }
  template <typename T>
  hipError_t get(T** out, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) ? count * sizeof(T) : 16);
    if (e != hipSuccess) return e;
    p[used++] = q;
    *out = (T*)q;
    return hipSuccess;
  }
};

}  // namespace

#define LBVH_TRY(x)                   \
  do {                                \
    hipError_t e_ = (x);              \
    if (e_ != hipSuccess) return e_;  \
  } while (0)

hipError_t lbvh_build(uint32_t n, const float* boxes, const float cent_lo[3], const float cent_hi[3], LbvhNode* nodes) {
  if (n < 2u || n > (1u << 30)) return hipErrorInvalidValue;
  DeviceArrays mem;
  float *d_boxes, *d_node_box;
  uint64_t *d_keys, *d_keys_sorted;
  uint32_t *d_order, *d_order_sorted, *d_child, *d_node_parent, *d_leaf_parent, *d_arrived;
  LbvhNode* d_nodes;
  LBVH_TRY(mem.get(&d_boxes, (size_t)n * 6));
  LBVH_TRY(mem.get(&d_keys, n));
  LBVH_TRY(mem.get(&d_keys_sorted, n));
  LBVH_TRY(mem.get(&d_order, n));
  LBVH_TRY(mem.get(&d_order_sorted, n));
  LBVH_TRY(mem.get(&d_child, (size_t)2 * (n - 1)));
  LBVH_TRY(mem.get(&d_node_parent, n - 1));
  LBVH_TRY(mem.get(&d_leaf_parent, n));
  LBVH_TRY(mem.get(&d_arrived, n - 1));
  LBVH_TRY(mem.get(&d_node_box, (size_t)(n - 1) * 6));
  LBVH_TRY(mem.get(&d_nodes, n - 1));
  LBVH_TRY(hipMemcpy(d_boxes, boxes, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice));
  LBVH_TRY(hipMemset(d_arrived, 0, (size_t)(n - 1) * sizeof(uint32_t)));
  // a node the kernels never wrote must not look like one: all ones fail the host's index checks
  LBVH_TRY(hipMemset(d_nodes, 0xFF, (size_t)(n - 1) * sizeof(LbvhNode)));
  LBVH_TRY(hipMemset(d_node_parent, 0, (size_t)(n - 1) * sizeof(uint32_t)));
  LBVH_TRY(hipMemset(d_leaf_parent, 0, (size_t)n * sizeof(uint32_t)));

  Grid g;
  for (int a = 0; a < 3; ++a) {
    const float ext = cent_hi[a] - cent_lo[a];
    g.lo[a] = cent_lo[a];
    g.scale[a] = ext > 0.0f && ext < 3.0e38f ? 2097152.0f / ext : 0.0f;
  }
  const uint32_t blocks = (n + 255u) / 256u;
  hipLaunchKernelGGL(lbvh_keys_kernel, dim3(blocks), dim3(256), 0, 0, n, d_boxes, g, d_keys, d_order);
  LBVH_TRY(hipGetLastError());

  size_t temp_bytes = 0;
  LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_order, d_order_sorted, (int)n, 0, 63, 0));
  void* d_temp = nullptr;
  LBVH_TRY(mem.get((unsigned char**)&d_temp, temp_bytes));
  LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_order, d_order_sorted, (int)n, 0, 63, 0));

  hipLaunchKernelGGL(lbvh_hierarchy_kernel, dim3((n - 1 + 255u) / 256u), dim3(256), 0, 0, (int)n, d_keys_sorted, d_child,
                     d_node_parent, d_leaf_parent);
  LBVH_TRY(hipGetLastError());
  hipLaunchKernelGGL(lbvh_fit_kernel, dim3(blocks), dim3(256), 0, 0, (int)n, d_boxes, d_order_sorted, d_child, d_node_parent,
                     d_leaf_parent, d_arrived, d_node_box, d_nodes);
  LBVH_TRY(hipGetLastError());
  LBVH_TRY(hipMemcpy(nodes, d_nodes, (size_t)(n - 1) * sizeof(LbvhNode), hipMemcpyDeviceToHost));
  return hipSuccess;
}

}  // namespace ht
