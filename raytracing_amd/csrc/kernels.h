// kernels.h — host-callable launchers of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scene_layout.h"
#include "work_queue.h"

namespace ht {

// Camera basis of Camera.scala:5-17 (reference), computed on the host in fp64
// and rounded to the kernel's arithmetic type.
// corner = lower_left_corner - origin, subtracted in fp64 before the rounding: a ray
// direction is corner + s*horizontal + t*vertical - lens offset, two fused multiply-adds per
// component, and carries about one ulp of error instead of the three of the reference's
// order of operations in fp32.
template <typename R>
struct CamDev {
  R origin[3], corner[3], horizontal[3], vertical[3], u[3], v[3], background[3];
  R lens_radius;
};

struct RenderArgs {
  const uint8_t* blob;     // scene image in global memory (copied to LDS per workgroup unless L.wide)
  SceneLayout L;
  const uint32_t* texels;  // image texels, 0x00RRGGBB
  long long* accum;        // [H*W*3] radiance sums, signed 32.32 fixed point, row 0 = bottom
  uint32_t* tile_counter;  // work-unit queue head, set to the number of waves of the launch before it starts
                           //   (wave w takes unit w first, then asks the counter)
  unsigned long long* counters;  // NUM_COUNTERS words, or nullptr
  int32_t W, H;
  int32_t sample_first;    // global index of the first sample of this launch
  int32_t sample_count;    // samples per pixel in this launch
  int32_t chunk;           // samples per work unit
  int32_t chunk_tail;      // samples per work unit among the last `chunk` samples of the launch
  int32_t max_depth;
  int32_t jitter;
  uint32_t seed_lo, seed_hi;
  uint32_t philox_keys[20];  // round keys of Philox4x32-10 for that seed: (lo + r*0x9E3779B9, hi + r*0xBB67AE85)
  int32_t wait_threshold;  // lanes that may wait for shading before traversal yields
  int32_t node_den;        // node steps go on while inner lanes * node_den >= travelling lanes
  int32_t reserved1, reserved2;
  float inv_w1, inv_h1;    // 1/(W-1), 1/(H-1) correctly rounded (jittered fp32 sample positions)
  int32_t tile_first;      // this launch renders the 8x8 tiles tile_first, tile_first + tile_stride, ... of the
  int32_t tile_stride;     //   frame (row-major tile numbers); 0 / 1: every tile
  uint32_t* timeline;      // nullptr, or 3 x TIMELINE_BINS counters of a profiling run: work units fetched,
                           //   waves that found the queue empty, waves that ended, per time bin
  QueueGeom q;             // geometry of the work queue of this launch (work_queue.h)
};
// a timeline bin is 1024 ticks of the 100 MHz s_memrealtime clock (10.24 us); bin = (ticks >> 10) mod TIMELINE_BINS
constexpr uint32_t TIMELINE_BINS = 4096;

// ---- wavefront variant: the same paths as render_kernel, one kernel per stage ----------
// Path state lives in global memory between the stages, structure-of-arrays with `cap`
// elements per array:
//   ray  [6][cap]  origin, direction            thr [3][cap]  throughput T
//   ids  [2][cap]  item (pixel + sample * W*H within the launch), bounce
//   hit_t [cap], hit_ref [cap]                  closest hit of the ray (hit_ref = empty: miss)
//   contrib [3][items]                          what the path added to its pixel, stored once
struct WfBuffers {
  float* ray;
  float* thr;
  uint32_t* ids;
};
struct WfArgs {
  const uint8_t* blob;
  SceneLayout L;
  const uint32_t* texels;
  WfBuffers cur, nxt;
  float* hit_t;
  uint32_t* hit_ref;
  float* contrib;
  uint32_t* counters;     // [0]: paths written to nxt by the shade stage, [1]: next ray of the trace stage
  long long* accum;
  uint32_t cap;           // elements per array
  uint32_t n;             // paths in cur
  uint32_t n_new, slot0;  // generate: paths to start and where the first one goes in cur
  uint32_t item0;         // generate: first item to start
  uint32_t items;         // items of the launch = W*H * samples
  uint32_t WH;
  int32_t W, H;
  int32_t sample_first;   // global index of the launch's first sample
  int32_t max_depth, jitter;
  float inv_w1, inv_h1;   // 1/(W-1), 1/(H-1)
  int32_t refill;         // trace: idle lanes of a wave that trigger a refill from the ray list
  uint32_t philox_keys[20];
};

struct FirstHitArgs {
  const uint8_t* blob;
  SceneLayout L;
  int32_t W, H;
  void* t;        // R [H*W]
  int32_t* prim;  // [H*W]
  void* normal;   // R [H*W*3]
};

enum {
  CTR_RAYS = 0, CTR_BOX, CTR_SPHERE, CTR_RECT, CTR_TRI,
  CTR_SHADE0, CTR_SHADE1, CTR_SHADE2, CTR_SHADE3,
  CTR_TEX0, CTR_TEX1, CTR_TEX2, CTR_TEX3,
  CTR_STEPS, CTR_ACTIVE, CTR_SHADE_ROUNDS, CTR_REGEN_ROUNDS, CTR_MISS,
  CTR_NODE_STEPS,                              // lane slots offered by node steps (the rest of CTR_STEPS: leaf steps)
  CTR_CYC_TRAV, CTR_CYC_SHADE, CTR_CYC_REGEN,  // wave cycles per phase (lane 0 of every wave)
  CTR_CLK_SHADER, CTR_CLK_REAL,  // s_memtime / s_memrealtime ticks of workgroup 0
  NUM_COUNTERS
};

constexpr int BLOCK_THREADS = 1024;
constexpr int TILE_W = 8, TILE_H = 8;

// dynamic LDS bytes a render / first-hit launch of `threads` threads needs for layout L
uint32_t lds_bytes_needed(const SceneLayout& L, uint32_t threads);
uint32_t max_block_threads(int precision);

// texture features a kernel variant is built for
constexpr uint32_t KFEAT_PERLIN = 1u, KFEAT_IMAGE = 2u;
constexpr uint32_t KFEAT_TRIS = 4u;  // the hierarchy holds triangles only (used when no texture bit is set)
constexpr uint32_t KFEAT_LEAN = 8u;  // the variant held to 80 scalar registers, eight waves per SIMD: fp32, on-chip store, plain textures
constexpr uint32_t KFEAT_FLAT = 16u;  // the scene has no hierarchy: variants without a traversal phase where they exist

// Number of workgroups of `threads` threads per CU the device can hold for this
// kernel variant and LDS size (0: does not fit).
int render_blocks_per_cu(int precision, bool counters, uint32_t feat, bool wide, int threads,
                         uint32_t lds_bytes);

// Waves of that kernel variant a CU really holds, by the scalar registers it is compiled to
// (whatever the occupancy query says): 32 up to 80 of them, 28 up to 96, 24 beyond.
int render_max_waves_per_cu(int precision, bool counters, uint32_t feat, bool wide);
// name of the kernel a (precision, counters build, features, store) runs, as a profiler prints it
// ("render_kernel_8<4u>"); returns the scalar-register budget it is compiled to (80, 96; 0: none)
int render_kernel_name(int precision, bool counters, uint32_t feat, bool wide, char* out, size_t n);
// whether `feat | KFEAT_LEAN` names a kernel of its own for this precision / store
bool render_has_lean_variant(int precision, bool counters, uint32_t feat, bool wide);

// a.L.wide / the layout's store selects the kernel variant
hipError_t launch_render(int precision, bool counters, uint32_t feat, const RenderArgs& a, const void* cam,
                         int grid_blocks, int threads, uint32_t lds_bytes, hipStream_t stream);
hipError_t launch_first_hit(int precision, const FirstHitArgs& a, const void* cam, int threads,
                            uint32_t lds_bytes, hipStream_t stream);
// dst[i] += src[i] over n radiance sums (combining the partial frames of sample shards)
hipError_t launch_accum_add(long long* dst, const long long* src, size_t n, hipStream_t stream);
// dst[i] += sum over j < count of src[j * n + i]: the other shards' copies of one slice of the frame
hipError_t launch_accum_add_many(long long* dst, const long long* src, size_t n, int count, hipStream_t stream);

// wavefront stages (fp32 arithmetic).  `feat` as for launch_render; the store is a.L.wide.
hipError_t launch_wf_generate(const WfArgs& a, const CamDev<float>& cam, hipStream_t stream);
hipError_t launch_wf_trace(uint32_t feat, const WfArgs& a, int grid_blocks, int threads, uint32_t lds_bytes,
                           hipStream_t stream);
hipError_t launch_wf_shade(uint32_t feat, const WfArgs& a, const CamDev<float>& cam, hipStream_t stream);
hipError_t launch_wf_reduce(const WfArgs& a, uint32_t samples, hipStream_t stream);

// The library's static device array (see kernels.hip): the address of the current device's
// copy, looked up once per device.  Layout: radiance sums of HT_NOALLOC_MAX_PIXELS pixels
// (3 x int64 each), then STATIC_QUEUE_SLOTS work-queue heads of 256 bytes.
constexpr uint32_t STATIC_QUEUE_SLOTS = 16;
hipError_t static_scratch(int device, void** out);
// mean + Vec3.toRGB + row flip (Vec3.scala:82-84, main.scala:93)
hipError_t launch_resolve(const long long* accum, int32_t W, int32_t H, double inv_samples,
                          float* rgb_mean, uint32_t* rgb_packed, hipStream_t stream);

// Tile sharding.  The frame's 8x8 tiles are numbered row-major; shard g of n owns the tiles
// g, g + n, g + 2n, ...  A shard's picture is kept compact: owned tile k (tile number
// g + k n) occupies pixels [64 k, 64 k + 64) of a tile buffer, pixel q = (row within the tile,
// bottom row first) * 8 + column; pixels beyond the frame's edge are zero.
inline uint32_t tiles_in_frame(int32_t W, int32_t H) {
  return (uint32_t)((W + TILE_W - 1) / TILE_W) * (uint32_t)((H + TILE_H - 1) / TILE_H);
}
inline uint32_t tiles_owned(uint32_t n_tiles, uint32_t first, uint32_t stride) {
  return first < n_tiles ? (n_tiles - first + stride - 1u) / stride : 0u;
}
// mean and Vec3.toRGB code of the owned tiles' pixels, in tile order (no row flip yet)
hipError_t launch_resolve_tiles(const long long* accum, int32_t W, int32_t H, double inv_samples, uint32_t tile_first,
                                uint32_t tile_stride, float* mean_tiles, uint32_t* packed_tiles, hipStream_t stream);
// the frame from the tile buffers of n shards (buffer g starts g * shard_pixels pixels into the arrays):
// rgb_mean row 0 = bottom, rgb_packed row 0 = top (the row flip of main.scala:93)
hipError_t launch_assemble_tiles(int32_t W, int32_t H, uint32_t n_shards, size_t shard_pixels, const float* mean_tiles,
                                 const uint32_t* packed_tiles, float* rgb_mean, uint32_t* rgb_packed,
                                 hipStream_t stream);

}  // namespace ht
