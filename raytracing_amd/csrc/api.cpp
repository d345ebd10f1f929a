// api.cpp — the C ABI of libhiptrace.so (include/hiptrace.h).
//
// Replaces the body of renderFrame (reference: jvm/src/main/scala/main.scala:73-99).
// There is no CPU path in here: without a HIP device every compute entry point
// fails with HT_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hiptrace.h"
#include "host_scene.h"
#include "bvh_gpu.h"
#include "kernels.h"

using namespace ht;

// Launch shape of the render kernel for one (precision, counters) variant of a scene.
struct LaunchShape {
  int threads = 0, bpc = 0;
  uint32_t lds = 0;
  uint32_t kfeat = 0;   // the kernel variant the shape is for
  int generation = -1;  // option generation it was worked out under
};

// Work-queue heads: every launch takes the next one of a ring, so launches of one scene
// handle may overlap on different streams (up to QUEUE_SLOTS of them in flight).
constexpr uint32_t QUEUE_SLOTS = 64, QUEUE_SLOT_BYTES = 256;

// buffers of a handle's share of a sharded render (ht_scene::d_shard)
enum { SHARD_ACCUM = 0, SHARD_STAGING, SHARD_MEAN_TILES, SHARD_PACKED_TILES, SHARD_GATHERED, SHARD_BUFFERS };

struct ht_scene {
  HostScene hs;
  int device = 0;
  int n_cu = 0;
  uint8_t* d_blob32 = nullptr;
  uint8_t* d_blob64 = nullptr;
  uint32_t* d_texels = nullptr;
  uint8_t* d_queue_heads = nullptr;  // QUEUE_SLOTS heads, QUEUE_SLOT_BYTES apart
  std::atomic<uint32_t> next_head{0};
  unsigned long long* d_counters = nullptr;
  std::atomic<int> cancel{0};
  uint32_t feat = 0;  // KFEAT_* texture kinds present in the scene
  std::mutex shape_mutex;
  LaunchShape shape[2][2];  // [precision][counters]
  // workspace of the wavefront variant (path state between the stage kernels), grown on demand;
  // wavefront renders of one handle run one at a time (wf_mutex)
  uint8_t* d_wf = nullptr;
  size_t wf_bytes = 0;
  std::mutex wf_mutex;
  // Device buffers of the host-buffer entry points (ht_render, ht_resolve, ht_frame_read), kept
  // between calls and grown on demand: a frame's sums, its means and its packed picture.  A call
  // holds host_mutex while it uses them (one host-buffer call per handle at a time).
  std::mutex host_mutex;
  int64_t* d_host_accum = nullptr;
  float* d_host_mean = nullptr;
  uint32_t* d_host_packed = nullptr;
  size_t host_accum_pixels = 0, host_mean_pixels = 0, host_packed_pixels = 0;
  // profiling: per-time-bin counters that the render kernels of this handle add to while the
  // option "timeline" is on (ht_debug_timeline reads and clears them)
  std::mutex timeline_mutex;
  uint32_t* d_timeline = nullptr;
  // Device buffers of ht_render_sharded (this handle's sums, staging and tile buffers), kept
  // between calls and grown on demand; a sharded render holds shard_mutex of every handle it uses
  std::mutex shard_mutex;
  void* d_shard[SHARD_BUFFERS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t shard_bytes[SHARD_BUFFERS] = {0, 0, 0, 0, 0};
  // HIP events of finished calls, reused by later ones
  std::mutex event_mutex;
  std::vector<hipEvent_t> timing_events, plain_events;
};

struct ht_frame {
  ht_scene* scene = nullptr;
  int32_t width = 0, height = 0;
  int64_t samples = 0;
  int64_t* d_accum = nullptr;
};

namespace {

// The calling thread's last error message: a fixed buffer, so that reporting an error
// allocates nothing.
thread_local char g_error[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof g_error, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (void)hipGetLastError();                                                                \
      return fail(HT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                  \
    }                                                                                         \
  } while (0)

// No C++ exception leaves an entry point: every body that can allocate host memory or
// start a thread runs inside these.
#define HT_API_TRY try {
#define HT_API_CATCH                                                         \
  }                                                                          \
  catch (const std::bad_alloc&) {                                            \
    return fail(HT_E_INVALID, "out of host memory");                         \
  }                                                                          \
  catch (const std::exception& ex) {                                         \
    return fail(HT_E_INVALID, "internal error: %s", ex.what());              \
  }                                                                          \
  catch (...) {                                                              \
    return fail(HT_E_INVALID, "internal error: unknown exception");          \
  }

// tuning knobs (ht_set_option)
std::atomic<int> g_wait_threshold{0};  // 0: by store (40 on chip, 32 in HBM, where a waiting lane costs less than a late round)
std::atomic<int> g_max_blocks_per_cu{8};
std::atomic<int> g_max_waves_per_cu{0};  // 0: what the kernel variant admits
std::atomic<int> g_block_threads{0};  // 0: choose
std::atomic<int> g_node_den{2};
std::atomic<int> g_default_batch{1024};
std::atomic<int> g_chunk{0};  // 0: by launch size
std::atomic<int> g_chunk_tail{4};
std::atomic<int> g_gpu_bvh_min{0};   // > 0: hierarchies over at least this many primitives are built on the device
std::atomic<int> g_force_hbm{0};     // scenes created while set keep their image in HBM
std::atomic<int> g_wavefront{0};      // 1: fp32 renders run the wavefront variant
std::atomic<int> g_wavefront_cap{1 << 24};  // paths in flight of the wavefront variant
std::atomic<int> g_wavefront_refill{32};    // idle lanes of a trace wave that trigger a refill
std::atomic<int> g_bvh_optimise_max_nodes{1 << 13};  // largest hierarchy (inner nodes) the optimiser is run over
std::atomic<int> g_flat_kernels{1};        // scenes without a hierarchy run the kernel variants without a traversal phase
std::atomic<int> g_bvh_optimise{4};        // passes of the hierarchy optimiser over a new scene's hierarchy (0: off)
std::atomic<int> g_timeline{0};            // 1: render launches record when their waves fetch units and end
std::atomic<int> g_xcd_bands{1};           // HBM store: a band of tiles and a queue head per XCD
std::atomic<int> g_shape_generation{0};  // bumped by the options a launch shape depends on

// Makes `device` current for the lifetime of the object, then restores.
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    ok = hipSetDevice(device) == hipSuccess;
    if (!ok) (void)hipGetLastError();
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

template <typename R>
void fill_cam(const CameraBasis& b, CamDev<R>& c) {
  for (int i = 0; i < 3; ++i) {
    c.origin[i] = (R)b.origin[i];
    c.corner[i] = (R)(b.llc[i] - b.origin[i]);
    c.horizontal[i] = (R)b.horizontal[i];
    c.vertical[i] = (R)b.vertical[i];
    c.u[i] = (R)b.u[i];
    c.v[i] = (R)b.v[i];
    c.background[i] = (R)b.background[i];
  }
  c.lens_radius = (R)b.lens_radius;
}

int check_camera(const ht_camera* cam) {
  if (!cam) return fail(HT_E_INVALID, "camera is NULL");
  const double* v = cam->origin;
  for (int i = 0; i < 3; ++i)
    if (!std::isfinite(v[i]) || !std::isfinite(cam->target[i]) || !std::isfinite(cam->vup[i]))
      return fail(HT_E_INVALID, "camera has a non-finite field");
  if (!std::isfinite(cam->vfov) || !std::isfinite(cam->aspect_ratio) || !std::isfinite(cam->aperture))
    return fail(HT_E_INVALID, "camera has a non-finite field");
  return HT_OK;
}

int check_params(const ht_scene* s, const ht_render_params* p) {
  if (p->width < 2 || p->height < 2)
    return fail(HT_E_INVALID, "width and height must be >= 2 (the reference divides by W-1 and H-1)");
  if ((int64_t)p->width * p->height > (int64_t)1 << 30) return fail(HT_E_INVALID, "image too large");
  if (p->samples < 1) return fail(HT_E_INVALID, "samples must be >= 1");
  if (p->sample_offset < 0) return fail(HT_E_INVALID, "sample_offset must be >= 0");
  if ((int64_t)p->sample_offset + (int64_t)p->samples > (int64_t)INT32_MAX)
    return fail(HT_E_INVALID, "sample_offset + samples exceeds 2^31 - 1");
  if (p->precision != HT_PRECISION_FP32 && p->precision != HT_PRECISION_FP64)
    return fail(HT_E_INVALID, "unknown precision %d", p->precision);
  if (s && p->device >= 0 && p->device != s->device)
    return fail(HT_E_INVALID, "scene lives on device %d, params ask for %d", s->device, p->device);
  if (p->tile_stride < 0 || p->tile_first < 0 || p->tile_first >= std::max(1, p->tile_stride))
    return fail(HT_E_INVALID, "tile_first must be in [0, max(tile_stride, 1)) (got first %d, stride %d)", p->tile_first,
                p->tile_stride);
  return HT_OK;
}

// Pixels of the frame inside the tiles first, first + stride, ... (edge tiles hold fewer than 64)
int64_t owned_pixels(int32_t W, int32_t H, int32_t first, int32_t stride) {
  if (stride <= 1) return (int64_t)W * H;
  const int32_t tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  int64_t n = 0;
  for (int64_t t = first; t < (int64_t)tiles_x * tiles_y; t += stride) {
    const int32_t x0 = (int32_t)(t % tiles_x) * TILE_W, y0 = (int32_t)(t / tiles_x) * TILE_H;
    n += (int64_t)std::min(TILE_W, W - x0) * std::min(TILE_H, H - y0);
  }
  return n;
}

// Events are taken from the scene's pool and handed back, so that a call creates none once
// the pool is warm.  Timing events only for callers that want statistics.
struct EventLease {
  ht_scene* s;
  bool timing;
  hipEvent_t ev = nullptr;
  EventLease(ht_scene* s_, bool timing_) : s(s_), timing(timing_) {}
  hipError_t acquire() {
    {
      std::lock_guard<std::mutex> lock(s->event_mutex);
      std::vector<hipEvent_t>& pool = timing ? s->timing_events : s->plain_events;
      if (!pool.empty()) { ev = pool.back(); pool.pop_back(); return hipSuccess; }
    }
    return timing ? hipEventCreate(&ev) : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  }
  ~EventLease() {
    if (!ev) return;
    try {
      std::lock_guard<std::mutex> lock(s->event_mutex);
      (timing ? s->timing_events : s->plain_events).push_back(ev);
    } catch (...) {
      (void)hipEventDestroy(ev);
    }
  }
};

// Grows one of the scene's cached device buffers to `pixels` pixels of `elem` bytes.
template <typename T>
hipError_t grow(T*& buf, size_t& have, size_t pixels, size_t elem) {
  if (have >= pixels && buf) return hipSuccess;
  if (buf) (void)hipFree(buf);
  buf = nullptr;
  have = 0;
  const hipError_t e = hipMalloc((void**)&buf, pixels * elem);
  if (e == hipSuccess) have = pixels;
  return e;
}

// One of a handle's sharded-render buffers, at least `bytes` large (the handle's device is current,
// its shard_mutex held).
hipError_t shard_buffer(ht_scene* s, int which, size_t bytes, void** out) {
  *out = nullptr;
  if (bytes == 0) return hipSuccess;
  if (s->shard_bytes[which] < bytes) {
    if (s->d_shard[which]) (void)hipFree(s->d_shard[which]);
    s->d_shard[which] = nullptr;
    s->shard_bytes[which] = 0;
    const hipError_t e = hipMalloc(&s->d_shard[which], bytes);
    if (e != hipSuccess) return e;
    s->shard_bytes[which] = bytes;
  }
  *out = s->d_shard[which];
  return hipSuccess;
}

// Locks the shard workspaces of all handles of a sharded render, in address order (two renders over
// the same handles in different orders cannot deadlock).
struct ShardLocks {
  std::vector<ht_scene*> held;
  ShardLocks(ht_scene* const* scenes, int n) {
    held.assign(scenes, scenes + n);
    std::sort(held.begin(), held.end());
    for (ht_scene* s : held) s->shard_mutex.lock();
  }
  ~ShardLocks() {
    for (auto it = held.rbegin(); it != held.rend(); ++it) (*it)->shard_mutex.unlock();
  }
};

// Workgroup size and count of the render kernel for this scene: the combination that keeps
// the most waves resident per CU; smaller groups win ties (measured on the Cornell box, 32
// waves per CU either way: 256 x 8 34.2, 512 x 4 32.8, 1024 x 2 27.7 G samples/s).  Worked out
// once per scene, precision and kernel variant (the occupancy queries cost host time on every
// batch of a progressive frame otherwise) and again only after ht_set_option changed one of
// the options it depends on.  The device must be current.
int launch_shape(ht_scene* s, int prec, bool counters, LaunchShape& out) {
  std::lock_guard<std::mutex> lock(s->shape_mutex);
  LaunchShape& c = s->shape[prec][counters ? 1 : 0];
  const int gen = g_shape_generation.load();
  if (c.generation != gen) {
    const SceneLayout& L = prec == HT_PRECISION_FP32 ? s->hs.layout32 : s->hs.layout64;
    const uint32_t base_feat = s->feat | (L.tree_kinds == TREE_TRIANGLES ? KFEAT_TRIS : 0u) | (L.n_nodes == 0u && g_flat_kernels.load() ? KFEAT_FLAT : 0u);
    LaunchShape best;
    const int forced = g_block_threads.load();
    // The production kernels are compiled to 96 scalar registers (seven waves per SIMD); the
    // on-chip ones with plain textures also to 80 (eight waves), 1-3 % slower at equal
    // occupancy because more scalars live in vector lanes.  That variant is taken when it buys
    // a larger shape (dragon: 1024 x 2 instead of 768 x 2, +5 %; scenes without an LDS image:
    // 256 x 8 instead of 256 x 7, within 1 %), not otherwise (random.json: 768 x 2 either way).
    for (int lean = 0; lean <= (render_has_lean_variant(prec, counters, base_feat, L.wide != 0u) ? 1 : 0); ++lean) {
      const uint32_t kfeat = base_feat | (lean ? KFEAT_LEAN : 0u);
      for (int t = 256; t <= (int)max_block_threads(prec); t += 128) {
        if (forced > 0 && t != forced) continue;
        const uint32_t need = lds_bytes_needed(L, (uint32_t)t);
        int b = render_blocks_per_cu(prec, counters, kfeat, L.wide != 0u, t, need);
        b = std::min(b, g_max_blocks_per_cu.load());
        // several workgroups share a CU well only when each fills the four SIMDs evenly: two
        // groups of 14 waves (measured on the Cornell box) ran 20 % slower than two of 12
        if (b > 1 && t % 256 != 0) b = 1;
        // A SIMD admits fewer waves of a kernel with many scalar registers than the occupancy
        // API answers (it says eight for all of these): a workgroup promised beyond the real
        // limit is accepted and never becomes resident (measured on the dragon: "1024 x 2" of
        // the 96-register kernel 20.1, 768 x 2 24.3 G samples/s).
        const int forced_waves = g_max_waves_per_cu.load();
        const int max_waves = forced_waves > 0 ? forced_waves : render_max_waves_per_cu(prec, counters, kfeat, L.wide != 0u);
        if (b * t > max_waves * 64) b = max_waves * 64 / t;
        if (b * t > best.bpc * best.threads) { best.threads = t; best.bpc = b; best.lds = need; best.kfeat = kfeat; }
      }
    }
    best.generation = gen;
    c = best;
  }
  if (c.bpc <= 0) {
    const SceneLayout& L = prec == HT_PRECISION_FP32 ? s->hs.layout32 : s->hs.layout64;
    return fail(HT_E_TOO_LARGE, "the render kernel cannot be resident with this scene (%u bytes of scene image)",
                L.total_bytes);
  }
  out = c;
  return HT_OK;
}

// Everything the launches of one call share.
struct RenderPlan {
  RenderArgs a;
  CamDev<float> cam32;
  CamDev<double> cam64;
  const void* camp = nullptr;
  LaunchShape shape;
  uint32_t kfeat = 0;
  int prec = 0;
  bool counters = false;
  int grid = 0;
  int batch = 0;
};

int make_plan(ht_scene* s, const ht_camera* cam, const ht_render_params* p, int64_t* accum_dev, RenderPlan& pl) {
  pl.prec = p->precision;
  pl.counters = p->collect_counters != 0;
  int rc = launch_shape(s, pl.prec, pl.counters, pl.shape);
  if (rc != HT_OK) return rc;
  const SceneLayout& L = pl.prec == HT_PRECISION_FP32 ? s->hs.layout32 : s->hs.layout64;
  pl.kfeat = pl.shape.kfeat;
  pl.grid = pl.shape.bpc * s->n_cu;

  CameraBasis basis;
  camera_basis(*cam, basis);
  fill_cam(basis, pl.cam32);
  fill_cam(basis, pl.cam64);
  pl.camp = pl.prec == HT_PRECISION_FP32 ? (const void*)&pl.cam32 : (const void*)&pl.cam64;

  RenderArgs& a = pl.a;
  memset(&a, 0, sizeof a);
  a.blob = pl.prec == HT_PRECISION_FP32 ? s->d_blob32 : s->d_blob64;
  a.L = L;
  a.texels = s->d_texels;
  a.accum = (long long*)accum_dev;
  // Samples per work unit.  A unit's tile sums are flushed to the frame once, so larger
  // units mean fewer 64-bit atomics (the only HBM traffic of an on-chip scene); they balance
  // worse when a launch holds few of them per tile.  Measured on 1024-spp launches: 16 is
  // 1 % faster than 8 on the dragon and the Cornell box, 32 equals 8; on 256-spp launches 8 wins.
  const int launch_spp = std::min(p->batch > 0 ? p->batch : g_default_batch.load(), p->samples);
  // (a tile shard holds 1/tile_stride of the frame's units: it is sized like a launch of that many fewer samples)
  a.chunk = g_chunk.load() > 0 ? g_chunk.load() : (launch_spp / std::max(1, (int)p->tile_stride) >= 1024 ? 16 : 8);
  if (g_chunk.load() <= 0) {
    // A small launch gets units small enough that every wave has one: the reference's own 400 x 225 x 16
    // picture is 1 450 tiles x 16 samples for 7 168 waves, and with units of eight samples most waves
    // found the queue empty while the others walked 512 paths each, one after the other (0.45 ms per
    // frame; units of 4: 0.34, of 2: 0.35; of 1: 0.46, 23 200 units through one queue head).
    const int64_t tiles = (int64_t)tiles_owned(tiles_in_frame(p->width, p->height), (uint32_t)p->tile_first,
                                               (uint32_t)std::max(1, (int)p->tile_stride));
    const int64_t n_waves = std::max<int64_t>(1, (int64_t)pl.grid * (pl.shape.threads / 64));
    a.chunk = (int)std::max<int64_t>(1, std::min<int64_t>(a.chunk, (tiles * launch_spp + n_waves - 1) / n_waves));
  }
  a.chunk_tail = std::min(a.chunk, std::max(1, g_chunk_tail.load()));
  a.counters = pl.counters ? s->d_counters : nullptr;
  a.W = p->width;
  a.H = p->height;
  a.max_depth = p->max_depth > 0 ? p->max_depth : 50;
  a.jitter = p->jitter ? 1 : 0;
  a.seed_lo = (uint32_t)p->seed;
  a.seed_hi = (uint32_t)(p->seed >> 32);
  for (uint32_t r = 0; r < 10; ++r) {
    a.philox_keys[2 * r] = a.seed_lo + r * 0x9E3779B9u;
    a.philox_keys[2 * r + 1] = a.seed_hi + r * 0xBB67AE85u;
  }
  a.wait_threshold = g_wait_threshold.load();
  if (a.wait_threshold <= 0) a.wait_threshold = a.L.wide ? 32 : 40;
  a.node_den = g_node_den.load();
  a.reserved1 = a.reserved2 = 0;
  a.inv_w1 = 1.0f / (float)(p->width - 1);
  a.inv_h1 = 1.0f / (float)(p->height - 1);
  a.tile_stride = std::max(1, p->tile_stride);
  a.tile_first = p->tile_first;
  a.timeline = nullptr;
  if (g_timeline.load() != 0) {
    std::lock_guard<std::mutex> lock(s->timeline_mutex);
    if (!s->d_timeline) {
      hipError_t e = hipMalloc((void**)&s->d_timeline, 3u * TIMELINE_BINS * sizeof(uint32_t));
      if (e == hipSuccess) e = hipMemset(s->d_timeline, 0, 3u * TIMELINE_BINS * sizeof(uint32_t));
      if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(HT_E_HIP, "allocation of the timeline failed: %s", hipGetErrorString(e));
      }
    }
    a.timeline = s->d_timeline;
  }

  int batch = p->batch > 0 ? p->batch : g_default_batch.load();
  // the work queue of one launch is indexed with 32 bits: tiles x sample chunks
  const int64_t tiles = (int64_t)((p->width + TILE_W - 1) / TILE_W) * ((p->height + TILE_H - 1) / TILE_H);
  const int64_t max_chunks = std::max<int64_t>(1, ((int64_t)1 << 31) / tiles - 1 - a.chunk);
  pl.batch = (int)std::min<int64_t>(batch, max_chunks * a.chunk);
  return HT_OK;
}

// One launch of the plan: samples [first, first + count) of every pixel, its own queue head.
hipError_t enqueue_launch(const RenderPlan& pl, uint32_t* queue_head, int first, int count, hipStream_t stream) {
  RenderArgs a = pl.a;
  a.sample_first = first;
  a.sample_count = count;
  a.tile_counter = queue_head;
  // HBM store: a band of tiles and a queue head per XCD (kernels.hip, the unit fetch)
  a.q = plan_queue(a.W, a.H, (uint32_t)a.tile_first, (uint32_t)a.tile_stride, (uint32_t)count, (uint32_t)a.chunk,
                   (uint32_t)a.chunk_tail, a.L.wide != 0u && g_xcd_bands.load() != 0);
  // One queue (on-chip store): the head starts behind the units the waves take without asking,
  // one per wave.  Banded queues: every head starts at zero.
  const int n_waves = pl.grid * (pl.shape.threads / 64);
  hipError_t e = hipMemsetD32Async((hipDeviceptr_t)queue_head, a.L.wide ? 0 : n_waves, QUEUE_SLOT_BYTES / 4, stream);
  if (e == hipSuccess)
    e = launch_render(pl.prec, pl.counters, pl.kfeat, a, pl.camp, pl.grid, pl.shape.threads, pl.shape.lds, stream);
  return e;
}


// The wavefront variant of one launch: samples [first, first + count) of every pixel, added to
// the plan's sum buffer.  Stage kernels over path state in global memory; the number of paths
// that survive a shade stage is read back each round, so the call synchronises with `stream`.
int render_wavefront(ht_scene* s, const RenderPlan& pl, int first, int count, hipStream_t stream) {
  const RenderArgs& ra = pl.a;
  if (ra.tile_stride > 1) return fail(HT_E_INVALID, "the wavefront variant renders whole frames only (tile_stride > 1)");
  // the workspace below belongs to the handle: one wavefront render of it at a time (every
  // round synchronises with the stream anyway)
  std::lock_guard<std::mutex> wf_lock(s->wf_mutex);
  const uint64_t WH = (uint64_t)ra.W * (uint64_t)ra.H;
  // a sub-launch holds at most 2^28 items: their contributions (12 bytes each, 3.2 GB at most)
  // are kept until the reduce, and every sub-launch pays the drain of its last long paths
  // (tens of rounds with few rays) once
  const int max_count = (int)std::max<uint64_t>(1, ((uint64_t)1 << 28) / WH);
  for (int done = 0; done < count; done += max_count) {
    const int n_s = std::min(max_count, count - done);
    const uint64_t items = WH * (uint64_t)n_s;
    if (items > 0xFFFFFFF0ull) return fail(HT_E_TOO_LARGE, "picture too large for the wavefront variant");
    const uint64_t cap = std::min<uint64_t>((uint64_t)g_wavefront_cap.load(), (items + 255) & ~(uint64_t)255);
    // [ray 6][thr 3][ids 2] x 2 buffers, hit_t, hit_ref: 24 words per path; contributions: 3 per item
    const size_t bytes = (size_t)cap * 24 * 4 + (size_t)items * 3 * 4 + 256;
    if (bytes > s->wf_bytes) {
      if (s->d_wf) (void)hipFree(s->d_wf);
      s->d_wf = nullptr;
      s->wf_bytes = 0;
      HIP_TRY(hipMalloc((void**)&s->d_wf, bytes));
      s->wf_bytes = bytes;
    }
    WfArgs a;
    memset(&a, 0, sizeof a);
    a.blob = ra.blob;
    a.L = ra.L;
    a.texels = ra.texels;
    a.accum = ra.accum;
    a.cap = (uint32_t)cap;
    a.items = (uint32_t)items;
    a.WH = (uint32_t)WH;
    a.W = ra.W;
    a.H = ra.H;
    a.sample_first = first + done;
    a.max_depth = ra.max_depth;
    a.jitter = ra.jitter;
    a.refill = g_wavefront_refill.load();
    a.inv_w1 = ra.inv_w1;
    a.inv_h1 = ra.inv_h1;
    memcpy(a.philox_keys, ra.philox_keys, sizeof a.philox_keys);
    float* w = (float*)s->d_wf;
    WfBuffers b0, b1;
    b0.ray = w; w += 6 * cap;
    b0.thr = w; w += 3 * cap;
    b0.ids = (uint32_t*)w; w += 2 * cap;
    b1.ray = w; w += 6 * cap;
    b1.thr = w; w += 3 * cap;
    b1.ids = (uint32_t*)w; w += 2 * cap;
    a.hit_t = w; w += cap;
    a.hit_ref = (uint32_t*)w; w += cap;
    a.contrib = w; w += 3 * items;
    a.counters = (uint32_t*)w;
    a.cur = b0;
    a.nxt = b1;
    HIP_TRY(hipMemsetAsync(a.contrib, 0, (size_t)items * 3 * 4, stream));
    uint64_t next = 0;
    uint32_t live = 0;
    for (int round = 0; live > 0 || next < items; ++round) {
      if (round > 1000000) return fail(HT_E_HIP, "wavefront variant: no progress");
      const uint32_t g = (uint32_t)std::min<uint64_t>(cap - live, items - next);
      a.n_new = g;
      a.slot0 = live;
      a.item0 = (uint32_t)next;
      HIP_TRY(launch_wf_generate(a, pl.cam32, stream));
      live += g;
      next += g;
      a.n = live;
      HIP_TRY(hipMemsetAsync(a.counters, 0, 256, stream));
      // no more workgroups than the rays can feed (every wave starts with a block of 64 or more)
      const int wg_rays = pl.shape.threads * 4;
      const int trace_grid = (int)std::min<uint64_t>((uint64_t)pl.grid, ((uint64_t)live + wg_rays - 1) / wg_rays);
      HIP_TRY(launch_wf_trace(pl.kfeat, a, std::max(1, trace_grid), pl.shape.threads, pl.shape.lds, stream));
      HIP_TRY(launch_wf_shade(pl.kfeat, a, pl.cam32, stream));
      uint32_t survivors = 0;
      HIP_TRY(hipMemcpyAsync(&survivors, a.counters, 4, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      if (survivors > live) return fail(HT_E_HIP, "wavefront variant: compaction counter out of range");
      live = survivors;
      std::swap(a.cur, a.nxt);
    }
    HIP_TRY(launch_wf_reduce(a, (uint32_t)n_s, stream));
  }
  return HT_OK;
}

uint32_t* next_queue_head(ht_scene* s) {
  return (uint32_t*)(s->d_queue_heads + (size_t)(s->next_head.fetch_add(1) % QUEUE_SLOTS) * QUEUE_SLOT_BYTES);
}

// ht_render_accumulate without the argument checks that its callers share.
// clear_cancel: a cancel request that arrived before this call belongs to an earlier frame
// (the reference's shouldBreak only stops the frame in flight, main.scala:80) and is dropped.
int render_accumulate_impl(ht_scene* s, const ht_camera* cam, const ht_render_params* p, int64_t* accum_dev,
                           hipStream_t stream, ht_stats* stats, bool clear_cancel) {
  if (clear_cancel) s->cancel.store(0);
  const auto t0 = std::chrono::steady_clock::now();
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);

  RenderPlan pl;
  int rc = make_plan(s, cam, p, accum_dev, pl);
  if (rc != HT_OK) return rc;
  if (pl.counters) HIP_TRY(hipMemsetAsync(s->d_counters, 0, NUM_COUNTERS * sizeof(unsigned long long), stream));

  // Timing events only for a caller that wants statistics; the two pacing events only when
  // the call makes more than two launches (they keep at most two batches in flight so that a
  // cancel request is seen soon).  A plain one-launch call creates and records none.
  const int n_launches = (p->samples + pl.batch - 1) / pl.batch;
  EventLease l_start(s, true), l_stop(s, true), l_batch0(s, false), l_batch1(s, false);
  {
    hipError_t ce = hipSuccess;
    if (stats) {
      ce = l_start.acquire();
      if (ce == hipSuccess) ce = l_stop.acquire();
    }
    if (ce == hipSuccess && n_launches > 2) {
      ce = l_batch0.acquire();
      if (ce == hipSuccess) ce = l_batch1.acquire();
    }
    if (ce != hipSuccess) {
      (void)hipGetLastError();
      return fail(HT_E_HIP, "hipEventCreate failed: %s", hipGetErrorString(ce));
    }
  }
  const hipEvent_t ev_start = l_start.ev, ev_stop = l_stop.ev;
  const hipEvent_t ev_batch[2] = {l_batch0.ev, l_batch1.ev};
  int launches = 0, done = 0;
  bool cancelled = false;
  const bool wavefront = g_wavefront.load() != 0 && pl.prec == HT_PRECISION_FP32 && !pl.counters;
  hipError_t e = ev_start ? hipEventRecord(ev_start, stream) : hipSuccess;
  while (e == hipSuccess && done < p->samples) {
    // keep at most two batches in flight so that a cancel request is seen soon
    if (launches >= 2 && ev_batch[0]) e = hipEventSynchronize(ev_batch[launches & 1]);
    if (e != hipSuccess) break;
    if (s->cancel.exchange(0)) { cancelled = true; break; }
    const int count = std::min(pl.batch, p->samples - done);
    if (wavefront) {
      const int wrc = render_wavefront(s, pl, p->sample_offset + done, count, stream);
      if (wrc != HT_OK) return wrc;
    } else {
      e = enqueue_launch(pl, next_queue_head(s), p->sample_offset + done, count, stream);
    }
    if (e == hipSuccess && ev_batch[0]) e = hipEventRecord(ev_batch[launches & 1], stream);
    if (e == hipSuccess) {
      done += count;
      launches++;
    }
  }
  if (e == hipSuccess && ev_stop) e = hipEventRecord(ev_stop, stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HT_E_HIP, "render launch failed: %s", hipGetErrorString(e));
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    e = hipEventSynchronize(ev_stop);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev_start, ev_stop);
    if (e == hipSuccess && pl.counters) {
      unsigned long long h[NUM_COUNTERS];
      e = hipMemcpy(h, s->d_counters, sizeof h, hipMemcpyDeviceToHost);
      if (e == hipSuccess) {
        stats->n_rays = (int64_t)h[CTR_RAYS];
        stats->n_box = (int64_t)h[CTR_BOX];
        stats->n_sphere = (int64_t)h[CTR_SPHERE];
        stats->n_rect = (int64_t)h[CTR_RECT];
        stats->n_tri = (int64_t)h[CTR_TRI];
        for (int k = 0; k < 4; ++k) {
          stats->n_shade[k] = (int64_t)h[CTR_SHADE0 + k];
          stats->n_tex[k] = (int64_t)h[CTR_TEX0 + k];
        }
        stats->n_steps = (int64_t)h[CTR_STEPS];
        stats->n_active = (int64_t)h[CTR_ACTIVE];
        stats->n_shade_rounds = (int64_t)h[CTR_SHADE_ROUNDS];
        stats->n_regen_rounds = (int64_t)h[CTR_REGEN_ROUNDS];
        stats->n_miss = (int64_t)h[CTR_MISS];
        stats->n_node_steps = (int64_t)h[CTR_NODE_STEPS];
        stats->cyc_trav = (int64_t)h[CTR_CYC_TRAV];
        stats->cyc_shade = (int64_t)h[CTR_CYC_SHADE];
        stats->cyc_regen = (int64_t)h[CTR_CYC_REGEN];
        if (h[CTR_CLK_REAL]) stats->shader_clock_mhz = 100.0 * (double)h[CTR_CLK_SHADER] / (double)h[CTR_CLK_REAL];
      }
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(HT_E_HIP, "render failed on the device: %s", hipGetErrorString(e));
    }
    stats->block_threads = pl.shape.threads;
    stats->blocks_per_cu = pl.shape.bpc;
    stats->lds_bytes = (int32_t)pl.shape.lds;
    stats->n_cu = s->n_cu;
    stats->seconds_kernel = ms * 1e-3;
    stats->launches = launches;
    stats->batch = pl.batch;
    if (wavefront) {
      snprintf(stats->kernel, sizeof stats->kernel, "wavefront");
    } else {
      stats->kernel_sgprs = render_kernel_name(pl.prec, pl.counters, pl.kfeat, pl.a.L.wide != 0u, stats->kernel,
                                               sizeof stats->kernel);
    }
    // what was launched, also when cancelled: the pixels of the call's own tiles x samples
    stats->samples = owned_pixels(p->width, p->height, pl.a.tile_first, pl.a.tile_stride) * done;
    stats->samples_per_pixel = done;
    stats->seconds_total =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (cancelled) return fail(HT_E_CANCELLED, "render cancelled after %d batches (%d samples per pixel)", launches, done);
  return HT_OK;
}

}  // namespace

extern "C" {

int ht_version(void) { return HT_VERSION; }

int ht_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* ht_last_error(void) { return g_error; }

int ht_set_option(const char* name, int value) {
  if (!name) return fail(HT_E_INVALID, "option name is NULL");
  if (!strcmp(name, "wait_threshold")) { g_wait_threshold = std::min(64, std::max(0, value)); return HT_OK; }
  if (!strcmp(name, "max_waves_per_cu")) { g_max_waves_per_cu = value <= 0 ? 0 : std::min(32, std::max(4, value)); g_shape_generation++; return HT_OK; }
  if (!strcmp(name, "max_blocks_per_cu")) { g_max_blocks_per_cu = std::max(1, value); g_shape_generation++; return HT_OK; }
  if (!strcmp(name, "default_batch")) { g_default_batch = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "chunk")) { g_chunk = std::max(0, value); return HT_OK; }
  if (!strcmp(name, "chunk_tail")) { g_chunk_tail = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "xcd_bands")) { g_xcd_bands = value != 0; return HT_OK; }
  if (!strcmp(name, "block_threads")) { g_block_threads = value; g_shape_generation++; return HT_OK; }
  if (!strcmp(name, "node_den")) { g_node_den = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "flat_kernels")) { g_flat_kernels = value ? 1 : 0; g_shape_generation++; return HT_OK; }
  if (!strcmp(name, "bvh_optimise_max_nodes")) { g_bvh_optimise_max_nodes = std::max(0, value); return HT_OK; }
  if (!strcmp(name, "bvh_optimise")) { g_bvh_optimise = std::min(64, std::max(0, value)); return HT_OK; }
  if (!strcmp(name, "timeline")) { g_timeline = value ? 1 : 0; return HT_OK; }
  if (!strcmp(name, "gpu_bvh")) { g_gpu_bvh_min = std::max(0, value); return HT_OK; }
  if (!strcmp(name, "force_hbm_scene")) { g_force_hbm = value ? 1 : 0; return HT_OK; }
  if (!strcmp(name, "wavefront")) { g_wavefront = value ? 1 : 0; return HT_OK; }
  if (!strcmp(name, "wavefront_refill")) { g_wavefront_refill = std::min(64, std::max(1, value)); return HT_OK; }
  if (!strcmp(name, "wavefront_paths")) { g_wavefront_cap = std::min(1 << 26, std::max(1 << 12, value)); return HT_OK; }
  return fail(HT_E_INVALID, "unknown option '%s'", name);
}

void ht_scene_destroy(ht_scene* s) {
  if (!s) return;
  {
    DeviceGuard g(s->device);
    if (s->d_blob32) (void)hipFree(s->d_blob32);
    if (s->d_blob64) (void)hipFree(s->d_blob64);
    if (s->d_texels) (void)hipFree(s->d_texels);
    if (s->d_queue_heads) (void)hipFree(s->d_queue_heads);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->d_wf) (void)hipFree(s->d_wf);
    if (s->d_timeline) (void)hipFree(s->d_timeline);
    for (int k = 0; k < SHARD_BUFFERS; ++k)
      if (s->d_shard[k]) (void)hipFree(s->d_shard[k]);
    if (s->d_host_accum) (void)hipFree(s->d_host_accum);
    if (s->d_host_mean) (void)hipFree(s->d_host_mean);
    if (s->d_host_packed) (void)hipFree(s->d_host_packed);
    for (hipEvent_t ev : s->timing_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : s->plain_events) (void)hipEventDestroy(ev);
  }
  delete s;
}

namespace {
// BuildOptions::external_bvh: the linear BVH of bvh_gpu.hip on the scene's device
bool device_bvh(void* ctx, uint32_t n, const float* boxes, const float cent_lo[3], const float cent_hi[3],
                ht::ExternalNode* nodes) {
  static_assert(sizeof(ht::ExternalNode) == sizeof(ht::LbvhNode) && ht::EXTERNAL_LEAF == ht::LBVH_LEAF, "one layout");
  DeviceGuard g(*static_cast<const int*>(ctx));
  if (!g.ok) return false;
  const hipError_t e = ht::lbvh_build(n, boxes, cent_lo, cent_hi, reinterpret_cast<ht::LbvhNode*>(nodes));
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess;
}
}  // namespace

int ht_scene_create(const ht_scene_desc* desc, int device, ht_scene** out) {
  if (!out) return fail(HT_E_INVALID, "out is NULL");
  *out = nullptr;
  HT_API_TRY
  const int ndev = ht_device_count();
  if (ndev <= 0) return fail(HT_E_NO_DEVICE, "no HIP device is visible; libhiptrace has no CPU path");
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = 0; }
  }
  if (device >= ndev) return fail(HT_E_NO_DEVICE, "device %d requested, %d visible", device, ndev);

  hipDeviceProp_t prop;
  {
    hipError_t pe = hipGetDeviceProperties(&prop, device);
    if (pe != hipSuccess) {
      (void)hipGetLastError();
      return fail(HT_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(pe));
    }
  }
  ht_scene* s = new (std::nothrow) ht_scene;
  if (!s) return fail(HT_E_INVALID, "out of host memory");
  std::string err;
  int rc;
  try {
    BuildOptions opt;
    opt.lds_bytes = (uint32_t)prop.sharedMemPerBlock;  // 160 KiB on gfx950
    opt.force_hbm = g_force_hbm.load() != 0;
    opt.external_bvh = device_bvh;
    opt.external_ctx = &device;
    opt.external_min_prims = g_gpu_bvh_min.load();
    opt.optimise_passes = g_bvh_optimise.load();
    opt.optimise_max_nodes = g_bvh_optimise_max_nodes.load();
    rc = build_host_scene(desc, s->hs, err, opt);
  } catch (const std::exception& e) {
    delete s;
    return fail(HT_E_INVALID, "scene flattening failed: %s", e.what());
  }
  if (rc != HT_OK) {
    delete s;
    return fail(rc, "%s", err.c_str());
  }
  s->device = device;
  DeviceGuard g(device);
  if (!g.ok) { delete s; return fail(HT_E_NO_DEVICE, "cannot select device %d", device); }
  hipError_t e = hipSuccess;
  s->n_cu = prop.multiProcessorCount;
  for (int t : s->hs.tex_type) {
    if (t == HT_TEX_PERLIN) s->feat |= KFEAT_PERLIN;
    if (t == HT_TEX_IMAGE) s->feat |= KFEAT_IMAGE;
  }
  auto upload = [&](const void* src, size_t bytes, void** dst) -> hipError_t {
    hipError_t r = hipMalloc(dst, bytes ? bytes : 16);
    if (r != hipSuccess) return r;
    if (bytes) r = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return r;
  };
  e = upload(s->hs.blob32.data(), s->hs.blob32.size(), (void**)&s->d_blob32);
  if (e == hipSuccess) e = upload(s->hs.blob64.data(), s->hs.blob64.size(), (void**)&s->d_blob64);
  if (e == hipSuccess)
    e = upload(s->hs.image_texels.data(), s->hs.image_texels.size() * 4, (void**)&s->d_texels);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_queue_heads, (size_t)QUEUE_SLOTS * QUEUE_SLOT_BYTES);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_counters, NUM_COUNTERS * sizeof(unsigned long long));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ht_scene_destroy(s);
    return fail(HT_E_HIP, "device allocation / upload failed: %s", hipGetErrorString(e));
  }
  // Set-up for the entry point that may not allocate or query later on: the launch shapes of
  // the two production variants and this device's address of the static scratch array.
  {
    LaunchShape sh;
    (void)launch_shape(s, HT_PRECISION_FP32, false, sh);  // a scene too large for a variant fails when rendered
    (void)launch_shape(s, HT_PRECISION_FP64, false, sh);
    void* scratch = nullptr;
    e = static_scratch(device, &scratch);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      ht_scene_destroy(s);
      return fail(HT_E_HIP, "hipGetSymbolAddress of the static scratch array failed: %s", hipGetErrorString(e));
    }
  }
  *out = s;
  return HT_OK;
  HT_API_CATCH
}

int ht_scene_get_info(const ht_scene* s, ht_scene_info* out) {
  if (!s || !out) return fail(HT_E_INVALID, "NULL argument");
  out->n_prims = (int32_t)s->hs.prim_type.size();
  out->n_nodes = (int32_t)s->hs.nodes.size();
  out->bvh_depth = s->hs.depth;
  out->n_materials = (int32_t)s->hs.mat_type.size();
  out->n_textures = (int32_t)s->hs.tex_type.size();
  out->lds_bytes_fp32 = (int32_t)s->hs.layout32.total_bytes;
  out->lds_bytes_fp64 = (int32_t)s->hs.layout64.total_bytes;
  out->hbm_store_fp32 = (int32_t)s->hs.layout32.wide;
  out->hbm_store_fp64 = (int32_t)s->hs.layout64.wide;
  out->device = s->device;
  out->sah_before = s->hs.sah_before;
  out->sah_after = s->hs.sah_after;
  return HT_OK;
}

int ht_debug_timeline(ht_scene* s, uint32_t* out, int32_t n_words) {
  if (!s || !out) return fail(HT_E_INVALID, "NULL argument");
  if (n_words != (int32_t)(3u * TIMELINE_BINS)) return fail(HT_E_INVALID, "the timeline has %u words", 3u * TIMELINE_BINS);
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  std::lock_guard<std::mutex> lock(s->timeline_mutex);
  if (!s->d_timeline) return fail(HT_E_INVALID, "no launch of this scene ran with the option \"timeline\" on");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, s->d_timeline, 3u * TIMELINE_BINS * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(s->d_timeline, 0, 3u * TIMELINE_BINS * sizeof(uint32_t)));
  return HT_OK;
}

int ht_cancel(ht_scene* s) {
  if (!s) return fail(HT_E_INVALID, "scene is NULL");
  s->cancel.store(1);
  return HT_OK;
}

int ht_render_accumulate(ht_scene* s, const ht_camera* cam, const ht_render_params* p,
                         int64_t* accum_dev, void* stream_v, ht_stats* stats) {
  if (!s || !p || !accum_dev) return fail(HT_E_INVALID, "NULL argument");
  int rc = check_camera(cam);
  if (rc != HT_OK) return rc;
  rc = check_params(s, p);
  if (rc != HT_OK) return rc;
  HT_API_TRY
  return render_accumulate_impl(s, cam, p, accum_dev, (hipStream_t)stream_v, stats, true);
  HT_API_CATCH
}

int ht_resolve_device(ht_scene* s, const int64_t* accum_dev, int32_t W, int32_t H, int64_t total_samples,
                      float* rgb_linear_dev, uint32_t* rgb_packed_dev, void* stream_v) {
  if (!s || !accum_dev) return fail(HT_E_INVALID, "NULL argument");
  if (W < 1 || H < 1 || total_samples < 1) return fail(HT_E_INVALID, "bad resolve dimensions");
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  HIP_TRY(launch_resolve((const long long*)accum_dev, W, H, 1.0 / (double)total_samples, rgb_linear_dev,
                         rgb_packed_dev, (hipStream_t)stream_v));
  return HT_OK;
}

namespace {
// ht_resolve with the scene's host_mutex held by the caller: mean / toRGB / flip into the
// scene's cached device pictures, then the copies into the caller's host buffers.
int resolve_to_host_locked(ht_scene* s, const int64_t* accum_dev, int32_t W, int32_t H, int64_t total_samples,
                           float* rgb_linear, uint32_t* rgb_packed, hipStream_t stream) {
  const size_t n = (size_t)W * H;
  if (rgb_linear) HIP_TRY(grow(s->d_host_mean, s->host_mean_pixels, n, 3 * sizeof(float)));
  if (rgb_packed) HIP_TRY(grow(s->d_host_packed, s->host_packed_pixels, n, sizeof(uint32_t)));
  float* d_mean = rgb_linear ? s->d_host_mean : nullptr;
  uint32_t* d_packed = rgb_packed ? s->d_host_packed : nullptr;
  hipError_t e = launch_resolve((const long long*)accum_dev, W, H, 1.0 / (double)total_samples, d_mean, d_packed, stream);
  // the packed picture first: it is what renderFrame's caller shows (main.scala:93)
  if (e == hipSuccess && rgb_packed)
    e = hipMemcpyAsync(rgb_packed, d_packed, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess && rgb_linear)
    e = hipMemcpyAsync(rgb_linear, d_mean, n * 3 * sizeof(float), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HT_E_HIP, "resolve failed: %s", hipGetErrorString(e));
  }
  return HT_OK;
}
}  // namespace

int ht_resolve(ht_scene* s, const int64_t* accum_dev, int32_t W, int32_t H, int64_t total_samples,
               float* rgb_linear, uint32_t* rgb_packed, void* stream_v) {
  if (!s || !accum_dev) return fail(HT_E_INVALID, "NULL argument");
  if (W < 1 || H < 1 || total_samples < 1) return fail(HT_E_INVALID, "bad resolve dimensions");
  HT_API_TRY
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  std::lock_guard<std::mutex> lock(s->host_mutex);
  return resolve_to_host_locked(s, accum_dev, W, H, total_samples, rgb_linear, rgb_packed, (hipStream_t)stream_v);
  HT_API_CATCH
}

int64_t ht_shard_tile_pixels(int32_t W, int32_t H, int32_t n_shards) {
  if (W < 1 || H < 1 || n_shards < 1) return 0;
  const int64_t tiles = (int64_t)tiles_in_frame(W, H);
  return ((tiles + n_shards - 1) / n_shards) * (int64_t)(TILE_W * TILE_H);
}

int ht_resolve_tiles_device(ht_scene* s, const int64_t* accum_dev, int32_t W, int32_t H, int64_t total_samples,
                            int32_t tile_first, int32_t tile_stride, float* mean_tiles_dev,
                            uint32_t* packed_tiles_dev, void* stream_v) {
  if (!s || !accum_dev) return fail(HT_E_INVALID, "NULL argument");
  if (W < 1 || H < 1 || total_samples < 1) return fail(HT_E_INVALID, "bad resolve dimensions");
  if ((int64_t)W * H > (int64_t)1 << 30) return fail(HT_E_INVALID, "image too large");
  if (tile_stride < 0 || tile_first < 0 || tile_first >= std::max(1, tile_stride))
    return fail(HT_E_INVALID, "tile_first must be in [0, max(tile_stride, 1))");
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  HIP_TRY(launch_resolve_tiles((const long long*)accum_dev, W, H, 1.0 / (double)total_samples, (uint32_t)tile_first,
                               (uint32_t)std::max(1, tile_stride), mean_tiles_dev, packed_tiles_dev,
                               (hipStream_t)stream_v));
  return HT_OK;
}

int ht_assemble_tiles_device(ht_scene* s, int32_t W, int32_t H, int32_t n_shards, const float* mean_tiles_all_dev,
                             const uint32_t* packed_tiles_all_dev, float* rgb_linear_dev, uint32_t* rgb_packed_dev,
                             void* stream_v) {
  if (!s) return fail(HT_E_INVALID, "scene is NULL");
  if (W < 1 || H < 1 || n_shards < 1) return fail(HT_E_INVALID, "bad assemble dimensions");
  if ((int64_t)W * H > (int64_t)1 << 30) return fail(HT_E_INVALID, "image too large");
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  HIP_TRY(launch_assemble_tiles(W, H, (uint32_t)n_shards, (size_t)ht_shard_tile_pixels(W, H, n_shards), mean_tiles_all_dev,
                                packed_tiles_all_dev, rgb_linear_dev, rgb_packed_dev, (hipStream_t)stream_v));
  return HT_OK;
}

int ht_render(ht_scene* s, const ht_camera* cam, const ht_render_params* p, float* rgb_linear,
              uint32_t* rgb_packed, ht_stats* stats_out) {
  if (!s || !p) return fail(HT_E_INVALID, "NULL argument");
  int rc = check_camera(cam);
  if (rc != HT_OK) return rc;
  rc = check_params(s, p);
  if (rc != HT_OK) return rc;
  HT_API_TRY
  const auto t0 = std::chrono::steady_clock::now();
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  const size_t n = (size_t)p->width * p->height;
  // the sums live in the handle's cached buffer: no allocation once it has the frame's size
  std::lock_guard<std::mutex> lock(s->host_mutex);
  HIP_TRY(grow(s->d_host_accum, s->host_accum_pixels, n, 3 * sizeof(int64_t)));
  int64_t* d_accum = s->d_host_accum;
  HIP_TRY(hipMemsetAsync(d_accum, 0, n * 3 * sizeof(int64_t), nullptr));
  ht_stats st;
  rc = render_accumulate_impl(s, cam, p, d_accum, nullptr, stats_out ? &st : nullptr, true);
  // (a tile shard through this call resolves the whole frame: the tiles it did not render are black)
  if (rc == HT_OK)
    rc = resolve_to_host_locked(s, d_accum, p->width, p->height, p->samples, rgb_linear, rgb_packed, nullptr);
  if (rc == HT_OK && stats_out) {
    *stats_out = st;
    stats_out->seconds_total =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return rc;
  HT_API_CATCH
}

int ht_render_noalloc(ht_scene* s, const ht_camera* cam, const ht_render_params* p, float* rgb_linear_dev,
                      uint32_t* rgb_packed_dev, void* stream_v) {
  if (!s || !p) return fail(HT_E_INVALID, "NULL argument");
  int rc = check_camera(cam);
  if (rc != HT_OK) return rc;
  rc = check_params(s, p);
  if (rc != HT_OK) return rc;
  if (p->collect_counters) return fail(HT_E_INVALID, "ht_render_noalloc does not collect counters");
  const int64_t pixels = (int64_t)p->width * p->height;
  if (pixels > (int64_t)HT_NOALLOC_MAX_PIXELS)
    return fail(HT_E_TOO_LARGE, "ht_render_noalloc: %d x %d = %lld pixels, the static scratch array holds %d",
                p->width, p->height, (long long)pixels, HT_NOALLOC_MAX_PIXELS);
  hipStream_t stream = (hipStream_t)stream_v;
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  // this device's copy of the static array: [radiance sums of HT_NOALLOC_MAX_PIXELS pixels][queue heads]
  void* scratch = nullptr;
  HIP_TRY(static_scratch(s->device, &scratch));
  int64_t* sums = (int64_t*)scratch;
  uint8_t* heads = (uint8_t*)scratch + (size_t)HT_NOALLOC_MAX_PIXELS * 3 * sizeof(int64_t);

  RenderPlan pl;
  rc = make_plan(s, cam, p, sums, pl);  // launch shape: cached since ht_scene_create
  if (rc != HT_OK) return rc;
  // the array keeps the last call's sums: clear what this call adds to
  HIP_TRY(hipMemsetAsync(sums, 0, (size_t)pixels * 3 * sizeof(int64_t), stream));
  int k = 0;
  for (int done = 0; done < p->samples; done += pl.batch, ++k) {
    uint32_t* head = (uint32_t*)(heads + (size_t)(k % (int)STATIC_QUEUE_SLOTS) * QUEUE_SLOT_BYTES);
    HIP_TRY(enqueue_launch(pl, head, p->sample_offset + done, std::min(pl.batch, p->samples - done), stream));
  }
  HIP_TRY(launch_resolve((const long long*)sums, p->width, p->height, 1.0 / (double)p->samples, rgb_linear_dev,
                         rgb_packed_dev, stream));
  return HT_OK;
}

// ht_render_sharded with HT_SHARD_TILES: handle k renders every sample of the tiles k, k + n, ...
// into a sum buffer on its own device and resolves them there into tile buffers (1/n of the
// frame each); the tile buffers go to the first device (peer copies, all at once, each over its
// own link), which puts the frame together.  No sums cross a link: 16 bytes per pixel of the
// frame in all instead of 24 bytes per pixel per device.
static int render_sharded_tiles(ht_scene* const* scenes, int32_t n, const ht_camera* cam, const ht_render_params* p,
                                float* rgb_linear, uint32_t* rgb_packed, ht_stats* stats_out) {
  const auto t0 = std::chrono::steady_clock::now();
  const int32_t W = p->width, H = p->height;
  const size_t pixels = (size_t)W * H;
  const size_t shard_px = (size_t)ht_shard_tile_pixels(W, H, n);
  for (int k = 0; k < n; ++k) scenes[k]->cancel.store(0);  // requests older than this frame
  struct Shard {
    int64_t* d_accum = nullptr;
    float* d_mean_tiles = nullptr;       // shards 1..n-1: this shard's tile buffers on its own device
    uint32_t* d_packed_tiles = nullptr;
    ht_render_params params;
    ht_stats stats;
    int rc = HT_OK;
    std::string error;
  };
  std::vector<Shard> shards((size_t)n);
  // The buffers live in the handles' shard workspaces (allocated by the first call of a size, kept).
  ShardLocks locks(scenes, n);
  auto release = []() {};
  float* d_mean_all = nullptr;       // on the first device: every shard's tile buffers, shard g at g * shard_px
  uint32_t* d_packed_all = nullptr;
  {
    DeviceGuard g(scenes[0]->device);
    hipError_t e = g.ok ? hipSuccess : hipErrorInvalidDevice;
    const size_t mean_bytes = rgb_linear ? shard_px * (size_t)n * 3 * sizeof(float) : 0;
    const size_t packed_bytes = rgb_packed ? shard_px * (size_t)n * sizeof(uint32_t) : 0;
    void* all = nullptr;  // means first (a multiple of 256 bytes: shard_px is a multiple of 64), then the packed codes
    if (e == hipSuccess) e = shard_buffer(scenes[0], SHARD_GATHERED, mean_bytes + packed_bytes, &all);
    if (e == hipSuccess && rgb_linear) d_mean_all = (float*)all;
    if (e == hipSuccess && rgb_packed) d_packed_all = (uint32_t*)((uint8_t*)all + mean_bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      release();
      return fail(HT_E_HIP, "allocation of the gathered tile buffers failed: %s", hipGetErrorString(e));
    }
  }
  for (int k = 0; k < n; ++k) {
    Shard& sh = shards[(size_t)k];
    sh.params = *p;
    sh.params.tile_first = k;
    sh.params.tile_stride = n;
    sh.params.device = -1;
    memset(&sh.stats, 0, sizeof sh.stats);
    DeviceGuard g(scenes[k]->device);
    hipError_t e = g.ok ? shard_buffer(scenes[k], SHARD_ACCUM, pixels * 3 * sizeof(int64_t), (void**)&sh.d_accum) : hipErrorInvalidDevice;
    if (e == hipSuccess) e = hipMemset(sh.d_accum, 0, pixels * 3 * sizeof(int64_t));
    if (k == 0) {  // the first shard resolves straight into its place in the gathered arrays
      sh.d_mean_tiles = d_mean_all;
      sh.d_packed_tiles = d_packed_all;
    } else {
      if (e == hipSuccess && rgb_linear)
        e = shard_buffer(scenes[k], SHARD_MEAN_TILES, shard_px * 3 * sizeof(float), (void**)&sh.d_mean_tiles);
      if (e == hipSuccess && rgb_packed)
        e = shard_buffer(scenes[k], SHARD_PACKED_TILES, shard_px * sizeof(uint32_t), (void**)&sh.d_packed_tiles);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      release();
      return fail(HT_E_HIP, "shard %d: allocation failed: %s", k, hipGetErrorString(e));
    }
  }
  // one host thread per shard: render, then resolve the shard's own tiles on its own device
  {
    std::vector<std::thread> workers;
    for (int k = 0; k < n; ++k) {
      Shard* sh = &shards[(size_t)k];
      ht_scene* sc = scenes[k];
      const int64_t total = p->samples;
      workers.emplace_back([sh, sc, cam, W, H, total]() {
        try {
          sh->rc = render_accumulate_impl(sc, cam, &sh->params, sh->d_accum, nullptr, &sh->stats, false);
          if (sh->rc == HT_OK)
            sh->rc = ht_resolve_tiles_device(sc, sh->d_accum, W, H, total, sh->params.tile_first, sh->params.tile_stride,
                                             sh->d_mean_tiles, sh->d_packed_tiles, nullptr);
          if (sh->rc == HT_OK) {
            DeviceGuard g(sc->device);
            const hipError_t e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) {
              (void)hipGetLastError();
              sh->rc = fail(HT_E_HIP, "resolve of the shard's tiles failed: %s", hipGetErrorString(e));
            }
          }
          if (sh->rc != HT_OK) sh->error = ht_last_error();  // the message is per thread: carry it over
        } catch (...) {
          sh->rc = HT_E_INVALID;
        }
      });
    }
    for (auto& w : workers) w.join();
  }
  for (int k = 0; k < n; ++k)
    if (shards[(size_t)k].rc != HT_OK) {
      const int code = shards[(size_t)k].rc;
      const std::string msg = shards[(size_t)k].error;
      release();
      return fail(code, "shard %d: %s", k, msg.c_str());
    }
  // gather: every other device sends its tile buffers to the first one, all at once
  hipError_t e = hipSuccess;
  for (int k = 1; k < n && e == hipSuccess; ++k) {
    const int dk = scenes[k]->device, d0 = scenes[0]->device;
    DeviceGuard g(dk);
    if (dk != d0) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, dk, d0) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d0, 0);
      (void)hipGetLastError();
    }
    if (rgb_packed)
      e = hipMemcpyPeerAsync(d_packed_all + (size_t)k * shard_px, d0, shards[(size_t)k].d_packed_tiles, dk,
                             shard_px * sizeof(uint32_t), nullptr);
    if (e == hipSuccess && rgb_linear)
      e = hipMemcpyPeerAsync(d_mean_all + (size_t)k * shard_px * 3, d0, shards[(size_t)k].d_mean_tiles, dk,
                             shard_px * 3 * sizeof(float), nullptr);
  }
  for (int k = 1; k < n; ++k) {
    DeviceGuard g(scenes[k]->device);
    const hipError_t se = hipStreamSynchronize(nullptr);
    if (e == hipSuccess) e = se;
  }
  int rc = HT_OK;
  if (e != hipSuccess) {
    (void)hipGetLastError();
    rc = fail(HT_E_HIP, "gathering the shards' tiles failed: %s", hipGetErrorString(e));
  } else {
    ht_scene* s0 = scenes[0];
    DeviceGuard g(s0->device);
    std::lock_guard<std::mutex> lock(s0->host_mutex);
    if (rgb_linear) e = grow(s0->d_host_mean, s0->host_mean_pixels, pixels, 3 * sizeof(float));
    if (e == hipSuccess && rgb_packed) e = grow(s0->d_host_packed, s0->host_packed_pixels, pixels, sizeof(uint32_t));
    if (e == hipSuccess)
      e = launch_assemble_tiles(W, H, (uint32_t)n, shard_px, d_mean_all, d_packed_all, rgb_linear ? s0->d_host_mean : nullptr,
                                rgb_packed ? s0->d_host_packed : nullptr, nullptr);
    if (e == hipSuccess && rgb_packed)
      e = hipMemcpyAsync(rgb_packed, s0->d_host_packed, pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, nullptr);
    if (e == hipSuccess && rgb_linear)
      e = hipMemcpyAsync(rgb_linear, s0->d_host_mean, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(HT_E_HIP, "assembling the frame failed: %s", hipGetErrorString(e));
    }
  }
  release();
  if (rc == HT_OK && stats_out) {
    *stats_out = shards[0].stats;
    for (int k = 1; k < n; ++k) {
      const ht_stats& st = shards[(size_t)k].stats;
      stats_out->samples += st.samples;
      stats_out->launches += st.launches;
      stats_out->seconds_kernel = std::max(stats_out->seconds_kernel, st.seconds_kernel);
    }
    stats_out->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return rc;
}

int ht_render_sharded(ht_scene* const* scenes, int32_t n, const ht_camera* cam, const ht_render_params* p,
                      float* rgb_linear, uint32_t* rgb_packed, ht_stats* stats_out) {
  if (!scenes || !p) return fail(HT_E_INVALID, "NULL argument");
  if (n < 1 || n > 64) return fail(HT_E_INVALID, "n_scenes must be 1..64");
  for (int k = 0; k < n; ++k) {
    if (!scenes[k]) return fail(HT_E_INVALID, "scene %d is NULL", k);
    for (int j = 0; j < k; ++j)
      if (scenes[j] == scenes[k]) return fail(HT_E_INVALID, "scene handle %d is listed twice (one render per handle)", k);
    if (scenes[k]->hs.prim_type.size() != scenes[0]->hs.prim_type.size() ||
        scenes[k]->hs.nodes.size() != scenes[0]->hs.nodes.size())
      return fail(HT_E_INVALID, "scene %d is not the scene of handle 0", k);
  }
  int rc = check_camera(cam);
  if (rc != HT_OK) return rc;
  rc = check_params(nullptr, p);
  if (rc != HT_OK) return rc;
  if (p->shard_mode != HT_SHARD_SAMPLES && p->shard_mode != HT_SHARD_TILES)
    return fail(HT_E_INVALID, "unknown shard_mode %d", p->shard_mode);
  if (p->tile_stride > 1) return fail(HT_E_INVALID, "ht_render_sharded cuts the frame itself: tile_stride must be 0 or 1");
  HT_API_TRY
  if (p->shard_mode == HT_SHARD_TILES && n > 1) return render_sharded_tiles(scenes, n, cam, p, rgb_linear, rgb_packed, stats_out);
  const auto t0 = std::chrono::steady_clock::now();
  const size_t words = (size_t)p->width * p->height * 3;
  const size_t bytes = words * sizeof(int64_t);
  for (int k = 0; k < n; ++k) scenes[k]->cancel.store(0);  // requests older than this frame

  struct Shard {
    int64_t* d_accum = nullptr;
    int64_t* d_staging = nullptr;  // the other shards' copies of this shard's slice
    ht_render_params params;
    ht_stats stats;
    int rc = HT_OK;
    std::string error;
  };
  std::vector<Shard> shards((size_t)n);
  // The buffers live in the handles' shard workspaces (allocated by the first call of a size, kept).
  ShardLocks locks(scenes, n);
  auto release = []() {};
  // Slice k of the frame is summed on shard k's device (words [lo_k, hi_k), 16-byte aligned):
  // every device pulls its slice from every other one at the same time, each pair over its
  // own xGMI link, and adds what arrived; then the summed slices go to the first device.
  // That is a reduce-scatter and a gather written out as peer copies: the combine costs two
  // messages of 1/n of the frame per link instead of n - 1 whole frames through one device.
  std::vector<size_t> lo((size_t)n + 1);
  for (int k = 0; k <= n; ++k) lo[(size_t)k] = std::min(words, ((words * (size_t)k / (size_t)n) + 1) & ~(size_t)1);
  lo[(size_t)n] = words;
  for (int k = 0; k < n; ++k) {
    Shard& sh = shards[(size_t)k];
    const int64_t first = (int64_t)k * p->samples / n, last = (int64_t)(k + 1) * p->samples / n;
    sh.params = *p;
    sh.params.sample_offset = p->sample_offset + (int32_t)first;
    sh.params.samples = (int32_t)(last - first);
    sh.params.device = -1;
    memset(&sh.stats, 0, sizeof sh.stats);
    DeviceGuard g(scenes[k]->device);
    const size_t slice = lo[(size_t)k + 1] - lo[(size_t)k];
    hipError_t e = g.ok ? shard_buffer(scenes[k], SHARD_ACCUM, bytes, (void**)&sh.d_accum) : hipErrorInvalidDevice;
    if (e == hipSuccess) e = hipMemset(sh.d_accum, 0, bytes);
    if (e == hipSuccess && n > 1 && slice > 0)
      e = shard_buffer(scenes[k], SHARD_STAGING, slice * (size_t)(n - 1) * sizeof(int64_t), (void**)&sh.d_staging);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      release();
      return fail(HT_E_HIP, "shard %d: allocation of the sum buffer failed: %s", k, hipGetErrorString(e));
    }
  }
  // one host thread per shard: every call sets its own device and blocks until its launches are done
  {
    std::vector<std::thread> workers;
    for (int k = 0; k < n; ++k) {
      Shard* sh = &shards[(size_t)k];
      ht_scene* sc = scenes[k];
      if (sh->params.samples < 1) continue;  // more shards than samples
      workers.emplace_back([sh, sc, cam]() {
        try {
          sh->rc = render_accumulate_impl(sc, cam, &sh->params, sh->d_accum, nullptr, &sh->stats, false);
          if (sh->rc != HT_OK) sh->error = ht_last_error();  // the message is per thread: carry it over
        } catch (...) {
          sh->rc = HT_E_INVALID;
        }
      });
    }
    for (auto& w : workers) w.join();
  }
  for (int k = 0; k < n; ++k)
    if (shards[(size_t)k].rc != HT_OK) {
      const int code = shards[(size_t)k].rc;
      const std::string msg = shards[(size_t)k].error;
      release();
      return fail(code, "shard %d: %s", k, msg.c_str());
    }
  if (n > 1) {
    hipError_t e = hipSuccess;
    // peer access where the links allow it (already enabled, or not possible: the copies then
    // go through the host, which is still correct)
    for (int k = 0; k < n; ++k)
      for (int j = 0; j < n; ++j) {
        const int dk = scenes[k]->device, dj = scenes[j]->device;
        if (dk == dj) continue;
        DeviceGuard g(dk);
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dk, dj) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(dj, 0);
        (void)hipGetLastError();
      }
    // reduce-scatter: all devices at once, nothing waits on the host in between
    for (int k = 0; k < n && e == hipSuccess; ++k) {
      Shard& sh = shards[(size_t)k];
      const size_t slice = lo[(size_t)k + 1] - lo[(size_t)k];
      if (slice == 0) continue;
      const int dk = scenes[k]->device;
      DeviceGuard g(dk);
      int slot = 0;
      for (int j = 0; j < n && e == hipSuccess; ++j) {
        if (j == k || shards[(size_t)j].params.samples < 1) continue;
        e = hipMemcpyPeerAsync(sh.d_staging + (size_t)slot * slice, dk, shards[(size_t)j].d_accum + lo[(size_t)k],
                               scenes[j]->device, slice * sizeof(int64_t), nullptr);
        slot++;
      }
      if (e == hipSuccess && slot > 0)
        e = launch_accum_add_many((long long*)sh.d_accum + lo[(size_t)k], (const long long*)sh.d_staging, slice, slot,
                                  nullptr);
    }
    for (int k = 0; k < n; ++k) {
      DeviceGuard g(scenes[k]->device);
      const hipError_t se = hipStreamSynchronize(nullptr);
      if (e == hipSuccess) e = se;
    }
    // gather: every device sends its summed slice to the first one, all at once
    for (int k = 1; k < n && e == hipSuccess; ++k) {
      const size_t slice = lo[(size_t)k + 1] - lo[(size_t)k];
      if (slice == 0) continue;
      DeviceGuard g(scenes[k]->device);
      e = hipMemcpyPeerAsync(shards[0].d_accum + lo[(size_t)k], scenes[0]->device,
                             shards[(size_t)k].d_accum + lo[(size_t)k], scenes[k]->device,
                             slice * sizeof(int64_t), nullptr);
    }
    for (int k = 1; k < n; ++k) {
      DeviceGuard g(scenes[k]->device);
      const hipError_t se = hipStreamSynchronize(nullptr);
      if (e == hipSuccess) e = se;
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      release();
      return fail(HT_E_HIP, "combining the shards failed: %s", hipGetErrorString(e));
    }
  }
  rc = ht_resolve(scenes[0], shards[0].d_accum, p->width, p->height, p->samples, rgb_linear, rgb_packed, nullptr);
  release();
  if (rc == HT_OK && stats_out) {
    *stats_out = shards[0].stats;
    for (int k = 1; k < n; ++k) {
      const ht_stats& st = shards[(size_t)k].stats;
      stats_out->samples += st.samples;
      stats_out->launches += st.launches;
      stats_out->seconds_kernel = std::max(stats_out->seconds_kernel, st.seconds_kernel);
    }
    stats_out->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return rc;
  HT_API_CATCH
}

int ht_frame_create(ht_scene* s, int32_t W, int32_t H, ht_frame** out) {
  if (!s || !out) return fail(HT_E_INVALID, "NULL argument");
  *out = nullptr;
  if (W < 2 || H < 2)
    return fail(HT_E_INVALID, "width and height must be >= 2 (the reference divides by W-1 and H-1)");
  if ((int64_t)W * H > (int64_t)1 << 30) return fail(HT_E_INVALID, "image too large");
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  ht_frame* f = new (std::nothrow) ht_frame;
  if (!f) return fail(HT_E_INVALID, "out of host memory");
  f->scene = s;
  f->width = W;
  f->height = H;
  const size_t bytes = (size_t)W * H * 3 * sizeof(int64_t);
  hipError_t e = hipMalloc((void**)&f->d_accum, bytes);
  if (e == hipSuccess) e = hipMemset(f->d_accum, 0, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (f->d_accum) (void)hipFree(f->d_accum);
    delete f;
    return fail(HT_E_HIP, "frame allocation failed: %s", hipGetErrorString(e));
  }
  *out = f;
  return HT_OK;
}

void ht_frame_destroy(ht_frame* f) {
  if (!f) return;
  {
    DeviceGuard g(f->scene->device);
    if (f->d_accum) (void)hipFree(f->d_accum);
  }
  delete f;
}

int ht_frame_clear(ht_frame* f) {
  if (!f) return fail(HT_E_INVALID, "frame is NULL");
  DeviceGuard g(f->scene->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", f->scene->device);
  HIP_TRY(hipMemset(f->d_accum, 0, (size_t)f->width * f->height * 3 * sizeof(int64_t)));
  f->samples = 0;
  return HT_OK;
}

int ht_frame_accumulate(ht_frame* f, const ht_camera* cam, const ht_render_params* p, ht_stats* stats_out) {
  if (!f || !p) return fail(HT_E_INVALID, "NULL argument");
  if (p->width != f->width || p->height != f->height)
    return fail(HT_E_INVALID, "params are %dx%d, the frame is %dx%d", p->width, p->height, f->width, f->height);
  int rc = check_camera(cam);
  if (rc != HT_OK) return rc;
  rc = check_params(f->scene, p);
  if (rc != HT_OK) return rc;
  HT_API_TRY
  ht_stats st;
  rc = render_accumulate_impl(f->scene, cam, p, f->d_accum, nullptr, &st, true);
  if (rc == HT_OK || rc == HT_E_CANCELLED) {
    // a cancelled call still added the batches it had launched
    f->samples += st.samples_per_pixel;
    if (stats_out) *stats_out = st;
  }
  return rc;
  HT_API_CATCH
}

int ht_frame_read(ht_frame* f, float* rgb_linear, uint32_t* rgb_packed) {
  if (!f) return fail(HT_E_INVALID, "frame is NULL");
  if (f->samples < 1) return fail(HT_E_INVALID, "the frame holds no samples yet");
  return ht_resolve(f->scene, f->d_accum, f->width, f->height, f->samples, rgb_linear, rgb_packed, nullptr);
}

int64_t ht_frame_samples(const ht_frame* f) { return f ? f->samples : 0; }

int ht_first_hit(ht_scene* s, const ht_camera* cam, int32_t W, int32_t H, int32_t precision,
                 float* t, int32_t* prim_id, float* normal, double* t64, double* normal64) {
  if (!s) return fail(HT_E_INVALID, "scene is NULL");
  int rc = check_camera(cam);
  if (rc != HT_OK) return rc;
  if (W < 2 || H < 2) return fail(HT_E_INVALID, "width and height must be >= 2");
  if ((int64_t)W * H > (int64_t)1 << 30) return fail(HT_E_INVALID, "image too large");
  if (precision != HT_PRECISION_FP32 && precision != HT_PRECISION_FP64)
    return fail(HT_E_INVALID, "unknown precision %d", precision);
  HT_API_TRY
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  const size_t n = (size_t)W * H;
  const size_t rs = precision == HT_PRECISION_FP32 ? 4 : 8;
  std::vector<uint8_t> ht_(n * rs), hn(n * 3 * rs);  // host staging, before anything is held on the device
  void *d_t = nullptr, *d_n = nullptr;
  int32_t* d_p = nullptr;
  hipError_t e = hipMalloc(&d_t, n * rs);
  if (e == hipSuccess) e = hipMalloc(&d_n, n * 3 * rs);
  if (e == hipSuccess) e = hipMalloc((void**)&d_p, n * 4);
  auto release = [&]() {
    if (d_t) (void)hipFree(d_t);
    if (d_n) (void)hipFree(d_n);
    if (d_p) (void)hipFree(d_p);
  };
  if (e != hipSuccess) { release(); return fail(HT_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e)); }

  CameraBasis basis;
  camera_basis(*cam, basis);
  CamDev<float> cam32;
  CamDev<double> cam64;
  fill_cam(basis, cam32);
  fill_cam(basis, cam64);
  FirstHitArgs a;
  memset(&a, 0, sizeof a);
  a.blob = precision == HT_PRECISION_FP32 ? s->d_blob32 : s->d_blob64;
  a.L = precision == HT_PRECISION_FP32 ? s->hs.layout32 : s->hs.layout64;
  a.W = W;
  a.H = H;
  a.t = d_t;
  a.prim = d_p;
  a.normal = d_n;
  uint32_t fh_threads = max_block_threads(precision);
  while (fh_threads > 64 && lds_bytes_needed(a.L, fh_threads) > 160u * 1024u) fh_threads -= 64;
  const uint32_t lds = lds_bytes_needed(a.L, fh_threads);
  e = launch_first_hit(precision, a, precision == HT_PRECISION_FP32 ? (const void*)&cam32 : (const void*)&cam64,
                       (int)fh_threads, lds, nullptr);
  if (e == hipSuccess) e = hipMemcpy(ht_.data(), d_t, n * rs, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(hn.data(), d_n, n * 3 * rs, hipMemcpyDeviceToHost);
  if (e == hipSuccess && prim_id) e = hipMemcpy(prim_id, d_p, n * 4, hipMemcpyDeviceToHost);
  release();
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HT_E_HIP, "first-hit failed: %s", hipGetErrorString(e));
  }
  if (precision == HT_PRECISION_FP32) {
    const float* pt = (const float*)ht_.data();
    const float* pn = (const float*)hn.data();
    for (size_t i = 0; i < n; ++i) {
      if (t) t[i] = pt[i];
      if (t64) t64[i] = pt[i];
    }
    for (size_t i = 0; i < n * 3; ++i) {
      if (normal) normal[i] = pn[i];
      if (normal64) normal64[i] = pn[i];
    }
  } else {
    const double* pt = (const double*)ht_.data();
    const double* pn = (const double*)hn.data();
    for (size_t i = 0; i < n; ++i) {
      if (t) t[i] = (float)pt[i];
      if (t64) t64[i] = pt[i];
    }
    for (size_t i = 0; i < n * 3; ++i) {
      if (normal) normal[i] = (float)pn[i];
      if (normal64) normal64[i] = pn[i];
    }
  }
  return HT_OK;
  HT_API_CATCH
}

}  // extern "C"
