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

#include <atomic>
#include <chrono>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hiptrace.h"
#include "host_scene.h"
#include "kernels.h"

using namespace ht;

struct ht_scene {
  HostScene hs;
  int device = 0;
  int n_cu = 0;
  uint8_t* d_blob32 = nullptr;
  uint8_t* d_blob64 = nullptr;
  uint32_t* d_texels = nullptr;
  uint32_t* d_tile_counter = nullptr;
  unsigned long long* d_counters = nullptr;
  std::atomic<int> cancel{0};
  uint32_t feat = 0;  // KFEAT_* texture kinds present in the scene
};

struct ht_frame {
  ht_scene* scene = nullptr;
  int32_t width = 0, height = 0;
  int64_t samples = 0;
  int64_t* d_accum = nullptr;
};

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
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

// tuning knobs (ht_set_option)
std::atomic<int> g_wait_threshold{40};
std::atomic<int> g_max_blocks_per_cu{8};
std::atomic<int> g_block_threads{0};  // 0: choose
std::atomic<int> g_node_den{2}, g_wait_num{1000}, g_wait_den{1};
std::atomic<int> g_default_batch{1024};
std::atomic<int> g_chunk{8};
std::atomic<int> g_chunk_tail{4};
std::atomic<int> g_force_hbm{0};  // scenes created while set keep their image in HBM

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
    c.llc[i] = (R)b.llc[i];
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

const char* ht_last_error(void) { return g_error.c_str(); }

int ht_set_option(const char* name, int value) {
  if (!name) return fail(HT_E_INVALID, "option name is NULL");
  if (!strcmp(name, "wait_threshold")) { g_wait_threshold = std::min(64, std::max(1, value)); return HT_OK; }
  if (!strcmp(name, "max_blocks_per_cu")) { g_max_blocks_per_cu = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "default_batch")) { g_default_batch = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "chunk")) { g_chunk = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "chunk_tail")) { g_chunk_tail = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "block_threads")) { g_block_threads = value; return HT_OK; }
  if (!strcmp(name, "node_den")) { g_node_den = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "wait_ratio_num")) { g_wait_num = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "wait_ratio_den")) { g_wait_den = std::max(1, value); return HT_OK; }
  if (!strcmp(name, "force_hbm_scene")) { g_force_hbm = value ? 1 : 0; return HT_OK; }
  return fail(HT_E_INVALID, "unknown option '%s'", name);
}

void ht_scene_destroy(ht_scene* s) {
  if (!s) return;
  {
    DeviceGuard g(s->device);
    if (s->d_blob32) (void)hipFree(s->d_blob32);
    if (s->d_blob64) (void)hipFree(s->d_blob64);
    if (s->d_texels) (void)hipFree(s->d_texels);
    if (s->d_tile_counter) (void)hipFree(s->d_tile_counter);
    if (s->d_counters) (void)hipFree(s->d_counters);
  }
  delete s;
}

int ht_scene_create(const ht_scene_desc* desc, int device, ht_scene** out) {
  if (!out) return fail(HT_E_INVALID, "out is NULL");
  *out = nullptr;
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
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_tile_counter, 256);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_counters, NUM_COUNTERS * sizeof(unsigned long long));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ht_scene_destroy(s);
    return fail(HT_E_HIP, "device allocation / upload failed: %s", hipGetErrorString(e));
  }
  *out = s;
  return HT_OK;
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
  if (p->width < 2 || p->height < 2)
    return fail(HT_E_INVALID, "width and height must be >= 2 (the reference divides by W-1 and H-1)");
  if ((int64_t)p->width * p->height > (int64_t)1 << 30) return fail(HT_E_INVALID, "image too large");
  if (p->samples < 1) return fail(HT_E_INVALID, "samples must be >= 1");
  if (p->sample_offset < 0) return fail(HT_E_INVALID, "sample_offset must be >= 0");
  if (p->precision != HT_PRECISION_FP32 && p->precision != HT_PRECISION_FP64)
    return fail(HT_E_INVALID, "unknown precision %d", p->precision);
  if (p->device >= 0 && p->device != s->device)
    return fail(HT_E_INVALID, "scene lives on device %d, params ask for %d", s->device, p->device);
  const auto t0 = std::chrono::steady_clock::now();
  hipStream_t stream = (hipStream_t)stream_v;
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);

  const int prec = p->precision;
  const bool counters = p->collect_counters != 0;
  const SceneLayout& L = prec == HT_PRECISION_FP32 ? s->hs.layout32 : s->hs.layout64;
  const uint32_t kfeat = s->feat | (L.tree_kinds == TREE_TRIANGLES ? KFEAT_TRIS : 0u);
  // workgroup size and count: the combination that keeps the most waves resident per CU;
  // smaller groups win ties (measured on the Cornell box, 32 waves per CU either way:
  // 256 x 8 34.2, 512 x 4 32.8, 1024 x 2 27.7 G samples/s)
  int threads = 0, bpc = 0;
  uint32_t lds = 0;
  const int forced = g_block_threads.load();
  for (int t = 256; t <= (int)max_block_threads(prec); t += 128) {
    if (forced > 0 && t != forced) continue;
    const uint32_t need = lds_bytes_needed(L, (uint32_t)t);
    int b = render_blocks_per_cu(prec, counters, kfeat, L.wide != 0u, t, need);
    b = std::min(b, g_max_blocks_per_cu.load());
    // several workgroups share a CU well only when each fills the four SIMDs evenly: two
    // groups of 14 waves (measured on the Cornell box) ran 20 % slower than two of 12
    if (b > 1 && t % 256 != 0) b = 1;
    if (b * t > bpc * threads) { threads = t; bpc = b; lds = need; }
  }
  if (bpc <= 0)
    return fail(HT_E_TOO_LARGE, "the render kernel cannot be resident with this scene (%u bytes of scene image)",
                L.total_bytes);
  const int grid = bpc * s->n_cu;

  CameraBasis basis;
  camera_basis(*cam, basis);
  CamDev<float> cam32;
  CamDev<double> cam64;
  fill_cam(basis, cam32);
  fill_cam(basis, cam64);
  const void* camp = prec == HT_PRECISION_FP32 ? (const void*)&cam32 : (const void*)&cam64;

  RenderArgs a;
  memset(&a, 0, sizeof a);
  a.blob = prec == HT_PRECISION_FP32 ? s->d_blob32 : s->d_blob64;
  a.L = L;
  a.texels = s->d_texels;
  a.accum = (long long*)accum_dev;
  a.chunk = std::max(1, g_chunk.load());
  a.chunk_tail = std::min(a.chunk, std::max(1, g_chunk_tail.load()));
  a.tile_counter = s->d_tile_counter;
  a.counters = counters ? s->d_counters : nullptr;
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
  a.node_den = g_node_den.load();
  a.wait_ratio_num = g_wait_num.load();
  a.wait_ratio_den = g_wait_den.load();

  if (counters) HIP_TRY(hipMemsetAsync(s->d_counters, 0, NUM_COUNTERS * sizeof(unsigned long long), stream));

  hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_batch[2] = {nullptr, nullptr};
  auto cleanup = [&]() {
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (ev_stop) (void)hipEventDestroy(ev_stop);
    if (ev_batch[0]) (void)hipEventDestroy(ev_batch[0]);
    if (ev_batch[1]) (void)hipEventDestroy(ev_batch[1]);
  };
  {
    hipError_t ce = hipEventCreate(&ev_start);
    if (ce == hipSuccess) ce = hipEventCreate(&ev_stop);
    if (ce == hipSuccess) ce = hipEventCreateWithFlags(&ev_batch[0], hipEventDisableTiming);
    if (ce == hipSuccess) ce = hipEventCreateWithFlags(&ev_batch[1], hipEventDisableTiming);
    if (ce != hipSuccess) {
      (void)hipGetLastError();
      cleanup();
      return fail(HT_E_HIP, "hipEventCreate failed: %s", hipGetErrorString(ce));
    }
  }

  int batch = p->batch > 0 ? p->batch : g_default_batch.load();
  {
    // the work queue of one launch is indexed with 32 bits: tiles x sample chunks
    const int64_t tiles = (int64_t)((p->width + TILE_W - 1) / TILE_W) * ((p->height + TILE_H - 1) / TILE_H);
    const int64_t max_chunks = std::max<int64_t>(1, ((int64_t)1 << 31) / tiles - 1 - a.chunk);
    batch = (int)std::min<int64_t>(batch, max_chunks * a.chunk);
  }
  int launches = 0;
  bool cancelled = false;
  hipError_t e = hipEventRecord(ev_start, stream);
  for (int done = 0; e == hipSuccess && done < p->samples; done += batch) {
    // keep at most two batches in flight so that a cancel request is seen soon
    if (launches >= 2) e = hipEventSynchronize(ev_batch[launches & 1]);
    if (e != hipSuccess) break;
    if (s->cancel.exchange(0)) { cancelled = true; break; }
    a.sample_first = p->sample_offset + done;
    a.sample_count = std::min(batch, p->samples - done);
    e = hipMemsetAsync(s->d_tile_counter, 0, 256, stream);
    if (e == hipSuccess) e = launch_render(prec, counters, kfeat, a, camp, grid, threads, lds, stream);
    if (e == hipSuccess) e = hipEventRecord(ev_batch[launches & 1], stream);
    launches++;
  }
  if (e == hipSuccess) e = hipEventRecord(ev_stop, stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    return fail(HT_E_HIP, "render launch failed: %s", hipGetErrorString(e));
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    e = hipEventSynchronize(ev_stop);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev_start, ev_stop);
    if (e == hipSuccess && counters) {
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
        stats->cyc_trav = (int64_t)h[CTR_CYC_TRAV];
        stats->cyc_shade = (int64_t)h[CTR_CYC_SHADE];
        stats->cyc_regen = (int64_t)h[CTR_CYC_REGEN];
        if (h[CTR_CLK_REAL]) stats->shader_clock_mhz = 100.0 * (double)h[CTR_CLK_SHADER] / (double)h[CTR_CLK_REAL];
      }
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      cleanup();
      return fail(HT_E_HIP, "render failed on the device: %s", hipGetErrorString(e));
    }
    stats->block_threads = threads;
    stats->blocks_per_cu = bpc;
    stats->lds_bytes = (int32_t)lds;
    stats->n_cu = s->n_cu;
    stats->seconds_kernel = ms * 1e-3;
    stats->launches = launches;
    stats->samples = (int64_t)p->width * p->height * (cancelled ? (int64_t)(launches)*batch : p->samples);
    stats->seconds_total =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  cleanup();
  if (cancelled) return fail(HT_E_CANCELLED, "render cancelled after %d batches", launches);
  return HT_OK;
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

int ht_resolve(ht_scene* s, const int64_t* accum_dev, int32_t W, int32_t H, int64_t total_samples,
               float* rgb_linear, uint32_t* rgb_packed, void* stream_v) {
  if (!s || !accum_dev) return fail(HT_E_INVALID, "NULL argument");
  if (W < 1 || H < 1 || total_samples < 1) return fail(HT_E_INVALID, "bad resolve dimensions");
  hipStream_t stream = (hipStream_t)stream_v;
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  const size_t n = (size_t)W * H;
  float* d_mean = nullptr;
  uint32_t* d_packed = nullptr;
  if (rgb_linear) HIP_TRY(hipMalloc((void**)&d_mean, n * 3 * sizeof(float)));
  if (rgb_packed) {
    hipError_t e = hipMalloc((void**)&d_packed, n * sizeof(uint32_t));
    if (e != hipSuccess) {
      if (d_mean) (void)hipFree(d_mean);
      return fail(HT_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
  }
  hipError_t e = launch_resolve((const long long*)accum_dev, W, H, 1.0 / (double)total_samples, d_mean, d_packed, stream);
  if (e == hipSuccess && rgb_linear)
    e = hipMemcpyAsync(rgb_linear, d_mean, n * 3 * sizeof(float), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess && rgb_packed)
    e = hipMemcpyAsync(rgb_packed, d_packed, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (d_mean) (void)hipFree(d_mean);
  if (d_packed) (void)hipFree(d_packed);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HT_E_HIP, "resolve failed: %s", hipGetErrorString(e));
  }
  return HT_OK;
}

int ht_render(ht_scene* s, const ht_camera* cam, const ht_render_params* p, float* rgb_linear,
              uint32_t* rgb_packed, ht_stats* stats_out) {
  if (!s || !p) return fail(HT_E_INVALID, "NULL argument");
  if (p->width < 2 || p->height < 2)
    return fail(HT_E_INVALID, "width and height must be >= 2 (the reference divides by W-1 and H-1)");
  const auto t0 = std::chrono::steady_clock::now();
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  const size_t n = (size_t)p->width * p->height;
  int64_t* d_accum = nullptr;
  HIP_TRY(hipMalloc((void**)&d_accum, n * 3 * sizeof(int64_t)));
  hipError_t e = hipMemsetAsync(d_accum, 0, n * 3 * sizeof(int64_t), nullptr);
  if (e != hipSuccess) {
    (void)hipFree(d_accum);
    return fail(HT_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
  }
  ht_stats st;
  int rc = ht_render_accumulate(s, cam, p, d_accum, nullptr, &st);
  if (rc == HT_OK)
    rc = ht_resolve(s, d_accum, p->width, p->height, p->samples, rgb_linear, rgb_packed, nullptr);
  (void)hipFree(d_accum);
  if (rc == HT_OK && stats_out) {
    *stats_out = st;
    stats_out->seconds_total =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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
  if (p->width < 2 || p->height < 2)
    return fail(HT_E_INVALID, "width and height must be >= 2 (the reference divides by W-1 and H-1)");
  if ((int64_t)p->width * p->height > (int64_t)1 << 30) return fail(HT_E_INVALID, "image too large");
  if (p->samples < 1) return fail(HT_E_INVALID, "samples must be >= 1");
  const auto t0 = std::chrono::steady_clock::now();
  const size_t words = (size_t)p->width * p->height * 3;
  const size_t bytes = words * sizeof(int64_t);

  struct Shard {
    int64_t* d_accum = nullptr;
    ht_render_params params;
    ht_stats stats;
    int rc = HT_OK;
    std::string error;
  };
  std::vector<Shard> shards((size_t)n);
  auto release = [&]() {
    for (int k = 0; k < n; ++k)
      if (shards[(size_t)k].d_accum) {
        DeviceGuard g(scenes[k]->device);
        (void)hipFree(shards[(size_t)k].d_accum);
      }
  };
  for (int k = 0; k < n; ++k) {
    Shard& sh = shards[(size_t)k];
    const int64_t first = (int64_t)k * p->samples / n, last = (int64_t)(k + 1) * p->samples / n;
    sh.params = *p;
    sh.params.sample_offset = p->sample_offset + (int32_t)first;
    sh.params.samples = (int32_t)(last - first);
    sh.params.device = -1;
    memset(&sh.stats, 0, sizeof sh.stats);
    DeviceGuard g(scenes[k]->device);
    hipError_t e = g.ok ? hipMalloc((void**)&sh.d_accum, bytes) : hipErrorInvalidDevice;
    if (e == hipSuccess) e = hipMemset(sh.d_accum, 0, bytes);
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
        sh->rc = ht_render_accumulate(sc, cam, &sh->params, sh->d_accum, nullptr, &sh->stats);
        if (sh->rc != HT_OK) sh->error = ht_last_error();  // the message is per thread: carry it over
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
  // bring the partial sums to device 0 of the list and add them there
  {
    const int dev0 = scenes[0]->device;
    DeviceGuard g(dev0);
    if (!g.ok) { release(); return fail(HT_E_NO_DEVICE, "cannot select device %d", dev0); }
    int64_t* staging = nullptr;
    hipError_t e = hipSuccess;
    for (int k = 1; k < n && e == hipSuccess; ++k) {
      if (shards[(size_t)k].params.samples < 1) continue;
      const int devk = scenes[k]->device;
      const long long* src = (const long long*)shards[(size_t)k].d_accum;
      if (devk != dev0) {
        // direct device-to-device copies where the link allows them (already enabled, or
        // not possible: the copy then goes through the host, which is still correct)
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev0, devk) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devk, 0);
        (void)hipGetLastError();
        if (!staging) e = hipMalloc((void**)&staging, bytes);
        if (e == hipSuccess) e = hipMemcpyPeer(staging, dev0, shards[(size_t)k].d_accum, devk, bytes);
        src = (const long long*)staging;
      }
      if (e == hipSuccess) e = launch_accum_add((long long*)shards[0].d_accum, src, words, nullptr);
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // staging is reused by the next shard
    }
    if (staging) (void)hipFree(staging);
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
  ht_stats st;
  int rc = ht_render_accumulate(f->scene, cam, p, f->d_accum, nullptr, &st);
  if (rc == HT_OK || rc == HT_E_CANCELLED) {
    // a cancelled call still added the batches it had launched
    f->samples += st.samples / ((int64_t)f->width * f->height);
    if (stats_out) *stats_out = st;
  }
  return rc;
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
  if (precision != HT_PRECISION_FP32 && precision != HT_PRECISION_FP64)
    return fail(HT_E_INVALID, "unknown precision %d", precision);
  DeviceGuard g(s->device);
  if (!g.ok) return fail(HT_E_NO_DEVICE, "cannot select device %d", s->device);
  const size_t n = (size_t)W * H;
  const size_t rs = precision == HT_PRECISION_FP32 ? 4 : 8;
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
  std::vector<uint8_t> ht_(n * rs), hn(n * 3 * rs);
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
}

}  // extern "C"
