/* hiptrace.h — C ABI of libhiptrace.so: the per-pixel trace / intersect / shade
 * loop of the Ivoah/RayTracing renderer, rebuilt for AMD MI355X (gfx950).
 *
 * The reference has no FFI of its own; the seam this ABI replaces is the body of
 *   renderFrame(camera, world, img, samples, update, finish, shouldBreak)
 *   (reference: jvm/src/main/scala/main.scala:73-99)
 * Everything the JVM host builds before that call (loadScene, main.scala:32-71;
 * the Camera case class, Camera.scala:3-4; the Hittable / Material / Texture
 * objects, Hittable.scala, Material.scala, Texture.scala) crosses this boundary
 * as plain arrays of doubles and ints.  All geometry is passed in fp64, exactly
 * the values the reference holds; the library does its own conversion.
 *
 * Conventions
 *  - every entry point returns 0 on success or a negative HT_E_* code; a
 *    human-readable message for the calling thread is in ht_last_error().
 *  - no C++ exception, abort or exit crosses the boundary.
 *  - all buffers are caller-allocated and caller-freed; the scene handle owns
 *    only device memory.
 *  - one render may be in flight per scene handle; ht_cancel may be called
 *    from any other thread.
 *  - the library never falls back to the CPU: without a HIP device every
 *    compute entry point fails with HT_E_NO_DEVICE.
 */
#ifndef HIPTRACE_H
#define HIPTRACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HT_VERSION 100 /* 0.1.0 */

/* ---- error codes ------------------------------------------------------- */
enum {
  HT_OK = 0,
  HT_E_INVALID = -1,   /* bad argument / malformed scene description        */
  HT_E_NO_DEVICE = -2, /* no HIP device, or the requested one does not exist */
  HT_E_HIP = -3,       /* a HIP runtime call failed                          */
  HT_E_TOO_LARGE = -4, /* scene beyond the library's limits (1 Mi primitives) */
  HT_E_CANCELLED = -5  /* ht_cancel() was observed between sample batches    */
};

/* ---- primitive / material / texture type tags --------------------------- */
/* Primitive kinds (reference: Hittable.scala:25-115). */
enum {
  HT_PRIM_SPHERE = 0,   /* data: cx cy cz radius                    Hittable.scala:25 */
  HT_PRIM_XYRECT = 1,   /* data: x0 x1 y0 y1 z                      Hittable.scala:54 */
  HT_PRIM_XZRECT = 2,   /* data: x0 x1 z0 z1 y                      Hittable.scala:68 */
  HT_PRIM_YZRECT = 3,   /* data: y0 y1 z0 z1 x                      Hittable.scala:82 */
  HT_PRIM_TRIANGLE = 4  /* data: ax ay az bx by bz cx cy cz         Hittable.scala:96 */
};
#define HT_PRIM_STRIDE 9 /* doubles per primitive in ht_scene_desc.prim_data */

/* Material kinds (reference: Material.scala:9-55). */
enum {
  HT_MAT_DIFFUSE = 0,  /* param unused            Material.scala:9  */
  HT_MAT_GLOSSY = 1,   /* param = roughness       Material.scala:15 */
  HT_MAT_GLASS = 2,    /* param = ior             Material.scala:23 */
  HT_MAT_EMISSION = 3  /* param = strength        Material.scala:52 */
};

/* Texture kinds (reference: Texture.scala:12-107).  Textures form a forest:
 * a Checker names two child textures by index. */
enum {
  HT_TEX_SOLID = 0,   /* data: r g b -          Texture.scala:12 */
  HT_TEX_CHECKER = 1, /* data: size - - - ; child = {t1, t2}   Texture.scala:20 */
  HT_TEX_PERLIN = 2,  /* data: scale - - - ; child = {perlin table index, -}  Texture.scala:35 */
  HT_TEX_IMAGE = 3    /* child = {image index, -}  Texture.scala:97 */
};
#define HT_TEX_STRIDE 4 /* doubles per texture in ht_scene_desc.tex_data */
#define HT_PERLIN_POINTS 256 /* Texture.scala:36 */

/* ---- scene description -------------------------------------------------- */
/* The flattened world.  Primitive i of prim_* is "primitive id i" everywhere in
 * this ABI: JSON `world` array order, an STL entry expanded to its facets in
 * file order (JsonFormats.scala:87-99, BVH.scala:35-55). */
typedef struct ht_scene_desc {
  int32_t n_prims;
  const int32_t* prim_type;     /* [n_prims] HT_PRIM_*                                   */
  const int32_t* prim_material; /* [n_prims] index into mat_*                            */
  const double* prim_data;      /* [n_prims * HT_PRIM_STRIDE]                            */
  const int32_t* prim_group;    /* [n_prims] or NULL: index of the `world` array entry the
                                   primitive came from.  Informational (the reference nests
                                   an STL's own BVH as one world entry, main.scala:60); the
                                   closest hit does not depend on it.                   */

  int32_t n_materials;
  const int32_t* mat_type;    /* [n_materials] HT_MAT_*            */
  const int32_t* mat_texture; /* [n_materials] index into tex_*    */
  const double* mat_param;    /* [n_materials]                     */

  int32_t n_textures;
  const int32_t* tex_type;  /* [n_textures] HT_TEX_*               */
  const double* tex_data;   /* [n_textures * HT_TEX_STRIDE]        */
  const int32_t* tex_child; /* [n_textures * 2]                    */

  /* Decoded images, as BufferedImage.getRGB delivers them to Vec3.fromRGB
   * (jvm/src/main/scala/FileLoader.scala:14-17, Vec3.scala:36-40):
   * one 0x00RRGGBB word per texel, row 0 = top of the picture. */
  int32_t n_images;
  const int32_t* image_width;   /* [n_images] */
  const int32_t* image_height;  /* [n_images] */
  const int64_t* image_offset;  /* [n_images] first texel of image k in image_texels */
  const uint32_t* image_texels;

  /* Perlin tables (Texture.scala:38-42).  The reference draws them from an
   * unseeded RNG at load time, so the host passes the ones it drew. */
  int32_t n_perlin;
  const double* perlin_ranvec; /* [n_perlin * 256 * 3] unit vectors           */
  const int32_t* perlin_perm;  /* [n_perlin * 3 * 256] perm_x, perm_y, perm_z */
} ht_scene_desc;

/* The eight fields of the reference's Camera case class (Camera.scala:3-4,
 * JsonFormats.scala:10).  The derived basis (Camera.scala:5-17) is computed by
 * the library. */
typedef struct ht_camera {
  double origin[3];
  double target[3];
  double vup[3];
  double vfov;           /* degrees */
  double aspect_ratio;   /* from the scene file, NOT width/height */
  double aperture;
  double focus_distance; /* < 0: |target - origin| */
  double background[3];
} ht_camera;

/* ---- render parameters --------------------------------------------------- */
enum { HT_PRECISION_FP32 = 0, HT_PRECISION_FP64 = 1 };

typedef struct ht_render_params {
  int32_t width;          /* img.getWidth()   main.scala:81 */
  int32_t height;         /* img.getHeight()  main.scala:79 */
  int32_t samples;        /* camera paths per pixel rendered by THIS call          */
  int32_t sample_offset;  /* global index of this call's first sample: a node that
                             shards spp over G devices gives rank g the range
                             [g*spp/G, (g+1)*spp/G) so the sample set is the same
                             for every G                                           */
  int32_t max_depth;      /* 50 in the reference (Camera.scala:25); 0 -> 50        */
  int32_t jitter;         /* 1: sub-pixel jitter + lens sampling as main.scala:88-89,
                             Camera.scala:20.  0: pixel offset 0.5, aperture 0     */
  int32_t precision;      /* HT_PRECISION_*: arithmetic type of the device path    */
  int32_t batch;          /* samples per kernel launch (cancel granularity); 0 -> default */
  uint64_t seed;          /* Philox key                                             */
  int32_t device;         /* -1, or the ordinal the scene lives on (checked)       */
  int32_t collect_counters; /* 1: run the instrumented kernel and fill ht_stats.n_* */
  /* Tile sharding (main.scala:79-81 loops over independent pixels).  The frame's 8x8-pixel
   * tiles are numbered row-major from the bottom-left corner (tiles per row = ceil(width/8));
   * the call renders only the tiles tile_first, tile_first + tile_stride, ... and leaves the
   * other pixels' sums untouched.  tile_stride 0 or 1: every tile (tile_first must be 0).
   * A node that shards by tiles gives rank g of G the pair (g, G) and all samples. */
  int32_t tile_first;
  int32_t tile_stride;
  int32_t shard_mode;     /* ht_render_sharded only: HT_SHARD_SAMPLES or HT_SHARD_TILES   */
  int32_t reserved;       /* 0 */
} ht_render_params;

/* How ht_render_sharded cuts a frame between its scene handles. */
enum {
  HT_SHARD_SAMPLES = 0, /* handle k renders samples [k*S/n, (k+1)*S/n) of every pixel; the partial sums
                           (24 bytes per pixel) are added across devices                             */
  HT_SHARD_TILES = 1    /* handle k renders every sample of the tiles k, k+n, k+2n, ...; nothing is
                           added, each device resolves its own tiles and the first one gathers the
                           pictures (4 + 12 bytes per pixel, 1/n of the frame per device)             */
};

/* Work counters (per call; n_* only when collect_counters = 1). */
typedef struct ht_stats {
  double seconds_total;   /* wall time of the call                                 */
  double seconds_kernel;  /* device time of the path-tracing launches (HIP events) */
  int64_t samples;        /* camera paths traced: the call's pixels (all of the frame, or those of
                             its own tiles) x samples_per_pixel                     */
  int32_t launches;       /* path-tracing kernel launches                           */
  int32_t samples_per_pixel; /* samples per pixel the call launched (fewer than asked for when cancelled) */
  int64_t n_rays;         /* rays traced (camera + scattered)                       */
  int64_t n_box;          /* ray / AABB slab tests                                  */
  int64_t n_sphere;       /* ray / sphere tests                                     */
  int64_t n_rect;         /* ray / rect tests                                       */
  int64_t n_tri;          /* ray / triangle tests                                   */
  int64_t n_shade[4];     /* scatter evaluations per HT_MAT_*                       */
  int64_t n_tex[4];       /* texture evaluations per HT_TEX_* (leaf or checker)     */
  int64_t n_steps;        /* wave-loop iterations x 64 (lane slots offered)         */
  int64_t n_active;       /* lane slots that held a live path                       */
  int64_t n_shade_rounds; /* wave-level shading rounds                               */
  int64_t n_regen_rounds; /* wave-level camera-path start rounds                     */
  int64_t n_miss;         /* paths that left the scene (background)                  */
  int64_t cyc_trav, cyc_shade, cyc_regen; /* wave cycles spent per phase, summed over waves */
  int32_t block_threads;  /* launch shape of the path-tracing kernel: threads per workgroup, */
  int32_t blocks_per_cu;  /*   resident workgroups per compute unit,                        */
  int32_t lds_bytes;      /*   LDS bytes per workgroup (scene image + stacks + tile sums)   */
  int32_t n_cu;           /*   compute units of the device                                  */
  double shader_clock_mhz; /* in-kernel clock: s_memtime ticks per 100 MHz s_memrealtime tick */
  int32_t batch;          /* samples per pixel of a full launch as the library planned it   */
  int32_t kernel_sgprs;   /* scalar-register budget the chosen kernel is compiled to (80 / 96; 0: not capped) */
  int64_t n_node_steps;   /* the part of n_steps offered by node steps (the rest: leaf steps)       */
  char kernel[64];        /* the path-tracing kernel the library chose for this scene, as a profiler
                             shows it: "render_kernel_8<4u>", "render_kernel_7<0u, true>",
                             "render_kernel<float, false, 1u, false>", or "wavefront"           */
} ht_stats;

typedef struct ht_scene ht_scene; /* opaque */

/* ---- entry points -------------------------------------------------------- */

/* Library version (HT_VERSION). */
int ht_version(void);

/* Number of HIP devices visible to the process (0 when there is none). */
int ht_device_count(void);

/* Flatten `desc` into device buffers on `device` (-1: current) and build the
 * BVH.  Replaces the object graph loadScene hands to renderFrame
 * (main.scala:57-60, BVH.scala:18-33).  The arrays in `desc` are copied. */
int ht_scene_create(const ht_scene_desc* desc, int device, ht_scene** out);
void ht_scene_destroy(ht_scene* scene);

/* Render `params->samples` paths per pixel and write the picture.
 * Replaces the loop main.scala:79-94.
 *   rgb_linear   [width*height*3] mean linear radiance, row 0 = j = 0 = BOTTOM
 *                row of the picture (the reference's loop order); may be NULL.
 *   rgb_packed   [width*height] 0x00RRGGBB after Vec3.toRGB (Vec3.scala:82-84),
 *                row 0 = TOP, i.e. what img.setRGB(i, H-j-1, ..) leaves in a
 *                TYPE_INT_RGB BufferedImage (main.scala:93); may be NULL.
 * Both are host pointers. */
int ht_render(ht_scene* scene, const ht_camera* camera, const ht_render_params* params,
              float* rgb_linear, uint32_t* rgb_packed, ht_stats* stats_out);

/* Radiance sums are signed 32.32 fixed point (value = sum * 2^-32).  Integer
 * sums commute, so a frame is bit-identical however it is cut into batches,
 * launches, calls or GPUs. */
#define HT_ACCUM_ONE 4294967296.0

/* An entry point that allocates nothing.  It has no counterpart in the reference; it is for
 * callers that cannot allocate device memory or wait on the device: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The same frame as
 * ht_render (same kernels, same sums), with these differences:
 *   - rgb_linear_dev / rgb_packed_dev are DEVICE buffers on the scene's device (layouts as in
 *     ht_render; either may be NULL);
 *   - all work is enqueued on `stream` (a hipStream_t, NULL: default stream) and the call
 *     returns without waiting for it; there are no statistics and no cancel point;
 *   - the radiance sums and the work-queue heads live in a static device array of the
 *     library (48 MiB + 4 KiB that every GPU the library has run on holds until the process
 *     ends), so width * height may not exceed HT_NOALLOC_MAX_PIXELS (HT_E_TOO_LARGE);
 *   - the call makes no hipMalloc / hipFree and no allocating library call (error
 *     messages go to a fixed per-thread buffer).
 * Every call on one GPU uses that GPU's one array: calls that overlap in time on the same
 * GPU (two streams, two threads, two scene handles) race and give wrong pictures; order
 * them on one stream or with events.  The array keeps the last call's sums; each call
 * clears the part it uses on `stream` before rendering into it. */
#define HT_NOALLOC_MAX_PIXELS (1 << 21) /* 2 097 152 >= 1920 x 1080 */
int ht_render_noalloc(ht_scene* scene, const ht_camera* camera, const ht_render_params* params,
                      float* rgb_linear_dev, uint32_t* rgb_packed_dev, void* stream);

/* Multi-device building block: add this call's radiance SUMS (not means) into
 * `accum_dev`, a DEVICE buffer of width*height*3 int64 on the scene's device
 * (row 0 = bottom; the caller zeroes it before the first call).  `stream` is a
 * hipStream_t (NULL: default stream).  The call returns after the work is enqueued
 * and, if stats_out is non-NULL, complete.  The caller reduces the buffers of all
 * ranks (RCCL reduce(sum) of int64 over xGMI) and then calls ht_resolve on the
 * root. */
int ht_render_accumulate(ht_scene* scene, const ht_camera* camera,
                         const ht_render_params* params, int64_t* accum_dev, void* stream,
                         ht_stats* stats_out);

/* One frame rendered by several scene handles at once, one host thread each:
 * handle k renders the global sample indices [k*S/n, (k+1)*S/n) of every pixel
 * (S = params->samples, starting at params->sample_offset) into a sum buffer on
 * its own device; the partial sums are then copied to scenes[0]'s device
 * (device-to-device, over xGMI between GPUs of one node) and added there, and the
 * frame is resolved as in ht_render (params->shard_mode = HT_SHARD_SAMPLES, the default).
 * With HT_SHARD_TILES handle k renders every sample of the tiles k, k + n, ... instead,
 * resolves them on its own device, and only the pictures travel to scenes[0]'s device.
 * The samples of a pixel are a plain sum in the
 * reference (main.scala:87-91), so any split gives the picture ht_render gives for
 * the same seed, bit for bit.  The handles must hold the same scene; they may live
 * on different devices (one per GPU of the node: the single-process form of spp
 * sharding, for hosts like the JVM that cannot run one process per GPU) or share
 * one.  params->device is ignored.  stats_out: samples and launches are totals,
 * seconds_kernel is the slowest shard's. */
int ht_render_sharded(ht_scene* const* scenes, int32_t n_scenes, const ht_camera* camera,
                      const ht_render_params* params, float* rgb_linear, uint32_t* rgb_packed,
                      ht_stats* stats_out);

/* Divide a DEVICE sum buffer by total_samples and write the host pictures as
 * ht_render does (mean, then Vec3.toRGB + row flip). */
int ht_resolve(ht_scene* scene, const int64_t* accum_dev, int32_t width, int32_t height,
               int64_t total_samples, float* rgb_linear, uint32_t* rgb_packed, void* stream);

/* As ht_resolve, but the two pictures are DEVICE buffers on the scene's device and
 * the call only enqueues work on `stream` (no allocation, no host copy): for
 * callers that keep the frame on the GPU. */
int ht_resolve_device(ht_scene* scene, const int64_t* accum_dev, int32_t width, int32_t height,
                      int64_t total_samples, float* rgb_linear_dev, uint32_t* rgb_packed_dev,
                      void* stream);

/* Tile sharding, the pictures.  A shard that rendered the tiles (tile_first, tile_stride) of a
 * frame resolves them into TILE BUFFERS on its own device: owned tile k (tile number
 * tile_first + k * tile_stride) fills pixels [64 k, 64 k + 64) of the buffer, pixel
 * q = 8 * (row within the tile, bottom row first) + column; pixels outside the frame are 0.
 *   packed_tiles_dev  0x00RRGGBB codes (Vec3.toRGB, Vec3.scala:82-84), one uint32 per pixel
 *   mean_tiles_dev    mean linear radiance, three floats per pixel
 * Either may be NULL.  ht_shard_tile_pixels(width, height, n_shards) is the number of pixels a
 * shard's buffer must hold (64 * ceil(tiles / n_shards), the same for every shard). */
int64_t ht_shard_tile_pixels(int32_t width, int32_t height, int32_t n_shards);
int ht_resolve_tiles_device(ht_scene* scene, const int64_t* accum_dev, int32_t width, int32_t height,
                            int64_t total_samples, int32_t tile_first, int32_t tile_stride,
                            float* mean_tiles_dev, uint32_t* packed_tiles_dev, void* stream);
/* The frame from the tile buffers of all n_shards shards, gathered into one DEVICE array each
 * (shard g's buffer starts g * ht_shard_tile_pixels(..) pixels in: the layout an all-gather or a
 * gather to the root leaves).  Outputs as in ht_resolve_device: rgb_linear_dev row 0 = bottom,
 * rgb_packed_dev row 0 = top (the flip of img.setRGB(i, H-j-1, ..), main.scala:93); a NULL input
 * skips its output. */
int ht_assemble_tiles_device(ht_scene* scene, int32_t width, int32_t height, int32_t n_shards,
                             const float* mean_tiles_all_dev, const uint32_t* packed_tiles_all_dev,
                             float* rgb_linear_dev, uint32_t* rgb_packed_dev, void* stream);

/* A frame: a device accumulation buffer owned by the library, for callers that
 * cannot allocate device memory themselves (a JVM host) and for progressive
 * rendering.  The reference's renderFrame fills its BufferedImage in one go
 * (main.scala:79-96); with a frame the host adds sample batches, reads the picture
 * after any of them (the GUI's live view, main.scala:181) and can stop between
 * batches (shouldBreak, main.scala:80). */
typedef struct ht_frame ht_frame; /* opaque */
int ht_frame_create(ht_scene* scene, int32_t width, int32_t height, ht_frame** out);
void ht_frame_destroy(ht_frame* frame);
/* Forget all samples (the sums become zero). */
int ht_frame_clear(ht_frame* frame);
/* Add params->samples paths per pixel starting at global sample index
 * params->sample_offset; width / height must match the frame.  Synchronous. */
int ht_frame_accumulate(ht_frame* frame, const ht_camera* camera, const ht_render_params* params,
                        ht_stats* stats_out);
/* The picture of everything accumulated so far (host buffers as in ht_render;
 * either may be NULL).  Fails with HT_E_INVALID while the frame holds no samples. */
int ht_frame_read(ht_frame* frame, float* rgb_linear, uint32_t* rgb_packed);
/* Samples per pixel accumulated so far. */
int64_t ht_frame_samples(const ht_frame* frame);

/* First-hit buffers with jitter off (u = (i+0.5)/(W-1), v = (j+0.5)/(H-1),
 * aperture forced to 0): what world.hit(ray, 0.001, +inf) returns for the
 * camera ray of every pixel (Camera.scala:28, BVH.scala:7-14).
 *   t        [W*H]   ray parameter of the hit (+inf on a miss)
 *   prim_id  [W*H]   flattened primitive id (-1 on a miss)
 *   normal   [W*H*3] Hit.normal (Hittable.scala:7): the primitive's outward
 *                    normal, flipped to face the ray, NOT normalised
 * Row 0 = j = 0 = bottom.  `precision` as in ht_render_params.  In fp64
 * precision the library converts to the float buffers at the end; t64/normal64
 * (may be NULL) receive the unrounded values. */
int ht_first_hit(ht_scene* scene, const ht_camera* camera, int32_t width, int32_t height,
                 int32_t precision, float* t, int32_t* prim_id, float* normal,
                 double* t64, double* normal64);

/* Ask a running ht_render / ht_render_accumulate on `scene` to stop at the next
 * sample-batch boundary (the analogue of shouldBreak(), main.scala:80). */
int ht_cancel(ht_scene* scene);

/* Message for the last failed call on the calling thread ("" if none). */
const char* ht_last_error(void);

/* Introspection used by tests and the benchmark: sizes of the flattened scene
 * (nodes of the library's own BVH, bytes of the on-chip scene image, depth). */
typedef struct ht_scene_info {
  int32_t n_prims, n_nodes, bvh_depth, n_materials, n_textures;
  int32_t lds_bytes_fp32, lds_bytes_fp64; /* size of the scene image per precision      */
  int32_t device;
  int32_t hbm_store_fp32, hbm_store_fp64; /* 0: the image is traversed in LDS (on-chip
                                             store); 1: it is too large for that and is
                                             traversed where it lies in HBM               */
  double sah_before, sah_after;           /* surface-area cost of the hierarchy (sum of the half-areas
                                             of its inner nodes' boxes) as built and after the
                                             optimiser (option "bvh_optimise"); 0: not run  */
} ht_scene_info;
int ht_scene_get_info(const ht_scene* scene, ht_scene_info* out);

/* Process-wide tuning knobs of the device path (no counterpart in the reference):
 *   "wait_threshold"    lanes of a wave that may wait for shading before
 *                       traversal yields (1..64; 0, the default: 40 for the on-chip store, 32
 *                       for the HBM store)
 *   "max_blocks_per_cu" cap on resident workgroups per compute unit
 *   "max_waves_per_cu"  cap on resident waves per compute unit (4..32; 0, the default: what the
 *                       kernel variant's scalar registers admit, 28 or 32; the occupancy query
 *                       answers 32 for all)
 *   "flat_kernels"      1 (default): a scene without a hierarchy (every primitive tested when a ray
 *                       starts) runs a kernel variant compiled without the traversal phase; 0: the
 *                       general variant (same sums; for comparing the two)
 *   "bvh_optimise_max_nodes"  largest hierarchy (inner nodes) the optimiser is run over (8192)
 *   "block_threads"     force the workgroup size (multiple of 128; 0 = choose for occupancy)
 *   "node_den"          node steps continue while inner-node lanes * node_den >= travelling lanes
 *   "default_batch"     samples per launch when ht_render_params.batch = 0
 *   "chunk"             samples of one 8x8 tile that form one unit of the work queue
 *                       (0, the default: 16 for launches of 1024 samples or more, else 8)
 *   "chunk_tail"        the same for the last "chunk" samples of a launch (smaller units
 *                       at the end shorten the tail of the launch)
 *   "gpu_bvh"           n > 0: scenes created from now on whose hierarchy holds at least n primitives
 *                       get it built on the device (linear BVH) instead of by the host's SAH
 *                       builder: scene creation 0.37 instead of 0.45 s at a million triangles,
 *                       frames 1.5x slower; 0, the default: always the host
 *   "force_hbm_scene"   scenes created while this is 1 keep their image in HBM even if it
 *                       would fit on chip (testing the large-scene path on small scenes) */
/*   "wavefront"         1: fp32 renders run the wavefront variant of the path tracer (one kernel
 *                       per stage: generate, trace, shade with compaction, reduce; path state in
 *                       device memory between them) instead of the persistent kernel; same paths,
 *                       same sums up to the compiler's placement of fused multiply-adds (a hit
 *                       point may differ in its last bit).  The calls then synchronise with the stream.
 *   "wavefront_paths"   paths in flight of the wavefront variant (2^12..2^26, default 2^24)
 *   "wavefront_refill"  idle lanes of a trace wave that trigger a refill from the ray list (default 32) */
/*   "bvh_optimise"      passes of the insertion-based optimiser over the hierarchy of scenes created from
 *                       now on (default 4; 0: the hierarchy stays as the top-down builder made it)
 *   "timeline"          1: render launches record, per 10.24 us bin of the device's 100 MHz clock, how
 *                       many work units their waves fetched, how many waves found the queue empty and
 *                       how many ended (profiling; read with ht_debug_timeline)
 *   "xcd_bands"         1 (default): a launch over a scene in the HBM store cuts its tiles into eight bands
 *                       (rows of the picture) with a queue head each, and a wave takes its work from the
 *                       band of the XCD it runs on, so that one part of the mesh stays in one XCD's L2;
 *                       0: one queue for all.  Order of the work only: the sums are the same. */
int ht_set_option(const char* name, int value);

/* Profiling aid (no counterpart in the reference): the counters of the option "timeline" for the
 * launches of this scene since the last call, then cleared.  out receives 3 x 4096 words: units
 * fetched, waves that found the queue empty, waves that ended; word k of each block counts the
 * events in the bin (device clock ticks / 1024) mod 4096.  Waits for the device. */
int ht_debug_timeline(ht_scene* scene, uint32_t* out, int32_t n_words);

#ifdef __cplusplus
}
#endif
#endif /* HIPTRACE_H */
