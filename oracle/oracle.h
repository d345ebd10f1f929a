/* oracle.h — CPU restatement (fp64, plain C) of the reference renderer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (Scala 3 on the JVM) ships no tests, golden
 * vectors or fixtures for this path and cannot be compiled or run where this
 * was written (no JDK / sbt / dependency jars).  This file is therefore pinned
 * only by closed-form known-answer tests and by line-by-line review against
 * the cited Scala, not by outputs of the reference itself.
 *
 * The scene description is the one the product's C ABI takes (hiptrace.h), so a
 * test builds it once and hands it to both sides.
 */
#ifndef HT_ORACLE_H
#define HT_ORACLE_H

#include <stdint.h>
#include "../include/hiptrace.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct or_scene or_scene;

/* How random numbers are drawn. */
enum {
  /* The reference's own procedure: a sequential stream per camera path and
   * the rejection loops of Vec3.scala:17-27, doubles made from 53 random bits
   * the way java.util.Random.nextDouble does.  (The reference's stream itself,
   * an unseeded global java.util.Random, cannot be matched by anything.) */
  OR_SAMPLER_REFERENCE = 0,
  /* Same distributions, but every random decision is a pure function of
   * (seed, pixel, sample, bounce): one Philox4x32-10 block per event, 24-bit
   * uniforms, uniform ball / disk sampled directly instead of by rejection.
   * This is the mapping the device kernels use, so a device render and an
   * oracle render follow the same paths. */
  OR_SAMPLER_MATCHED = 1
};

/* How world.hit finds the closest hit. */
enum {
  /* The reference's structure: BVH.apply's random-axis median split
   * (BVH.scala:18-33) with AABB.hit's slab test (AABB.scala:5-19), an STL
   * entry nested as one world entry (main.scala:60). */
  OR_WORLD_REFERENCE_BVH = 0,
  /* A linear scan in flattened order with the same per-primitive tests;
   * used to cross-check the BVH restatement. */
  OR_WORLD_LINEAR = 1
};

typedef struct or_counters {
  int64_t n_rays, n_box, n_sphere, n_rect, n_tri;
  int64_t n_shade[4];
} or_counters;

/* bvh_seed seeds the axis choice Random.between(0, 3) of BVH.scala:19. */
int or_scene_create(const ht_scene_desc* desc, uint64_t bvh_seed, or_scene** out);
void or_scene_destroy(or_scene* s);

/* world.hit(camera ray, 0.001, +inf) per pixel with the lens closed
 * (Camera.scala:19-28).  u = (i+off_x)/(W-1), v = (j+off_y)/(H-1).
 * Row 0 = j = 0 = bottom.  Outputs may be NULL.
 *   t [W*H] (+inf on miss), prim_id [W*H] (-1 on miss),
 *   normal [W*H*3] = Hit.normal (Hittable.scala:7), uv [W*H*2] = Hit.uv. */
int or_first_hit(const or_scene* s, const ht_camera* cam, int32_t W, int32_t H, int world_mode,
                 double off_x, double off_y, double* t, int32_t* prim_id, double* normal,
                 double* uv);
/* or_first_hit for camera rays rounded to fp32 (origin and direction, component-wise to the
 * nearest float) before the fp64 trace: the best an fp32 ray can do. */
int or_first_hit_fp32_rays(const or_scene* s, const ht_camera* cam, int32_t W, int32_t H, int world_mode,
                           double off_x, double off_y, double* t, int32_t* prim_id, double* normal,
                           double* uv);

/* renderFrame (main.scala:79-94).  rgb_linear [W*H*3] = mean radiance, row 0 =
 * bottom; rgb_packed [W*H] = toRGB codes, row 0 = top (BufferedImage order).
 * nthreads <= 0: all cores (rows are split over threads; the result does not
 * depend on the thread count because every path owns its random stream). */
int or_render(const or_scene* s, const ht_camera* cam, int32_t W, int32_t H, int32_t samples,
              int32_t sample_offset, int32_t max_depth, int32_t jitter, int sampler,
              int world_mode, uint64_t seed, int nthreads, double* rgb_linear,
              uint32_t* rgb_packed, or_counters* counters);

/* Radiance of ONE camera path, for path-by-path comparisons. */
int or_path_radiance(const or_scene* s, const ht_camera* cam, int32_t W, int32_t H, int32_t i,
                     int32_t j, int32_t sample, int32_t max_depth, int32_t jitter, int sampler,
                     int world_mode, uint64_t seed, double rgb[3]);

/* Single-function probes for known-answer tests. */
void or_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* One primitive of the scene against one ray; returns 1 on hit. */
int or_prim_hit(const or_scene* s, int32_t prim, const double origin[3], const double dir[3],
                double t_min, double t_max, double* t, double normal_out[3], double uv[2],
                int* front_face);
/* texture(uv, p) for texture index tex. */
void or_texture(const or_scene* s, int32_t tex, const double uv[2], const double p[3],
                double rgb[3]);
/* Perlin noise / turbulence of table k (Texture.scala:58-88). */
double or_perlin_noise(const or_scene* s, int32_t k, const double p[3]);
double or_perlin_turb(const or_scene* s, int32_t k, const double p[3]);
/* material.scatter with explicit uniforms u[4] in the MATCHED mapping.
 * Returns 1 if scattered; fills direction and attenuation; emit always filled. */
int or_scatter_matched(const or_scene* s, int32_t material, const double ray_o[3],
                       const double ray_d[3], double t, const double outward_normal[3],
                       const double uv[2], const double u[4], double dir_out[3],
                       double atten[3], double emit[3]);
/* Vec3.toRGB (Vec3.scala:82-84). */
uint32_t or_to_rgb(const double rgb[3]);
/* Camera basis (Camera.scala:5-17): origin, lower_left_corner, horizontal,
 * vertical, u, v, w (3 doubles each) then lens_radius -> out[22]. */
void or_camera_basis(const ht_camera* cam, double out[22]);
/* Reference-BVH statistics: number of inner nodes and depth. */
void or_bvh_stats(const or_scene* s, int32_t* n_inner, int32_t* depth);

#ifdef __cplusplus
}
#endif
#endif
