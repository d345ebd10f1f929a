/* asan_smoke.c — drives the CPU restatement (oracle.c) under AddressSanitizer / UBSan:
 * a small world with every primitive, material and texture kind, first-hit buffers and
 * renders in both world modes and with both samplers.  Test infrastructure only
 * (`make -C oracle asan`, run by tests/test_sanitizers.py); prints "ok" when the two world
 * modes agree on every primitive id. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define NP 9
#define NM 5
#define NT 6

int main(void) {
  static int32_t prim_type[NP] = {HT_PRIM_SPHERE, HT_PRIM_SPHERE, HT_PRIM_SPHERE, HT_PRIM_XYRECT, HT_PRIM_XZRECT,
                                  HT_PRIM_YZRECT, HT_PRIM_TRIANGLE, HT_PRIM_TRIANGLE, HT_PRIM_SPHERE};
  static int32_t prim_material[NP] = {0, 1, 2, 3, 0, 4, 1, 0, 4};
  static double prim_data[NP * HT_PRIM_STRIDE] = {
      0, -1000, 0, 1000, 0, 0, 0, 0, 0,   /* ground */
      0, 1, 0, 1, 0, 0, 0, 0, 0,          /* glass */
      -3, 1, 0.5, 1, 0, 0, 0, 0, 0,       /* glossy */
      -1, 1, 2, 3, -4, 0, 0, 0, 0,        /* light, z = -4 */
      -6, 6, -6, 6, 5, 0, 0, 0, 0,        /* ceiling, y = 5 */
      0, 3, -3, 3, 4, 0, 0, 0, 0,         /* wall, x = 4 */
      2, 0, 1, 3, 0, 2, 2.5, 2, 1.5,      /* triangles */
      -2, 0.2, 2, -1, 0.2, 2.5, -1.5, 1.5, 2.2,
      2.5, 0.6, -1, 0.6, 0, 0, 0, 0, 0};
  static int32_t mat_type[NM] = {HT_MAT_DIFFUSE, HT_MAT_GLASS, HT_MAT_GLOSSY, HT_MAT_EMISSION, HT_MAT_DIFFUSE};
  static int32_t mat_texture[NM] = {1, 0, 2, 0, 4};
  static double mat_param[NM] = {0, 1.5, 0.2, 4.0, 0};
  static int32_t tex_type[NT] = {HT_TEX_SOLID, HT_TEX_CHECKER, HT_TEX_SOLID, HT_TEX_PERLIN, HT_TEX_IMAGE, HT_TEX_SOLID};
  static double tex_data[NT * HT_TEX_STRIDE] = {1, 1, 1, 0, 3, 0, 0, 0, 0.7, 0.6, 0.5, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0.2, 0.3, 0.1, 0};
  static int32_t tex_child[NT * 2] = {0, 0, 3, 5, 0, 0, 0, 0, 0, 0, 0, 0};
  static int32_t image_width[1] = {2}, image_height[1] = {2};
  static int64_t image_offset[1] = {0};
  static uint32_t image_texels[4] = {0xFF0000, 0x00FF00, 0x0000FF, 0xFFFFFF};
  static double ranvec[256 * 3];
  static int32_t perm[3 * 256];
  uint32_t x = 12345u;
  for (int i = 0; i < 256; ++i) {
    double v[3], n = 0;
    for (int k = 0; k < 3; ++k) { x = x * 1664525u + 1013904223u; v[k] = (double)(x >> 8) / 8388608.0 - 1.0; n += v[k] * v[k]; }
    n = sqrt(n);
    for (int k = 0; k < 3; ++k) ranvec[i * 3 + k] = v[k] / n;
  }
  for (int a = 0; a < 3; ++a) {
    for (int i = 0; i < 256; ++i) perm[a * 256 + i] = i;
    for (int i = 255; i > 0; --i) {
      x = x * 1664525u + 1013904223u;
      int j = (int)((x >> 8) % (uint32_t)(i + 1)), t = perm[a * 256 + i];
      perm[a * 256 + i] = perm[a * 256 + j];
      perm[a * 256 + j] = t;
    }
  }
  ht_scene_desc d;
  memset(&d, 0, sizeof d);
  d.n_prims = NP; d.prim_type = prim_type; d.prim_material = prim_material; d.prim_data = prim_data;
  d.n_materials = NM; d.mat_type = mat_type; d.mat_texture = mat_texture; d.mat_param = mat_param;
  d.n_textures = NT; d.tex_type = tex_type; d.tex_data = tex_data; d.tex_child = tex_child;
  d.n_images = 1; d.image_width = image_width; d.image_height = image_height; d.image_offset = image_offset;
  d.image_texels = image_texels;
  d.n_perlin = 1; d.perlin_ranvec = ranvec; d.perlin_perm = perm;
  /* (a view axis that is not aimed at an edge of the x = 4 wall: a hit exactly on a box face is
   * decided by rounding, and the two world modes may then legitimately differ) */
  ht_camera cam = {{8.1, 3.2, 6.3}, {0.13, 1.07, -0.21}, {0, 1, 0}, 35, 16.0 / 9.0, 0.1, -1, {0.6, 0.7, 0.9}};

  or_scene* s = NULL;
  if (or_scene_create(&d, 7, &s) != HT_OK) { fprintf(stderr, "or_scene_create failed\n"); return 1; }
  enum { W = 48, H = 27 };
  double* t = malloc(sizeof(double) * W * H), *n = malloc(sizeof(double) * W * H * 3), *uv = malloc(sizeof(double) * W * H * 2);
  int32_t* id0 = malloc(sizeof(int32_t) * W * H), *id1 = malloc(sizeof(int32_t) * W * H);
  double* rgb = malloc(sizeof(double) * W * H * 3);
  uint32_t* packed = malloc(sizeof(uint32_t) * W * H);
  int bad = 0;
  if (or_first_hit(s, &cam, W, H, OR_WORLD_REFERENCE_BVH, 0.5, 0.5, t, id0, n, uv) != HT_OK) bad = 1;
  if (or_first_hit(s, &cam, W, H, OR_WORLD_LINEAR, 0.5, 0.5, t, id1, n, uv) != HT_OK) bad = 1;
  if (or_first_hit_fp32_rays(s, &cam, W, H, OR_WORLD_LINEAR, 0.5, 0.5, t, id1, n, uv) != HT_OK) bad = 1;
  for (int i = 0; i < W * H; ++i) if (id0[i] != id1[i]) { bad = 1; fprintf(stderr, "id %d: %d vs %d\n", i, id0[i], id1[i]); }
  for (int sampler = 0; sampler < 2; ++sampler)
    for (int mode = 0; mode < 2; ++mode) {
      or_counters c;
      if (or_render(s, &cam, W, H, 3, 5, 50, 1, sampler, mode, 99, 2, rgb, packed, &c) != HT_OK) bad = 1;
      for (int i = 0; i < W * H * 3; ++i) if (!(rgb[i] >= 0.0)) { bad = 1; fprintf(stderr, "rgb %d %g (sampler %d mode %d)\n", i, rgb[i], sampler, mode); break; }
    }
  or_scene_destroy(s);
  free(t); free(n); free(uv); free(id0); free(id1); free(rgb); free(packed);
  puts(bad ? "MISMATCH" : "ok");
  return bad;
}
