/* oracle.c — fp64 CPU restatement of the reference renderer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the reference has
 * no tests / golden vectors for this path and cannot be run here; this file is
 * pinned by closed-form known-answer tests and review against the cited Scala.
 *
 * Citations are to the reference tree (Ivoah/RayTracing):
 *   S = shared/src/main/scala, J = jvm/src/main/scala.
 * The file is compiled with -ffp-contract=off so that, like JVM doubles, no
 * multiply-add is fused; expressions keep the reference's association order.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* JVM arithmetic helpers                                              */
/* ------------------------------------------------------------------ */

/* java.lang.Math.max / min: NaN if either argument is NaN. */
static double jmax(double a, double b) {
  if (a != a) return a;
  if (b != b) return b;
  if (a == 0.0 && b == 0.0) return signbit(a) ? b : a;
  return a > b ? a : b;
}
static double jmin(double a, double b) {
  if (a != a) return a;
  if (b != b) return b;
  if (a == 0.0 && b == 0.0) return signbit(a) ? a : b;
  return a < b ? a : b;
}
/* Double.toInt (JLS 5.1.3): NaN -> 0, saturating, truncation toward zero. */
static int32_t d2i(double d) {
  if (d != d) return 0;
  if (d >= 2147483647.0) return 2147483647;
  if (d <= -2147483648.0) return (int32_t)(-2147483647 - 1);
  return (int32_t)d;
}
/* Util.clamp (S/Util.scala:4-5) */
static double clampd(double v, double lo, double hi) { return jmin(jmax(v, lo), hi); }
static int32_t clampi(int32_t v, int32_t lo, int32_t hi) {
  int32_t m = v > lo ? v : lo;
  return m < hi ? m : hi;
}

/* ------------------------------------------------------------------ */
/* Vec3 (S/Vec3.scala:49-68), Vec2, Ray (S/Ray.scala:1-3)             */
/* ------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;
typedef struct { double x, y; } v2;
typedef struct { v3 o, d; } ray_t;

static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* scalar forms go through the implicit Double -> Vec3(v,v,v) (Vec3.scala:7) */
static v3 smul(double s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static v3 sdiv(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 vcross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static double vlen2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static double vlen(v3 a) { return sqrt(vlen2(a)); }
static v3 vunit(v3 a) { return sdiv(a, vlen(a)); }
/* reflect (Vec3.scala:64): this - 2*dot(other)*other */
static v3 vreflect(v3 a, v3 n) { return vsub(a, smul(2 * vdot(a, n), n)); }
static double vcomp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static v3 ray_at(const ray_t* r, double t) { return vadd(r->o, smul(t, r->d)); }

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; public algorithm)              */
/* ------------------------------------------------------------------ */
void or_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* java.util.Random (48-bit LCG; public JDK specification) — only used for the
 * axis choice of the reference BVH builder. */
typedef struct { uint64_t seed; } jrandom;
static void jr_init(jrandom* r, uint64_t seed) {
  r->seed = (seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
}
static int32_t jr_next(jrandom* r, int bits) {
  r->seed = (r->seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
  return (int32_t)((int64_t)r->seed >> (48 - bits));
}
static int32_t jr_next_int(jrandom* r, int32_t bound) {
  int32_t x = jr_next(r, 31);
  int32_t m = bound - 1;
  if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)x) >> 31);
  for (int32_t u = x; (int32_t)((uint32_t)u - (uint32_t)(x = u % bound) + (uint32_t)m) < 0;
       u = jr_next(r, 31)) {
  }
  return x;
}

/* ------------------------------------------------------------------ */
/* Random sources for one camera path                                  */
/* ------------------------------------------------------------------ */
typedef struct {
  int mode;            /* OR_SAMPLER_* */
  uint32_t key[2];
  uint32_t pixel, sample;
  /* REFERENCE mode: sequential stream of doubles */
  uint32_t block;      /* next Philox block index */
  uint32_t buf[4];
  int have;            /* unread words in buf */
  /* MATCHED mode: uniforms of the current event */
  double u[4];
} path_rng;

static void rng_begin(path_rng* g, int mode, uint64_t seed, uint32_t pixel, uint32_t sample) {
  g->mode = mode;
  g->key[0] = (uint32_t)seed;
  g->key[1] = (uint32_t)(seed >> 32);
  g->pixel = pixel;
  g->sample = sample;
  g->block = 0;
  g->have = 0;
}
/* MATCHED: load the four 24-bit uniforms of event `bounce` (0 = camera). */
static void rng_event(path_rng* g, uint32_t bounce) {
  uint32_t ctr[4] = {g->pixel, g->sample, bounce, 0u};
  uint32_t out[4];
  or_philox4x32_10(ctr, g->key, out);
  for (int i = 0; i < 4; ++i) g->u[i] = (double)(out[i] >> 8) * (1.0 / 16777216.0);
}
/* REFERENCE: Random.nextDouble — 26 + 27 random bits * 2^-53. */
static double rng_next_double(path_rng* g) {
  if (g->have < 2) {
    uint32_t ctr[4] = {g->pixel, g->sample, 0x80000000u | g->block, 1u};
    or_philox4x32_10(ctr, g->key, g->buf);
    g->block++;
    g->have = 4;
  }
  uint32_t a = g->buf[4 - g->have], b = g->buf[5 - g->have];
  g->have -= 2;
  return (double)(((uint64_t)(a >> 6) << 27) + (uint64_t)(b >> 5)) * 0x1.0p-53;
}
/* scala.util.Random.between(Double, Double) */
static double rng_between(path_rng* g, double lo, double hi) {
  double next = rng_next_double(g) * (hi - lo) + lo;
  return next < hi ? next : nextafter(hi, -INFINITY);
}

static const double OR_PI = 3.14159265358979323846; /* scala.math.Pi */

/* Vec3.random_unit_vector (Vec3.scala:29-34) */
static v3 sample_unit_vector(path_rng* g) {
  double a, z;
  if (g->mode == OR_SAMPLER_REFERENCE) {
    a = rng_between(g, 0, 2 * OR_PI);
    z = rng_between(g, -1.0, 1.0);
  } else {
    a = 2 * OR_PI * g->u[0];
    z = 2.0 * g->u[1] - 1.0;
  }
  double r = sqrt(1 - z * z);
  return V(r * cos(a), r * sin(a), z);
}
/* Vec3.random_in_unit_sphere (Vec3.scala:17-21) */
static v3 sample_in_unit_sphere(path_rng* g) {
  if (g->mode == OR_SAMPLER_REFERENCE) {
    for (;;) {
      v3 v;
      v.x = rng_between(g, -1, 1);
      v.y = rng_between(g, -1, 1);
      v.z = rng_between(g, -1, 1);
      if (vlen2(v) <= 1) return v;
    }
  }
  /* same distribution, sampled directly: uniform direction times cbrt(U) */
  v3 d = sample_unit_vector(g);
  return smul(cbrt(g->u[2]), d);
}
/* Vec3.random_in_unit_disk (Vec3.scala:23-27) */
static v3 sample_in_unit_disk(path_rng* g) {
  if (g->mode == OR_SAMPLER_REFERENCE) {
    for (;;) {
      v3 v;
      v.x = rng_between(g, -1.0, 1.0);
      v.y = rng_between(g, -1.0, 1.0);
      v.z = 0;
      if (vlen2(v) <= 1) return v;
    }
  }
  double r = sqrt(g->u[2]);
  double phi = 2 * OR_PI * g->u[3];
  return V(r * cos(phi), r * sin(phi), 0);
}

/* ------------------------------------------------------------------ */
/* Scene                                                               */
/* ------------------------------------------------------------------ */
typedef struct { v3 mn, mx; } aabb;

enum { OBJ_BVH = 100 };
typedef struct obj {
  int kind;   /* HT_PRIM_* or OBJ_BVH */
  int prim;   /* primitive id for leaves */
  aabb box;
  struct obj *left, *right;
} obj;

struct or_scene {
  int32_t n_prims;
  int32_t* prim_type;
  int32_t* prim_material;
  double* prim_data;
  int32_t n_materials;
  int32_t* mat_type;
  int32_t* mat_texture;
  double* mat_param;
  int32_t n_textures;
  int32_t* tex_type;
  double* tex_data;
  int32_t* tex_child;
  int32_t n_images;
  int32_t *image_width, *image_height;
  int64_t* image_offset;
  uint32_t* image_texels;
  int32_t n_perlin;
  double* perlin_ranvec;
  int32_t* perlin_perm;
  obj* world;       /* reference-style BVH */
  obj* pool;        /* node storage */
  int32_t pool_used, pool_cap;
  int32_t n_inner, depth;
};

typedef struct {
  ray_t ray;
  v3 outward_normal;
  double t;
  v2 uv;
  int material;
  int prim;
} hit_t;

/* Hit (S/Hittable.scala:4-8) */
static v3 hit_position(const hit_t* h) { return ray_at(&h->ray, h->t); }
static int hit_front_face(const hit_t* h) { return vdot(h->ray.d, h->outward_normal) < 0; }
static v3 hit_normal(const hit_t* h) {
  return hit_front_face(h) ? h->outward_normal : vneg(h->outward_normal);
}

typedef struct {
  int64_t n_rays, n_box, n_sphere, n_rect, n_tri, n_shade[4];
} ctr_t;

/* ---------------- AABB (S/AABB.scala) ---------------- */
/* AABB.hit (AABB.scala:5-19), tail recursion written as a loop.  Note the
 * reference returns true at dim >= 3 without looking at the z interval. */
static int aabb_hit(const aabb* b, const ray_t* r, double t_min, double t_max) {
  for (int dim = 0;; ++dim) {
    if (dim >= 3) return 1;
    if (t_max <= t_min) return 0;
    double invD = 1.0 / vcomp(r->d, dim);
    double t0, t1;
    if (invD >= 0) {
      t0 = (vcomp(b->mn, dim) - vcomp(r->o, dim)) * invD;
      t1 = (vcomp(b->mx, dim) - vcomp(r->o, dim)) * invD;
    } else {
      t0 = (vcomp(b->mx, dim) - vcomp(r->o, dim)) * invD;
      t1 = (vcomp(b->mn, dim) - vcomp(r->o, dim)) * invD;
    }
    t_min = jmax(t0, t_min);
    t_max = jmin(t1, t_max);
  }
}
/* AABB.+ (AABB.scala:21-32) */
static aabb aabb_union(aabb a, aabb b) {
  aabb r;
  r.mn = V(jmin(a.mn.x, b.mn.x), jmin(a.mn.y, b.mn.y), jmin(a.mn.z, b.mn.z));
  r.mx = V(jmax(a.mx.x, b.mx.x), jmax(a.mx.y, b.mx.y), jmax(a.mx.z, b.mx.z));
  return r;
}

/* ---------------- primitives (S/Hittable.scala) ---------------- */
/* Sphere.get_uv (Hittable.scala:26-32) */
static v2 sphere_uv(v3 p) {
  double phi = atan2(p.z, p.x);
  double theta = asin(p.y);
  v2 r;
  r.x = 1 - (phi + OR_PI) / (2 * OR_PI);
  r.y = (theta + OR_PI / 2) / OR_PI;
  return r;
}

static int prim_hit(const or_scene* s, int prim, const ray_t* r, double t_min, double t_max,
                    hit_t* out, ctr_t* c) {
  const double* d = s->prim_data + (size_t)prim * HT_PRIM_STRIDE;
  int kind = s->prim_type[prim];
  out->ray = *r;
  out->material = s->prim_material[prim];
  out->prim = prim;
  switch (kind) {
    case HT_PRIM_SPHERE: { /* Hittable.scala:34-49 */
      if (c) c->n_sphere++;
      v3 center = V(d[0], d[1], d[2]);
      double radius = d[3];
      v3 oc = vsub(r->o, center);
      double a = vlen2(r->d);
      double half_b = vdot(oc, r->d);
      double cc = vlen2(oc) - radius * radius;
      double discriminant = half_b * half_b - a * cc;
      if (discriminant > 0) {
        double t1 = (-half_b - sqrt(discriminant)) / a;
        double t2 = (-half_b + sqrt(discriminant)) / a;
        if (t1 < t_max && t1 > t_min) {
          out->outward_normal = sdiv(vsub(ray_at(r, t1), center), radius);
          out->t = t1;
          out->uv = sphere_uv(sdiv(vsub(ray_at(r, t1), center), radius));
          return 1;
        } else if (t2 < t_max && t2 > t_min) {
          out->outward_normal = sdiv(vsub(ray_at(r, t2), center), radius);
          out->t = t2;
          /* the reference takes uv from the t1 point here as well (:46) */
          out->uv = sphere_uv(sdiv(vsub(ray_at(r, t1), center), radius));
          return 1;
        }
      }
      return 0;
    }
    case HT_PRIM_XYRECT: { /* Hittable.scala:55-64 */
      if (c) c->n_rect++;
      double x0 = d[0], x1 = d[1], y0 = d[2], y1 = d[3], z = d[4];
      double t = (z - r->o.z) / r->d.z;
      if (t < t_min || t > t_max) return 0;
      double x = r->o.x + t * r->d.x;
      double y = r->o.y + t * r->d.y;
      if (x < x0 || x > x1 || y < y0 || y > y1) return 0;
      out->outward_normal = V(0, 0, 1);
      out->t = t;
      out->uv.x = (x - x0) / (x1 - x0);
      out->uv.y = (y - y0) / (y1 - y0);
      return 1;
    }
    case HT_PRIM_XZRECT: { /* Hittable.scala:69-78 */
      if (c) c->n_rect++;
      double x0 = d[0], x1 = d[1], z0 = d[2], z1 = d[3], y = d[4];
      double t = (y - r->o.y) / r->d.y;
      if (t < t_min || t > t_max) return 0;
      double x = r->o.x + t * r->d.x;
      double z = r->o.z + t * r->d.z;
      if (x < x0 || x > x1 || z < z0 || z > z1) return 0;
      out->outward_normal = V(0, 0, 1); /* sic: the reference reports +z for every rect (:76) */
      out->t = t;
      out->uv.x = (x - x0) / (x1 - x0);
      out->uv.y = (z - z0) / (z1 - z0);
      return 1;
    }
    case HT_PRIM_YZRECT: { /* Hittable.scala:83-92 */
      if (c) c->n_rect++;
      double y0 = d[0], y1 = d[1], z0 = d[2], z1 = d[3], x = d[4];
      double t = (x - r->o.x) / r->d.x;
      if (t < t_min || t > t_max) return 0;
      double y = r->o.y + t * r->d.y;
      double z = r->o.z + t * r->d.z;
      if (y < y0 || y > y1 || z < z0 || z > z1) return 0;
      out->outward_normal = V(0, 0, 1); /* sic (:90) */
      out->t = t;
      out->uv.x = (y - y0) / (y1 - y0);
      out->uv.y = (z - z0) / (z1 - z0);
      return 1;
    }
    case HT_PRIM_TRIANGLE: { /* Hittable.scala:97-113 */
      if (c) c->n_tri++;
      v3 A = V(d[0], d[1], d[2]), B = V(d[3], d[4], d[5]), C = V(d[6], d[7], d[8]);
      v3 edgeAB = vsub(B, A);
      v3 edgeAC = vsub(C, A);
      v3 normalVector = vcross(edgeAB, edgeAC);
      v3 ao = vsub(r->o, A);
      v3 dao = vcross(ao, r->d);
      double determinant = -vdot(r->d, normalVector);
      double t = vdot(ao, normalVector) / determinant;
      double u = vdot(edgeAC, dao) / determinant;
      double v = -vdot(edgeAB, dao) / determinant;
      double w = 1 - u - v;
      if (determinant > 1e-6 && t >= t_min && t <= t_max && u >= 0 && v >= 0 && w >= 0) {
        out->outward_normal = normalVector; /* not normalised (:111) */
        out->t = t;
        out->uv.x = u;
        out->uv.y = v;
        return 1;
      }
      return 0;
    }
  }
  return 0;
}

/* bounding_box of a primitive (Hittable.scala:51, 65, 79, 93, 114) */
static aabb prim_box(const or_scene* s, int prim) {
  const double* d = s->prim_data + (size_t)prim * HT_PRIM_STRIDE;
  aabb b;
  switch (s->prim_type[prim]) {
    case HT_PRIM_SPHERE:
      b.mn = V(d[0] - d[3], d[1] - d[3], d[2] - d[3]);
      b.mx = V(d[0] + d[3], d[1] + d[3], d[2] + d[3]);
      break;
    case HT_PRIM_XYRECT:
      b.mn = V(d[0], d[2], d[4] - 0.0001);
      b.mx = V(d[1], d[3], d[4] + 0.0001);
      break;
    case HT_PRIM_XZRECT:
      b.mn = V(d[0], d[4] - 0.0001, d[2]);
      b.mx = V(d[1], d[4] + 0.0001, d[3]);
      break;
    case HT_PRIM_YZRECT:
      b.mn = V(d[4] - 0.0001, d[0], d[2]);
      b.mx = V(d[4] + 0.0001, d[1], d[3]);
      break;
    default: { /* triangle: AABB(Seq(v1,v2,v3)) = reduce of point boxes (AABB.scala:36) */
      aabb p0 = {V(d[0], d[1], d[2]), V(d[0], d[1], d[2])};
      aabb p1 = {V(d[3], d[4], d[5]), V(d[3], d[4], d[5])};
      aabb p2 = {V(d[6], d[7], d[8]), V(d[6], d[7], d[8])};
      b = aabb_union(aabb_union(p0, p1), p2);
    }
  }
  return b;
}

/* ---------------- BVH (S/BVH.scala) ---------------- */
/* Hittable.hit dispatch + BVH.hit (BVH.scala:7-14) */
static int obj_hit(const or_scene* s, const obj* o, const ray_t* r, double t_min, double t_max,
                   hit_t* out, ctr_t* c) {
  if (o->kind != OBJ_BVH) return prim_hit(s, o->prim, r, t_min, t_max, out, c);
  if (c) c->n_box++;
  if (!aabb_hit(&o->box, r, t_min, t_max)) return 0;
  hit_t hl, hr;
  int has_l = obj_hit(s, o->left, r, t_min, t_max, &hl, c);
  int has_r = obj_hit(s, o->right, r, t_min, has_l ? hl.t : t_max, &hr, c);
  if (has_r) { *out = hr; return 1; }  /* hit_right.orElse(hit_left) */
  if (has_l) { *out = hl; return 1; }
  return 0;
}

static obj* pool_new(or_scene* s) { return &s->pool[s->pool_used++]; }

/* stable merge sort of object pointers by v_min(axis), as Seq.sortWith */
static void stable_sort(obj** a, obj** tmp, int n, int axis) {
  if (n < 2) return;
  int h = n / 2;
  stable_sort(a, tmp, h, axis);
  stable_sort(a + h, tmp, n - h, axis);
  int i = 0, j = h, k = 0;
  while (i < h && j < n) {
    /* take from the right run only when strictly less: keeps equal keys in order */
    if (vcomp(a[j]->box.mn, axis) < vcomp(a[i]->box.mn, axis)) tmp[k++] = a[j++];
    else tmp[k++] = a[i++];
  }
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof(obj*));
}

/* BVH.apply (BVH.scala:18-33) */
static obj* bvh_build(or_scene* s, obj** objs, obj** tmp, int n, jrandom* rnd, int depth) {
  int axis = jr_next_int(rnd, 3); /* Random.between(0, 3) */
  stable_sort(objs, tmp, n, axis);
  int g = (int)ceil((double)n / 2);
  obj* node = pool_new(s);
  node->kind = OBJ_BVH;
  node->prim = -1;
  s->n_inner++;
  if (depth + 1 > s->depth) s->depth = depth + 1;
  if (n <= g) { /* one group: case Seq(only) => BVH(only, only) */
    obj* only = (n > 1) ? bvh_build(s, objs, tmp, n, rnd, depth + 1) : objs[0];
    node->left = only;
    node->right = only;
  } else {
    int nl = g, nr = n - g;
    /* the iterator builds the left group before the right one */
    obj* l = (nl > 1) ? bvh_build(s, objs, tmp, nl, rnd, depth + 1) : objs[0];
    obj* r = (nr > 1) ? bvh_build(s, objs + g, tmp + g, nr, rnd, depth + 1) : objs[g];
    node->left = l;
    node->right = r;
  }
  node->box = aabb_union(node->left->box, node->right->box); /* BVH.scala:5 */
  return node;
}

#define COPY_ARR(dst, src, count, type)                               \
  do {                                                                \
    size_t _n = (size_t)(count);                                      \
    (dst) = (type*)malloc((_n ? _n : 1) * sizeof(type));              \
    if (!(dst)) goto oom;                                             \
    if (_n) memcpy((dst), (src), _n * sizeof(type));                  \
  } while (0)

int or_scene_create(const ht_scene_desc* d, uint64_t bvh_seed, or_scene** out) {
  if (!d || !out || d->n_prims <= 0) return HT_E_INVALID;
  or_scene* s = (or_scene*)calloc(1, sizeof(or_scene));
  if (!s) return HT_E_INVALID;
  s->n_prims = d->n_prims;
  COPY_ARR(s->prim_type, d->prim_type, d->n_prims, int32_t);
  COPY_ARR(s->prim_material, d->prim_material, d->n_prims, int32_t);
  COPY_ARR(s->prim_data, d->prim_data, (size_t)d->n_prims * HT_PRIM_STRIDE, double);
  s->n_materials = d->n_materials;
  COPY_ARR(s->mat_type, d->mat_type, d->n_materials, int32_t);
  COPY_ARR(s->mat_texture, d->mat_texture, d->n_materials, int32_t);
  COPY_ARR(s->mat_param, d->mat_param, d->n_materials, double);
  s->n_textures = d->n_textures;
  COPY_ARR(s->tex_type, d->tex_type, d->n_textures, int32_t);
  COPY_ARR(s->tex_data, d->tex_data, (size_t)d->n_textures * HT_TEX_STRIDE, double);
  COPY_ARR(s->tex_child, d->tex_child, (size_t)d->n_textures * 2, int32_t);
  s->n_images = d->n_images;
  COPY_ARR(s->image_width, d->image_width, d->n_images, int32_t);
  COPY_ARR(s->image_height, d->image_height, d->n_images, int32_t);
  COPY_ARR(s->image_offset, d->image_offset, d->n_images, int64_t);
  {
    int64_t ntex = 0;
    for (int k = 0; k < d->n_images; ++k) {
      int64_t e = d->image_offset[k] + (int64_t)d->image_width[k] * d->image_height[k];
      if (e > ntex) ntex = e;
    }
    COPY_ARR(s->image_texels, d->image_texels, ntex, uint32_t);
  }
  s->n_perlin = d->n_perlin;
  COPY_ARR(s->perlin_ranvec, d->perlin_ranvec, (size_t)d->n_perlin * 256 * 3, double);
  COPY_ARR(s->perlin_perm, d->perlin_perm, (size_t)d->n_perlin * 3 * 256, int32_t);

  /* world list: consecutive primitives of one group with more than one member
   * are an STL entry, i.e. a nested BVH built first (JsonFormats.scala:92,
   * BVH.scala:54); everything else is the primitive itself. */
  s->pool_cap = 4 * d->n_prims + 8;
  s->pool = (obj*)calloc((size_t)s->pool_cap, sizeof(obj));
  obj** list = (obj**)malloc((size_t)(d->n_prims + 1) * sizeof(obj*));
  obj** tmp = (obj**)malloc((size_t)(d->n_prims + 1) * sizeof(obj*));
  obj** sub = (obj**)malloc((size_t)(d->n_prims + 1) * sizeof(obj*));
  if (!s->pool || !list || !tmp || !sub) { free(list); free(tmp); free(sub); goto oom; }
  jrandom rnd;
  jr_init(&rnd, bvh_seed);
  int n_world = 0;
  for (int i = 0; i < d->n_prims;) {
    int j = i + 1;
    if (d->prim_group)
      while (j < d->n_prims && d->prim_group[j] == d->prim_group[i]) ++j;
    int cnt = j - i;
    for (int k = 0; k < cnt; ++k) {
      obj* leaf = pool_new(s);
      leaf->kind = s->prim_type[i + k];
      leaf->prim = i + k;
      leaf->box = prim_box(s, i + k);
      leaf->left = leaf->right = NULL;
      sub[k] = leaf;
    }
    if (cnt > 1) list[n_world++] = bvh_build(s, sub, tmp, cnt, &rnd, 0);
    else list[n_world++] = sub[0];
    i = j;
  }
  int inner_depth = s->depth;
  s->depth = 0;
  s->world = bvh_build(s, list, tmp, n_world, &rnd, 0); /* main.scala:60 */
  s->depth += inner_depth;
  free(list); free(tmp); free(sub);
  *out = s;
  return HT_OK;
oom:
  or_scene_destroy(s);
  return HT_E_INVALID;
}

void or_scene_destroy(or_scene* s) {
  if (!s) return;
  free(s->prim_type); free(s->prim_material); free(s->prim_data);
  free(s->mat_type); free(s->mat_texture); free(s->mat_param);
  free(s->tex_type); free(s->tex_data); free(s->tex_child);
  free(s->image_width); free(s->image_height); free(s->image_offset); free(s->image_texels);
  free(s->perlin_ranvec); free(s->perlin_perm);
  free(s->pool);
  free(s);
}

void or_bvh_stats(const or_scene* s, int32_t* n_inner, int32_t* depth) {
  if (n_inner) *n_inner = s->n_inner;
  if (depth) *depth = s->depth;
}

/* world.hit in either mode */
static int world_hit(const or_scene* s, int world_mode, const ray_t* r, double t_min,
                     double t_max, hit_t* out, ctr_t* c) {
  if (world_mode == OR_WORLD_REFERENCE_BVH) return obj_hit(s, s->world, r, t_min, t_max, out, c);
  /* linear scan: fold with shrinking t_max, later primitives win inclusive ties */
  int has = 0;
  hit_t h;
  for (int p = 0; p < s->n_prims; ++p) {
    if (prim_hit(s, p, r, t_min, has ? out->t : t_max, &h, c)) { *out = h; has = 1; }
  }
  return has;
}

/* ---------------- textures (S/Texture.scala) ---------------- */
/* Perlin.noise (Texture.scala:44-76) */
static double perlin_noise(const or_scene* s, int k, v3 p) {
  const double* ranvec = s->perlin_ranvec + (size_t)k * 256 * 3;
  const int32_t* perm_x = s->perlin_perm + (size_t)k * 3 * 256;
  const int32_t* perm_y = perm_x + 256;
  const int32_t* perm_z = perm_y + 256;
  double u = p.x - floor(p.x);
  double v = p.y - floor(p.y);
  double w = p.z - floor(p.z);
  int32_t i = d2i(floor(p.x));
  int32_t j = d2i(floor(p.y));
  int32_t kk = d2i(floor(p.z));
  double uu = u * u * (3 - 2 * u);
  double vv = v * v * (3 - 2 * v);
  double ww = w * w * (3 - 2 * w);
  double sum = 0;
  for (int di = 0; di < 2; ++di)
    for (int dj = 0; dj < 2; ++dj)
      for (int dk = 0; dk < 2; ++dk) {
        int idx = perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(kk + dk) & 255];
        v3 c = V(ranvec[idx * 3], ranvec[idx * 3 + 1], ranvec[idx * 3 + 2]);
        v3 weight_v = V(u - di, v - dj, w - dk);
        sum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) *
               (dk * ww + (1 - dk) * (1 - ww)) * vdot(c, weight_v);
      }
  return sum;
}
/* Perlin.turb (Texture.scala:78-88), depth 7 */
static double perlin_turb(const or_scene* s, int k, v3 p) {
  double accum = 0, weight = 1;
  v3 temp_p = p;
  for (int i = 0; i < 7; ++i) {
    accum = accum + weight * perlin_noise(s, k, temp_p);
    temp_p = smul(2, temp_p);
    weight = weight * 0.5;
  }
  return fabs(accum);
}

static v3 texture_eval(const or_scene* s, int tex, v2 uv, v3 p) {
  for (;;) {
    const double* d = s->tex_data + (size_t)tex * HT_TEX_STRIDE;
    switch (s->tex_type[tex]) {
      case HT_TEX_SOLID: /* Texture.scala:13 */
        return V(d[0], d[1], d[2]);
      case HT_TEX_CHECKER: { /* Texture.scala:21-25 */
        double size = d[0];
        double sines = sin(size * p.x) * sin(size * p.y) * sin(size * p.z);
        tex = (sines < 0) ? s->tex_child[tex * 2] : s->tex_child[tex * 2 + 1];
        continue;
      }
      case HT_TEX_PERLIN: { /* Texture.scala:90 */
        double scale = d[0];
        double g = 0.5 * (1 + sin(scale * p.z + 10 * perlin_turb(s, s->tex_child[tex * 2], p)));
        return V(g, g, g);
      }
      default: { /* Image, Texture.scala:98-107; texel = Vec3.fromRGB (Vec3.scala:36-40) */
        int k = s->tex_child[tex * 2];
        int32_t width = s->image_width[k], height = s->image_height[k];
        double u = clampd(uv.x, 0.0, 1.0);
        double v = 1.0 - clampd(uv.y, 0.0, 1.0);
        int32_t i = clampi(d2i(u * width), 0, width - 1);
        int32_t j = clampi(d2i(v * height), 0, height - 1);
        uint32_t rgb = s->image_texels[s->image_offset[k] + (int64_t)j * width + i];
        return V(((rgb >> 16) & 0xFF) / 255.0, ((rgb >> 8) & 0xFF) / 255.0, (rgb & 0xFF) / 255.0);
      }
    }
  }
}

/* ---------------- materials (S/Material.scala) ---------------- */
/* returns 1 if scattered */
static int material_scatter(const or_scene* s, const hit_t* hit, path_rng* g, ray_t* scattered,
                            v3* attenuation, ctr_t* c) {
  int m = hit->material;
  double param = s->mat_param[m];
  int tex = s->mat_texture[m];
  const ray_t* ray = &hit->ray;
  if (c) c->n_shade[s->mat_type[m]]++;
  switch (s->mat_type[m]) {
    case HT_MAT_DIFFUSE: { /* Material.scala:10-12 */
      scattered->o = hit_position(hit);
      scattered->d = vadd(hit_normal(hit), sample_unit_vector(g));
      *attenuation = texture_eval(s, tex, hit->uv, hit_position(hit));
      return 1;
    }
    case HT_MAT_GLOSSY: { /* Material.scala:16-20 */
      scattered->o = hit_position(hit);
      scattered->d = vadd(vreflect(vunit(ray->d), hit_normal(hit)),
                          smul(param, sample_in_unit_sphere(g)));
      if (vdot(scattered->d, hit_normal(hit)) > 0) {
        *attenuation = texture_eval(s, tex, hit->uv, hit_position(hit));
        return 1;
      }
      return 0;
    }
    case HT_MAT_GLASS: { /* Material.scala:24-49 */
      double ior = param;
      double etai_over_etat = hit_front_face(hit) ? 1.0 / ior : ior;
      double cos_theta = jmin(-vdot(vunit(ray->d), hit_normal(hit)), 1.0);
      double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
      int reflect_it = etai_over_etat * sin_theta > 1;
      if (!reflect_it) {
        /* schlick (:24-28) */
        double r0 = (1 - etai_over_etat) / (1 + etai_over_etat);
        double r1 = r0 * r0;
        double sch = r1 + (1 - r1) * pow(1 - cos_theta, 5);
        double xi = (g->mode == OR_SAMPLER_REFERENCE) ? rng_next_double(g) : g->u[0];
        reflect_it = sch > xi;
      }
      scattered->o = hit_position(hit);
      if (reflect_it) {
        scattered->d = vreflect(vunit(ray->d), hit_normal(hit));
      } else { /* refract (:30-35) */
        v3 uvv = vunit(ray->d);
        v3 normal = hit_normal(hit);
        double ct = -vdot(uvv, normal);
        v3 r_out_perp = smul(etai_over_etat, vadd(uvv, smul(ct, normal)));
        v3 r_out_parallel = smul(-sqrt(fabs(1.0 - vlen2(r_out_perp))), normal);
        scattered->d = vadd(r_out_perp, r_out_parallel);
      }
      *attenuation = texture_eval(s, tex, hit->uv, hit_position(hit));
      return 1;
    }
    default: /* Emission never scatters (Material.scala:53) */
      return 0;
  }
}
/* Material.emit (Material.scala:5, 54) */
static v3 material_emit(const or_scene* s, const hit_t* hit) {
  int m = hit->material;
  if (s->mat_type[m] != HT_MAT_EMISSION) return V(0, 0, 0);
  return smul(s->mat_param[m], texture_eval(s, s->mat_texture[m], hit->uv, hit_position(hit)));
}

/* ---------------- camera (S/Camera.scala) ---------------- */
typedef struct {
  v3 origin, lower_left_corner, horizontal, vertical, u, v, w, background;
  double lens_radius;
} camera_t;

static void camera_init(camera_t* c, const ht_camera* in) { /* Camera.scala:5-17 */
  v3 origin = V(in->origin[0], in->origin[1], in->origin[2]);
  v3 target = V(in->target[0], in->target[1], in->target[2]);
  v3 vup = V(in->vup[0], in->vup[1], in->vup[2]);
  /* Math.toRadians = angdeg * (PI / 180) */
  double viewport_height = 2.0 * tan(in->vfov * 0.017453292519943295 / 2);
  double viewport_width = in->aspect_ratio * viewport_height;
  c->w = vunit(vsub(origin, target));
  c->u = vunit(vcross(vup, c->w));
  c->v = vcross(c->w, c->u);
  double focus = in->focus_distance < 0 ? vlen(vsub(target, origin)) : in->focus_distance;
  c->horizontal = smul(focus * viewport_width, c->u);
  c->vertical = smul(focus * viewport_height, c->v);
  c->lower_left_corner = vsub(vsub(vsub(origin, sdiv(c->horizontal, 2)), sdiv(c->vertical, 2)),
                              smul(focus, c->w));
  c->lens_radius = in->aperture / 2;
  c->origin = origin;
  c->background = V(in->background[0], in->background[1], in->background[2]);
}

void or_camera_basis(const ht_camera* cam, double out[22]) {
  camera_t c;
  camera_init(&c, cam);
  const v3* vs[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
  for (int i = 0; i < 7; ++i) {
    out[i * 3] = vs[i]->x; out[i * 3 + 1] = vs[i]->y; out[i * 3 + 2] = vs[i]->z;
  }
  out[21] = c.lens_radius;
}

/* Camera.get_ray (Camera.scala:19-23) with the disk sample supplied */
static ray_t camera_ray(const camera_t* c, double s, double t, v3 disk) {
  v3 rd = smul(c->lens_radius, disk);
  v3 offset = vadd(smul(rd.x, c->u), smul(rd.y, c->v));
  ray_t r;
  r.o = vadd(c->origin, offset);
  r.d = vsub(vsub(vadd(vadd(c->lower_left_corner, smul(s, c->horizontal)), smul(t, c->vertical)),
                  c->origin),
             offset);
  return r;
}

/* Camera.ray_color (Camera.scala:26-36), recursive as in the reference.
 * `bounce` numbers the scatter events of the path for the MATCHED sampler. */
static v3 ray_color(const or_scene* s, const camera_t* cam, int world_mode, const ray_t* ray,
                    int depth, uint32_t bounce, path_rng* g, ctr_t* c) {
  if (depth <= 0) return V(0, 0, 0);
  hit_t hit;
  if (c) c->n_rays++;
  if (world_hit(s, world_mode, ray, 0.001, INFINITY, &hit, c)) {
    ray_t scattered;
    v3 attenuation;
    if (g->mode == OR_SAMPLER_MATCHED) rng_event(g, bounce);
    if (material_scatter(s, &hit, g, &scattered, &attenuation, c)) {
      v3 emit = material_emit(s, &hit);
      v3 rest = ray_color(s, cam, world_mode, &scattered, depth - 1, bounce + 1, g, c);
      return vadd(emit, vmul(attenuation, rest));
    }
    return material_emit(s, &hit);
  }
  return cam->background;
}

/* one sample of one pixel: main.scala:88-90 */
static v3 pixel_sample(const or_scene* s, const camera_t* cam, int world_mode, int W, int H,
                       int i, int j, int sample, int max_depth, int jitter, int sampler,
                       uint64_t seed, ctr_t* c) {
  path_rng g;
  rng_begin(&g, sampler, seed, (uint32_t)(j * W + i), (uint32_t)sample);
  double xi1 = 0.5, xi2 = 0.5;
  v3 disk = V(0, 0, 0);
  if (jitter) {
    if (sampler == OR_SAMPLER_REFERENCE) {
      xi1 = rng_next_double(&g);
      xi2 = rng_next_double(&g);
      disk = sample_in_unit_disk(&g); /* drawn even when the aperture is 0 (Camera.scala:20) */
    } else {
      rng_event(&g, 0);
      xi1 = g.u[0];
      xi2 = g.u[1];
      disk = sample_in_unit_disk(&g);
    }
  }
  double u = (i + xi1) / (W - 1);
  double v = (j + xi2) / (H - 1);
  ray_t r = camera_ray(cam, u, v, disk);
  return ray_color(s, cam, world_mode, &r, max_depth, 1, &g, c);
}

/* Vec3.toRGB (Vec3.scala:82-84) */
uint32_t or_to_rgb(const double rgb[3]) {
  uint32_t r = (uint32_t)d2i(255 * clampd(sqrt(rgb[0]), 0, 0.999));
  uint32_t g = (uint32_t)d2i(255 * clampd(sqrt(rgb[1]), 0, 0.999));
  uint32_t b = (uint32_t)d2i(255 * clampd(sqrt(rgb[2]), 0, 0.999));
  return r << 16 | g << 8 | b;
}

int or_render(const or_scene* s, const ht_camera* cam_in, int32_t W, int32_t H, int32_t samples,
              int32_t sample_offset, int32_t max_depth, int32_t jitter, int sampler,
              int world_mode, uint64_t seed, int nthreads, double* rgb_linear,
              uint32_t* rgb_packed, or_counters* counters) {
  if (!s || !cam_in || W < 2 || H < 2 || samples < 1) return HT_E_INVALID;
  if (max_depth <= 0) max_depth = 50;
  camera_t cam;
  camera_init(&cam, cam_in);
  ctr_t total;
  memset(&total, 0, sizeof total);
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_num_procs();
#else
  nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
  {
    ctr_t local;
    memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
    for (int j = 0; j < H; ++j) {        /* main.scala:79 */
      for (int i = 0; i < W; ++i) {      /* main.scala:81 */
        v3 sum = V(0, 0, 0);
        for (int k = 0; k < samples; ++k) { /* main.scala:87-91: map then reduce(_ + _) */
          v3 col = pixel_sample(s, &cam, world_mode, W, H, i, j, sample_offset + k, max_depth,
                                jitter, sampler, seed, counters ? &local : NULL);
          sum = (k == 0) ? col : vadd(sum, col);
        }
        v3 pixel = sdiv(sum, (double)samples);
        if (rgb_linear) {
          double* o = rgb_linear + ((size_t)j * W + i) * 3;
          o[0] = pixel.x; o[1] = pixel.y; o[2] = pixel.z;
        }
        if (rgb_packed) { /* img.setRGB(i, H - j - 1, pixel.toRGB)  main.scala:93 */
          double p[3] = {pixel.x, pixel.y, pixel.z};
          rgb_packed[(size_t)(H - j - 1) * W + i] = or_to_rgb(p);
        }
      }
    }
#pragma omp critical
    {
      total.n_rays += local.n_rays; total.n_box += local.n_box;
      total.n_sphere += local.n_sphere; total.n_rect += local.n_rect; total.n_tri += local.n_tri;
      for (int k = 0; k < 4; ++k) total.n_shade[k] += local.n_shade[k];
    }
  }
  if (counters) {
    counters->n_rays = total.n_rays; counters->n_box = total.n_box;
    counters->n_sphere = total.n_sphere; counters->n_rect = total.n_rect;
    counters->n_tri = total.n_tri;
    for (int k = 0; k < 4; ++k) counters->n_shade[k] = total.n_shade[k];
  }
  return HT_OK;
}

int or_path_radiance(const or_scene* s, const ht_camera* cam_in, int32_t W, int32_t H, int32_t i,
                     int32_t j, int32_t sample, int32_t max_depth, int32_t jitter, int sampler,
                     int world_mode, uint64_t seed, double rgb[3]) {
  if (!s || !cam_in || W < 2 || H < 2) return HT_E_INVALID;
  if (max_depth <= 0) max_depth = 50;
  camera_t cam;
  camera_init(&cam, cam_in);
  v3 c = pixel_sample(s, &cam, world_mode, W, H, i, j, sample, max_depth, jitter, sampler, seed,
                      NULL);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
  return HT_OK;
}

static int first_hit_impl(const or_scene* s, const ht_camera* cam_in, int32_t W, int32_t H, int world_mode,
                          double off_x, double off_y, int round_ray_fp32, double* t, int32_t* prim_id,
                          double* normal, double* uv) {
  if (!s || !cam_in || W < 2 || H < 2) return HT_E_INVALID;
  camera_t cam;
  camera_init(&cam, cam_in);
  cam.lens_radius = 0;
#pragma omp parallel for schedule(dynamic, 4)
  for (int j = 0; j < H; ++j)
    for (int i = 0; i < W; ++i) {
      double u = (i + off_x) / (W - 1);
      double v = (j + off_y) / (H - 1);
      ray_t r = camera_ray(&cam, u, v, V(0, 0, 0));
      if (round_ray_fp32) {  /* the ray an fp32 device can at best hold: each component to the nearest float */
        r.o = V((float)r.o.x, (float)r.o.y, (float)r.o.z);
        r.d = V((float)r.d.x, (float)r.d.y, (float)r.d.z);
      }
      hit_t hit;
      size_t p = (size_t)j * W + i;
      if (world_hit(s, world_mode, &r, 0.001, INFINITY, &hit, NULL)) {
        v3 n = hit_normal(&hit);
        if (t) t[p] = hit.t;
        if (prim_id) prim_id[p] = hit.prim;
        if (normal) { normal[p * 3] = n.x; normal[p * 3 + 1] = n.y; normal[p * 3 + 2] = n.z; }
        if (uv) { uv[p * 2] = hit.uv.x; uv[p * 2 + 1] = hit.uv.y; }
      } else {
        if (t) t[p] = INFINITY;
        if (prim_id) prim_id[p] = -1;
        if (normal) { normal[p * 3] = 0; normal[p * 3 + 1] = 0; normal[p * 3 + 2] = 0; }
        if (uv) { uv[p * 2] = 0; uv[p * 2 + 1] = 0; }
      }
    }
  return HT_OK;
}

int or_first_hit(const or_scene* s, const ht_camera* cam_in, int32_t W, int32_t H, int world_mode,
                 double off_x, double off_y, double* t, int32_t* prim_id, double* normal,
                 double* uv) {
  return first_hit_impl(s, cam_in, W, H, world_mode, off_x, off_y, 0, t, prim_id, normal, uv);
}

/* The same query for camera rays whose origin and direction were rounded to fp32 first and
 * then traced in fp64: what separates "the device computes badly" from "a ray held in fp32
 * is not the reference's ray" when the fp32 device path is compared with this restatement. */
int or_first_hit_fp32_rays(const or_scene* s, const ht_camera* cam_in, int32_t W, int32_t H, int world_mode,
                           double off_x, double off_y, double* t, int32_t* prim_id, double* normal,
                           double* uv) {
  return first_hit_impl(s, cam_in, W, H, world_mode, off_x, off_y, 1, t, prim_id, normal, uv);
}

/* ---------------- probes for known-answer tests ---------------- */
int or_prim_hit(const or_scene* s, int32_t prim, const double origin[3], const double dir[3],
                double t_min, double t_max, double* t, double normal_out[3], double uv[2],
                int* front_face) {
  ray_t r;
  r.o = V(origin[0], origin[1], origin[2]);
  r.d = V(dir[0], dir[1], dir[2]);
  hit_t h;
  if (!prim_hit(s, prim, &r, t_min, t_max, &h, NULL)) return 0;
  v3 n = hit_normal(&h);
  if (t) *t = h.t;
  if (normal_out) { normal_out[0] = n.x; normal_out[1] = n.y; normal_out[2] = n.z; }
  if (uv) { uv[0] = h.uv.x; uv[1] = h.uv.y; }
  if (front_face) *front_face = hit_front_face(&h);
  return 1;
}

void or_texture(const or_scene* s, int32_t tex, const double uv[2], const double p[3],
                double rgb[3]) {
  v2 u = {uv[0], uv[1]};
  v3 c = texture_eval(s, tex, u, V(p[0], p[1], p[2]));
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

double or_perlin_noise(const or_scene* s, int32_t k, const double p[3]) {
  return perlin_noise(s, k, V(p[0], p[1], p[2]));
}
double or_perlin_turb(const or_scene* s, int32_t k, const double p[3]) {
  return perlin_turb(s, k, V(p[0], p[1], p[2]));
}

int or_scatter_matched(const or_scene* s, int32_t material, const double ray_o[3],
                       const double ray_d[3], double t, const double outward_normal[3],
                       const double uv[2], const double u[4], double dir_out[3],
                       double atten[3], double emit[3]) {
  hit_t h;
  h.ray.o = V(ray_o[0], ray_o[1], ray_o[2]);
  h.ray.d = V(ray_d[0], ray_d[1], ray_d[2]);
  h.outward_normal = V(outward_normal[0], outward_normal[1], outward_normal[2]);
  h.t = t;
  h.uv.x = uv[0]; h.uv.y = uv[1];
  h.material = material;
  h.prim = -1;
  path_rng g;
  rng_begin(&g, OR_SAMPLER_MATCHED, 0, 0, 0);
  for (int i = 0; i < 4; ++i) g.u[i] = u[i];
  ray_t sc;
  v3 a = V(0, 0, 0);
  sc.d = V(0, 0, 0);
  int did = material_scatter(s, &h, &g, &sc, &a, NULL);
  v3 e = material_emit(s, &h);
  dir_out[0] = sc.d.x; dir_out[1] = sc.d.y; dir_out[2] = sc.d.z;
  atten[0] = a.x; atten[1] = a.y; atten[2] = a.z;
  emit[0] = e.x; emit[1] = e.y; emit[2] = e.z;
  return did;
}
