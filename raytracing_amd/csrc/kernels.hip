// kernels.hip — gfx950 (MI355X, wave64) kernels of libhiptrace.
//
// One persistent "megakernel" does everything between a pixel index and its
// radiance sum: camera ray generation, BVH traversal, primitive intersection,
// material scatter and texture evaluation, for `sample_count` paths per pixel.
//
// Execution model
//  * a workgroup copies the scene image (scene_layout.h) into LDS once; every
//    scene of the reference fits (<= ~60 KB), so traversal never touches HBM.
//    Image texels are the only global reads.  Scenes beyond the LDS store (large
//    meshes) run the WIDE variants of the same kernels, which read the image where
//    it lies in global memory and keep only the stacks and tile sums in LDS.
//  * work is a global queue of units, unit = (8x8-pixel tile, chunk of samples);
//    a wave streams units: dead lanes claim the next (pixel, sample) items of the
//    wave's current unit, and the next unit is opened while stragglers of the old
//    one are still in flight.
//  * lanes are small state machines (TRAV / WAIT / DEAD / DONE).  The wave runs
//    traversal steps while enough lanes are traversing; when `wait_threshold`
//    lanes have finished their ray, the wave shades them, starts scattered rays
//    and new camera paths for the lanes that ended, and goes back to traversal
//    with the stragglers still in mid-flight, so no lane idles for the length of
//    the longest path in the wave.
//  * the per-lane traversal stack lives in LDS, [level][thread], 16-bit node
//    references (32-bit in the WIDE variants).
//  * radiance sums are signed 32.32 fixed point: per-wave tile accumulators in LDS
//    (64-bit LDS atomics), flushed to the frame with 64-bit global atomics, so the
//    frame does not depend on the order in which paths finish.
//  * random numbers: Philox4x32-10, counter = (pixel, sample, bounce, 0),
//    key = seed; one block per event.
//
// Reference semantics reproduced here (Ivoah/RayTracing, S = shared/src/main/scala):
//  S/Camera.scala:19-36, S/Hittable.scala:4-115, S/Material.scala:4-55,
//  S/Texture.scala:12-107, S/Vec3.scala:17-34,82-84 and the loop at
//  jvm/src/main/scala/main.scala:79-94.  Directions are never normalised, rect
//  normals are always +z and triangle normals are the raw cross product, as in
//  the reference.
#include <hip/hip_runtime.h>

#include <stdio.h>

#include <mutex>

#include "../../include/hiptrace.h"
#include "kernels.h"

namespace ht {

// ---------------------------------------------------------------------------
// arithmetic traits
// ---------------------------------------------------------------------------
template <typename R> struct M;

template <> struct M<float> {
  typedef float4 R4;
  static constexpr int BT = 1024;  // threads per workgroup
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float fmin(float a, float b) { return __builtin_fminf(a, b); }
  static __device__ __forceinline__ float fmax(float a, float b) { return __builtin_fmaxf(a, b); }
  static __device__ __forceinline__ float fabs(float a) { return __builtin_fabsf(a); }
  static __device__ __forceinline__ float floor(float a) { return __builtin_floorf(a); }
  static __device__ __forceinline__ float copysign(float a, float b) { return __builtin_copysignf(a, b); }
  // sin / cos of 2*pi*u, u in [0,1): v_sin_f32 / v_cos_f32 take revolutions
  static __device__ __forceinline__ void sincos2pi(float u, float& s, float& c) {
    s = __builtin_amdgcn_sinf(u);
    c = __builtin_amdgcn_cosf(u);
  }
  static __device__ __forceinline__ float sin(float x) {
    float r = x * 0.15915494309189535f;
    return __builtin_amdgcn_sinf(r - __builtin_floorf(r));
  }
  static __device__ __forceinline__ float cbrt(float x) { return ::cbrtf(x); }
  static __device__ __forceinline__ float atan2(float y, float x) { return ::atan2f(y, x); }
  static __device__ __forceinline__ float asin(float x) { return ::asinf(x); }
  static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
  static __device__ __forceinline__ uint32_t bits_lo(float x) { return __float_as_uint(x); }
  static __device__ __forceinline__ float to_float_down(float x) { return x; }
  static __device__ __forceinline__ float slab_widen() { return 1.0000005f; }
  static __device__ __forceinline__ float slab_abs() { return 2.3841858e-07f; }  // 2^-22
  static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
};

template <> struct M<double> {
  typedef double4 R4;
  static constexpr int BT = 512;
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
  static __device__ __forceinline__ double fmin(double a, double b) { return ::fmin(a, b); }
  static __device__ __forceinline__ double fmax(double a, double b) { return ::fmax(a, b); }
  static __device__ __forceinline__ double fabs(double a) { return ::fabs(a); }
  static __device__ __forceinline__ double floor(double a) { return ::floor(a); }
  static __device__ __forceinline__ double copysign(double a, double b) { return ::copysign(a, b); }
  static __device__ __forceinline__ void sincos2pi(double u, double& s, double& c) {
    ::sincos(6.283185307179586476925 * u, &s, &c);
  }
  static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
  static __device__ __forceinline__ double cbrt(double x) { return ::cbrt(x); }
  static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
  static __device__ __forceinline__ double asin(double x) { return ::asin(x); }
  static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
  static __device__ __forceinline__ uint32_t bits_lo(double x) {
    return (uint32_t)__double_as_longlong(x);
  }
  static __device__ __forceinline__ float to_float_down(double x) { return __double2float_rd(x); }
  static __device__ __forceinline__ double slab_widen() { return 1.0000000000001; }
  static __device__ __forceinline__ double slab_abs() { return 4.440892098500626e-16; }  // 2^-51
  static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
};

template <typename R> struct V3 { R x, y, z; };
template <typename R> __device__ __forceinline__ V3<R> mk(R x, R y, R z) { V3<R> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename R> __device__ __forceinline__ V3<R> operator+(V3<R> a, V3<R> b) { return mk<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename R> __device__ __forceinline__ V3<R> operator-(V3<R> a, V3<R> b) { return mk<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename R> __device__ __forceinline__ V3<R> operator*(V3<R> a, V3<R> b) { return mk<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename R> __device__ __forceinline__ V3<R> operator*(R s, V3<R> a) { return mk<R>(s * a.x, s * a.y, s * a.z); }
template <typename R> __device__ __forceinline__ V3<R> operator-(V3<R> a) { return mk<R>(-a.x, -a.y, -a.z); }
template <typename R> __device__ __forceinline__ R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename R> __device__ __forceinline__ V3<R> cross(V3<R> a, V3<R> b) {
  return mk<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename R, typename P> __device__ __forceinline__ V3<R> ld3(P p) { return mk<R>(p[0], p[1], p[2]); }
// a fused multiply-add that is written out (where the compiler's own contraction is switched off)
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
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
// The same with the ten round keys (k0 + r*0x9E3779B9, k1 + r*0xBB67AE85) read from a table
// the host prepared: kept as live values they would hold twenty scalar registers for the
// whole kernel.
template <typename KeyPtr>
__device__ __forceinline__ void philox4x32_10_keyed(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    KeyPtr keys, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ keys[2 * r];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ keys[2 * r + 1];
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// 24-bit uniform in [0,1): exactly representable in fp32 and fp64
template <typename R> __device__ __forceinline__ R u01(uint32_t x) {
  return (R)(x >> 8) * (R)(1.0 / 16777216.0);
}

// ---------------------------------------------------------------------------
// scene image accessors (LDS)
// ---------------------------------------------------------------------------
struct Scene {
  const uint8_t* base;  // LDS (on-chip store) or global memory (HBM store)
  SceneLayout L;
};

// child references of a node: packed in one word (on-chip store) or two words (HBM store)
template <bool WIDE>
__device__ __forceinline__ void load_children(const uint8_t* p, uint32_t& c0, uint32_t& c1) {
  if constexpr (WIDE) {
    const uint2 cc = *(const uint2*)p;
    c0 = cc.x;
    c1 = cc.y;
  } else {
    const uint32_t cc = *(const uint32_t*)p;
    c0 = cc & 0xFFFFu;
    c1 = cc >> 16;
  }
}
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void unpack_half2(uint32_t w, float& lo, float& hi) {
  const half2_t h = __builtin_bit_cast(half2_t, w);
  lo = (float)h.x;
  hi = (float)h.y;
}
// fp32 node.  On chip: 12 floats + packed child references, 64 bytes.  HBM store: 12
// half floats (the half-extents rounded up by the host) + two child references, 32 bytes.
template <bool WIDE>
__device__ __forceinline__ void load_node(const Scene& sc, uint32_t idx, float b[12], uint32_t& c0,
                                          uint32_t& c1) {
  if constexpr (WIDE) {
    const uint8_t* p = sc.base + sc.L.nodes + (size_t)idx * 32u;
    const uint4 q0 = *(const uint4*)(p);
    const uint4 q1 = *(const uint4*)(p + 16);
    unpack_half2(q0.x, b[0], b[1]);
    unpack_half2(q0.y, b[2], b[3]);
    unpack_half2(q0.z, b[4], b[5]);
    unpack_half2(q0.w, b[6], b[7]);
    unpack_half2(q1.x, b[8], b[9]);
    unpack_half2(q1.y, b[10], b[11]);
    c0 = q1.z;
    c1 = q1.w;
  } else {
    const uint8_t* p = sc.base + sc.L.nodes + idx * 64u;
    const float4 a0 = *(const float4*)(p);
    const float4 a1 = *(const float4*)(p + 16);
    const float4 a2 = *(const float4*)(p + 32);
    load_children<false>(p + 48, c0, c1);
    b[0] = a0.x; b[1] = a0.y; b[2] = a0.z; b[3] = a0.w;
    b[4] = a1.x; b[5] = a1.y; b[6] = a1.z; b[7] = a1.w;
    b[8] = a2.x; b[9] = a2.y; b[10] = a2.z; b[11] = a2.w;
  }
}
template <bool WIDE>
__device__ __forceinline__ void load_node(const Scene& sc, uint32_t idx, double b[12], uint32_t& c0,
                                          uint32_t& c1) {
  const uint8_t* p = sc.base + sc.L.nodes + (size_t)idx * 112u;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double2 v = *(const double2*)(p + 16 * i);
    b[2 * i] = v.x;
    b[2 * i + 1] = v.y;
  }
  load_children<WIDE>(p + 96, c0, c1);
}

template <typename R>
__device__ __forceinline__ typename M<R>::R4 load_r4(const Scene& sc, uint32_t off, uint32_t idx) {
  return *(const typename M<R>::R4*)(sc.base + off + idx * (uint32_t)(4 * sizeof(R)));
}
__device__ __forceinline__ uint32_t load_u32(const Scene& sc, uint32_t off, uint32_t idx) {
  return *(const uint32_t*)(sc.base + off + idx * 4u);
}

// ---------------------------------------------------------------------------
// ray + traversal state
// ---------------------------------------------------------------------------
template <typename R>
struct Ray {
  V3<R> o, d, inv;  // inv = 1/d per component
  V3<R> oi;         // o * inv per component (fp32 box tests: (centre - o)/d as one fused multiply-add)
  R a, inv_a, oo;   // |d|^2, 1/|d|^2, |o|^2
};

// What only the sphere tests and the shading of a sphere read.  A kernel whose hierarchy holds
// triangles alone works these out again before it shades, so that they hold no registers
// during the traversal.
template <typename R>
__device__ __forceinline__ void ray_sphere_terms(Ray<R>& r) {
  r.a = dot(r.d, r.d);
  r.inv_a = M<R>::rcp(r.a);
  r.oo = dot(r.o, r.o);
}
template <typename R, bool FLAT = false>  // FLAT: no hierarchy, so no box tests
__device__ __forceinline__ void ray_setup(Ray<R>& r) {
  r.inv = mk<R>(M<R>::rcp(r.d.x), M<R>::rcp(r.d.y), M<R>::rcp(r.d.z));
  if constexpr (sizeof(R) == 4 && !FLAT) {
    // |1/d| is held to 1e30 (a direction component below 1e-30, zero included, counts as 1e-30:
    // over any distance in a scene the ray does not move along that axis either way), so that
    // o/d and centre/d stay finite and their difference means something: an infinite o/d beside a
    // finite half-extent/|d| would put every box at minus infinity and end the ray without a hit
    r.inv = mk<R>(__builtin_amdgcn_fmed3f(r.inv.x, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(r.inv.y, -1e30f, 1e30f),
                  __builtin_amdgcn_fmed3f(r.inv.z, -1e30f, 1e30f));
  }
  r.oi = mk<R>(r.o.x * r.inv.x, r.o.y * r.inv.y, r.o.z * r.inv.z);
  ray_sphere_terms<R>(r);
}

template <typename R>
struct Trav {
  uint32_t cur;       // reference being processed; REF_EMPTY when the ray is finished
  uint32_t sp;        // LDS byte address of the lane's next free stack slot (layout [level][thread])
  uint32_t best_ref;  // REF_EMPTY = no hit yet
  R best_t;
};

// The traversal stacks in LDS: one row per level, [level][thread], so that the lanes of a wave
// never meet in a bank.  On-chip store: 16-bit entries.  HBM store: a reference is 24 bits
// (scene_layout.h) and an entry is those three bytes, a row being the threads' low
// halves followed by their third bytes: with 4-byte entries a hierarchy of 17 levels takes
// 100 bytes of LDS per lane (tile sums included) and six 256-thread workgroups fill a CU, at
// three bytes 81 and seven (the second LDS access costs 3 % at equal occupancy).
// `addr` is the LDS byte address of the lane's low half.
// Stack addresses are LDS byte addresses, not offsets from the workgroup's dynamic LDS: the base
// (a link-time constant, zero in these kernels) is added once when a lane's level-0 address is
// worked out, not by a vector add in front of every stack access.
typedef __attribute__((address_space(3))) uint8_t LdsU8;
typedef __attribute__((address_space(3))) uint16_t LdsU16;
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(const LdsU8*)p; }
template <bool WIDE>
__device__ __forceinline__ uint32_t stack_hi_offset() {  // from a lane's low half to its third byte
  return WIDE ? 2u * blockDim.x - threadIdx.x : 0u;
}
template <bool WIDE>
__device__ __forceinline__ uint32_t stack_load(const uint8_t*, uint32_t addr) {
  const uint32_t lo = *(const LdsU16*)(uintptr_t)addr;
  if constexpr (!WIDE) return lo;
  return lo | ((uint32_t)*(const LdsU8*)(uintptr_t)(addr + stack_hi_offset<WIDE>()) << 16);
}
template <bool WIDE>
__device__ __forceinline__ void stack_store(uint8_t*, uint32_t addr, uint32_t ref) {
  *(LdsU16*)(uintptr_t)addr = (uint16_t)ref;
  if constexpr (WIDE) *(LdsU8*)(uintptr_t)(addr + stack_hi_offset<WIDE>()) = (uint8_t)(ref >> 16);
}

template <typename R, bool WIDE>
__device__ __forceinline__ void trav_begin(Trav<R>& tv, uint32_t stack_level1) {
  tv.cur = 0;  // root is inner node 0
  tv.sp = stack_level1;  // level 0 of every lane's stack holds REF_EMPTY for good
  tv.best_ref = RefT<WIDE>::EMPTY;
  tv.best_t = M<R>::inf();
}

struct Cnt {
  uint32_t v[NUM_COUNTERS];
};

// Slab test of one box against [t_min, t_far]; returns hit flag and entry distance.
template <typename R>
__device__ __forceinline__ bool slab(const Ray<R>& r, R lox, R loy, R loz, R hix, R hiy, R hiz,
                                     R t_min, R t_far, R& tn) {
  R x0 = (lox - r.o.x) * r.inv.x, x1 = (hix - r.o.x) * r.inv.x;
  R y0 = (loy - r.o.y) * r.inv.y, y1 = (hiy - r.o.y) * r.inv.y;
  R z0 = (loz - r.o.z) * r.inv.z, z1 = (hiz - r.o.z) * r.inv.z;
  R nx = M<R>::fmin(x0, x1), fx = M<R>::fmax(x0, x1);
  R ny = M<R>::fmin(y0, y1), fy = M<R>::fmax(y0, y1);
  R nz = M<R>::fmin(z0, z1), fz = M<R>::fmax(z0, z1);
  tn = M<R>::fmax(M<R>::fmax(nx, ny), M<R>::fmax(nz, t_min));
  R tf = M<R>::fmin(M<R>::fmin(fx, fy), fz);
  // widen the exit by a few ulps so rounding never drops a box the exact ray touches
  tf = M<R>::fmin(tf * M<R>::slab_widen(), t_far);
  return tn <= tf;
}

// The same test for the two children of an inner node at once.  b = the node's bounds as
// the scene image stores them, per axis four numbers, one pair for each child:
//   fp64: (lo child 0, lo child 1, hi child 0, hi child 1);
//   fp32: (centre 0, centre 1, half-extent 0, half-extent 1), the half-extent rounded up
//         and widened by the host so that it also covers the rounding of this arithmetic.
// With c = (centre - o)/d = centre/d - o/d (one fused multiply-add, o/d kept with the ray; its
// rounding is of the size of an ulp of the ray's own origin) and e = half/|d| the slab is
// [c - e, c + e]: four operations per axis and child and no min/max to order the two planes (on gfx950 an
// fp32 add or multiply issues in about half the cycles of a min, max, compare or select).
template <typename R>
__device__ __forceinline__ void slab2(const Ray<R>& r, const R b[12], R t_min, R t_far, R& tn0,
                                      R& tn1, bool& h0, bool& h1) {
  if constexpr (sizeof(R) == 4) {
    const float aix = __builtin_fabsf(r.inv.x), aiy = __builtin_fabsf(r.inv.y), aiz = __builtin_fabsf(r.inv.z);
    const float cx0 = __builtin_fmaf(b[0], r.inv.x, -r.oi.x), cx1 = __builtin_fmaf(b[1], r.inv.x, -r.oi.x), ex0 = b[2] * aix, ex1 = b[3] * aix;
    const float cy0 = __builtin_fmaf(b[4], r.inv.y, -r.oi.y), cy1 = __builtin_fmaf(b[5], r.inv.y, -r.oi.y), ey0 = b[6] * aiy, ey1 = b[7] * aiy;
    const float cz0 = __builtin_fmaf(b[8], r.inv.z, -r.oi.z), cz1 = __builtin_fmaf(b[9], r.inv.z, -r.oi.z), ez0 = b[10] * aiz, ez1 = b[11] * aiz;
    // The entry distances are ordered as the signed integers of their bit patterns, which puts
    // every negative number below every positive one and the negative ones in reverse.  An entry
    // that is not ahead of the ray then needs no clamp to t_min: a box around the origin has a
    // negative entry and a positive exit and is taken, and a box behind the ray has a negative
    // exit on some axis k, where entry_k <= exit_k < 0 makes bits(entry_k) >= bits(exit_k), so the
    // largest entry pattern is not below the exit's and the box is left out (but for a box of no
    // thickness, which costs a visit and no hit: primitives test t >= t_min themselves).
    const int n0 = max(max(__float_as_int(cx0 - ex0), __float_as_int(cy0 - ey0)), __float_as_int(cz0 - ez0));
    const int n1 = max(max(__float_as_int(cx1 - ex1), __float_as_int(cy1 - ey1)), __float_as_int(cz1 - ez1));
    const float tf0 = __builtin_fminf(__builtin_fminf(cx0 + ex0, cy0 + ey0), cz0 + ez0);
    const float tf1 = __builtin_fminf(__builtin_fminf(cx1 + ex1, cy1 + ey1), cz1 + ez1);
    const float w = M<float>::slab_widen();
    h0 = n0 <= __float_as_int(__builtin_fminf(tf0 * w, t_far));
    h1 = n1 <= __float_as_int(__builtin_fminf(tf1 * w, t_far));
    tn0 = __int_as_float(n0);  // the caller orders the two children by these: as numbers where both
    tn1 = __int_as_float(n1);  // are ahead, in no particular order where the ray starts inside both
  } else {
    h0 = slab<R>(r, b[0], b[4], b[8], b[2], b[6], b[10], t_min, t_far, tn0);
    h1 = slab<R>(r, b[1], b[5], b[9], b[3], b[7], b[11], t_min, t_far, tn1);
  }
}

// The sphere quadratic in a cancellation-free form.
// Same acceptance rule as Sphere.hit (Hittable.scala:40-47): discriminant > 0,
// the near root if it lies in (t_min, t_max), else the far root.
template <typename R>
struct SphereSolve {
  V3<R> cen, l;  // centre; offset of the ray's closest-approach point from the centre
  R rad, t1, t2; // radius; near and far root
  R s;           // sqrt(disc)/|d|^2: the roots are t_closest -/+ s
  bool big;
};
template <typename R>
__device__ __forceinline__ bool sphere_solve(const Scene& sc, const Ray<R>& ray, uint32_t kind,
                                             uint32_t idx, SphereSolve<R>& o) {
  typename M<R>::R4 s = load_r4<R>(sc, sc.L.sph, idx);
  o.cen = mk<R>(s.x, s.y, s.z);
  o.rad = s.w;
  o.big = kind == KIND_BIG_SPHERE;
  V3<R> oc = ray.o - o.cen;
  R hb = dot(oc, ray.d);
  const R r2 = o.rad * o.rad;
  R c, disc;
  // two forms of the discriminant: from the closest-approach offset (accurate when
  // the origin is far from the sphere compared with its radius) or b^2 - a c
  // (accurate when the origin is on or near the surface)
  if (o.big) {
    // a sphere that dwarfs the scene: the rays start on or near it, the second form; the first
    // only behind a branch that the lanes of a wave practically never take
    R K = *(const R*)(sc.base + sc.L.sph_k + idx * (uint32_t)sizeof(R));
    c = ray.oo - (R)2 * dot(ray.o, o.cen) + K;
    disc = hb * hb - ray.a * c;
    o.l = oc;
    if (c > (R)3 * r2) {
      o.l = oc - (hb * ray.inv_a) * ray.d;
      disc = ray.a * (r2 - dot(o.l, o.l));
    }
  } else {
    c = dot(oc, oc) - r2;
    o.l = oc - (hb * ray.inv_a) * ray.d;
    disc = (c > (R)3 * r2) ? ray.a * (r2 - dot(o.l, o.l)) : hb * hb - ray.a * c;
  }
  if (!(disc > (R)0)) return false;
  R sq = M<R>::sqrt(disc);
  R q = -(hb + M<R>::copysign(sq, hb));
  R root_a = q * ray.inv_a, root_b = c * M<R>::rcp(q);
  o.t1 = M<R>::fmin(root_a, root_b);
  o.t2 = M<R>::fmax(root_a, root_b);
  o.s = sq * ray.inv_a;
  return true;
}

// Offset of the ray's point of closest approach from a sphere's centre, l = (o - c) - q d with
// q = (o - c).d / |d|^2, for shading.  |l| is of the size of the radius while o - c and q d are
// of the size of the distance travelled, so the two roundings that matter (of o - c and of the
// product q d) are carried along as error terms (TwoSum, TwoProduct by fused multiply-add):
// the result is as good as the ray itself.  Plain fp32 leaves an error of ~1e-6 of the
// distance, 1e-5 of a 0.2-radius sphere's normal at 20 units.  No contraction in here: the
// error terms are differences of rounded values.
template <typename R>
__device__ __forceinline__ V3<R> closest_approach_offset(const Ray<R>& ray, V3<R> cen, R& q) {
#pragma clang fp contract(off)
  const R o[3] = {ray.o.x, ray.o.y, ray.o.z}, c[3] = {cen.x, cen.y, cen.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z};
  R oc[3], oce[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // o - c = oc + oce exactly
    const R nb = -c[k];
    oc[k] = o[k] + nb;
    const R bp = oc[k] - o[k];
    const R ap = oc[k] - bp;
    oce[k] = (o[k] - ap) + (nb - bp);
  }
  q = (oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2]) * ray.inv_a;
  R l[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // q d = p + pe exactly
    const R p = q * d[k];
    const R pe = fma_r(q, d[k], -p);
    l[k] = (oc[k] - p) + (oce[k] - pe);
  }
  return mk<R>(l[0], l[1], l[2]);
}

template <typename R>
struct RectGeom {
  R a0, a1, b0, b1, k;
};
template <typename R>
__device__ __forceinline__ RectGeom<R> load_rect(const Scene& sc, uint32_t idx) {
  typename M<R>::R4 r0 = load_r4<R>(sc, sc.L.rect, idx * 2u);
  R k = *(const R*)(sc.base + sc.L.rect + (idx * 8u + 4u) * (uint32_t)sizeof(R));
  RectGeom<R> g;
  g.a0 = r0.x; g.a1 = r0.y; g.b0 = r0.z; g.b1 = r0.w; g.k = k;
  return g;
}
// components of v along the rect's (a, b, k) axes: XY -> (x,y,z), XZ -> (x,z,y), YZ -> (y,z,x)
template <typename R>
__device__ __forceinline__ void rect_axes(uint32_t kind, V3<R> v, R& a, R& b, R& k) {
  a = (kind == KIND_YZRECT) ? v.y : v.x;
  b = (kind == KIND_XYRECT) ? v.y : v.z;
  k = (kind == KIND_XYRECT) ? v.z : ((kind == KIND_XZRECT) ? v.y : v.x);
}

// fp64: {A, e1, e2, n = e1 x e2}.  fp32: {A, U, V, n} with U = (e2 x n)/|n|^2, V = (n x e1)/|n|^2, the
// dual basis of the edges: the barycentric coordinates of a point A + x of the plane are x.U and x.V.
template <typename R>
struct TriGeom {
  V3<R> A, e1, e2, n;  // fp32: e1, e2 hold U, V
};
// WHOLE: how much of the record is read in whole 16-byte words before any of it is used (fp32)
enum { TRI_AS_NEEDED = 0, TRI_FIRST_AND_LAST = 1, TRI_ALL = 2 };
template <typename R, int WHOLE = TRI_AS_NEEDED>
__device__ __forceinline__ TriGeom<R> load_tri(const Scene& sc, uint32_t idx) {
  typename M<R>::R4 q0 = load_r4<R>(sc, sc.L.tri, idx * 3u);
  typename M<R>::R4 q1 = load_r4<R>(sc, sc.L.tri, idx * 3u + 1u);
  typename M<R>::R4 q2 = load_r4<R>(sc, sc.L.tri, idx * 3u + 2u);
  if constexpr (sizeof(R) == 4 && WHOLE == TRI_ALL) {
    // HBM store: the record is read as the three aligned 16-byte words it is made of, all issued
    // before the first is used.  Left to itself the compiler reads a triangle in two rounds of two
    // narrower loads each (A and n, then the rest once t is known): four accesses of the vector
    // cache per lane where three do, and two waits where one does.
    asm volatile("" ::"v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w),
                 "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w));
  } else if constexpr (sizeof(R) == 4 && WHOLE == TRI_FIRST_AND_LAST) {
    // On chip, in the kernels of triangle hierarchies (LDS reads; the kernel held to 64 vector registers
    // has no room for all twelve values at once): the first and the last word whole (A and n, what t
    // needs), the middle one after them; the compiler then reads that one whole as well.  Three
    // ds_read_b128 and one wait per triangle instead of six narrower reads and two waits.
    asm volatile("" ::"v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w));
  }
  TriGeom<R> g;
  g.A = mk<R>(q0.x, q0.y, q0.z);
  g.e1 = mk<R>(q0.w, q1.x, q1.y);
  g.e2 = mk<R>(q1.z, q1.w, q2.x);
  g.n = mk<R>(q2.y, q2.z, q2.w);
  return g;
}
// Triangle.hit (Hittable.scala:97-113) with edges and normal precomputed.  fp64 follows the
// reference's arithmetic (u and v from (o - A) x d).  fp32 takes them from the hit point instead,
// u = (ao + t d).U, v = (ao + t d).V: the same numbers in exact arithmetic, errors of the same size
// (an ulp of the distance over the edge length), and five vector instructions fewer per test.
template <typename R>
__device__ __forceinline__ void tri_solve(const TriGeom<R>& g, const Ray<R>& ray, R& det, R& t, R& u,
                                          R& v) {
  V3<R> ao = ray.o - g.A;
  det = -dot(ray.d, g.n);
  R inv_det = M<R>::rcp(det);
  t = dot(ao, g.n) * inv_det;
  if constexpr (sizeof(R) == 4) {
    const V3<R> x = ao + t * ray.d;
    u = dot(x, g.e1);
    v = dot(x, g.e2);
  } else {
    V3<R> dao = cross(ao, ray.d);
    u = dot(g.e2, dao) * inv_det;
    v = -dot(g.e1, dao) * inv_det;
  }
}

// One primitive against the ray; a closer accepted hit replaces tv.best_*.
// The accepted hit is taken by selects, not branches: the traversal loop is short of
// scalar issue, not of vector work.
template <typename R, bool CNT>
__device__ __forceinline__ void sphere_test(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t ref,
                                            uint32_t kind, uint32_t idx, R t_min, Cnt& cnt) {
  if (CNT) cnt.v[CTR_SPHERE]++;
  SphereSolve<R> ss;
  if (sphere_solve<R>(sc, ray, kind, idx, ss)) {
    R t = (ss.t1 > t_min) ? ss.t1 : ss.t2;
    const bool ok = t > t_min && t < tv.best_t;
    tv.best_t = ok ? t : tv.best_t;
    tv.best_ref = ok ? ref : tv.best_ref;
  }
}
template <typename R, bool CNT, int WHOLE = TRI_AS_NEEDED>
__device__ __forceinline__ void triangle_test(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t ref,
                                              uint32_t idx, R t_min, Cnt& cnt) {
  if (CNT) cnt.v[CTR_TRI]++;
  TriGeom<R> g = load_tri<R, WHOLE>(sc, idx);
  R det, t, u, v;
  tri_solve<R>(g, ray, det, t, u, v);
  R w = (R)1 - u - v;
  const bool ok = det > (R)1e-6 && t >= t_min && t < tv.best_t && u >= (R)0 && v >= (R)0 && w >= (R)0;
  tv.best_t = ok ? t : tv.best_t;
  tv.best_ref = ok ? ref : tv.best_ref;
}
// A triangle leaf: triangle idx, and for a pair (KIND_TRI_PAIR) triangle idx + 1 as well.  The
// hit is recorded under the single triangle's own reference.  Sphere leaves likewise.
template <typename R, bool CNT, bool WIDE, int WHOLE = (WIDE ? TRI_ALL : TRI_AS_NEEDED)>
__device__ __forceinline__ void triangle_pair_test(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t idx,
                                                   bool pair, R t_min, Cnt& cnt) {
  const uint32_t single = RefT<WIDE>::LEAF | (KIND_TRIANGLE << RefT<WIDE>::KIND_SHIFT);
  triangle_test<R, CNT, WHOLE>(sc, ray, tv, single | idx, idx, t_min, cnt);
  if (pair) triangle_test<R, CNT, WHOLE>(sc, ray, tv, single | (idx + 1u), idx + 1u, t_min, cnt);
}
template <typename R, bool CNT, bool WIDE>
__device__ __forceinline__ void sphere_pair_test(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t idx,
                                                 bool pair, R t_min, Cnt& cnt) {
  const uint32_t single = RefT<WIDE>::LEAF | (KIND_SPHERE << RefT<WIDE>::KIND_SHIFT);
  sphere_test<R, CNT>(sc, ray, tv, single | idx, KIND_SPHERE, idx, t_min, cnt);
  if (pair) sphere_test<R, CNT>(sc, ray, tv, single | (idx + 1u), KIND_SPHERE, idx + 1u, t_min, cnt);
}
// XY / XZ / YZ rect, Hittable.scala:55-92.  K = the axis the rect is perpendicular to, a
// compile-time fact of each of the three copies: picking the components with selects on the
// kind cost eight of them per test, and 1/d of that axis is in the ray already.  The hoisted
// primitives are the same for every lane, so there the three copies sit behind scalar branches
// (the Cornell box tests six rects per ray this way).
template <typename R, bool CNT, int K>
__device__ __forceinline__ void rect_test_axis(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t ref,
                                               uint32_t idx, R t_min, Cnt& cnt) {
  if (CNT) cnt.v[CTR_RECT]++;
  RectGeom<R> g = load_rect<R>(sc, idx);
  // (a, b, k) axes: XY -> (x, y, z), XZ -> (x, z, y), YZ -> (y, z, x)
  const R oa = K == 0 ? ray.o.y : ray.o.x, ob = K == 2 ? ray.o.y : ray.o.z, ok_ = K == 2 ? ray.o.z : (K == 1 ? ray.o.y : ray.o.x);
  const R da = K == 0 ? ray.d.y : ray.d.x, db = K == 2 ? ray.d.y : ray.d.z;
  const R ik = K == 2 ? ray.inv.z : (K == 1 ? ray.inv.y : ray.inv.x);  // 1 / d_k
  R t = (g.k - ok_) * ik;
  if constexpr (sizeof(R) == 8) {
    // the validation path divides, as the reference does (Hittable.scala:56): a ray through the
    // line where two walls meet lies on the inclusive edge of both rects only in that arithmetic;
    // a product with 1/d_k puts it an ulp outside of both, and the ray leaves the box there
    const R dk = K == 2 ? ray.d.z : (K == 1 ? ray.d.y : ray.d.x);
    t = (g.k - ok_) / dk;
  }
  R a = oa + t * da, bb = ob + t * db;
  const bool ok = t >= t_min && t < tv.best_t && a >= g.a0 && a <= g.a1 && bb >= g.b0 && bb <= g.b1;
  tv.best_t = ok ? t : tv.best_t;
  tv.best_ref = ok ? ref : tv.best_ref;
}
template <typename R, bool CNT>
__device__ __forceinline__ void rect_test(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t ref,
                                          uint32_t kind, uint32_t idx, R t_min, Cnt& cnt) {
  if (kind == KIND_XYRECT) rect_test_axis<R, CNT, 2>(sc, ray, tv, ref, idx, t_min, cnt);
  else if (kind == KIND_XZRECT) rect_test_axis<R, CNT, 1>(sc, ray, tv, ref, idx, t_min, cnt);
  else rect_test_axis<R, CNT, 0>(sc, ray, tv, ref, idx, t_min, cnt);
}
template <typename R, bool CNT, bool WIDE>
__device__ __forceinline__ void prim_test(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t ref,
                                          R t_min, Cnt& cnt) {
  const uint32_t kind = (ref >> RefT<WIDE>::KIND_SHIFT) & 7u, idx = ref & RefT<WIDE>::INDEX_MASK;
  if (kind == KIND_TRIANGLE) triangle_test<R, CNT, (WIDE ? TRI_ALL : TRI_AS_NEEDED)>(sc, ray, tv, ref, idx, t_min, cnt);
  else if (kind == KIND_TRI_PAIR) triangle_pair_test<R, CNT, WIDE>(sc, ray, tv, idx, true, t_min, cnt);
  else if (kind == KIND_SPH_PAIR) sphere_pair_test<R, CNT, WIDE>(sc, ray, tv, idx, true, t_min, cnt);
  else if (kind == KIND_SPHERE || kind == KIND_BIG_SPHERE) sphere_test<R, CNT>(sc, ray, tv, ref, kind, idx, t_min, cnt);
  else rect_test<R, CNT>(sc, ray, tv, ref, kind, idx, t_min, cnt);
}
// A leaf of the hierarchy.  The host records which kinds of primitive the hierarchy
// holds (SceneLayout::tree_kinds, wave-uniform): a mesh or a field of spheres skips the
// dispatch on the kind altogether.
template <typename R, bool CNT, bool WIDE, bool TRIS>
__device__ __forceinline__ void leaf_test(const Scene& sc, const Ray<R>& ray, Trav<R>& tv, uint32_t ref,
                                          R t_min, Cnt& cnt) {
  const uint32_t idx = ref & RefT<WIDE>::INDEX_MASK;
  if (TRIS || sc.L.tree_kinds == TREE_TRIANGLES)
    triangle_pair_test<R, CNT, WIDE, (WIDE ? TRI_ALL : (TRIS ? TRI_FIRST_AND_LAST : TRI_AS_NEEDED))>(
        sc, ray, tv, idx, ((ref >> RefT<WIDE>::KIND_SHIFT) & 7u) == KIND_TRI_PAIR, t_min, cnt);
  else if (sc.L.tree_kinds == TREE_SPHERES)
    sphere_pair_test<R, CNT, WIDE>(sc, ray, tv, idx, ((ref >> RefT<WIDE>::KIND_SHIFT) & 7u) == KIND_SPH_PAIR, t_min, cnt);
  else prim_test<R, CNT, WIDE>(sc, ray, tv, ref, t_min, cnt);
}

// Next reference from the lane's stack.  (Entries carry no entry distance: with
// near-child-first descent, culling popped entries against the closest hit saved only
// ~2.5 % of the box tests on the reference's scenes, less than the LDS it costs.)
// Level 0 of a stack holds REF_EMPTY, written once per kernel: the pop that finds the
// stack empty reads it like any other entry and the traversal ends, without a compare
// and select on every pop.  A push always stores above the top and bumps the count only
// when it is meant; the stack has two slots more than the hierarchy has levels (the
// REF_EMPTY below, the store of a full stack above).  On gfx950 a compare or select
// issues at half the rate of an add, and the loop is also short of scalar issue, so both
// operations are free of branches.
template <typename R, bool WIDE>
__device__ __forceinline__ uint32_t trav_pop(Trav<R>& tv, const uint8_t* lds) {
  tv.sp -= blockDim.x * RefT<WIDE>::STACK_ENTRY;  // one level down
  return stack_load<WIDE>(lds, tv.sp);
}

// Inner node: slab-test both children, descend into the nearer one that is hit
// and push the other, or pop when neither is.
template <typename R, bool CNT, bool WIDE>
__device__ __forceinline__ void node_step(const Scene& sc, const Ray<R>& ray, Trav<R>& tv,
                                          uint8_t* lds, R t_min, Cnt& cnt) {
  const uint32_t LEVEL = blockDim.x * RefT<WIDE>::STACK_ENTRY;
  // what a pop would return, read before the arithmetic so that its latency hides there
  const uint32_t below = tv.sp - LEVEL;
  const uint32_t top = stack_load<WIDE>(lds, below);
  R b[12];
  uint32_t c0, c1;
  load_node<WIDE>(sc, tv.cur, b, c0, c1);
  R tn0, tn1;
  bool h0, h1;
  slab2<R>(ray, b, t_min, tv.best_t, tn0, tn1, h0, h1);
  if (CNT) cnt.v[CTR_BOX] += 2;
  const bool both = h0 && h1, none = !(h0 || h1);
  const bool swp = tn1 < tn0;                         // meaningful when both are hit
  const uint32_t farc = swp ? c0 : c1;
  const uint32_t nearc = both ? (swp ? c1 : c0) : (h0 ? c0 : c1);
  stack_store<WIDE>(lds, tv.sp, farc);  // kept only if the address moves past it
  tv.cur = none ? top : nearc;
  tv.sp = both ? tv.sp + LEVEL : (none ? below : tv.sp);
}

// Start of a ray: the hoisted primitives (the same primitive for every lane, so
// the wave runs one code path), then the bounds of the hierarchy.
template <typename R, bool CNT, bool WIDE, bool FLAT = false>
__device__ __forceinline__ void ray_start(const Scene& sc, const Ray<R>& ray, Trav<R>& tv,
                                          uint32_t stack_level1, R t_min, Cnt& cnt) {
  trav_begin<R, WIDE>(tv, stack_level1);
  const uint32_t* hl = (const uint32_t*)(sc.base + sc.L.hoist);
  for (uint32_t i = 0; i < sc.L.n_hoist; ++i) {
    const uint32_t ref = __builtin_amdgcn_readfirstlane(hl[i]);
    prim_test<R, CNT, WIDE>(sc, ray, tv, ref, t_min, cnt);
  }
  tv.cur = RefT<WIDE>::EMPTY;
  if (!FLAT && sc.L.n_nodes != 0u) {
    const R* rb = (const R*)(sc.base + sc.L.root_box);
    R tn;
    if (CNT) cnt.v[CTR_BOX]++;
    if (slab<R>(ray, rb[0], rb[1], rb[2], rb[3], rb[4], rb[5], t_min, tv.best_t, tn)) tv.cur = 0u;
  }
}

// The whole closest-hit query of one lane (first-hit kernel).
template <typename R, bool WIDE>
__device__ __forceinline__ void trace_closest(const Scene& sc, const Ray<R>& ray, Trav<R>& tv,
                                              uint8_t* lds, uint32_t stack_level1, R t_min) {
  Cnt cnt;
  ray_start<R, false, WIDE>(sc, ray, tv, stack_level1, t_min, cnt);
  while (tv.cur != RefT<WIDE>::EMPTY) {
    if (!(tv.cur & RefT<WIDE>::LEAF)) {
      node_step<R, false, WIDE>(sc, ray, tv, lds, t_min, cnt);
    } else {
      prim_test<R, false, WIDE>(sc, ray, tv, tv.cur, t_min, cnt);
      tv.cur = trav_pop<R, WIDE>(tv, lds);
    }
  }
}

// ---------------------------------------------------------------------------
// hit geometry, textures, materials
// ---------------------------------------------------------------------------
template <typename R>
struct HitGeom {
  uint32_t kind, idx, meta;
  V3<R> p;      // Hit.position (Hittable.scala:5)
  V3<R> n_out;  // outward_normal as the primitive reports it
};

template <typename R, bool WIDE>
__device__ __forceinline__ HitGeom<R> hit_geometry(const Scene& sc, const Ray<R>& ray, R t,
                                                   uint32_t ref) {
  HitGeom<R> h;
  h.kind = (ref >> RefT<WIDE>::KIND_SHIFT) & 7u;
  h.idx = ref & RefT<WIDE>::INDEX_MASK;
  h.p = ray.o + t * ray.d;
  if (h.kind == KIND_SPHERE || h.kind == KIND_BIG_SPHERE) {
    // outward normal (p - center)/radius (Hittable.scala:44).  For an ordinary
    // sphere p - center is rebuilt from the closest-approach offset and the
    // half-chord, both of the size of the radius, so its error scales with the
    // radius and not with the distance the ray travelled; the hit point is then
    // center + that vector.
    V3<R> pc;
    R rad;
    if (h.kind == KIND_BIG_SPHERE) {  // the hit point is as accurate as the radius is large
      typename M<R>::R4 s = load_r4<R>(sc, sc.L.sph, h.idx);
      pc = h.p - mk<R>(s.x, s.y, s.z);
      rad = s.w;
    } else {
      typename M<R>::R4 s = load_r4<R>(sc, sc.L.sph, h.idx);
      const V3<R> cen = mk<R>(s.x, s.y, s.z);
      rad = s.w;
      R q;
      const V3<R> l0 = closest_approach_offset<R>(ray, cen, q);
      // one re-orthogonalisation of l against d removes the along-ray error that the
      // rounding of the closest-approach parameter left in it
      const V3<R> l = l0 - (dot(l0, ray.d) * ray.inv_a) * ray.d;
      // half-chord from the same l, so that |pc| = radius; the accepted root is the near one
      // when t lies before the point of closest approach (parameter -q)
      const R hc = M<R>::sqrt(M<R>::fmax(rad * rad - dot(l, l), (R)0) * ray.inv_a);
      pc = l + ((t + q <= (R)0) ? -hc : hc) * ray.d;
      h.p = cen + pc;
    }
    h.n_out = M<R>::rcp(rad) * pc;
    h.meta = load_u32(sc, sc.L.meta_sph, h.idx);
  } else if (h.kind == KIND_TRIANGLE) {
    typename M<R>::R4 q2 = load_r4<R>(sc, sc.L.tri, h.idx * 3u + 2u);
    h.n_out = mk<R>(q2.y, q2.z, q2.w);  // raw cross product, Hittable.scala:100,111
    h.meta = load_u32(sc, sc.L.meta_tri, h.idx);
  } else {
    // every rect reports +z (Hittable.scala:62, 76, 90)
    h.n_out = mk<R>((R)0, (R)0, (R)1);
    // the hit point lies on the plane: pin that coordinate so that rounding of
    // o + t*d cannot put the next ray's origin off the rect
    R k = *(const R*)(sc.base + sc.L.rect + (h.idx * 8u + 4u) * (uint32_t)sizeof(R));
    if (h.kind == KIND_XYRECT) h.p.z = k;
    else if (h.kind == KIND_XZRECT) h.p.y = k;
    else h.p.x = k;
    h.meta = load_u32(sc, sc.L.meta_rect, h.idx);
  }
  return h;
}

// Hit.uv, evaluated only when an Image texture asks for it.
template <typename R>
__device__ __forceinline__ void hit_uv(const Scene& sc, const Ray<R>& ray, const HitGeom<R>& h, R& u,
                                    R& v) {
  if (h.kind == KIND_SPHERE || h.kind == KIND_BIG_SPHERE) {
    // Sphere.get_uv of the NEAR root's point, whichever root was accepted
    // (Hittable.scala:44, 46 both pass r.at(t1))
    SphereSolve<R> ss;
    sphere_solve<R>(sc, ray, h.kind, h.idx, ss);
    V3<R> q = M<R>::rcp(ss.rad) * (ss.big ? (ray.o + ss.t1 * ray.d) - ss.cen : ss.l - ss.s * ray.d);
    const R PI = (R)3.14159265358979323846;
    R phi = M<R>::atan2(q.z, q.x);
    R theta = M<R>::asin(M<R>::fmin(M<R>::fmax(q.y, (R)-1), (R)1));
    u = (R)1 - (phi + PI) / ((R)2 * PI);
    v = (theta + PI / (R)2) / PI;
  } else if (h.kind == KIND_TRIANGLE) {
    TriGeom<R> g = load_tri<R>(sc, h.idx);
    R det, t;
    tri_solve<R>(g, ray, det, t, u, v);
  } else {
    RectGeom<R> g = load_rect<R>(sc, h.idx);
    R a, b, k;
    rect_axes<R>(h.kind, h.p, a, b, k);
    u = (a - g.a0) / (g.a1 - g.a0);
    v = (b - g.b0) / (g.b1 - g.b0);
  }
}

// Perlin.noise (Texture.scala:44-76).  The fused multiply-adds are written out and the
// compiler's own contraction is off in these two functions: seven octaves of this sum feed a
// sine, and left to itself the compiler fuses them differently in every kernel the function
// is inlined into, so the variants of the renderer would disagree in the last bit of a fifth
// of a Perlin scene's sums.
template <typename R>
__device__ __forceinline__ R perlin_noise(const Scene& sc, uint32_t table, V3<R> p) {
#pragma clang fp contract(off)
  const uint8_t* tb = sc.base + sc.L.perlin + table * sc.L.perlin_stride;
  const uint8_t* perm = tb + 256u * 4u * (uint32_t)sizeof(R);
  R fx = M<R>::floor(p.x), fy = M<R>::floor(p.y), fz = M<R>::floor(p.z);
  R u = p.x - fx, v = p.y - fy, w = p.z - fz;
  int i = (int)fx, j = (int)fy, k = (int)fz;
  R uu = u * u * fma_r((R)-2, u, (R)3);
  R vv = v * v * fma_r((R)-2, v, (R)3);
  R ww = w * w * fma_r((R)-2, w, (R)3);
  R sum = 0;
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        uint32_t g = (uint32_t)perm[(i + di) & 255] ^ (uint32_t)perm[256 + ((j + dj) & 255)] ^
                     (uint32_t)perm[512 + ((k + dk) & 255)];
        typename M<R>::R4 c = *(const typename M<R>::R4*)(tb + g * (uint32_t)(4 * sizeof(R)));
        R wx = di ? uu : (R)1 - uu, wy = dj ? vv : (R)1 - vv, wz = dk ? ww : (R)1 - ww;
        const R d = fma_r(c.z, w - (R)dk, fma_r(c.y, v - (R)dj, c.x * (u - (R)di)));
        sum = fma_r(wx * wy * wz, d, sum);
      }
  return sum;
}
// Perlin.turb, depth 7 (Texture.scala:78-88)
template <typename R>
__device__ __forceinline__ R perlin_turb(const Scene& sc, uint32_t table, V3<R> p) {
#pragma clang fp contract(off)
  R accum = 0, weight = 1;
  for (int i = 0; i < 7; ++i) {
    accum = fma_r(weight, perlin_noise<R>(sc, table, p), accum);
    p = (R)2 * p;
    weight *= (R)0.5;
  }
  return M<R>::fabs(accum);
}

// texture(uv, p) (Texture.scala:13, 21-25, 90, 98-107)
// FEAT: which texture kinds beyond SolidColor / Checker the scene uses; the others
// are compiled out (the Perlin and Image paths alone cost ~50 VGPRs).
constexpr uint32_t FEAT_PERLIN = 1u, FEAT_IMAGE = 2u;
// Not a texture kind: the hierarchy holds triangles only (a mesh scene), known when the
// kernel is compiled, so a leaf step has no dispatch on the kind of primitive at all.
constexpr uint32_t FEAT_TRIS = 4u;
// The scene has no hierarchy (every primitive is hoisted and tested when a ray starts): the
// kernel has no traversal phase, no stacks and keeps no traversal state in registers.
constexpr uint32_t FEAT_FLAT = 16u;

template <typename R, bool CNT, uint32_t FEAT>
__device__ __forceinline__ V3<R> texture_eval(const Scene& sc, const uint32_t* __restrict__ texels,
                                              uint32_t tex, const Ray<R>& ray, const HitGeom<R>& h,
                                              Cnt& cnt) {
  for (;;) {
    typename M<R>::R4 t = load_r4<R>(sc, sc.L.tex, tex);
    uint32_t w3 = M<R>::bits_lo(t.w);
    uint32_t kind = w3 & 3u;
    if (CNT) cnt.v[CTR_TEX0 + kind]++;
    if (kind == 0u) return mk<R>(t.x, t.y, t.z);
    if (kind == 1u) {  // Checker
      R sines = M<R>::sin(t.x * h.p.x) * M<R>::sin(t.x * h.p.y) * M<R>::sin(t.x * h.p.z);
      tex = (sines < (R)0) ? ((w3 >> 2) & 0x7FFFu) : (w3 >> 17);
      continue;
    }
    if constexpr ((FEAT & FEAT_PERLIN) != 0u) {
      if (kind == 2u) {  // Perlin
        R g = (R)0.5 * ((R)1 + M<R>::sin(t.x * h.p.z + (R)10 * perlin_turb<R>(sc, (w3 >> 2) & 0x7FFFu, h.p)));
        return mk<R>(g, g, g);
      }
    }
    if constexpr ((FEAT & FEAT_IMAGE) != 0u) {
      // Image: nearest texel, no linearisation (Texture.scala:98-107, Vec3.scala:36-40)
      R u, v;
      hit_uv<R>(sc, ray, h, u, v);
      const uint32_t* im = (const uint32_t*)(sc.base + sc.L.images) + ((w3 >> 2) & 0x7FFFu) * 4u;
      int width = (int)im[0], height = (int)im[1];
      uint64_t off = (uint64_t)im[2] | ((uint64_t)im[3] << 32);
      R uc = M<R>::fmin(M<R>::fmax(u, (R)0), (R)1);
      R vc = (R)1 - M<R>::fmin(M<R>::fmax(v, (R)0), (R)1);
      int ii = min(max((int)(uc * (R)width), 0), width - 1);
      int jj = min(max((int)(vc * (R)height), 0), height - 1);
      uint32_t rgb = texels[off + (uint64_t)jj * (uint64_t)width + (uint64_t)ii];
      const R s = (R)(1.0 / 255.0);
      return mk<R>((R)((rgb >> 16) & 0xFFu) * s, (R)((rgb >> 8) & 0xFFu) * s, (R)(rgb & 0xFFu) * s);
    }
    return mk<R>((R)0, (R)0, (R)0);  // a kind this kernel variant was not built for (host never selects it)
  }
}

// The transcendental part of an event's sample: sine and cosine of one uniform (as an angle in
// revolutions) and the square root of a second number.  A camera event needs them for the
// point of the lens (Vec3.random_in_unit_disk, Vec3.scala:23-27: radius sqrt(u2), angle u3), a
// scatter event for the direction (Vec3.random_unit_vector, Vec3.scala:29-34: angle u0,
// radius sqrt(1 - z^2) with z = 2 u1 - 1).  The lanes of a wave are at both kinds of event in
// the same round; worked out here once for all of them, the three slow instructions issue
// once per round and not once per kind.
template <typename R>
struct EventTrig {
  R s, c, rr;
};
template <typename R>
__device__ __forceinline__ EventTrig<R> event_trig(R angle, R radius_squared) {
  EventTrig<R> t;
  M<R>::sincos2pi(angle, t.s, t.c);
  t.rr = M<R>::sqrt(radius_squared);
  return t;
}
// what event_trig takes for a scatter event with uniforms u0, u1: angle u0, radius^2 = 1 - z^2
template <typename R>
__device__ __forceinline__ R scatter_z(R u1) { return (R)2 * u1 - (R)1; }
template <typename R>
__device__ __forceinline__ V3<R> unit_vector_from(const EventTrig<R>& t, R z) {
  return mk<R>(t.rr * t.c, t.rr * t.s, z);
}

// Radiance sums are kept in signed 32.32 fixed point: integer adds commute, so a
// pixel's sum does not depend on the order in which its paths finish, on how
// the frame is cut into launches or on how many GPUs share it.
// (No fused multiply-add across this function: inlined behind `T * e`, a contraction of
// x - floor(x) with the product would convert the unrounded product, and the sum would then
// depend on whether the caller held x in a register or had stored it as a number first.)
template <typename R>
__device__ __forceinline__ long long to_fixed(R x) {
#pragma clang fp contract(off)
  asm volatile("" : "+v"(x));  // x is a rounded number of type R from here on
  const R f = M<R>::floor(x);
  const int hi = (int)f;                                        // saturates; NaN -> 0
  const unsigned lo = (unsigned)((x - f) * (R)4294967296.0);    // [0, 2^32)
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)lo);
}
// Where a finished path's contribution goes: the wave's LDS tile while the
// path's work unit is still the wave's current one, else (a straggler of an
// older unit) straight to the frame buffer.
struct Sink {
  unsigned long long* lds;     // this pixel's three LDS accumulators
  unsigned long long* global;  // this pixel's three frame accumulators
  bool current;
};
template <typename R>
__device__ __forceinline__ void deposit(const Sink& k, V3<R> c) {
  const unsigned long long r = (unsigned long long)to_fixed<R>(c.x);
  const unsigned long long g = (unsigned long long)to_fixed<R>(c.y);
  const unsigned long long b = (unsigned long long)to_fixed<R>(c.z);
  if (k.current) {
    atomicAdd(k.lds + 0, r);
    atomicAdd(k.lds + 1, g);
    atomicAdd(k.lds + 2, b);
  } else {
    atomicAdd(k.global + 0, r);
    atomicAdd(k.global + 1, g);
    atomicAdd(k.global + 2, b);
  }
}

// The wavefront kernels' sink: a path contributes once, when it ends, so its contribution
// is stored (not added) at the path's own place in a buffer that a later kernel sums per
// pixel; no atomics.
struct StoreSink {
  float* c;         // this path's red value; green and blue follow `stride` floats apart
  uint32_t stride;
};
template <typename R>
__device__ __forceinline__ void deposit(const StoreSink& k, V3<R> c) {
  k.c[0] = (float)c.x;
  k.c[k.stride] = (float)c.y;
  k.c[2u * k.stride] = (float)c.z;
}

// What the shading of a hit leaves for the scatter step.
template <typename R>
struct Surface {
  V3<R> p, n;     // Hit.position, Hit.normal (not normalised)
  R param;        // roughness / ior
  uint32_t mkind; // HT_MAT_*
  bool front;     // Hit.front_face
};

// First half of material.scatter / emit for the closest hit (Material.scala:9-55,
// Camera.scala:29-33): geometry, material record, texture.  Emission deposits and
// ends the path (returns false); otherwise T is multiplied by the attenuation and
// `sf` describes the surface for scatter_direction().
template <typename R, bool CNT, uint32_t FEAT, bool WIDE, typename SinkT>
__device__ __forceinline__ bool shade_surface(const Scene& sc, const uint32_t* __restrict__ texels, const Ray<R>& ray,
                                              const Trav<R>& tv, V3<R>& T, const SinkT& sink,
                                              Surface<R>& sf, Cnt& cnt) {
  HitGeom<R> h = hit_geometry<R, WIDE>(sc, ray, tv.best_t, tv.best_ref);
  struct alignas(2 * sizeof(R)) MatRec { R x, y; };  // (type | texture << 4) in the low word, parameter
  const MatRec m = *(const MatRec*)(sc.base + sc.L.mat + (h.meta >> META_MAT_SHIFT) * (uint32_t)sizeof(MatRec));
  const uint32_t m0 = M<R>::bits_lo(m.x);
  const uint32_t tex = m0 >> 4;
  sf.mkind = m0 & 15u;
  sf.param = m.y;
  if (CNT) cnt.v[CTR_SHADE0 + sf.mkind]++;
  sf.front = dot(ray.d, h.n_out) < (R)0;          // Hit.front_face
  sf.n = sf.front ? h.n_out : -h.n_out;           // Hit.normal (not normalised)
  sf.p = h.p;
  V3<R> col;
  if (sc.L.solid_textures_only) {  // wave-uniform: no texture loop, no kind dispatch
    typename M<R>::R4 t = load_r4<R>(sc, sc.L.tex, tex);
    col = mk<R>(t.x, t.y, t.z);
    if (CNT) cnt.v[CTR_TEX0]++;
  } else {
    col = texture_eval<R, CNT, FEAT>(sc, texels, tex, ray, h, cnt);
  }
  if (sf.mkind == 3u) {  // Emission: never scatters, emits on both faces
    deposit<R>(sink, T * (sf.param * col));
    return false;
  }
  T = T * col;
  return true;
}

// Second half: the scattered direction from the event's four uniforms.
// Returns false when a Glossy bounce is absorbed (Material.scala:18-19).
template <typename R>
__device__ __forceinline__ bool scatter_direction(const Surface<R>& sf, const Ray<R>& ray,
                                                  const uint32_t rnd[4], const EventTrig<R>& trig, R z, V3<R>& dir,
                                                  bool diffuse_only) {
  const V3<R> n = sf.n;
  if (diffuse_only || sf.mkind == 0u) {  // Diffuse
    dir = n + unit_vector_from<R>(trig, z);
    return true;
  }
  const V3<R> ud = M<R>::rsqrt(ray.a) * ray.d;  // ray.direction.unit_vector
  if (sf.mkind == 1u) {  // Glossy
    V3<R> refl = ud - ((R)2 * dot(ud, n)) * n;
    V3<R> ball = M<R>::cbrt(u01<R>(rnd[2])) * unit_vector_from<R>(trig, z);
    dir = refl + sf.param * ball;
    return dot(dir, n) > (R)0;
  }
  // Glass
  R eta = sf.front ? M<R>::rcp(sf.param) : sf.param;
  R cos_theta = M<R>::fmin(-dot(ud, n), (R)1);
  R sin_theta = M<R>::sqrt((R)1 - cos_theta * cos_theta);
  bool reflect_it = eta * sin_theta > (R)1;
  if (!reflect_it) {
    R r0 = ((R)1 - eta) / ((R)1 + eta);
    R r1 = r0 * r0;
    R x = (R)1 - cos_theta;
    R x2 = x * x;
    R sch = r1 + ((R)1 - r1) * (x2 * x2 * x);
    reflect_it = sch > u01<R>(rnd[0]);
  }
  if (reflect_it) {
    dir = ud - ((R)2 * dot(ud, n)) * n;
  } else {
    R ct = -dot(ud, n);
    V3<R> perp = eta * (ud + ct * n);
    V3<R> par = (-M<R>::sqrt(M<R>::fabs((R)1 - dot(perp, perp)))) * n;
    dir = perp + par;
  }
  return true;
}

// Camera.get_ray (Camera.scala:19-23) for one sample of pixel (i, j)
// (main.scala:88-89); rnd = the path's camera event.
// trig = event_trig(u3, u2) of the event's uniforms (unused with the jitter off).
template <typename R, typename ArgsT, typename CamPtr>
__device__ __forceinline__ void camera_ray(const ArgsT& A, CamPtr camp, Ray<R>& ray,
                                           uint32_t i, uint32_t j, const uint32_t rnd[4], const EventTrig<R>& trig,
                                           float inv_w1, float inv_h1) {
  const auto& cam = *camp;
  R xi1 = (R)0.5, xi2 = (R)0.5;
  V3<R> offset = mk<R>((R)0, (R)0, (R)0);
  if (A.jitter) {
    xi1 = u01<R>(rnd[0]);
    xi2 = u01<R>(rnd[1]);
    // uniform point of the unit disk, scaled by the lens radius
    R r = cam.lens_radius * trig.rr;
    offset = (r * trig.c) * ld3<R>(cam.u) + (r * trig.s) * ld3<R>(cam.v);
  }
  // main.scala:88-89 divides.  With the jitter off the division is kept: a reciprocal and a
  // multiply move pixel-centre rays across texel boundaries (24 of 2304 pixels of plane.json).
  // A jittered fp32 sample position is a random number whose last bit carries nothing (i + xi
  // has already lost eleven bits of xi at i ~ 1900), so there the host's correctly rounded
  // 1/(W-1), 1/(H-1) are multiplied in: ~30 instructions fewer per camera ray.
  R s, t;
  if (sizeof(R) == 4 && A.jitter) {
    s = ((R)i + xi1) * (R)inv_w1;
    t = ((R)j + xi2) * (R)inv_h1;
  } else {
    s = ((R)i + xi1) / (R)(A.W - 1);
    t = ((R)j + xi2) / (R)(A.H - 1);
  }
  V3<R> origin = ld3<R>(cam.origin);
  ray.o = origin + offset;
  ray.d = (ld3<R>(cam.corner) + s * ld3<R>(cam.horizontal) + t * ld3<R>(cam.vertical)) - offset;
}

// The camera of render_kernel, read from the kernel-argument segment where it is used.
// Taken as an ordinary by-value parameter its 22 numbers sit in scalar registers for the
// whole kernel, and the kernel is short of those: it spilled ~40 of them to vector lanes.
// The empty asm hides the address from the optimiser, which would otherwise hoist the
// loads back out of the main loop.
template <typename R>
__device__ __forceinline__ const __attribute__((address_space(4))) CamDev<R>* kernarg_camera() {
  typedef const __attribute__((address_space(4))) char* kptr;
  kptr p = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(RenderArgs);  // second parameter
  asm volatile("" : "+s"(p));
  return (const __attribute__((address_space(4))) CamDev<R>*)p;
}

// The arguments of render_kernel read the same way, for the values that only the rare
// parts of the kernel use.
__device__ __forceinline__ const __attribute__((address_space(4))) RenderArgs* kernarg_args() {
  typedef const __attribute__((address_space(4))) char* kptr;
  kptr p = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return (const __attribute__((address_space(4))) RenderArgs*)p;
}

// The XCD (accelerator complex die, 0-7) this wave runs on: HW_REG_XCC_ID (20), bits 3:0.
__device__ __forceinline__ uint32_t xcc_id() {
  return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ void copy_scene_to_lds(uint8_t* lds, const uint8_t* blob, uint32_t bytes,
                                                  uint32_t nthreads) {
  const uint4* src = (const uint4*)blob;
  uint4* dst = (uint4*)lds;
  for (uint32_t i = threadIdx.x; i < bytes / 16u; i += nthreads) dst[i] = src[i];
  __syncthreads();
}

// A lane with a path is LIVE: it traverses while tv.cur != REF_EMPTY and waits to be shaded
// once it is (every lane without a ray in flight holds REF_EMPTY there, so the traversal
// loop tells the lanes apart by tv.cur alone).  DEAD: no path, wants one.  DONE: the
// queue ran dry.
enum : uint32_t { ST_LIVE = 0, ST_DEAD = 2, ST_DONE = 3 };

// LDS of a render workgroup: [scene image (on-chip store only)][traversal stacks]
// [per-wave tile sums]  (a lane pushes at most one entry per level of inner nodes above it)
__host__ __device__ inline uint32_t stack_bytes(uint32_t threads, uint32_t depth, uint32_t wide) {
  return (threads * (wide ? 3u : 2u) * (depth + 2u) + 15u) & ~15u;  // REF_EMPTY below, one slot above the deepest push
}
__host__ __device__ inline uint32_t image_bytes_in_lds(const SceneLayout& L) { return L.wide ? 0u : L.total_bytes; }
constexpr uint32_t TILE_PIXELS = TILE_W * TILE_H;       // 64 = one wave
constexpr uint32_t TILE_ACC_BYTES = TILE_PIXELS * 3u * 8u;

// The path-tracing kernel.  Work is a global queue of units; unit u is CHUNK
// consecutive samples of one 8x8 tile (chunk-major order, so the units of an
// expensive tile are spread over the whole launch).  A wave streams units: its
// dead lanes claim the next (pixel, sample) items of the current unit, and when
// the unit runs out the wave moves on to the next one at once while the
// stragglers of the old unit are still in flight.
// Waves per SIMD the register allocator must leave room for: the plain fp32 variants
// are held to 80 VGPRs, so that 24 waves fit a CU next to the scene image.  (The variants
// that production launches go through render_kernel_7 / render_kernel_8 below.)
template <typename R, bool CNT, uint32_t FEAT, bool WIDE>
struct MinWaves { static constexpr int value = (sizeof(R) == 4 && !CNT) ? ((FEAT & 3u) == 0u ? 6 : ((FEAT & 3u) == 3u ? 4 : 5)) : 1; };

template <typename R, bool CNT, uint32_t FEAT, bool WIDE>
__device__ __forceinline__ void render_body(const RenderArgs& A) {
  extern __shared__ __align__(16) uint8_t lds[];
  typedef RefT<WIDE> RF;
  constexpr bool FLAT = (FEAT & FEAT_FLAT) != 0u;  // no hierarchy: a ray is finished when it starts
  const uint32_t BT = blockDim.x;  // <= M<R>::BT, a multiple of 64
  Scene sc;
  sc.L = A.L;
  uint32_t lds_image = 0;
  if constexpr (WIDE) {
    sc.base = A.blob;
  } else {
    copy_scene_to_lds(lds, A.blob, A.L.total_bytes, BT);
    sc.base = lds;
    lds_image = A.L.total_bytes;
  }
  // the lane's stack: level 0 (holds REF_EMPTY) and level 1, as byte offsets into lds
  const uint32_t stack_level0 = lds_address(lds) + lds_image + threadIdx.x * 2u;
  const uint32_t stack_level1 = stack_level0 + BT * RF::STACK_ENTRY;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long* tile_acc = (unsigned long long*)(lds + lds_image + stack_bytes(BT, A.L.depth, WIDE ? 1u : 0u) +
                                                       (threadIdx.x >> 6) * TILE_ACC_BYTES);
  unsigned long long* frame = (unsigned long long*)A.accum;
  const R t_min = (R)0.001;  // Camera.scala:28
  static_assert(sizeof(RenderArgs) % alignof(CamDev<R>) == 0, "camera follows RenderArgs without padding");
  Cnt cnt;
  uint64_t clk0 = 0, real0 = 0;
  if (CNT) {
#pragma unroll
    for (int k = 0; k < NUM_COUNTERS; ++k) cnt.v[k] = 0;
    clk0 = __builtin_amdgcn_s_memtime();
    real0 = __builtin_amdgcn_s_memrealtime();
  }
  tile_acc[lane * 3u + 0u] = 0ull;
  tile_acc[lane * 3u + 1u] = 0ull;
  tile_acc[lane * 3u + 2u] = 0ull;

  // wave-uniform state of the current unit
  uint32_t epoch = 0;          // number of units this wave has opened
  uint32_t next_item = 0, pool_size = 0;
  uint32_t tile_x0 = 0, tile_y0 = 0, sample_base = 0;
  bool queue_empty = false;
  // HBM store: the band of tiles this wave takes its units from (that of its XCD to begin with) and
  // the number of bands it has found used up
  uint32_t band = 0, bands_done = 0;
  if constexpr (WIDE) band = xcc_id() & (kernarg_args()->q.bands - 1u);

  // per-lane path state
  Ray<R> ray;
  Trav<R> tv;
  V3<R> T = mk<R>((R)1, (R)1, (R)1);
  uint32_t state = ST_DEAD;
  uint32_t my_epoch = 0, pix = 0, pixel = 0, sample = 0, bounce = 0;
  ray.o = ray.d = ray.inv = mk<R>((R)0, (R)0, (R)0);
  ray.oi = mk<R>((R)0, (R)0, (R)0);
  ray.a = ray.inv_a = ray.oo = (R)0;
  trav_begin<R, WIDE>(tv, stack_level1);
  tv.cur = RF::EMPTY;
  if constexpr (!FLAT) stack_store<WIDE>(lds, stack_level0, RF::EMPTY);

  // adds the wave's tile sums to the frame and clears them
  auto flush_tile = [&]() {
    const uint32_t px = tile_x0 + (lane & 7u), py = tile_y0 + (lane >> 3);
    if (px < (uint32_t)A.W && py < (uint32_t)A.H) {
      unsigned long long* out = frame + ((size_t)py * (uint32_t)A.W + px) * 3u;
#pragma unroll
      for (uint32_t c = 0; c < 3u; ++c) {
        const unsigned long long v = tile_acc[lane * 3u + c];
        if (v != 0ull) atomicAdd(out + c, v);
        tile_acc[lane * 3u + c] = 0ull;
      }
    }
  };

  for (;;) {
    uint64_t ts0 = 0;
    if (CNT) ts0 = __builtin_amdgcn_s_memtime();
    // ---- 1. shade the finished rays: deposit, or prepare a scatter ------------------
    // A lane leaves this block as DEAD (path ended) or with `scatter` set.
    if (CNT) {
      const uint64_t w = __ballot(state == ST_LIVE && (FLAT || tv.cur == RF::EMPTY));
      if (lane == 0 && w != 0ull) cnt.v[CTR_SHADE_ROUNDS]++;
    }
    bool scatter = false;
    Surface<R> sf;
    sf.p = sf.n = mk<R>((R)0, (R)0, (R)0);
    sf.param = (R)0;
    sf.mkind = 0u;
    sf.front = false;
    if (state == ST_LIVE && (FLAT || tv.cur == RF::EMPTY)) {
      if constexpr ((FEAT & FEAT_TRIS) != 0u && sizeof(R) == 4) ray_sphere_terms<R>(ray);
      Sink sink;
      sink.lds = tile_acc + pix * 3u;
      sink.global = frame + (size_t)pixel * 3u;
      sink.current = my_epoch == epoch;
      state = ST_DEAD;
      if (tv.best_ref == RF::EMPTY) {
        const auto* cp = kernarg_camera<R>();
        const V3<R> bg = mk<R>(cp->background[0], cp->background[1], cp->background[2]);
        deposit<R>(sink, T * bg);  // case None => background (Camera.scala:34)
        if (CNT) cnt.v[CTR_MISS]++;
      } else if (shade_surface<R, CNT, FEAT, WIDE>(sc, A.texels, ray, tv, T, sink, sf, cnt)) {
        // depth exhausted: the scattered ray's ray_color returns 0 (Camera.scala:27),
        // so the path contributes nothing more
        scatter = bounce < (uint32_t)A.max_depth;
      }
    }
    uint64_t ts1 = 0;
    if (CNT) ts1 = __builtin_amdgcn_s_memtime();
    // ---- 2. dead lanes claim the next (pixel, sample) items ---------------------------
    bool camera = false;
    uint32_t cpx = 0, cpy = 0;
    for (;;) {
      const uint64_t dead = __ballot(state == ST_DEAD && !scatter && !camera);
      if (dead == 0ull) break;
      if (next_item >= pool_size) {  // wave-uniform: the current unit is used up
        if (queue_empty) {
          if (state == ST_DEAD && !scatter && !camera) state = ST_DONE;
          break;
        }
        if (epoch != 0u) flush_tile();
        // The queue's geometry comes from the host (RenderArgs::q, work_queue.h) and is read here, once
        // per unit, on the spot: as values kept over the whole kernel it held ten scalar registers,
        // and worked out here it took four divisions.
        const auto* ca = kernarg_args();
        uint32_t ck = 0, k = 0;  // the unit: step ck of the launch's own tile number k
        if constexpr (WIDE) {
          // HBM store: the launch's own tiles are cut into bands of consecutive tiles (rows of the
          // picture), each with a queue head of its own, and a wave takes units from the band of the
          // XCD it runs on: the paths of one band meet one part of the mesh, which then stays in that
          // XCD's L2 (4 MB each; they share nothing).  A wave whose band is used up goes on to the
          // next, so the XCDs whose bands were cheap help with the others.  Which band a wave takes
          // is a matter of speed only: every unit is handed out once, whatever the placement.
          bool none_left = false;
          for (;;) {
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(ca->tile_counter + band * QUEUE_BAND_WORDS, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u < ca->q.band_units[band]) {
              ck = fastdiv(u, ca->q.band_div[band].m, ca->q.band_div[band].sh);
              k = ca->q.band_start[band] + (u - ck * ca->q.band_tiles[band]);
              break;
            }
            // this band is used up (a head only grows): on to the next one
            band = (band + 1u) & (ca->q.bands - 1u);
            if (++bands_done >= ca->q.bands) {
              none_left = true;
              break;
            }
          }
          if (ca->timeline != nullptr && lane == 0)  // profiling run: when units are fetched
            atomicAdd(ca->timeline + (none_left ? TIMELINE_BINS : 0u) +
                          (((uint32_t)(__builtin_amdgcn_s_memrealtime() >> 10)) & (TIMELINE_BINS - 1u)), 1u);
          if (none_left) {
            queue_empty = true;
            epoch++;  // whatever is still in flight deposits to the frame directly
            continue;
          }
        } else {
          // A wave's first unit is its own number in the launch (the host starts the queue head at
          // the number of waves): thousands of waves asking one counter at the same moment would
          // be served one at a time.  Later units come from the counter.
          uint32_t unit = blockIdx.x * (BT >> 6) + (threadIdx.x >> 6);
          if (epoch != 0u && lane == 0) unit = atomicAdd(ca->tile_counter, 1u);
          // every lane is here (the loop's control is wave-uniform); read as a scalar, so that
          // the unit's state lives in scalar registers and the loop branches on scalars
          unit = __builtin_amdgcn_readfirstlane(unit);
          if (ca->timeline != nullptr && lane == 0)  // profiling run: when units are fetched
            atomicAdd(ca->timeline + (unit >= ca->q.units ? TIMELINE_BINS : 0u) +
                          (((uint32_t)(__builtin_amdgcn_s_memrealtime() >> 10)) & (TIMELINE_BINS - 1u)), 1u);
          if (unit >= ca->q.units) {
            queue_empty = true;
            epoch++;  // whatever is still in flight deposits to the frame directly
            continue;
          }
          ck = fastdiv(unit, ca->q.div_tiles.m, ca->q.div_tiles.sh);
          k = unit - ck * ca->q.tiles;
        }
        uint32_t first, count;
        step_samples(ck, ca->q.big, ca->q.s_big, (uint32_t)ca->chunk, (uint32_t)ca->chunk_tail,
                     (uint32_t)ca->sample_count, first, count);
        {
          const uint32_t tile = (uint32_t)ca->tile_first + k * (uint32_t)ca->tile_stride;
          const uint32_t ty = fastdiv(tile, ca->q.div_tiles_x.m, ca->q.div_tiles_x.sh);
          tile_x0 = (tile - ty * ca->q.tiles_x) * TILE_W;
          tile_y0 = ty * TILE_H;
        }
        sample_base = (uint32_t)ca->sample_first + first;
        pool_size = TILE_PIXELS * count;
        next_item = 0;
        epoch++;
        continue;
      }
      if (CNT && lane == 0) cnt.v[CTR_REGEN_ROUNDS]++;
      if (state == ST_DEAD && !scatter && !camera) {
        const uint32_t item = next_item + (uint32_t)__popcll(dead & ((1ull << lane) - 1ull));
        if (item < pool_size) {
          const uint32_t q = item & (TILE_PIXELS - 1u);
          const uint32_t px = tile_x0 + (q & 7u), py = tile_y0 + (q >> 3);
          if (px < (uint32_t)A.W && py < (uint32_t)A.H) {  // else: claim again next round
            pix = q;
            cpx = px;
            cpy = py;
            pixel = py * (uint32_t)A.W + px;
            sample = sample_base + (item >> 6);
            my_epoch = epoch;
            bounce = 0;
            camera = true;
          }
        }
      }
      next_item = min(pool_size, next_item + (uint32_t)__popcll(dead));
    }
    // ---- 3. one random event for every lane that starts a ray --------------------------
    // (scatter event `bounce` of the path, or its camera event 0)
    if (scatter || camera) {
      uint32_t rnd[4];
      philox4x32_10_keyed(pixel, sample, bounce, 0u, kernarg_args()->philox_keys, rnd);
      bool go = true;
      // sine, cosine and square root for both kinds of event at once
      const R z = scatter_z<R>(u01<R>(rnd[1]));
      const EventTrig<R> trig = event_trig<R>(camera ? u01<R>(rnd[3]) : u01<R>(rnd[0]),
                                              camera ? u01<R>(rnd[2]) : (R)1 - z * z);
      if (camera) {
        {
          const auto* ca = kernarg_args();
          camera_ray<R, RenderArgs>(A, kernarg_camera<R>(), ray, cpx, cpy, rnd, trig, ca->inv_w1, ca->inv_h1);
        }
        T = mk<R>((R)1, (R)1, (R)1);
      } else {
        V3<R> dir;
        go = scatter_direction<R>(sf, ray, rnd, trig, z, dir, sc.L.scatter_diffuse_only != 0u);
        ray.o = sf.p;
        ray.d = dir;
      }
      if (go) {
        bounce++;
        ray_setup<R, FLAT>(ray);
        ray_start<R, CNT, WIDE, FLAT>(sc, ray, tv, stack_level1, t_min, cnt);
        state = ST_LIVE;
        if (CNT) cnt.v[CTR_RAYS]++;
      }  // else: absorbed, the lane stays DEAD and claims new work next round
    }
    uint64_t ts2 = 0;
    if (CNT) {
      ts2 = __builtin_amdgcn_s_memtime();
      if (lane == 0) {
        cnt.v[CTR_CYC_SHADE] += (uint32_t)(ts1 - ts0);
        cnt.v[CTR_CYC_REGEN] += (uint32_t)(ts2 - ts1);
      }
    }
    if (__ballot(state != ST_DONE) == 0ull) break;  // the queue is empty and every path ended
    // ---- 4. traversal: node steps while most lanes are at inner nodes, then one leaf
    //         step for the lanes that reached a primitive; until enough rays are done ----
    if constexpr (FLAT) continue;
    const int n_live = __popcll(__ballot(state == ST_LIVE));
    // The traversal yields to the next round once `wait_threshold` of the live lanes have finished
    // their rays, i.e. when the travelling lanes are down to `stop` (or to none).  The loop is
    // written for few scalar instructions: the scalar unit is one per CU, shared by its four
    // SIMDs, and this kernel keeps it about as busy as the vector ALUs.
    const int stop = max(n_live - A.wait_threshold, 0);
    const int leaf_weight = A.node_den - 1;
    for (;;) {
      bool leaf;
      uint64_t ml;
      for (;;) {
        const bool inner = tv.cur < RF::LEAF;  // REF_EMPTY has the leaf bit set
        leaf = (uint32_t)(tv.cur - RF::LEAF) < (RF::EMPTY - RF::LEAF);
        const uint64_t mi = __ballot(inner);
        ml = __ballot(leaf);
        // leaves wait while at least 1/node_den of the travelling lanes still descend
        if (mi == 0ull || leaf_weight * __popcll(mi) < __popcll(ml)) break;
        if (inner) {
          node_step<R, CNT, WIDE>(sc, ray, tv, lds, t_min, cnt);
          if (CNT) cnt.v[CTR_ACTIVE]++;
        }
        if (CNT) cnt.v[CTR_STEPS]++;
        if (CNT) cnt.v[CTR_NODE_STEPS]++;
        // a second step for the lanes that are still at inner nodes, without asking the
        // wave again: the ballots and counts of the loop head cost as much scalar issue
        // as half a node step
        if (tv.cur < RF::LEAF) {
          node_step<R, CNT, WIDE>(sc, ray, tv, lds, t_min, cnt);
          if (CNT) cnt.v[CTR_ACTIVE]++;
        }
        if (CNT) cnt.v[CTR_STEPS]++;
        if (CNT) cnt.v[CTR_NODE_STEPS]++;
      }
      if (ml != 0ull) {
        if (leaf) {
          leaf_test<R, CNT, WIDE, (FEAT & FEAT_TRIS) != 0u>(sc, ray, tv, tv.cur, t_min, cnt);
          tv.cur = trav_pop<R, WIDE>(tv, lds);
          if (CNT) cnt.v[CTR_ACTIVE]++;
        }
        if (CNT) cnt.v[CTR_STEPS]++;
      }
      if (__popcll(__ballot(tv.cur != RF::EMPTY)) <= stop) break;
    }
    if (CNT && lane == 0) cnt.v[CTR_CYC_TRAV] += (uint32_t)(__builtin_amdgcn_s_memtime() - ts2);
  }
  // the last unit's sums (epoch was bumped when the queue ran dry, so every later
  // deposit went to the frame; what is left in LDS belongs to the last real unit)
  flush_tile();
  {
    const auto* ca = kernarg_args();
    if (ca->timeline != nullptr && lane == 0)  // profiling run: when waves end
      atomicAdd(ca->timeline + 2u * TIMELINE_BINS +
                    (((uint32_t)(__builtin_amdgcn_s_memrealtime() >> 10)) & (TIMELINE_BINS - 1u)), 1u);
  }

  if (CNT && A.counters) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      A.counters[CTR_CLK_SHADER] = __builtin_amdgcn_s_memtime() - clk0;
      A.counters[CTR_CLK_REAL] = __builtin_amdgcn_s_memrealtime() - real0;
    }
#pragma unroll
    for (int k = 0; k < CTR_CLK_SHADER; ++k) {
      // wave reduction, then one atomic per wave
      uint32_t v = cnt.v[k];
      unsigned long long tot = v;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
      if (lane == 0) atomicAdd(&A.counters[k], tot);
    }
  }
}

template <typename R, bool CNT, uint32_t FEAT, bool WIDE>
__global__ void __launch_bounds__(M<R>::BT) __attribute__((amdgpu_waves_per_eu(MinWaves<R, CNT, FEAT, WIDE>::value)))
render_kernel(RenderArgs A, CamDev<R> cam) {
  render_body<R, CNT, FEAT, WIDE>(A);
}

// Scalar registers decide how many waves of a kernel a SIMD really holds: a wave's own count,
// rounded up to 16, plus 16 that the hardware keeps per wave, out of 800 per SIMD.  Up to 80:
// eight waves; up to 96: seven; up to 112: six.  The occupancy query does not know (it answers
// eight for all of these kernels), and a workgroup launched beyond the real limit is accepted
// and never becomes resident: shapes of 256 x 8 and 256 x 7 ran at exactly the rate of 256 x 6
// while the kernels used 98-106 scalar registers.  The compiler left to itself uses 96-106
// here, so the production kernels are compiled to a stated budget (the attribute takes a
// literal only, hence kernels of their own and not cases of the template above; the body is
// the same function).  Scalars over the budget live in lanes of a vector register.
//
// Seven waves per SIMD, 96 scalar registers (0-10 of them in lanes) and at most 72 vector
// registers: the fp32 production kernels of both stores with plain textures, and the
// on-chip one with image textures (which needs 74 vector registers: two spilled; scene.json
// 34.0 -> 35.2 G samples/s against six waves).  In the HBM store this is what makes the
// seventh 256-thread workgroup real that the three-byte stack entries make room for.
template <uint32_t FEAT, bool WIDE>
__global__ void __launch_bounds__(M<float>::BT) __attribute__((amdgpu_waves_per_eu(7), amdgpu_num_sgpr(96)))
render_kernel_7(RenderArgs A, CamDev<float> cam) {
  render_body<float, false, FEAT, WIDE>(A);
}
// Eight waves per SIMD, 80 scalar registers (about eleven in lanes: 1-3 % slower at equal
// occupancy): on-chip store, plain textures.  Taken where 32 waves fit a CU: a second
// 1024-thread workgroup for scenes whose image leaves room for it (dragon), 256 x 8 for
// scenes without a hierarchy (Cornell box).
template <uint32_t FEAT>
__global__ void __launch_bounds__(M<float>::BT) __attribute__((amdgpu_waves_per_eu(8), amdgpu_num_sgpr(80)))
render_kernel_8(RenderArgs A, CamDev<float> cam) {
  render_body<float, false, FEAT, false>(A);
}

// world.hit(camera ray through the pixel centre, 0.001, +inf) for every pixel.
template <typename R, bool WIDE>
__global__ void __launch_bounds__(M<R>::BT) first_hit_kernel(FirstHitArgs A, CamDev<R> cam) {
  extern __shared__ __align__(16) uint8_t lds[];
  const uint32_t BT = blockDim.x;
  Scene sc;
  sc.L = A.L;
  uint32_t lds_image = 0;
  if constexpr (WIDE) {
    sc.base = A.blob;
  } else {
    copy_scene_to_lds(lds, A.blob, A.L.total_bytes, BT);
    sc.base = lds;
    lds_image = A.L.total_bytes;
  }
  const uint32_t stack_level0 = lds_address(lds) + lds_image + threadIdx.x * 2u;
  stack_store<WIDE>(lds, stack_level0, RefT<WIDE>::EMPTY);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tiles_x = (uint32_t)(A.W + TILE_W - 1) / TILE_W;
  const uint32_t tiles_y = (uint32_t)(A.H + TILE_H - 1) / TILE_H;
  const uint32_t tile = blockIdx.x * (BT / 64u) + threadIdx.x / 64u;
  if (tile >= tiles_x * tiles_y) return;
  const uint32_t px = (tile % tiles_x) * TILE_W + (lane & 7u);
  const uint32_t py = (tile / tiles_x) * TILE_H + (lane >> 3);
  if (px >= (uint32_t)A.W || py >= (uint32_t)A.H) return;
  const uint32_t pixel = py * (uint32_t)A.W + px;

  RenderArgs ra;
  ra.jitter = 0;
  ra.W = A.W;
  ra.H = A.H;
  const uint32_t no_rnd[4] = {0u, 0u, 0u, 0u};
  Ray<R> ray;
  EventTrig<R> no_trig;
  no_trig.s = no_trig.c = no_trig.rr = (R)0;
  camera_ray<R, RenderArgs>(ra, &cam, ray, px, py, no_rnd, no_trig, 0.f, 0.f);  // jitter off: pixel centre, lens closed
  ray_setup<R>(ray);
  Trav<R> tv;
  trace_closest<R, WIDE>(sc, ray, tv, lds, stack_level0 + BT * RefT<WIDE>::STACK_ENTRY, (R)0.001);

  R* t_out = (R*)A.t;
  R* n_out = (R*)A.normal;
  if (tv.best_ref == RefT<WIDE>::EMPTY) {
    t_out[pixel] = M<R>::inf();
    A.prim[pixel] = -1;
    n_out[pixel * 3u] = n_out[pixel * 3u + 1u] = n_out[pixel * 3u + 2u] = (R)0;
  } else {
    HitGeom<R> h = hit_geometry<R, WIDE>(sc, ray, tv.best_t, tv.best_ref);
    const bool front = dot(ray.d, h.n_out) < (R)0;
    const V3<R> n = front ? h.n_out : -h.n_out;
    t_out[pixel] = tv.best_t;
    A.prim[pixel] = (int32_t)(h.meta & META_PRIM_MASK);
    n_out[pixel * 3u] = n.x;
    n_out[pixel * 3u + 1u] = n.y;
    n_out[pixel * 3u + 2u] = n.z;
  }
}

// Vec3.toRGB of one pixel's mean radiance (Vec3.scala:82-84): 255 * clamp(sqrt(x), 0, 0.999) truncated,
// NaN -> 0, packed 0x00RRGGBB; also stores the three means when `mean` is given.
__device__ __forceinline__ uint32_t resolve_pixel(const long long* __restrict__ sums, double inv_samples,
                                                  float* __restrict__ mean) {
  uint32_t code = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double m = (double)sums[c] * (1.0 / 4294967296.0) * inv_samples;
    if (mean) mean[c] = (float)m;
    double g = ::sqrt(m);
    g = ::fmin(::fmax(g, 0.0), 0.999);  // fmax(NaN, 0) = 0, the value NaN.toInt gives
    code = (code << 8) | (uint32_t)(int)(255.0 * g);
  }
  return code;
}

// mean radiance from the 32.32 fixed-point sums, Vec3.toRGB (Vec3.scala:82-84:
// 255 * clamp(sqrt(x), 0, 0.999), truncated; NaN -> 0) and the row flip of
// img.setRGB(i, H-j-1, ..) (main.scala:93)
__global__ void resolve_kernel(const long long* __restrict__ accum, int32_t W, int32_t H,
                               double inv_samples, float* __restrict__ rgb_mean,
                               uint32_t* __restrict__ rgb_packed) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (uint32_t)(W * H)) return;
  const uint32_t i = p % (uint32_t)W, j = p / (uint32_t)W;
  const uint32_t code = resolve_pixel(accum + (size_t)p * 3u, inv_samples, rgb_mean ? rgb_mean + (size_t)p * 3u : nullptr);
  if (rgb_packed) rgb_packed[(size_t)((uint32_t)H - 1u - j) * (uint32_t)W + i] = code;
}

// The same for the tiles one shard owns (tile_first, tile_first + tile_stride, ...), written
// compactly in tile order: thread = (owned tile k, pixel q of the tile).  No row flip here.
__global__ void resolve_tiles_kernel(const long long* __restrict__ accum, int32_t W, int32_t H, double inv_samples,
                                     uint32_t tile_first, uint32_t tile_stride, uint32_t n_owned,
                                     float* __restrict__ mean_tiles, uint32_t* __restrict__ packed_tiles) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = g / TILE_PIXELS, q = g % TILE_PIXELS;
  if (k >= n_owned) return;
  const uint32_t tiles_x = (uint32_t)(W + TILE_W - 1) / TILE_W;
  const uint32_t tile = tile_first + k * tile_stride;
  const uint32_t px = (tile % tiles_x) * TILE_W + (q & 7u), py = (tile / tiles_x) * TILE_H + (q >> 3);
  uint32_t code = 0;
  float m[3] = {0.f, 0.f, 0.f};
  if (px < (uint32_t)W && py < (uint32_t)H)
    code = resolve_pixel(accum + ((size_t)py * (uint32_t)W + px) * 3u, inv_samples, m);
  if (packed_tiles) packed_tiles[g] = code;
  if (mean_tiles) {
    mean_tiles[(size_t)g * 3u] = m[0];
    mean_tiles[(size_t)g * 3u + 1u] = m[1];
    mean_tiles[(size_t)g * 3u + 2u] = m[2];
  }
}

// The frame from the shards' tile buffers: thread = pixel (i, j), j = 0 the bottom row.
// Tile t belongs to shard t % n_shards and is that shard's tile number t / n_shards.
__global__ void assemble_tiles_kernel(int32_t W, int32_t H, uint32_t n_shards, size_t shard_pixels,
                                      const float* __restrict__ mean_tiles, const uint32_t* __restrict__ packed_tiles,
                                      float* __restrict__ rgb_mean, uint32_t* __restrict__ rgb_packed) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (uint32_t)(W * H)) return;
  const uint32_t i = p % (uint32_t)W, j = p / (uint32_t)W;
  const uint32_t tiles_x = (uint32_t)(W + TILE_W - 1) / TILE_W;
  const uint32_t tile = (j / TILE_H) * tiles_x + i / TILE_W;
  const size_t src = (size_t)(tile % n_shards) * shard_pixels + (size_t)(tile / n_shards) * TILE_PIXELS +
                     (j % TILE_H) * TILE_W + i % TILE_W;
  if (rgb_packed && packed_tiles) rgb_packed[(size_t)((uint32_t)H - 1u - j) * (uint32_t)W + i] = packed_tiles[src];
  if (rgb_mean && mean_tiles) {
    rgb_mean[(size_t)p * 3u] = mean_tiles[src * 3u];
    rgb_mean[(size_t)p * 3u + 1u] = mean_tiles[src * 3u + 1u];
    rgb_mean[(size_t)p * 3u + 2u] = mean_tiles[src * 3u + 2u];
  }
}

// dst += src: integer sums, so the order in which shards are combined does not matter
__global__ void accum_add_kernel(long long* __restrict__ dst, const long long* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}


// ---------------------------------------------------------------------------
// wavefront variant: generate -> trace -> shade (+ compaction) -> ... -> reduce
// ---------------------------------------------------------------------------
// The same paths as render_kernel (same Philox counters, same arithmetic), cut into one
// kernel per stage with the path state in global memory between them.  What it buys is
// full waves in every stage: the trace stage refills idle lanes from the list of rays, the
// shade stage runs one thread per path and writes the surviving paths compacted.  What it
// costs is the traffic of the path state (about 180 bytes per ray) and of the per-path
// contributions.  Built so that the two designs can be measured against each other.
template <typename ArgsT>
__device__ __forceinline__ void wf_pixel_of(const ArgsT& A, uint32_t item, uint32_t& pixel, uint32_t& sample) {
  const uint32_t s = item / A.WH;
  pixel = item - s * A.WH;
  sample = (uint32_t)A.sample_first + s;
}

// Camera paths for items [item0, item0 + n_new) into slots [slot0, ...) of `cur`.
__global__ void __launch_bounds__(256) wf_generate_kernel(WfArgs A, CamDev<float> cam) {
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= A.n_new) return;
  const uint32_t item = A.item0 + g, slot = A.slot0 + g;
  uint32_t pixel, sample;
  wf_pixel_of(A, item, pixel, sample);
  const uint32_t py = pixel / (uint32_t)A.W, px = pixel - py * (uint32_t)A.W;
  uint32_t rnd[4];
  philox4x32_10_keyed(pixel, sample, 0u, 0u, A.philox_keys, rnd);
  Ray<float> ray;
  camera_ray<float, WfArgs>(A, &cam, ray, px, py, rnd, event_trig<float>(u01<float>(rnd[3]), u01<float>(rnd[2])), A.inv_w1,
                            A.inv_h1);
  const uint32_t cap = A.cap;
  A.cur.ray[slot] = ray.o.x; A.cur.ray[cap + slot] = ray.o.y; A.cur.ray[2u * cap + slot] = ray.o.z;
  A.cur.ray[3u * cap + slot] = ray.d.x; A.cur.ray[4u * cap + slot] = ray.d.y; A.cur.ray[5u * cap + slot] = ray.d.z;
  A.cur.thr[slot] = 1.0f; A.cur.thr[cap + slot] = 1.0f; A.cur.thr[2u * cap + slot] = 1.0f;
  A.cur.ids[slot] = item;
  A.cur.ids[cap + slot] = 1u;  // bounce counter after the camera event, as in render_kernel
}

// Closest hit of rays [0, n) of `cur`.  Persistent waves: a lane whose ray is finished
// keeps its result until the wave refills, which it does as soon as REFILL lanes are idle
// (one atomic per wave on the ray counter); the traversal is render_kernel's.
template <bool WIDE, bool TRIS>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
wf_trace_kernel(WfArgs A) {
  extern __shared__ __align__(16) uint8_t lds[];
  typedef RefT<WIDE> RF;
  const uint32_t BT = blockDim.x;
  Scene sc;
  sc.L = A.L;
  uint32_t lds_image = 0;
  if constexpr (WIDE) {
    sc.base = A.blob;
  } else {
    copy_scene_to_lds(lds, A.blob, A.L.total_bytes, BT);
    sc.base = lds;
    lds_image = A.L.total_bytes;
  }
  const uint32_t stack_level0 = lds_address(lds) + lds_image + threadIdx.x * 2u;
  const uint32_t stack_level1 = stack_level0 + BT * RF::STACK_ENTRY;
  stack_store<WIDE>(lds, stack_level0, RF::EMPTY);
  const uint32_t lane = threadIdx.x & 63u;
  const float t_min = 0.001f;
  const int REFILL = A.refill;  // idle lanes that trigger a refill
  Cnt cnt;
  Ray<float> ray;
  ray.o = ray.d = ray.inv = mk<float>(0.f, 0.f, 0.f);
  ray.oi = mk<float>(0.f, 0.f, 0.f);
  ray.a = ray.inv_a = ray.oo = 0.f;
  Trav<float> tv;
  trav_begin<float, WIDE>(tv, stack_level1);
  tv.cur = RF::EMPTY;
  uint32_t my_ray = 0xFFFFFFFFu;
  bool list_empty = false;
  const uint32_t cap = A.cap, n = A.n;
  // The wave's block of the ray list, [blk_next, blk_end): one atomic on the shared counter
  // reserves a block (a single word takes about 90 atomics per microsecond, far fewer than
  // the refills of six thousand waves), and the refills hand it out 64 rays or fewer at a time.
  uint32_t blk_next = 0, blk_end = 0;
  const uint32_t n_waves = gridDim.x * (BT >> 6);
  for (;;) {
    const uint64_t idle = __ballot(tv.cur == RF::EMPTY);
    const int n_idle = __popcll(idle);
    if (n_idle >= REFILL && !(list_empty && n_idle < 64)) {
      // results of the finished rays
      if (tv.cur == RF::EMPTY && my_ray != 0xFFFFFFFFu) {
        A.hit_t[my_ray] = tv.best_t;
        A.hit_ref[my_ray] = tv.best_ref;
        my_ray = 0xFFFFFFFFu;
      }
      if (list_empty) break;  // every lane idle and nothing left to take
      if (blk_next >= blk_end) {
        // an even share of the list in eight blocks per wave, 64..4096 rays
        uint32_t want = min(4096u, max(64u, (n / n_waves / 8u + 63u) & ~63u));
        if (lane == 0) blk_next = atomicAdd(A.counters + 1, want);
        blk_next = __builtin_amdgcn_readfirstlane(blk_next);
        want = __builtin_amdgcn_readfirstlane(want);
        blk_end = min(n, blk_next + want);
        if (blk_next >= n) {
          list_empty = true;
          blk_next = blk_end = 0;
          continue;
        }
      }
      if (tv.cur == RF::EMPTY) {
        const uint32_t idx = blk_next + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        if (idx < blk_end) {
          ray.o = mk<float>(A.cur.ray[idx], A.cur.ray[cap + idx], A.cur.ray[2u * cap + idx]);
          ray.d = mk<float>(A.cur.ray[3u * cap + idx], A.cur.ray[4u * cap + idx], A.cur.ray[5u * cap + idx]);
          ray_setup<float>(ray);
          ray_start<float, false, WIDE>(sc, ray, tv, stack_level1, t_min, cnt);
          my_ray = idx;
        }
      }
      blk_next = min(blk_end, blk_next + (uint32_t)n_idle);
      continue;
    }
    // node steps while at least half of the travelling lanes are at inner nodes, then one
    // leaf step, as in render_kernel
    for (;;) {
      const bool inner = tv.cur < RF::LEAF;
      const uint64_t mi = __ballot(inner);
      const uint64_t mt = __ballot(tv.cur != RF::EMPTY);
      if (mi == 0ull || 2 * __popcll(mi) < __popcll(mt)) break;
      if (inner) node_step<float, false, WIDE>(sc, ray, tv, lds, t_min, cnt);
      if (tv.cur < RF::LEAF) node_step<float, false, WIDE>(sc, ray, tv, lds, t_min, cnt);
    }
    const bool leaf = (uint32_t)(tv.cur - RF::LEAF) < (RF::EMPTY - RF::LEAF);
    if (leaf) {
      leaf_test<float, false, WIDE, TRIS>(sc, ray, tv, tv.cur, t_min, cnt);
      tv.cur = trav_pop<float, WIDE>(tv, lds);
    }
  }
}

// One thread per path of `cur`: deposit or scatter (Camera.scala:29-34, Material.scala:9-55);
// the paths that go on are written to `nxt` without gaps (one atomic per wave).
template <uint32_t FEAT, bool WIDE>
__global__ void __launch_bounds__(1024) wf_shade_kernel(WfArgs A, CamDev<float> cam) {
  typedef RefT<WIDE> RF;
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t cap = A.cap;
  Scene sc;
  sc.L = A.L;
  sc.base = A.blob;  // shading touches one primitive, one material and its textures: read in place
  bool go = false;
  Ray<float> ray;
  V3<float> T = mk<float>(1.f, 1.f, 1.f), dir = mk<float>(0.f, 0.f, 0.f);
  Surface<float> sf;
  sf.p = sf.n = mk<float>(0.f, 0.f, 0.f);
  sf.param = 0.f; sf.mkind = 0u; sf.front = false;
  uint32_t item = 0, bounce = 0;
  if (i < A.n) {
    ray.o = mk<float>(A.cur.ray[i], A.cur.ray[cap + i], A.cur.ray[2u * cap + i]);
    ray.d = mk<float>(A.cur.ray[3u * cap + i], A.cur.ray[4u * cap + i], A.cur.ray[5u * cap + i]);
    ray_setup<float>(ray);
    T = mk<float>(A.cur.thr[i], A.cur.thr[cap + i], A.cur.thr[2u * cap + i]);
    item = A.cur.ids[i];
    bounce = A.cur.ids[cap + i];
    Trav<float> tv;
    tv.cur = RF::EMPTY;
    tv.sp = 0;
    tv.best_t = A.hit_t[i];
    tv.best_ref = A.hit_ref[i];
    StoreSink sink;
    sink.c = A.contrib + item;
    sink.stride = A.items;
    Cnt cnt;
    bool scatter = false;
    if (tv.best_ref == RF::EMPTY) {
      deposit<float>(sink, T * mk<float>(cam.background[0], cam.background[1], cam.background[2]));
    } else if (shade_surface<float, false, FEAT, WIDE>(sc, A.texels, ray, tv, T, sink, sf, cnt)) {
      scatter = bounce < (uint32_t)A.max_depth;
    }
    if (scatter) {
      uint32_t pixel, sample, rnd[4];
      wf_pixel_of(A, item, pixel, sample);
      philox4x32_10_keyed(pixel, sample, bounce, 0u, A.philox_keys, rnd);
      const float z = scatter_z<float>(u01<float>(rnd[1]));
      go = scatter_direction<float>(sf, ray, rnd, event_trig<float>(u01<float>(rnd[0]), 1.f - z * z), z, dir,
                                    sc.L.scatter_diffuse_only != 0u);
    }
  }
  // compaction: the waves' survivor counts meet in LDS and one lane of the workgroup takes
  // the block of `nxt` for all of them (one atomic per 1024 paths on the shared counter)
  __shared__ uint32_t wave_count[16], wave_base[16];
  const uint64_t alive = __ballot(go);
  const uint32_t wave = threadIdx.x >> 6;
  if (lane == 0) wave_count[wave] = (uint32_t)__popcll(alive);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) { wave_base[w] = total; total += wave_count[w]; }
    const uint32_t start = total ? atomicAdd(A.counters, total) : 0u;
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) wave_base[w] += start;
  }
  __syncthreads();
  const uint32_t base = wave_base[wave];
  if (go) {
    const uint32_t o = base + (uint32_t)__popcll(alive & ((1ull << lane) - 1ull));
    A.nxt.ray[o] = sf.p.x; A.nxt.ray[cap + o] = sf.p.y; A.nxt.ray[2u * cap + o] = sf.p.z;
    A.nxt.ray[3u * cap + o] = dir.x; A.nxt.ray[4u * cap + o] = dir.y; A.nxt.ray[5u * cap + o] = dir.z;
    A.nxt.thr[o] = T.x; A.nxt.thr[cap + o] = T.y; A.nxt.thr[2u * cap + o] = T.z;
    A.nxt.ids[o] = item;
    A.nxt.ids[cap + o] = bounce + 1u;
  }
}

// accum[pixel][c] += sum over the launch's samples of the stored contributions, converted to
// fixed point one by one exactly as render_kernel's deposits are.
__global__ void __launch_bounds__(256) wf_reduce_kernel(WfArgs A, uint32_t samples) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= A.WH) return;
#pragma unroll
  for (uint32_t c = 0; c < 3u; ++c) {
    const float* src = A.contrib + (size_t)c * A.items + p;
    long long sum = 0;
    for (uint32_t s = 0; s < samples; ++s) sum += to_fixed<float>(src[(size_t)s * A.WH]);
    A.accum[(size_t)p * 3u + c] += sum;
  }
}

// dst[i] += src[0][i] + src[1][i] + ... : one slice of the frame, `count` copies n words apart
__global__ void accum_add_many_kernel(long long* __restrict__ dst, const long long* __restrict__ src, size_t n,
                                      int count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long v = dst[i];
  for (int j = 0; j < count; ++j) v += src[(size_t)j * n + i];
  dst[i] = v;
}

// ---------------------------------------------------------------------------
// static scratch array (ht_render_noalloc)
// ---------------------------------------------------------------------------
// Device memory for callers that may not allocate: the radiance sums of one frame of up to
// HT_NOALLOC_MAX_PIXELS pixels and a ring of work-queue heads.  Every GPU the library runs
// on holds its own copy (48 MiB + 4 KiB) from the moment the code object is loaded until the
// process ends.  In host code the name `scratch` is only a handle for hipGetSymbolAddress:
// kernels, memsets and copies get the per-device address that static_scratch() returns.
constexpr size_t STATIC_SCRATCH_BYTES =
    (size_t)HT_NOALLOC_MAX_PIXELS * 3u * sizeof(long long) + (size_t)STATIC_QUEUE_SLOTS * 256u;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

hipError_t static_scratch(int device, void** out) {
  constexpr int MAX_DEVICES = 64;
  static std::mutex mu;
  static void* ptr[MAX_DEVICES];  // zero-initialised: not looked up yet
  *out = nullptr;
  if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!ptr[device]) {
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    if (cur != device) return hipErrorInvalidDevice;  // the caller makes `device` current first
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    ptr[device] = p;
  }
  *out = ptr[device];
  return hipSuccess;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
uint32_t lds_bytes_needed(const SceneLayout& L, uint32_t threads) {
  return image_bytes_in_lds(L) + stack_bytes(threads, L.depth, L.wide) + (threads / 64u) * TILE_ACC_BYTES;
}
uint32_t max_block_threads(int precision) { return precision == 0 ? M<float>::BT : M<double>::BT; }

template <typename K>
static hipError_t allow_big_lds(K kernel, uint32_t lds_bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

namespace {
// KFEAT_* bits -> column of the kernel tables
inline uint32_t feat_index(uint32_t feat) { return (feat & 3u) == 0u && (feat & FEAT_TRIS) ? 4u : (feat & 3u); }
typedef void (*RenderKernel32)(RenderArgs, CamDev<float>);
typedef void (*RenderKernel64)(RenderArgs, CamDev<double>);
// the fifth entry: plain textures and a triangle-only hierarchy (fp32 production variant only;
// elsewhere the generic kernel stands in)
#define HT_FEAT_ROW(R, CNT, WIDE)                                                              \
  {render_kernel<R, CNT, 0, WIDE>, render_kernel<R, CNT, 1, WIDE>, render_kernel<R, CNT, 2, WIDE>, \
   render_kernel<R, CNT, 3, WIDE>,                                                             \
   render_kernel<R, CNT, (sizeof(R) == 4 && !CNT) ? FEAT_TRIS : 0u, WIDE>}
// [store][counters][feat]
const RenderKernel32 k32[2][2][5] = {{HT_FEAT_ROW(float, false, false), HT_FEAT_ROW(float, true, false)},
                                     {HT_FEAT_ROW(float, false, true), HT_FEAT_ROW(float, true, true)}};
const RenderKernel64 k64[2][2][5] = {{HT_FEAT_ROW(double, false, false), HT_FEAT_ROW(double, true, false)},
                                     {HT_FEAT_ROW(double, false, true), HT_FEAT_ROW(double, true, true)}};
#undef HT_FEAT_ROW
// Which kernel a (precision, counters build, features, store) runs:
//   8: render_kernel_8, asked for with KFEAT_LEAN: fp32, on-chip store, plain textures
//   7: render_kernel_7: fp32, plain textures in either store, image textures on chip
//   0: the generic template (counters builds, fp64, Perlin: six waves per SIMD or fewer anyway)
int kernel_tier(int precision, bool counters, uint32_t feat, bool wide) {
  if (precision != 0 || counters) return 0;
  const uint32_t tex = feat & 3u;
  if (tex == 0u) return ((feat & KFEAT_LEAN) && !wide) ? 8 : 7;
  if (tex == 1u && (feat & FEAT_FLAT) && !(feat & FEAT_TRIS) && !wide) return 7;  // Perlin textures, no hierarchy
  return (tex == 2u && !wide) ? 7 : 0;
}
// the variants without a traversal phase: on-chip store (a scene without a hierarchy is small)
bool flat_variant(int precision, bool counters, uint32_t feat, bool wide) {
  return (feat & FEAT_FLAT) != 0u && (feat & FEAT_TRIS) == 0u && precision == 0 && !counters && !wide;
}
RenderKernel32 tier_kernel(int tier, uint32_t feat, bool wide) {
  const bool tris = (feat & FEAT_TRIS) != 0u;
  const bool flat = flat_variant(0, false, feat, wide);
  if (tier == 8) return flat ? render_kernel_8<FEAT_FLAT> : (tris ? render_kernel_8<FEAT_TRIS> : render_kernel_8<0>);
  if ((feat & 3u) == 2u) return flat ? render_kernel_7<2u | FEAT_FLAT, false> : render_kernel_7<2u, false>;
  if ((feat & 3u) == 1u) return render_kernel_7<1u | FEAT_FLAT, false>;  // kernel_tier() grants this to flat scenes only
  if (wide) return tris ? render_kernel_7<FEAT_TRIS, true> : render_kernel_7<0, true>;
  return tris ? render_kernel_7<FEAT_TRIS, false> : render_kernel_7<0, false>;
}
const void* render_kernel_ptr(int precision, bool counters, uint32_t feat, bool wide) {
  const int tier = kernel_tier(precision, counters, feat, wide);
  if (tier) return reinterpret_cast<const void*>(tier_kernel(tier, feat, wide));
  if (flat_variant(precision, counters, feat, wide) && (feat & 3u) == 3u)  // Perlin and image textures, no hierarchy
    return reinterpret_cast<const void*>((RenderKernel32)render_kernel<float, false, 3u | FEAT_FLAT, false>);
  return precision == 0 ? reinterpret_cast<const void*>(k32[wide ? 1 : 0][counters ? 1 : 0][feat_index(feat)])
                        : reinterpret_cast<const void*>(k64[wide ? 1 : 0][counters ? 1 : 0][feat_index(feat)]);
}
}  // namespace

// LDS a workgroup may ask for on gfx950; the kernels' dynamic-LDS limit is raised to it and
// never lowered, so threads that work out launch shapes for different scenes cannot undo
// each other's setting between the query and the launch.
constexpr uint32_t LDS_LIMIT_BYTES = 160u * 1024u;

int render_max_waves_per_cu(int precision, bool counters, uint32_t feat, bool wide) {
  const int tier = kernel_tier(precision, counters, feat, wide);
  return tier == 8 ? 32 : (tier == 7 ? 28 : 24);
}
// The chosen kernel's name as a profiler prints it (without namespace and arguments); returns
// the scalar-register budget it is compiled to (0: the generic template, not capped).
int render_kernel_name(int precision, bool counters, uint32_t feat, bool wide, char* out, size_t n) {
  const int tier = kernel_tier(precision, counters, feat, wide);
  const bool flat = flat_variant(precision, counters, feat, wide);
  unsigned f = (feat & 3u) == 0u && (feat & FEAT_TRIS) ? FEAT_TRIS : (feat & 3u);
  if (flat && (tier == 8 || (tier == 7 && f != 0u) || (tier == 0 && f == 3u))) f |= FEAT_FLAT;
  if (tier == 8) {
    snprintf(out, n, "render_kernel_8<%uu>", f);
    return 80;
  }
  if (tier == 7) {
    snprintf(out, n, "render_kernel_7<%uu, %s>", f, wide ? "true" : "false");
    return 96;
  }
  // the generic table has the triangle-only variant for the fp32 production build only
  const unsigned fg = (precision == 0 && !counters) ? f : (feat & 3u);
  snprintf(out, n, "render_kernel<%s, %s, %uu, %s>", precision == 0 ? "float" : "double", counters ? "true" : "false", fg,
           wide ? "true" : "false");
  return 0;
}
bool render_has_lean_variant(int precision, bool counters, uint32_t feat, bool wide) {
  return kernel_tier(precision, counters, feat | KFEAT_LEAN, wide) == 8;
}

int render_blocks_per_cu(int precision, bool counters, uint32_t feat, bool wide, int threads,
                         uint32_t lds_bytes) {
  const void* k = render_kernel_ptr(precision, counters, feat, wide);
  if (lds_bytes > LDS_LIMIT_BYTES) return 0;
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, threads, lds_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

hipError_t launch_render(int precision, bool counters, uint32_t feat, const RenderArgs& a, const void* cam,
                         int grid_blocks, int threads, uint32_t lds_bytes, hipStream_t stream) {
  const int w = a.L.wide ? 1 : 0;
  const int tier = kernel_tier(precision, counters, feat, a.L.wide != 0u);
  if (tier) {
    hipLaunchKernelGGL(tier_kernel(tier, feat, a.L.wide != 0u), dim3(grid_blocks), dim3(threads), lds_bytes, stream, a,
                       *static_cast<const CamDev<float>*>(cam));
  } else if (precision == 0) {
    hipLaunchKernelGGL(k32[w][counters ? 1 : 0][feat_index(feat)], dim3(grid_blocks), dim3(threads), lds_bytes, stream,
                       a, *static_cast<const CamDev<float>*>(cam));
  } else {
    hipLaunchKernelGGL(k64[w][counters ? 1 : 0][feat_index(feat)], dim3(grid_blocks), dim3(threads), lds_bytes, stream,
                       a, *static_cast<const CamDev<double>*>(cam));
  }
  return hipGetLastError();
}

template <typename R, bool WIDE>
static hipError_t launch_first_hit_t(const FirstHitArgs& a, const void* cam, uint32_t blocks, int threads,
                                     uint32_t lds_bytes, hipStream_t stream) {
  if (lds_bytes > LDS_LIMIT_BYTES) return hipErrorInvalidValue;
  hipError_t e = allow_big_lds(first_hit_kernel<R, WIDE>, LDS_LIMIT_BYTES);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((first_hit_kernel<R, WIDE>), dim3(blocks), dim3(threads), lds_bytes, stream, a,
                     *static_cast<const CamDev<R>*>(cam));
  return hipGetLastError();
}

hipError_t launch_first_hit(int precision, const FirstHitArgs& a, const void* cam, int threads,
                            uint32_t lds_bytes, hipStream_t stream) {
  const uint32_t tiles = (uint32_t)((a.W + TILE_W - 1) / TILE_W) * (uint32_t)((a.H + TILE_H - 1) / TILE_H);
  const uint32_t wpb = (uint32_t)threads / 64u;
  const uint32_t blocks = (tiles + wpb - 1) / wpb;
  if (precision == 0)
    return a.L.wide ? launch_first_hit_t<float, true>(a, cam, blocks, threads, lds_bytes, stream)
                    : launch_first_hit_t<float, false>(a, cam, blocks, threads, lds_bytes, stream);
  return a.L.wide ? launch_first_hit_t<double, true>(a, cam, blocks, threads, lds_bytes, stream)
                  : launch_first_hit_t<double, false>(a, cam, blocks, threads, lds_bytes, stream);
}


hipError_t launch_wf_generate(const WfArgs& a, const CamDev<float>& cam, hipStream_t stream) {
  if (a.n_new == 0) return hipSuccess;
  hipLaunchKernelGGL(wf_generate_kernel, dim3((a.n_new + 255u) / 256u), dim3(256), 0, stream, a, cam);
  return hipGetLastError();
}

template <bool WIDE, bool TRIS>
static hipError_t launch_wf_trace_t(const WfArgs& a, int grid_blocks, int threads, uint32_t lds_bytes, hipStream_t stream) {
  if (lds_bytes > LDS_LIMIT_BYTES) return hipErrorInvalidValue;
  hipError_t e = allow_big_lds(wf_trace_kernel<WIDE, TRIS>, LDS_LIMIT_BYTES);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((wf_trace_kernel<WIDE, TRIS>), dim3(grid_blocks), dim3(threads), lds_bytes, stream, a);
  return hipGetLastError();
}
hipError_t launch_wf_trace(uint32_t feat, const WfArgs& a, int grid_blocks, int threads, uint32_t lds_bytes,
                           hipStream_t stream) {
  if (a.n == 0) return hipSuccess;
  const bool tris = (feat & KFEAT_TRIS) != 0u;
  if (a.L.wide) return tris ? launch_wf_trace_t<true, true>(a, grid_blocks, threads, lds_bytes, stream)
                            : launch_wf_trace_t<true, false>(a, grid_blocks, threads, lds_bytes, stream);
  return tris ? launch_wf_trace_t<false, true>(a, grid_blocks, threads, lds_bytes, stream)
              : launch_wf_trace_t<false, false>(a, grid_blocks, threads, lds_bytes, stream);
}

template <bool WIDE>
static hipError_t launch_wf_shade_t(uint32_t feat, const WfArgs& a, const CamDev<float>& cam, hipStream_t stream) {
  const dim3 grid((a.n + 1023u) / 1024u), block(1024);
  switch (feat & 3u) {
    case 0: hipLaunchKernelGGL((wf_shade_kernel<0u, WIDE>), grid, block, 0, stream, a, cam); break;
    case 1: hipLaunchKernelGGL((wf_shade_kernel<1u, WIDE>), grid, block, 0, stream, a, cam); break;
    case 2: hipLaunchKernelGGL((wf_shade_kernel<2u, WIDE>), grid, block, 0, stream, a, cam); break;
    default: hipLaunchKernelGGL((wf_shade_kernel<3u, WIDE>), grid, block, 0, stream, a, cam); break;
  }
  return hipGetLastError();
}
hipError_t launch_wf_shade(uint32_t feat, const WfArgs& a, const CamDev<float>& cam, hipStream_t stream) {
  if (a.n == 0) return hipSuccess;
  return a.L.wide ? launch_wf_shade_t<true>(feat, a, cam, stream) : launch_wf_shade_t<false>(feat, a, cam, stream);
}

hipError_t launch_wf_reduce(const WfArgs& a, uint32_t samples, hipStream_t stream) {
  hipLaunchKernelGGL(wf_reduce_kernel, dim3((a.WH + 255u) / 256u), dim3(256), 0, stream, a, samples);
  return hipGetLastError();
}

hipError_t launch_accum_add(long long* dst, const long long* src, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(accum_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
  return hipGetLastError();
}

hipError_t launch_accum_add_many(long long* dst, const long long* src, size_t n, int count, hipStream_t stream) {
  if (n == 0 || count <= 0) return hipSuccess;
  hipLaunchKernelGGL(accum_add_many_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n,
                     count);
  return hipGetLastError();
}

hipError_t launch_resolve_tiles(const long long* accum, int32_t W, int32_t H, double inv_samples, uint32_t tile_first,
                                uint32_t tile_stride, float* mean_tiles, uint32_t* packed_tiles, hipStream_t stream) {
  const uint32_t n_owned = tiles_owned(tiles_in_frame(W, H), tile_first, tile_stride);
  if (n_owned == 0u) return hipSuccess;
  hipLaunchKernelGGL(resolve_tiles_kernel, dim3((n_owned * TILE_PIXELS + 255u) / 256u), dim3(256), 0, stream, accum, W, H,
                     inv_samples, tile_first, tile_stride, n_owned, mean_tiles, packed_tiles);
  return hipGetLastError();
}

hipError_t launch_assemble_tiles(int32_t W, int32_t H, uint32_t n_shards, size_t shard_pixels, const float* mean_tiles,
                                 const uint32_t* packed_tiles, float* rgb_mean, uint32_t* rgb_packed,
                                 hipStream_t stream) {
  const uint32_t n = (uint32_t)(W * H);
  hipLaunchKernelGGL(assemble_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, W, H, n_shards, shard_pixels,
                     mean_tiles, packed_tiles, rgb_mean, rgb_packed);
  return hipGetLastError();
}

hipError_t launch_resolve(const long long* accum, int32_t W, int32_t H, double inv_samples, float* rgb_mean,
                          uint32_t* rgb_packed, hipStream_t stream) {
  const uint32_t n = (uint32_t)(W * H);
  hipLaunchKernelGGL(resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, W, H,
                     inv_samples, rgb_mean, rgb_packed);
  return hipGetLastError();
}

}  // namespace ht
