// Issue-rate probe for gfx950: how many shader cycles one wave64 instruction of each
// kind occupies its SIMD for, with all lanes and with part of the exec mask.
// One workgroup per CU; WPS waves per SIMD; every wave runs ITER x 64 instructions of
// one kind on eight independent register chains; wave 0 of block 0 reports s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define CHK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { K_ADD, K_FMA, K_PKADD, K_PKFMA, K_MAD64, K_MULLO, K_MIN3, K_CNDMASK, K_CMP, K_RCP, K_SQRT, K_SALU, K_ADD_SALU, K_DSREAD, K_CVT, K_MUL, K_MAX, K_AND, K_CND64, K_CMPCND, K_MIN, K_SUB, K_LSHLADD, K_MED3, K_FMAC, K_FMA_ABC, K_FMA_ACC, K_NKINDS };
static const char* names[] = {"v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_mad_u64_u32", "v_mul_lo_u32", "v_min3_f32", "v_cndmask_b32", "v_cmp_lt_f32", "v_rcp_f32", "v_sqrt_f32", "s_add_u32", "v_add+s_add pairs", "ds_read_b128", "v_cvt_f32_f16", "v_mul_f32", "v_max_f32", "v_and_b32", "v_cndmask_e64 sgpr", "v_cmp+v_cndmask", "v_min_f32", "v_sub_f32", "v_lshl_add_u32", "v_med3_f32", "v_fmac_f32 a,b,c", "v_fma_f32 a,a,b,c", "v_fma_f32 a,b,c,a"};

template <int K>
__global__ void __launch_bounds__(1024) probe(uint64_t* out, int iters, uint64_t mask, float seed) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = seed;
  lds[threadIdx.x + 1024] = seed;
  lds[threadIdx.x + 2048] = seed;
  lds[threadIdx.x + 3072] = seed;
  __syncthreads();
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = 1.0f, b1 = 1.5f, b2 = 0.5f, b3 = 2.5f, b4 = 1.0f, b5 = 1.5f, b6 = 0.5f, b7 = 2.5f;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  const uint32_t addr = (threadIdx.x & 63u) * 16u;
  const bool on = (mask >> (threadIdx.x & 63u)) & 1ull;
  uint64_t t0 = 0, t1 = 0, r0 = 0;
  if (on) {  // the exec mask under test
    r0 = __builtin_amdgcn_s_memrealtime();
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      if (K == K_ADD) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_MUL) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_MAX) { REP8(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_AND) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_CND64) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "s20", "s21");) }
      if (K == K_CMPCND) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");) }
      if (K == K_MIN) { REP8(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_SUB) { REP8(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_LSHLADD) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_MED3) { REP8(asm volatile("v_med3_f32 %0, %0, %8, %1\n v_med3_f32 %1, %1, %8, %2\n v_med3_f32 %2, %2, %8, %3\n v_med3_f32 %3, %3, %8, %4\n v_med3_f32 %4, %4, %8, %5\n v_med3_f32 %5, %5, %8, %6\n v_med3_f32 %6, %6, %8, %7\n v_med3_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_FMA) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_FMAC) { REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
      if (K == K_FMA_ABC) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
      if (K == K_FMA_ACC) { REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) }
      if (K == K_MIN3) { REP8(asm volatile("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_CNDMASK) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");) }
      if (K == K_CMP) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");) }
      if (K == K_RCP) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_SQRT) { REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_CVT) { REP8(asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n v_cvt_f32_f16 %4, %4\n v_cvt_f32_f16 %5, %5\n v_cvt_f32_f16 %6, %6\n v_cvt_f32_f16 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_MULLO) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));) }
      if (K == K_SALU) { REP8(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");) }
      if (K == K_ADD_SALU) { REP8(asm volatile("v_add_f32 %0, %0, %8\n s_add_u32 %4, %4, 1\n v_add_f32 %1, %1, %8\n s_add_u32 %5, %5, 1\n v_add_f32 %2, %2, %8\n s_add_u32 %6, %6, 1\n v_add_f32 %3, %3, %8\n s_add_u32 %7, %7, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b0) : "scc");) }
      if (K == K_PKADD || K == K_PKFMA || K == K_MAD64) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q0 = {b0, b1}, q1 = {b2, b3}, q2 = {b4, b5}, q3 = {b6, b7}, c = {b0, b1};
        if (K == K_PKADD) { REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(c));) }
        if (K == K_PKFMA) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(c));) }
        if (K == K_MAD64) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, 0\n v_mad_u64_u32 %1, vcc, %8, %9, 0\n v_mad_u64_u32 %2, vcc, %8, %9, 0\n v_mad_u64_u32 %3, vcc, %8, %9, 0\n v_mad_u64_u32 %4, vcc, %8, %9, 0\n v_mad_u64_u32 %5, vcc, %8, %9, 0\n v_mad_u64_u32 %6, vcc, %8, %9, 0\n v_mad_u64_u32 %7, vcc, %8, %9, 0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b0), "v"(b1) : "vcc");) }
        a0 = p0.x + p1.x + p2.x + p3.x + q0.x + q1.y + q2.x + q3.y;
      }
      if (K == K_DSREAD) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 r0, r1, r2, r3;
        REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");)
        a0 += r0.x + r1.y + r2.z + r3.w;
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  float sum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
  if (sum == 123.456f) out[3] = 1;  // keep the chains alive
  if (blockIdx.x == 0 && threadIdx.x == (uint32_t)__builtin_ctzll(mask)) {
    out[0] = t1 - t0;
    out[1] = __builtin_amdgcn_s_memrealtime() - r0;  // 100 MHz ticks over the same span
  }
}

template <int K>
static void run_kind(uint64_t* d_out, int wps, uint64_t mask, const char* mask_name, int blocks_per_cu = 1) {
  const int iters = 2000, blocks = 256 * blocks_per_cu, threads = 256 * wps;
  uint64_t h = 0, hv[2] = {0, 0};
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  probe<K><<<blocks, threads>>>(d_out, 10, mask, 1.0f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  probe<K><<<blocks, threads>>>(d_out, iters, mask, 1.0f);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  CHK(hipMemcpy(hv, d_out, 16, hipMemcpyDeviceToHost));
  h = hv[0];
  const double ghz = hv[1] ? (double)hv[0] / (double)hv[1] * 0.1 : 0.0;  // shader clock during the probe
  const double n = (double)iters * 64.0;
  // wave-instructions per SIMD per nanosecond, from the wall clock of the launch
  const double rate = n * wps * blocks_per_cu / (ms * 1e6);
  printf("%-20s waves/SIMD=%-2d lanes=%-12s memtime ticks/instr/wave=%7.2f  launch %.3f ms -> %.3f wave-instr/ns/SIMD"
         "  clock %.2f GHz -> %.2f cycles/instr/SIMD\n", names[K],
         wps * blocks_per_cu, mask_name, (double)h / n, ms, rate, ghz, ghz / rate);
  fflush(stdout);
  CHK(hipEventDestroy(e0));
  CHK(hipEventDestroy(e1));
}
template <int K>
static void run_all(uint64_t* d_out) {
  run_kind<K>(d_out, 1, ~0ull, "64");
  run_kind<K>(d_out, 2, ~0ull, "64");
  run_kind<K>(d_out, 4, ~0ull, "64");
  run_kind<K>(d_out, 4, ~0ull, "64", 2);
  run_kind<K>(d_out, 4, 0xFFFFull, "16 (one row)");
  run_kind<K>(d_out, 4, 0x0001000100010001ull, "1 per row");
  run_kind<K>(d_out, 4, 0xFFFFFFFFull, "32 (low)");
}
template <int K>
static void mask_scan(uint64_t* d_out) {
  for (int rep = 0; rep < 2; ++rep) {
    run_kind<K>(d_out, 4, ~0ull, "64", 2);
    run_kind<K>(d_out, 4, 0x0001000100010001ull, "1 per row", 2);
    run_kind<K>(d_out, 4, 0x5555555555555555ull, "32 alternate", 2);
    run_kind<K>(d_out, 4, 0x00000000FFFFFFFFull, "32 low", 2);
    run_kind<K>(d_out, 4, 0x9A3C51E6B04D27F1ull, "31 scattered", 2);
    run_kind<K>(d_out, 4, 0x000000000000FFFFull, "16 one row", 2);
    run_kind<K>(d_out, 4, 0x000F000F000F000Full, "4 per row", 2);
    run_kind<K>(d_out, 4, 0x000000000000000Full, "4 in a row", 2);
    run_kind<K>(d_out, 4, 0x0000000000000001ull, "1", 2);
  }
}
int main(int argc, char** argv) {
  uint64_t* d_out;
  CHK(hipMalloc(&d_out, 64));
  CHK(hipMemset(d_out, 0, 64));
  if (argc > 1 && argv[1][0] == 's') {  // selects and the other short ops
    for (int rep = 0; rep < 2; ++rep) {
      run_kind<K_CNDMASK>(d_out, 4, ~0ull, "64", 2); run_kind<K_CND64>(d_out, 4, ~0ull, "64", 2); run_kind<K_CMPCND>(d_out, 4, ~0ull, "64", 2);
      run_kind<K_CMP>(d_out, 4, ~0ull, "64", 2); run_kind<K_MIN>(d_out, 4, ~0ull, "64", 2); run_kind<K_MAX>(d_out, 4, ~0ull, "64", 2);
      run_kind<K_SUB>(d_out, 4, ~0ull, "64", 2); run_kind<K_ADD>(d_out, 4, ~0ull, "64", 2); run_kind<K_LSHLADD>(d_out, 4, ~0ull, "64", 2);
      run_kind<K_MED3>(d_out, 4, ~0ull, "64", 2); run_kind<K_MIN3>(d_out, 4, ~0ull, "64", 2); run_kind<K_FMA>(d_out, 4, ~0ull, "64", 2);
      run_kind<K_MUL>(d_out, 4, ~0ull, "64", 2); run_kind<K_PKADD>(d_out, 4, ~0ull, "64", 2); run_kind<K_SALU>(d_out, 4, ~0ull, "64", 2);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'f') {  // the forms of the fp32 multiply-add at 1, 2, 4, 8 waves per SIMD
    for (int w = 1; w <= 8; w *= 2) {
      const int wps = w > 4 ? 4 : w, bpc = w > 4 ? 2 : 1;
      run_kind<K_ADD>(d_out, wps, ~0ull, "64", bpc); run_kind<K_FMAC>(d_out, wps, ~0ull, "64", bpc);
      run_kind<K_FMA_ACC>(d_out, wps, ~0ull, "64", bpc); run_kind<K_FMA_ABC>(d_out, wps, ~0ull, "64", bpc);
      run_kind<K_FMA>(d_out, wps, ~0ull, "64", bpc); run_kind<K_MAX>(d_out, wps, ~0ull, "64", bpc);
      run_kind<K_CNDMASK>(d_out, wps, ~0ull, "64", bpc);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'k') {  // how many active lanes it takes to leave the slow mode
    static char nm[64][16];
    for (int k = 1; k <= 20; ++k) { snprintf(nm[k], 16, "%d low", k); run_kind<K_FMA>(d_out, 4, (1ull << k) - 1ull, nm[k], 2); }
    for (int k = 1; k <= 8; ++k) { snprintf(nm[20 + k], 16, "%d spread", k); uint64_t m = 0; for (int j = 0; j < k; ++j) m |= 1ull << (j * 8); run_kind<K_MAX>(d_out, 4, m, nm[20 + k], 2); }
    return 0;
  }
  if (argc > 1) {  // exec-mask scan
    mask_scan<K_FMA>(d_out); mask_scan<K_MUL>(d_out); mask_scan<K_ADD>(d_out); mask_scan<K_MAX>(d_out);
    mask_scan<K_MIN3>(d_out); mask_scan<K_PKADD>(d_out); mask_scan<K_AND>(d_out); mask_scan<K_CNDMASK>(d_out);
    return 0;
  }
  run_all<K_MUL>(d_out); run_all<K_MAX>(d_out); run_all<K_AND>(d_out);
  run_all<K_ADD>(d_out); run_all<K_FMA>(d_out); run_all<K_PKADD>(d_out); run_all<K_PKFMA>(d_out);
  run_all<K_MAD64>(d_out); run_all<K_MULLO>(d_out); run_all<K_MIN3>(d_out); run_all<K_CNDMASK>(d_out);
  run_all<K_CMP>(d_out); run_all<K_RCP>(d_out); run_all<K_SQRT>(d_out); run_all<K_CVT>(d_out);
  run_all<K_SALU>(d_out); run_all<K_ADD_SALU>(d_out); run_all<K_DSREAD>(d_out);
  CHK(hipFree(d_out));
  return 0;
}
