// Throughput class of single VALU opcodes on gfx950: 8 waves per SIMD, eight independent
// chains, wall clock of the launch -> wave-instructions per ns per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP8(x) x x x x x x x x
// OP(d, s) expands to one instruction text with destination/source chain register d and operand s
#define CHAIN8(FMT) asm volatile(FMT(0) "\n" FMT(1) "\n" FMT(2) "\n" FMT(3) "\n" FMT(4) "\n" FMT(5) "\n" FMT(6) "\n" FMT(7) \
  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
#define KERNEL(NAME, FMT) \
  __global__ void __launch_bounds__(1024) k_##NAME(float* out, int iters, float seed) { \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    float b0 = 1.0f + seed * 1e-9f, b1 = 3.0f; \
    for (int i = 0; i < iters; ++i) { REP8(CHAIN8(FMT)) } \
    float sum = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    if (sum == 123.456f) out[0] = 1; \
  }
#define F_mov(i) "v_mov_b32 %" #i ", %8"
#define F_xor(i) "v_xor_b32 %" #i ", %" #i ", %8"
#define F_or(i) "v_or_b32 %" #i ", %" #i ", %8"
#define F_addu(i) "v_add_u32 %" #i ", %" #i ", %8"
#define F_subu(i) "v_sub_u32 %" #i ", %" #i ", %8"
#define F_lshl(i) "v_lshlrev_b32 %" #i ", 1, %" #i
#define F_lshr(i) "v_lshrrev_b32 %" #i ", 1, %" #i
#define F_bfe(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5"
#define F_add3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9"
#define F_andor(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9"
#define F_fmac(i) "v_fmac_f32 %" #i ", %8, %9"
#define F_mul24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8"
#define F_mad24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
#define F_cvtfu(i) "v_cvt_f32_u32 %" #i ", %" #i
#define F_cvtuf(i) "v_cvt_u32_f32 %" #i ", %" #i
#define F_floor(i) "v_floor_f32 %" #i ", %" #i
#define F_fract(i) "v_fract_f32 %" #i ", %" #i
#define F_sin(i) "v_sin_f32 %" #i ", %" #i
#define F_rsq(i) "v_rsq_f32 %" #i ", %" #i
#define F_addf(i) "v_add_f32 %" #i ", %" #i ", %8"
#define F_mulf(i) "v_mul_f32 %" #i ", %" #i ", %8"
#define F_maxf(i) "v_max_f32 %" #i ", %" #i ", %8"
#define F_add64(i) "v_add_f32_e64 %" #i ", %" #i ", %8"
#define F_addneg(i) "v_add_f32_e64 %" #i ", %" #i ", -%8"
#define F_mulhi(i) "v_mul_hi_u32 %" #i ", %" #i ", %8"
#define F_perm(i) "v_perm_b32 %" #i ", %" #i ", %8, %9"
#define F_alignbit(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7"
#define F_ldexp(i) "v_ldexp_f32 %" #i ", %" #i ", 1"
#define F_cvtpk(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8"
#define F_cvtf16(i) "v_cvt_f32_f16 %" #i ", %" #i
#define F_cvtf16hi(i) "v_cvt_f32_f16_sdwa %" #i ", %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define F_fmamix(i) "v_fma_mix_f32 %" #i ", %8, %9, %" #i " op_sel_hi:[1,0,0]"
#define F_fmamixhi(i) "v_fma_mix_f32 %" #i ", %8, %9, %" #i " op_sel:[1,0,0] op_sel_hi:[1,0,0]"
#define F_fma3(i) "v_fma_f32 %" #i ", %8, %9, %" #i
#define F_med3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9"
#define F_mac16(i) "v_pk_mul_f32 %" #i ", %" #i ", %8"
KERNEL(mov, F_mov) KERNEL(xor_, F_xor) KERNEL(or_, F_or) KERNEL(addu, F_addu) KERNEL(subu, F_subu) KERNEL(lshl, F_lshl)
KERNEL(lshr, F_lshr) KERNEL(bfe, F_bfe) KERNEL(add3, F_add3) KERNEL(andor, F_andor) KERNEL(fmac, F_fmac) KERNEL(mul24, F_mul24)
KERNEL(mad24, F_mad24) KERNEL(cvtfu, F_cvtfu) KERNEL(cvtuf, F_cvtuf) KERNEL(floor_, F_floor) KERNEL(fract, F_fract) KERNEL(sin_, F_sin)
KERNEL(rsq, F_rsq) KERNEL(addf, F_addf) KERNEL(mulf, F_mulf) KERNEL(maxf, F_maxf) KERNEL(add64, F_add64) KERNEL(addneg, F_addneg)
KERNEL(cvtf16, F_cvtf16) KERNEL(cvtf16hi, F_cvtf16hi) KERNEL(fmamix, F_fmamix) KERNEL(fmamixhi, F_fmamixhi) KERNEL(fma3, F_fma3) KERNEL(med3, F_med3)
KERNEL(mulhi, F_mulhi) KERNEL(perm, F_perm) KERNEL(alignbit, F_alignbit) KERNEL(ldexp_, F_ldexp) KERNEL(cvtpk, F_cvtpk)

typedef void (*kfn)(float*, int, float);
static void run(const char* name, kfn k, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  k<<<512, 1024>>>(d, 10, 1.0f);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  k<<<512, 1024>>>(d, iters, 1.0f);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-22s %.3f ms  %.3f wave-instr/ns/SIMD\n", name, ms, iters * 64.0 * 8 / (ms * 1e6));
  fflush(stdout);
}
int main() {
  float* d;
  CHK(hipMalloc(&d, 64));
#define R(NAME) run(#NAME, k_##NAME, d);
  for (int rep = 0; rep < 2; ++rep) {
    R(addf) R(mulf) R(maxf) R(add64) R(addneg) R(mov) R(xor_) R(or_) R(addu) R(subu) R(lshl) R(lshr) R(bfe) R(add3) R(andor) R(fmac)
    R(mul24) R(mad24) R(cvtfu) R(cvtuf) R(floor_) R(fract) R(sin_) R(rsq) R(mulhi) R(perm) R(alignbit) R(ldexp_) R(cvtpk) R(cvtf16) R(cvtf16hi) R(fmamix) R(fmamixhi) R(fma3) R(med3)
  }
  return 0;
}
