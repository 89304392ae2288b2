// Shared device helpers for the gfx950 (CDNA4) kernels of libcruller_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/crl.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

extern "C" void crl_set_error(const char* fmt, ...);
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: set once per (kernel, device), under a mutex (capi.cpp).
// Returns 0, or -2 with the error text set.
int crl_enable_lds(const void* kernel, int bytes, const char* who);

#define CRL_CHECK(cond, ...)                 \
  do {                                       \
    if (!(cond)) {                           \
      crl_set_error(__VA_ARGS__);            \
      return -1;                             \
    }                                        \
  } while (0)

#define CRL_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      crl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return -2;                                                            \
    }                                                                       \
  } while (0)

// ---- bf16 <-> f32 -------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(u16 x) { return __uint_as_float(((uint32_t)x) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
__device__ __forceinline__ u16 f2bf(float x) {
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ float round_bf(float x) { return bf2f(f2bf(x)); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// erf-GELU and its derivative in fp32.  erf by Abramowitz-Stegun 7.1.26 (|abs error| < 1.5e-7, far below the bf16
// rounding of the result) with hardware rcp / exp2: ~14 VALU issues per element instead of libm erff's ~40 -- the
// GEMM epilogues run 128 of these per lane with no second workgroup on the CU to hide them behind.
__device__ __forceinline__ void erf_parts(float x, float& erf_v, float& gauss) {  // erf(x/sqrt2), exp(-x^2/2)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  gauss = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);   // exp(-x^2/2) = 2^(-x^2 * log2(e)/2)
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * gauss;
  erf_v = copysignf(e, x);
}
__device__ __forceinline__ float gelu_f(float x) {
  float e, g;
  erf_parts(x, e, g);
  return 0.5f * x * (1.0f + e);
}
// Two elements at a time on the packed-fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): the GEMM epilogues run 128 of
// these per lane with the matrix pipe idle, and the s_memtime timeline of the persistent kernel shows the GELU / dGELU epilogues
// bound by VALU issue (about 124 clocks per element in the scalar form: ~21 simple ops + rcp + exp2).  Same formulas as above
// (the two constants of z and of the rcp argument are folded into one).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void erf_parts2(f32x2 x, f32x2& erf_v, f32x2& gauss) {
  f32x2 ax; ax.x = fabsf(x.x); ax.y = fabsf(x.y);
  const f32x2 den = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
  f32x2 t; t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  const f32x2 ge = (x * x) * -0.72134752044448170f;
  gauss.x = __builtin_amdgcn_exp2f(ge.x); gauss.y = __builtin_amdgcn_exp2f(ge.y);
  f32x2 p = t * 1.061405429f + -1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t + -0.284496736f;
  p = p * t + 0.254829592f;
  const f32x2 e = 1.0f - (p * t) * gauss;
  erf_v.x = copysignf(e.x, x.x); erf_v.y = copysignf(e.y, x.y);
}
// gelu(x) AND gelu'(x) = Phi(x) + x phi(x) from one erf / gauss evaluation (round 6: the forward epilogue saves the derivative, the
// backward epilogue only multiplies -- include/crl.h CRL_EPI_BF16_GELU / _DGELU): two packed operations more than gelu alone
__device__ __forceinline__ void gelu_grad2(f32x2 x, f32x2& y, f32x2& d) {
  f32x2 e, g;
  erf_parts2(x, e, g);
  const f32x2 cdf = (e + 1.0f) * 0.5f;       // (scaling by 0.5 is exact: x * cdf == (x * 0.5) * (e + 1) bit for bit)
  y = x * cdf;
  d = (x * 0.39894228040143268f) * g + cdf;
}
__device__ __forceinline__ void gelu_grad_f(float x, float& y, float& d) {
  float e, g;
  erf_parts(x, e, g);
  const float cdf = 0.5f * (1.0f + e);
  y = x * cdf;
  d = __builtin_fmaf(x * 0.39894228040143268f, g, cdf);
}
// the saved derivative travels as IEEE fp16 (round to nearest even): gelu' lies in [-0.13, 1.13], so fp16's 11 significant bits give it a
// relative rounding of 2^-11, a quarter of the bf16 rounding the product dy * gelu'(h) gets anyway (a bf16 copy would double that
// tensor's rounding noise); below 6e-5 fp16 goes subnormal: absolute error <= 3e-8
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) { const f16x2 r = {(_Float16)lo, (_Float16)hi}; return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ uint32_t pack_h2v(f32x2 v) { return pack_h2(v.x, v.y); }
__device__ __forceinline__ f32x2 unpack_h2(uint32_t w) { const f16x2 r = __builtin_bit_cast(f16x2, w); return f32x2{(float)r.x, (float)r.y}; }
__device__ __forceinline__ float h2f_lo(uint32_t w) { return unpack_h2(w).x; }
__device__ __forceinline__ float h2f_hi(uint32_t w) { return unpack_h2(w).y; }
// bf16 pair <-> two floats: one v_cvt_pk_bf16_f32 / one shift + one mask
__device__ __forceinline__ uint32_t pack_bf2v(f32x2 v) { return pack_bf2(v.x, v.y); }
__device__ __forceinline__ f32x2 unpack_bf2(uint32_t w) { f32x2 r; r.x = __uint_as_float(w << 16); r.y = __uint_as_float(w & 0xffff0000u); return r; }

// ---- wave64 reductions ----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- buffer resource (raw, bounds-checked: out-of-range 16-B loads return zeros) --------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// ---- LDS-DMA issued from inline asm.  hipcc does not see these loads, so it neither drains them with a
// conservative `s_waitcnt vmcnt(0)` in front of later LDS reads (it does that for the builtin form whenever the
// read is a ds_read_b64_tr_b16) nor counts them: every wait on them is an explicit, counted s_waitcnt in the kernel.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 make_srd(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  u32x4 r;
  r[0] = (uint32_t)a; r[1] = (uint32_t)(a >> 32) & 0xffffu; r[2] = bytes; r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// 16 B per lane: LDS[lds_base + 16*lane] <- buffer[voff + soff]   (zeros when out of range)
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t lds_base, uint32_t voff, uint32_t soff) {
#ifndef DMA_NOP
#define DMA_NOP "s_nop 4\n\t"
#endif
  asm volatile(DMA_NOP "s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD a contiguous chunk of the logical tile order (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// live per-kernel timing hooks (capi.cpp); `work` = algorithmic FLOPs (or bytes) of the launch
extern int g_crl_prof_on;
extern "C" void crl_prof_mark(int id, int phase, void* stream, double work);
#define CRL_PROF_START(id, s, work) do { if (g_crl_prof_on) crl_prof_mark((id), 0, (void*)(s), (work)); } while (0)
#define CRL_PROF_STOP(id, s) do { if (g_crl_prof_on) crl_prof_mark((id), 1, (void*)(s), 0.0); } while (0)
