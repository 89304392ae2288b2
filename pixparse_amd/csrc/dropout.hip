// Dropout for the BART decoder's hidden states (include/crl.h: crl_dropout / crl_dropout_add / crl_dropout_mask).
// Replaces nn.functional.dropout(hidden_states, p=self.dropout, training=self.training) of transformers' BartDecoder /
// BartDecoderLayer (modeling_bart.py:362,377,384-386,654), which is live in the reference only when the decoder is built with
// pretrained=False (from_config leaves train mode).  Opt-in: parity runs and bench.py keep it off.
//
// The mask is a pure function of (seed, step, site, element index): Philox4x32-10, one call per 8 elements (eight 16-bit uniforms;
// keep iff u16 >= p * 65536), so nothing is stored for the backward pass -- it regenerates the mask from the same triple.  All
// kernels are HBM-bound elementwise passes with 16-byte accesses.
#include "common.h"

namespace {

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
// keep flags (bit j = element 8 g + j) of element group g
__device__ __forceinline__ uint32_t keep8(uint64_t g, uint64_t seed, uint32_t step, uint32_t site, uint32_t thr) {
  const U4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), site, step, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    m |= ((w[j] & 0xffffu) >= thr ? 1u : 0u) << (2 * j);
    m |= ((w[j] >> 16) >= thr ? 1u : 0u) << (2 * j + 1);
  }
  return m;
}

struct DropArgs { uint64_t seed; uint32_t step, site, thr; float scale; };

// y = dropout(x) on bf16 (in place allowed)
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const u16* __restrict__ x, u16* __restrict__ y, uint64_t ngroups, DropArgs a) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const uint32_t m = keep8(g, a.seed, a.step, a.site, a.thr);
  const uint4 v = *reinterpret_cast<const uint4*>(x + g * 8);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = (m >> (2 * j)) & 1u ? bf2f(w[j] & 0xffff) * a.scale : 0.f;
    const float hi = (m >> (2 * j + 1)) & 1u ? bf2f(w[j] >> 16) * a.scale : 0.f;
    o[j] = pack_bf2(lo, hi);
  }
  *reinterpret_cast<uint4*>(y + g * 8) = uint4{o[0], o[1], o[2], o[3]};
}
// y = dropout(x) on fp32 (in place allowed), optional bf16 copy of the result
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, u16* __restrict__ yb, uint64_t ngroups, DropArgs a) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const uint32_t m = keep8(g, a.seed, a.step, a.site, a.thr);
  const float4 v0 = *reinterpret_cast<const float4*>(x + g * 8), v1 = *reinterpret_cast<const float4*>(x + g * 8 + 4);
  float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (m >> j) & 1u ? f[j] * a.scale : 0.f;
  *reinterpret_cast<float4*>(y + g * 8) = float4{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<float4*>(y + g * 8 + 4) = float4{f[4], f[5], f[6], f[7]};
  if (yb) *reinterpret_cast<uint4*>(yb + g * 8) = uint4{pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7])};
}
// out = resid + bf16(dropout(x)): the residual join behind a dropped bf16 branch
__global__ __launch_bounds__(256) void dropout_add_kernel(const u16* __restrict__ x, const float* __restrict__ resid, float* __restrict__ out, uint64_t ngroups,
                                                          DropArgs a) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const uint32_t m = keep8(g, a.seed, a.step, a.site, a.thr);
  const uint4 v = *reinterpret_cast<const uint4*>(x + g * 8);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const float4 r0 = *reinterpret_cast<const float4*>(resid + g * 8), r1 = *reinterpret_cast<const float4*>(resid + g * 8 + 4);
  const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  float f[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = r[2 * j] + ((m >> (2 * j)) & 1u ? round_bf(bf2f(w[j] & 0xffff) * a.scale) : 0.f);
    f[2 * j + 1] = r[2 * j + 1] + ((m >> (2 * j + 1)) & 1u ? round_bf(bf2f(w[j] >> 16) * a.scale) : 0.f);
  }
  *reinterpret_cast<float4*>(out + g * 8) = float4{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<float4*>(out + g * 8 + 4) = float4{f[4], f[5], f[6], f[7]};
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ keep, uint64_t ngroups, DropArgs a) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const uint32_t m = keep8(g, a.seed, a.step, a.site, a.thr);
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { lo |= ((m >> j) & 1u) << (8 * j); hi |= ((m >> (4 + j)) & 1u) << (8 * j); }
  *reinterpret_cast<uint2*>(keep + g * 8) = uint2{lo, hi};
}

// ---- drop-path (stochastic depth, timm DropPath on the two residual branches of a Swin block): ONE keep decision per sample
// scale[b] = keep(b) / (1 - p): the first 16-bit uniform of Philox(counter b; site, step; seed)
__global__ void droppath_scale_kernel(float* __restrict__ scale, int B, DropArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const U4 r = philox4x32_10((uint32_t)b, 0u, a.site, a.step, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  scale[b] = (r.x & 0xffffu) >= a.thr ? a.scale : 0.f;
}
// out(f32) = resid + float(bf16(x * scale[row / rows_per_sample]))      (8 elements per thread; C % 8 == 0)
__global__ __launch_bounds__(256) void rowscale_add_kernel(const u16* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ resid,
                                                           float* __restrict__ out, uint64_t ngroups, uint64_t groups_per_sample) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const float sc = scale[g / groups_per_sample];
  const uint4 v = *reinterpret_cast<const uint4*>(x + g * 8);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const float4 r0 = *reinterpret_cast<const float4*>(resid + g * 8), r1 = *reinterpret_cast<const float4*>(resid + g * 8 + 4);
  const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  float f[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = r[2 * j] + round_bf(bf2f(w[j] & 0xffff) * sc);
    f[2 * j + 1] = r[2 * j + 1] + round_bf(bf2f(w[j] >> 16) * sc);
  }
  *reinterpret_cast<float4*>(out + g * 8) = float4{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<float4*>(out + g * 8 + 4) = float4{f[4], f[5], f[6], f[7]};
}
// y(bf16) = bf16(x * scale[row / rows_per_sample])      (the gradient entering a dropped branch)
__global__ __launch_bounds__(256) void rowscale_bf16_kernel(const u16* __restrict__ x, const float* __restrict__ scale, u16* __restrict__ y, uint64_t ngroups,
                                                            uint64_t groups_per_sample) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const float sc = scale[g / groups_per_sample];
  const uint4 v = *reinterpret_cast<const uint4*>(x + g * 8);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack_bf2(bf2f(w[j] & 0xffff) * sc, bf2f(w[j] >> 16) * sc);
  *reinterpret_cast<uint4*>(y + g * 8) = uint4{o[0], o[1], o[2], o[3]};
}

int drop_args(const char* who, int64_t n, float p, uint64_t seed, uint32_t step, uint32_t site, DropArgs& a) {
  CRL_CHECK(n > 0 && (n % 8) == 0, "%s: element count %lld must be a positive multiple of 8", who, (long long)n);
  CRL_CHECK(p >= 0.f && p < 1.f, "%s: p = %g outside [0, 1)", who, p);
  a.seed = seed; a.step = step; a.site = site;
  a.thr = (uint32_t)(p * 65536.f + 0.5f);
  a.scale = 1.f / (1.f - p);
  return 0;
}
inline unsigned grid_for(int64_t n) { return (unsigned)((n / 8 + 255) / 256); }

}  // namespace

extern "C" int crl_dropout(const void* x, void* y, int64_t n, int is_f32, void* y_bf16, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream) {
  DropArgs a;
  if (drop_args("crl_dropout", n, p, seed, step, site, a)) return -1;
  CRL_CHECK(x && y && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "crl_dropout: null / unaligned pointer");
  CRL_CHECK(is_f32 || !y_bf16, "crl_dropout: the bf16 copy exists for fp32 inputs only");
  if (is_f32) dropout_f32_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>((const float*)x, (float*)y, (u16*)y_bf16, (uint64_t)n / 8, a);
  else dropout_bf16_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>((const u16*)x, (u16*)y, (uint64_t)n / 8, a);
  CRL_LAUNCH_CHECK("crl_dropout");
  return 0;
}

extern "C" int crl_dropout_add(const void* x_bf16, const float* resid, float* out, int64_t n, float p, uint64_t seed, uint32_t step, uint32_t site,
                               void* stream) {
  DropArgs a;
  if (drop_args("crl_dropout_add", n, p, seed, step, site, a)) return -1;
  CRL_CHECK(x_bf16 && resid && out, "crl_dropout_add: null pointer");
  dropout_add_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>((const u16*)x_bf16, resid, out, (uint64_t)n / 8, a);
  CRL_LAUNCH_CHECK("crl_dropout_add");
  return 0;
}

extern "C" int crl_dropout_mask(void* keep_u8, int64_t n, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream) {
  DropArgs a;
  if (drop_args("crl_dropout_mask", n, p, seed, step, site, a)) return -1;
  CRL_CHECK(keep_u8 && ((uintptr_t)keep_u8 % 8) == 0, "crl_dropout_mask: null / unaligned pointer");
  dropout_mask_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>((uint8_t*)keep_u8, (uint64_t)n / 8, a);
  CRL_LAUNCH_CHECK("crl_dropout_mask");
  return 0;
}

extern "C" int crl_droppath_scale(float* scale, int B, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream) {
  DropArgs a;
  if (drop_args("crl_droppath_scale", 8, p, seed, step, site, a)) return -1;
  CRL_CHECK(scale && B > 0, "crl_droppath_scale: bad arguments");
  droppath_scale_kernel<<<(unsigned)((B + 63) / 64), 64, 0, as_stream(stream)>>>(scale, B, a);
  CRL_LAUNCH_CHECK("crl_droppath_scale");
  return 0;
}

extern "C" int crl_rowscale_add(const void* x_bf16, const float* scale, const float* resid, float* out, int64_t rows, int64_t rows_per_sample, int64_t C,
                                void* stream) {
  CRL_CHECK(x_bf16 && scale && resid && out && rows > 0 && rows_per_sample > 0 && C > 0 && (C % 8) == 0 && (rows % rows_per_sample) == 0,
            "crl_rowscale_add: bad arguments (C must be a multiple of 8, rows a multiple of rows_per_sample)");
  const int64_t n = rows * C;
  rowscale_add_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>((const u16*)x_bf16, scale, resid, out, (uint64_t)n / 8, (uint64_t)(rows_per_sample * C / 8));
  CRL_LAUNCH_CHECK("crl_rowscale_add");
  return 0;
}

extern "C" int crl_rowscale_bf16(const void* x_bf16, const float* scale, void* y_bf16, int64_t rows, int64_t rows_per_sample, int64_t C, void* stream) {
  CRL_CHECK(x_bf16 && scale && y_bf16 && rows > 0 && rows_per_sample > 0 && C > 0 && (C % 8) == 0 && (rows % rows_per_sample) == 0,
            "crl_rowscale_bf16: bad arguments (C must be a multiple of 8, rows a multiple of rows_per_sample)");
  const int64_t n = rows * C;
  rowscale_bf16_kernel<<<grid_for(n), 256, 0, as_stream(stream)>>>((const u16*)x_bf16, scale, (u16*)y_bf16, (uint64_t)n / 8, (uint64_t)(rows_per_sample * C / 8));
  CRL_LAUNCH_CHECK("crl_rowscale_bf16");
  return 0;
}
