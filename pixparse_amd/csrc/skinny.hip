// Skinny linear layer for decode-time projections (include/crl.h: crl_linear_skinny_bf16):
//   out[m, n] = epilogue( sum_k x[m, k] * W[n, k] + bias[n] ),   M <= 16 rows (one per sequence being generated).
// HBM-bound on W (every weight byte is read once, 2 FLOP per 2 bytes per row): no LDS staging, no tile reuse to win --
// a workgroup owns 16 output columns, its four waves split K and stream W straight from global memory as the A operand
// of v_mfma_f32_16x16x32_bf16 (lane -> column n0 + (lane & 15), 8 consecutive k at 8 (lane >> 4)); the <= 16 rows of x
// are the B operand (L2-resident, re-read per workgroup). The four partial 16x16 tiles are summed through LDS in wave
// order (deterministic) and wave 0 applies the same epilogues as the GEMM (bias rounded to bf16, fp32 accumulate,
// bf16 output / exact-erf GELU / fp32 residual add).
#include "common.h"

namespace {

struct SkArgs {
  const u16* x; const u16* W; const float* bias; void* out; const float* resid;
  int64_t ldx, ldw, ldo, ldr;
  int M, N, K;
  const int* out_row; int64_t out_row_stride;   // optional: out += *out_row * out_row_stride (KV-cache row of this step)
  // LN variant: x is LayerNorm(xf) computed by every workgroup for itself (<= 16 rows of K floats: 64 KiB of L2 reads against one
  // kernel launch + its dependent-launch gap); workgroup b also stores columns [16 b, 16 b + 16) of the fp32 result to h32
  const float* xf; int64_t ldxf; const float* gamma; const float* beta; float eps; float* h32; int64_t ldh;
};
constexpr int SK_PAD = 32;    // bf16 elements between the LDS rows of the normalised input: row stride = 64 B mod 256 B

// NW waves split K.  4 for the LM head (thousands of workgroups: throughput-bound); 16 for the decoder-layer projections, which have only
// 64-256 workgroups and are LATENCY-bound (round 3: 7-10 us per launch for 2-8 MB of weights): with 16 waves a wave owns 2 (K = 1024) to 8
// (K = 4096) k steps, so all of its loads are one batch in flight, and the LayerNorm variant normalises one row per wave.
template <int EPI, bool LN = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void linear_skinny_kernel(const SkArgs a) {
  __shared__ f32x4 red[NW - 1][64];
  extern __shared__ __attribute__((aligned(16))) u16 xs[];          // LN: [M][K + SK_PAD] normalised rows, bf16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (LN) {
    // the arithmetic of ln_fwd_kernel (rowops.hip), operation for operation: lane layout c = (i * 64 + lane) * 4, two-pass variance
    constexpr int NV = 8;                                           // K <= 2048
    for (int row = wave; row < a.M; row += NW) {
      const float* xr = a.xf + (int64_t)row * a.ldxf;
      float4 xv[NV];
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < a.K) { xv[i] = *reinterpret_cast<const float4*>(xr + c); sm += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w); }
      }
      const float mean = wave_sum(sm) / (float)a.K;
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < a.K) {
          const float d0 = xv[i].x - mean, d1 = xv[i].y - mean, d2 = xv[i].z - mean, d3 = xv[i].w - mean;
          qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      const float rstd = rsqrtf(wave_sum(qq) / (float)a.K + a.eps);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < a.K) {
          const float4 gm = *reinterpret_cast<const float4*>(a.gamma + c);
          const float4 bt = *reinterpret_cast<const float4*>(a.beta + c);
          float4 o;
          o.x = (xv[i].x - mean) * rstd * gm.x + bt.x; o.y = (xv[i].y - mean) * rstd * gm.y + bt.y;
          o.z = (xv[i].z - mean) * rstd * gm.z + bt.z; o.w = (xv[i].w - mean) * rstd * gm.w + bt.w;
          *reinterpret_cast<uint2*>(xs + (size_t)row * (a.K + SK_PAD) + c) = uint2{pack_bf2(o.x, o.y), pack_bf2(o.z, o.w)};
          if (a.h32 && (c >> 4) == (int)blockIdx.x) *reinterpret_cast<float4*>(a.h32 + (int64_t)row * a.ldh + c) = o;
        }
      }
    }
    __syncthreads();
  }
  const int i = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int ksteps = (a.K + 31) / 32;                 // 32-wide k steps, split over the NW waves
  const int per = (ksteps + NW - 1) / NW;
  const int s0 = wave * per, s1 = min(ksteps, s0 + per);
  const bool col_ok = n0 + i < a.N, row_ok = i < a.M;
  const u16* wp = a.W + (int64_t)min(n0 + i, a.N - 1) * a.ldw + 8 * kq;
  const u16* xp = LN ? xs + (size_t)min(i, a.M - 1) * (a.K + SK_PAD) + 8 * kq : a.x + (int64_t)min(i, a.M - 1) * a.ldx + 8 * kq;
  const bf16x8 zero = {};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll NW == 4 ? 4 : 8
  for (int s = s0; s < s1; ++s) {
    const int k = 32 * s + 8 * kq;
    const bool k_ok = k < a.K;                          // K % 8 == 0: a chunk is entirely in or out
    const bf16x8 wf = (col_ok && k_ok) ? *reinterpret_cast<const bf16x8*>(wp + 32 * s) : zero;
    const bf16x8 xf = (row_ok && k_ok) ? *reinterpret_cast<const bf16x8*>(xp + 32 * s) : zero;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc, 0, 0, 0);   // D[row = column n (A), col = row m (B)]
  }
  if (wave) red[wave - 1][lane] = acc;
  __syncthreads();
  if (wave) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w) {
    const f32x4 o = red[w][lane];
    acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
  }
  const int m = lane & 15, n = n0 + 4 * (lane >> 4);   // this lane: row m, columns n .. n+3
  if (m >= a.M || n >= a.N) return;                     // N % 4 == 0
  const int64_t row_off = a.out_row ? (int64_t)(*a.out_row) * a.out_row_stride : 0;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (a.bias) {
    const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
    v[0] += round_bf(b.x); v[1] += round_bf(b.y); v[2] += round_bf(b.z); v[3] += round_bf(b.w);
  }
  if constexpr (EPI == CRL_EPI_BF16) {
    *reinterpret_cast<uint2*>((u16*)a.out + row_off + (int64_t)m * a.ldo + n) = uint2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = gelu_f(round_bf(v[r]));
    *reinterpret_cast<uint2*>((u16*)a.out + row_off + (int64_t)m * a.ldo + n) = uint2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
  } else {   // CRL_EPI_F32_RESID
    const float4 r = *reinterpret_cast<const float4*>(a.resid + (int64_t)m * a.ldr + n);
    *reinterpret_cast<float4*>((float*)a.out + row_off + (int64_t)m * a.ldo + n) =
        float4{r.x + round_bf(v[0]), r.y + round_bf(v[1]), r.z + round_bf(v[2]), r.w + round_bf(v[3])};
  }
}

}  // namespace

extern "C" int crl_linear_skinny_bf16(int epilogue, int M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* W,
                                      int64_t ldw, const float* bias, void* out, int64_t ldo, const float* resid, int64_t ldr,
                                      const int* out_row_dev, int64_t out_row_stride, void* stream) {
  const char* who = "crl_linear_skinny_bf16";
  CRL_CHECK(x && W && out, "%s: null pointer", who);
  CRL_CHECK(M >= 1 && M <= 16, "%s: M = %d rows (1..16 supported; use crl_gemm_bf16 beyond)", who, M);
  CRL_CHECK(N > 0 && K > 0 && N % 4 == 0 && K % 8 == 0, "%s: need N %% 4 == 0 and K %% 8 == 0 (N=%lld K=%lld)", who, (long long)N, (long long)K);
  CRL_CHECK(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K && ldo >= N && ldo % 4 == 0, "%s: bad leading dimensions", who);
  CRL_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)W % 16) == 0, "%s: x and W must be 16-byte aligned", who);
  CRL_CHECK(((uintptr_t)out % (epilogue == CRL_EPI_F32_RESID ? 16 : 8)) == 0, "%s: out must be 8-byte (bf16) / 16-byte (fp32) aligned", who);
  CRL_CHECK(N < (1ll << 31) - 16 && K < (1ll << 31) - 32, "%s: extent too large", who);
  CRL_CHECK(!out_row_dev || (out_row_stride % 4) == 0, "%s: out_row_stride must be a multiple of 4 elements", who);
  SkArgs a{(const u16*)x, (const u16*)W, bias, out, resid, ldx, ldw, ldo, ldr, M, (int)N, (int)K, out_row_dev, out_row_stride,
           nullptr, 0, nullptr, nullptr, 0.f, nullptr, 0};
  const unsigned grid = (unsigned)((N + 15) / 16);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool wide = grid <= 512 && K >= 512;          // few workgroups: 16 waves per workgroup (see the kernel)
#define SK_LAUNCH(E) do { if (wide) linear_skinny_kernel<E, false, 16><<<grid, 1024, 0, s>>>(a); else linear_skinny_kernel<E, false, 4><<<grid, 256, 0, s>>>(a); } while (0)
  if (epilogue == CRL_EPI_BF16) SK_LAUNCH(CRL_EPI_BF16);
  else if (epilogue == CRL_EPI_BF16_GELU) SK_LAUNCH(CRL_EPI_BF16_GELU);
  else if (epilogue == CRL_EPI_F32_RESID) {
    CRL_CHECK(resid && ldr >= N && ldr % 4 == 0 && ((uintptr_t)resid % 16) == 0, "%s: F32_RESID needs an aligned residual", who);
    SK_LAUNCH(CRL_EPI_F32_RESID);
  } else {
    crl_set_error("%s: epilogue %d not supported (BF16, BF16_GELU, F32_RESID)", who, epilogue);
    return -1;
  }
  CRL_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int crl_linear_skinny_ln_bf16(int epilogue, int M, int64_t N, int64_t K, const float* x_f32, int64_t ldx, const float* gamma,
                                         const float* beta, float eps, float* h_f32, int64_t ldh, const void* W, int64_t ldw,
                                         const float* bias, void* out, int64_t ldo, const int* out_row_dev, int64_t out_row_stride,
                                         void* stream) {
  const char* who = "crl_linear_skinny_ln_bf16";
  CRL_CHECK(x_f32 && gamma && beta && W && out, "%s: null pointer", who);
  CRL_CHECK(M >= 1 && M <= 16, "%s: M = %d rows (1..16 supported)", who, M);
  CRL_CHECK(N > 0 && N % 4 == 0 && K >= 64 && K % 8 == 0 && K <= 2048, "%s: need N %% 4 == 0, K %% 8 == 0, 64 <= K <= 2048 (N=%lld K=%lld)", who, (long long)N, (long long)K);
  CRL_CHECK(!h_f32 || N >= K, "%s: the fp32 LayerNorm output is written by the first K / 16 workgroups: needs N >= K", who);
  CRL_CHECK(ldx % 4 == 0 && ldx >= K && ldw % 8 == 0 && ldw >= K && ldo >= N && ldo % 4 == 0 && (!h_f32 || (ldh % 4 == 0 && ldh >= K)), "%s: bad leading dimensions", who);
  CRL_CHECK(((uintptr_t)x_f32 % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 &&
                ((uintptr_t)h_f32 % 16) == 0 && ((uintptr_t)out % 8) == 0, "%s: operands must be 16-byte aligned (out: 8)", who);
  CRL_CHECK(N < (1ll << 31) - 16, "%s: extent too large", who);
  CRL_CHECK(!out_row_dev || (out_row_stride % 4) == 0, "%s: out_row_stride must be a multiple of 4 elements", who);
  SkArgs a{nullptr, (const u16*)W, bias, out, nullptr, 0, ldw, ldo, 0, M, (int)N, (int)K, out_row_dev, out_row_stride,
           x_f32, ldx, gamma, beta, eps, h_f32, ldh};
  const unsigned grid = (unsigned)((N + 15) / 16);
  const size_t lds = (size_t)M * (K + SK_PAD) * sizeof(u16);
  CRL_CHECK(lds <= 60 * 1024, "%s: %d rows of %lld features need %zu bytes of LDS (limit 60 KiB: fewer rows, or crl_layernorm_fwd + crl_linear_skinny_bf16)",
            who, M, (long long)K, lds);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool wide = grid <= 512 && K >= 512;
#define SK_LAUNCH_LN(E) do { if (wide) linear_skinny_kernel<E, true, 16><<<grid, 1024, lds, s>>>(a); else linear_skinny_kernel<E, true, 4><<<grid, 256, lds, s>>>(a); } while (0)
  if (epilogue == CRL_EPI_BF16) SK_LAUNCH_LN(CRL_EPI_BF16);
  else if (epilogue == CRL_EPI_BF16_GELU) SK_LAUNCH_LN(CRL_EPI_BF16_GELU);
  else {
    crl_set_error("%s: epilogue %d not supported (BF16, BF16_GELU)", who, epilogue);
    return -1;
  }
  CRL_LAUNCH_CHECK(who);
  return 0;
}
