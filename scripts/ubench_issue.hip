// Issue-rate micro-benchmarks for gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
// One workgroup on one CU; cycles from s_memtime. Answers the questions the attention kernels' VALU/MFMA balance
// depends on: cycles per v_exp_f32 / v_mul_f32 / v_pk_mul_f32 / v_pk_fma_f32 / v_cvt_pk_bf16_f32 / v_max3_f32, the
// rate of dependent MFMA accumulate chains, and how much VALU throughput a wave keeps while its SIMD partner issues
// MFMAs back to back (waves w and w+4 of a workgroup share a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)

enum { K_EXP, K_MUL, K_PKMUL, K_PKFMA, K_CVT, K_MAX3, K_FMA, K_MFMA1, K_MFMA2, K_MFMA4, K_MFMA16_1, K_MFMA16_4, K_MFMA16_4L, K_MIX_1_4, K_MIX_1_2, K_SLOT, K_SLOT_NOLDS, K_SLOT_NOEXP, K_SLOT_NOMFMA, K_N };
static const char* NAMES[K_N] = {"v_exp_f32", "v_mul_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_fma_f32",
                                 "mfma32x32x16 1 chain", "mfma32x32x16 2 chains", "mfma32x32x16 4 chains", "mfma16x16x32 1 chain", "mfma16x16x32 4 chains", "mfma16x16x32 4 chains x4 per iteration",
                                 "1 wave: (mfma32 + 4 v_mul) x4 [per mfma]", "1 wave: (mfma32 + 2 v_mul) x4 [per mfma]",
                                 "slot: mfma32 + 2 ds_read + 2 fma + 2 exp + 2 mul", "slot without ds_read", "slot without exp (2 more mul)", "slot without mfma"};
static const int PER_ITER[K_N] = {16, 16, 16, 16, 16, 16, 16, 4, 4, 4, 4, 4, 16, 4, 4, 4, 4, 4, 4};

template <int KIND>
__device__ __forceinline__ void body(float (&r)[16], f32x2 (&p)[8], f32x4 (&q)[4], f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
  if constexpr (KIND == K_EXP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
  } else if constexpr (KIND == K_MUL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 5) & 15]));
  } else if constexpr (KIND == K_FMA) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(r[(i + 9) & 15]));
  } else if constexpr (KIND == K_PKMUL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]));
  } else if constexpr (KIND == K_PKFMA) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]), "v"(p[(i + 5) & 7]));
  } else if constexpr (KIND == K_CVT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(r[(i + 9) & 15]));
  } else if constexpr (KIND == K_MAX3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(r[(i + 9) & 15]));
  } else if constexpr (KIND == K_MFMA1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
  } else if constexpr (KIND == K_MFMA2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(a), "v"(b));
  } else if constexpr (KIND == K_MFMA4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  } else if constexpr (KIND == K_MFMA16_1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q[0]) : "v"(a), "v"(b));
  } else if constexpr (KIND == K_MFMA16_4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q[i]) : "v"(a), "v"(b));
  } else if constexpr (KIND == K_MFMA16_4L) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q[i & 3]) : "v"(a), "v"(b));
  } else if constexpr (KIND == K_SLOT || KIND == K_SLOT_NOLDS || KIND == K_SLOT_NOEXP || KIND == K_SLOT_NOMFMA) {
    // the per-MFMA slot of the software-pipelined attention backward: x4 per iteration, two accumulator chains
    const uint32_t la = (threadIdx.x & 63) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (KIND != K_SLOT_NOMFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(a), "v"(b));
      if constexpr (KIND != K_SLOT_NOLDS) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(p[2 * (i & 1)]) : "v"(la));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(p[2 * (i & 1) + 1]) : "v"(la));
      }
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[4 * i]) : "v"(r[13]), "v"(r[14]));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[4 * i + 1]) : "v"(r[13]), "v"(r[14]));
      if constexpr (KIND != K_SLOT_NOEXP) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[(4 * i + 8) & 15]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[(4 * i + 9) & 15]));
      } else {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[(4 * i + 8) & 15]) : "v"(r[15]));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[(4 * i + 9) & 15]) : "v"(r[15]));
      }
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[(4 * i + 2) & 15]) : "v"(r[15]));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[(4 * i + 3) & 15]) : "v"(r[15]));
    }
    if constexpr (KIND != K_SLOT_NOLDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if constexpr (KIND == K_MIX_1_4 || KIND == K_MIX_1_2) {
    constexpr int NV = KIND == K_MIX_1_4 ? 4 : 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[4 * i + v]) : "v"(r[(4 * i + v + 5) & 15]));
    }
  }
}

// waves 0,4,8,12 of a 1024-thread workgroup share SIMD 0: slot s = wave>>2 runs kind Ks (or idles when Ks < 0);
// out[s] = cycles of that wave
template <int K>
__device__ __forceinline__ void slot(int iters, uint64_t& t0, uint64_t& t1, float (&r)[16], f32x2 (&p)[8], f32x4 (&q)[4], f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
  if constexpr (K >= 0) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) body<K>(r, p, q, acc, a, b);
    asm volatile("s_nop 0" ::: "memory");
    t1 = __builtin_readcyclecounter();
  }
}
template <int K0, int K1, int K2, int K3>
__global__ __launch_bounds__(1024) void quad_kernel(int i0, int i1, int i2, int i3, uint64_t* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = 0.f;
  float r[16];
  f32x2 p[8];
  f32x4 q[4];
  f32x16 acc[4];
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f32x2{r[i], r[i + 8]};
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.001f; b[i] = (__bf16)0.002f; }
  __syncthreads();
  uint64_t t0 = 0, t1 = 0;
  if (wave == 0) slot<K0>(i0, t0, t1, r, p, q, acc, a, b);
  else if (wave == 4) slot<K1>(i1, t0, t1, r, p, q, acc, a, b);
  else if (wave == 8) slot<K2>(i2, t0, t1, r, p, q, acc, a, b);
  else if (wave == 12) slot<K3>(i3, t0, t1, r, p, q, acc, a, b);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i] + acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i] + p[i & 7][i >> 3] + q[i & 3][i >> 2];
  sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && (wave & 3) == 0) out[wave >> 2] = t1 - t0;
}

template <int K0, int K1, int K2, int K3>
void run(const char* label, int i0, int i1, int i2, int i3, uint64_t* d_out, float* d_sink) {
  uint64_t h[4] = {0, 0, 0, 0};
  const int ks[4] = {K0, K1, K2, K3}, is[4] = {i0, i1, i2, i3};
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((quad_kernel<K0, K1, K2, K3>), dim3(1), dim3(1024), 0, 0, i0, i1, i2, i3, d_out, d_sink);
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  }
  printf("%-44s", label);
  for (int s = 0; s < 4; ++s)
    if (ks[s] >= 0) printf("  [%d] %6.2f cyc/instr (%8llu)", s, (double)h[s] / ((double)is[s] * PER_ITER[ks[s]]), (unsigned long long)h[s]);
  printf("\n");
}

int main() {
  uint64_t* d_out; float* d_sink;
  hipMalloc(&d_out, 32); hipMalloc(&d_sink, 1024 * 4);
  const int N = 20000;
#define SOLO(K) run<K, -1, -1, -1>(NAMES[K], N, 0, 0, 0, d_out, d_sink)
  SOLO(K_EXP); SOLO(K_MUL); SOLO(K_FMA); SOLO(K_PKMUL); SOLO(K_PKFMA); SOLO(K_CVT); SOLO(K_MAX3);
  SOLO(K_MFMA1); SOLO(K_MFMA2); SOLO(K_MFMA4); SOLO(K_MFMA16_1); SOLO(K_MFMA16_4); SOLO(K_MFMA16_4L);
  // several waves of one SIMD
  run<K_MUL, K_MUL, -1, -1>("2 waves v_mul_f32", N, N, 0, 0, d_out, d_sink);
  run<K_MUL, K_MUL, K_MUL, -1>("3 waves v_mul_f32", N, N, N, 0, d_out, d_sink);
  run<K_MUL, K_MUL, K_MUL, K_MUL>("4 waves v_mul_f32", N, N, N, N, d_out, d_sink);
  run<K_EXP, K_EXP, -1, -1>("2 waves v_exp_f32", N, N, 0, 0, d_out, d_sink);
  run<K_EXP, K_EXP, K_EXP, K_EXP>("4 waves v_exp_f32", N, N, N, N, d_out, d_sink);
  run<K_PKFMA, K_PKFMA, K_PKFMA, K_PKFMA>("4 waves v_pk_fma_f32", N, N, N, N, d_out, d_sink);
  run<K_CVT, K_CVT, K_CVT, K_CVT>("4 waves v_cvt_pk_bf16_f32", N, N, N, N, d_out, d_sink);
  run<K_MFMA4, K_MFMA4, -1, -1>("2 waves mfma32", N, N, 0, 0, d_out, d_sink);
  // co-issue: one wave on the MFMA pipe (4 per iter = 128 ideal cycles), partners on the VALU (16 per iter)
  run<K_MFMA4, K_MUL, -1, -1>("mfma32 | mul", N, 2 * N, 0, 0, d_out, d_sink);
  run<K_MFMA4, K_MUL, K_MUL, -1>("mfma32 | mul | mul", N, 2 * N, 2 * N, 0, d_out, d_sink);
  run<K_MFMA4, K_MUL, K_MUL, K_MUL>("mfma32 | mul | mul | mul", N, 2 * N, 2 * N, 2 * N, d_out, d_sink);
  run<K_MFMA4, K_EXP, K_EXP, K_EXP>("mfma32 | exp | exp | exp", N, N, N, N, d_out, d_sink);
  run<K_MFMA4, K_PKFMA, K_PKFMA, -1>("mfma32 | pk_fma | pk_fma", N, 2 * N, 2 * N, 0, d_out, d_sink);
  run<K_MFMA4, K_CVT, K_CVT, -1>("mfma32 | cvt | cvt", N, 2 * N, 2 * N, 0, d_out, d_sink);
  run<K_MFMA16_4L, K_MUL, K_MUL, -1>("mfma16 | mul | mul", N, 2 * N, 2 * N, 0, d_out, d_sink);
  // one wave alternating its own MFMAs and independent VALU
  SOLO(K_MIX_1_4); SOLO(K_MIX_1_2);
  SOLO(K_SLOT); SOLO(K_SLOT_NOLDS); SOLO(K_SLOT_NOEXP); SOLO(K_SLOT_NOMFMA);
  run<K_SLOT, K_SLOT, -1, -1>("2 waves: slot", N, N, 0, 0, d_out, d_sink);
  run<K_SLOT_NOLDS, K_SLOT_NOLDS, -1, -1>("2 waves: slot without ds_read", N, N, 0, 0, d_out, d_sink);
  run<K_SLOT_NOEXP, K_SLOT_NOEXP, -1, -1>("2 waves: slot without exp", N, N, 0, 0, d_out, d_sink);
  run<K_SLOT_NOMFMA, K_SLOT_NOMFMA, -1, -1>("2 waves: slot without mfma", N, N, 0, 0, d_out, d_sink);
  run<K_SLOT, K_SLOT, K_SLOT, -1>("3 waves: slot", N, N, N, 0, d_out, d_sink);
  run<K_MIX_1_4, K_MIX_1_4, -1, -1>("2 waves: mfma32 + 4 v_mul", N, N, 0, 0, d_out, d_sink);
  return 0;
}
