// Sustained (power-limited) MFMA rate of the whole chip for the two bf16 MFMA shapes, operands in registers, no memory traffic:
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma_power.hip -o /tmp/ubench_mfma_power && /tmp/ubench_mfma_power
// 256 CUs x 4 SIMDs x W waves, each wave issues independent accumulate chains for ~100 ms per measurement; prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rnd(unsigned x) {   // cheap hash -> roughly N(0,1)-ish spread in [-2, 2]
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return ((float)(x & 0xffff) / 16384.0f) - 2.0f;
}
template <int SHAPE, int RANDOM>   // SHAPE 0: 32x32x16 (4 chains), 1: 16x16x32 (8 chains); RANDOM: operands differ from MFMA to MFMA and lane to lane
__global__ __launch_bounds__(256) void k(int iters, float* sink) {
  bf16x8 av[8], bv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      av[q][i] = RANDOM ? (__bf16)rnd(threadIdx.x * 131u + q * 17u + i) : (__bf16)(0.001f * (threadIdx.x % 7 + i));
      bv[q][i] = RANDOM ? (__bf16)(0.05f * rnd(threadIdx.x * 257u + q * 29u + i + 7u)) : (__bf16)(0.002f * (threadIdx.x % 5 + i));
    }
#define a av[(r * 2 + j) & 7]
#define b bv[(r * 3 + j) & 7]
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][7];
  } else {
    f32x4 c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][3];
  }
  if (s == 123.456f) sink[0] = s;
}

#undef a
#undef b
template <int SHAPE, int RANDOM>
void run(const char* name, int waves_per_simd, float* sink) {
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves = one per SIMD
  const int iters = 200000;   // ~50-100 ms per launch: long enough for the power controller
  const double flop_per_iter_wave = SHAPE == 0 ? 16.0 * 2 * 32 * 32 * 16 : 32.0 * 2 * 16 * 16 * 32;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(256), 0, 0, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s %s, %d wave(s)/SIMD, rep %d: %8.2f ms  %7.1f TFLOP/s\n", name, RANDOM ? "random operands" : "benign operands", waves_per_simd, rep, ms, flop_per_iter_wave * iters * blocks * 4 / (ms * 1e-3) / 1e12);
  }
}

int main() {
  float* sink; (void)hipMalloc(&sink, 4);
  run<0, 0>("v_mfma_f32_32x32x16_bf16", 2, sink);
  run<1, 0>("v_mfma_f32_16x16x32_bf16", 2, sink);
  run<0, 1>("v_mfma_f32_32x32x16_bf16", 2, sink);
  run<1, 1>("v_mfma_f32_16x16x32_bf16", 2, sink);
  run<0, 1>("v_mfma_f32_32x32x16_bf16", 1, sink);
  run<1, 1>("v_mfma_f32_16x16x32_bf16", 1, sink);
  return 0;
}
