// Store / load-modify-store rate of ONE workgroup per CU writing 256 x 256 fp32 tiles (the GEMM epilogue's traffic) with different
// lane -> address maps.  hipcc --offload-arch=gfx950 -O3 scripts/ubench_store.hip -o /tmp/ubench_store && /tmp/ubench_store
//   P0  lane l: row l & 15, 16-byte chunk l >> 4      (MFMA accumulator layout: a quarter-wave touches 16 rows)
//   P1  row l >> 2, chunk l & 3                        (16 rows x 64 contiguous bytes per wave instruction)
//   P2  row l >> 3, chunk l & 7                        ( 8 rows x 128 bytes: full L2 lines)
//   P3  row l >> 4, chunk l & 15                       ( 4 rows x 256 bytes)
//   P4  row 0, chunk l                                 ( 1 row x 1 KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int P, int RMW>
__global__ __launch_bounds__(512) void k(float* base, int pitch_f, int tiles_per_wg, int tile_cols, unsigned long long* cyc) {
  extern __shared__ char smem[];   // 128 KiB: forces one workgroup per CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPI = P == 0 ? 16 : P == 1 ? 16 : P == 2 ? 8 : P == 3 ? 4 : 1;     // rows per instruction
  constexpr int BPR = 1024 / RPI;                                                   // bytes per row per instruction
  int row, chunk;
  if (P == 0) { row = lane & 15; chunk = lane >> 4; } else { row = lane / (BPR / 16); chunk = lane % (BPR / 16); }
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x + t * gridDim.x;
    const int tr = tile / tile_cols, tc = tile % tile_cols;
    // wave w owns rows 32 w .. 32 w + 31 of the 256-row tile, all 1024 bytes of each row
    const uint32_t tbase = (uint32_t)((tr * 256 + 32 * wave) * pitch_f + tc * 256) * 4u;
    u4 v[32];
    if (RMW) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int seg = i % (1024 / BPR), rg = i / (1024 / BPR);
        const uint32_t off = tbase + (uint32_t)((rg * RPI + row) * pitch_f) * 4u + seg * BPR + chunk * 16;
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int seg = i % (1024 / BPR), rg = i / (1024 / BPR);
      const uint32_t off = tbase + (uint32_t)((rg * RPI + row) * pitch_f) * 4u + seg * BPR + chunk * 16;
      u4 o;
      if (RMW) { o = v[i]; o[0] += 1; } else { o = u4{(unsigned)i, (unsigned)lane, (unsigned)t, 1u}; }
      __builtin_amdgcn_raw_buffer_store_b128(o, r, off, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int P, int RMW>
void run(float* d, unsigned long long* dc, int grid) {
  const int pitch = 1024, tile_cols = 4;                 // [49152 x 1024] fp32 = 201 MB: the proj/fc2 output
  const int tiles = 192 * tile_cols, per = tiles / grid; // grid 256 -> 3 tiles per workgroup; grid 32 -> 24
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<P, RMW>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<P, RMW>), dim3(grid), dim3(512), 131072, 0, d, pitch, per, tile_cols, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  const double bytes = (double)per * 256 * 1024 * (RMW ? 2 : 1);
  printf("P%d %s grid %3d: %7.1f us, %8llu cycles / %d tiles = %6.0f cycles per tile, %5.1f B/clk/CU, %6.2f TB/s chip\n", P, RMW ? "load+store" : "store     ", grid,
         ms * 1e3, c, per, (double)c / per, bytes / c, bytes * grid / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; unsigned long long* dc;
  hipMalloc(&d, (size_t)49152 * 1024 * 4); hipMalloc(&dc, 8);
  hipMemset(d, 0, (size_t)49152 * 1024 * 4);
  for (int grid : {256, 32}) {
    run<0, 0>(d, dc, grid); run<1, 0>(d, dc, grid); run<2, 0>(d, dc, grid); run<3, 0>(d, dc, grid); run<4, 0>(d, dc, grid);
    run<0, 1>(d, dc, grid); run<1, 1>(d, dc, grid); run<2, 1>(d, dc, grid); run<3, 1>(d, dc, grid); run<4, 1>(d, dc, grid);
  }
  return 0;
}
