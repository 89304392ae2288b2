// The dQ hand-off of the single-pass attention backward priced AS A PIPELINE (VERDICT r4 item 6 / Weak #9): round 4 measured the latency of
// one hop (3.3 us, scripts/ubench_handoff.hip) and compared it with the 2.2 us a workgroup spends per query tile -- but in a chain where
// workgroup kb consumes tile t from kb - 1, the 25 workgroups of a head settle into a skew of one hop each: the hop sits on the critical path
// once per workgroup (pipeline fill), not once per tile, provided each link keeps up with the tile rate.  This program runs exactly that:
//   H heads x 25 workgroups (one per CU: each asks for 100 KiB of LDS), workgroup (h, kb) walks 97 tiles; per tile it "computes" for WORK_NS
//   (a timed spin), then -- kb > 0 -- waits until flag[h][kb - 1] >= t + 1 (one lane polls, relaxed, s_sleep between polls), reads the running
//   16-KiB fp32 tile of (h, t) with sc1 loads, adds its own contribution, writes it back with sc1 (write-through) stores, drains, and
//   publishes flag[h][kb] = t + 1 (guide Guideline 16, form R1: no release fence, no acquire).  The last workgroup of a head leaves the sum.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_handoff_pipeline.hip -o /tmp/ubench_pipe && /tmp/ubench_pipe
// Reported: kernel time against the no-hand-off time (97 x WORK_NS), the time per tile of the LAST link of a chain, checksum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int TILE_F = 4096;          // floats per tile (16 KiB): 256 threads x 16
constexpr int NKB = 25, NT = 97;
constexpr int MAX_SPIN = 1 << 22;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void chain_kernel(float* tiles, unsigned* flags, int work_ns, int mode, unsigned long long* out, int* fail) {
  extern __shared__ char lds[];       // 100 KiB requested: one workgroup per CU, like the 512-register attention stream
  const int h = blockIdx.x / NKB, kb = blockIdx.x % NKB;
  unsigned* myflag = flags + 64 * (h * NKB + kb);
  unsigned* prev = flags + 64 * (h * NKB + kb - 1);
  __shared__ int ok;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tiles + (size_t)h * NT * TILE_F, 0, NT * TILE_F * 4, 0x00020000);
  unsigned long long t_first = 0, t_last = 0;
  for (int t = 0; t < NT; ++t) {
    // ---- the tile pass of the attention stream: WORK_NS of "matrix work" (s_memrealtime runs at 100 MHz)
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - w0 < (unsigned long long)(work_ns / 10)) __builtin_amdgcn_s_sleep(4);
    f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = f4{(float)(kb + 1), (float)(kb + 1), (float)(kb + 1), (float)(kb + 1)};
    if (mode == 1) {                  // ---- ordered hand-off
      if (kb > 0) {
        if (threadIdx.x == 0) {
          int spins = 0;
          while (__hip_atomic_load(prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(t + 1) && ++spins < MAX_SPIN) __builtin_amdgcn_s_sleep(1);
          ok = spins < MAX_SPIN;
        }
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // every load of the handed-off bytes is an sc1 load (aux 16)
          const u4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (t * TILE_F + 4 * (threadIdx.x + 256 * j)) * 4, 0, 16);
          v[j] += __builtin_bit_cast(f4, r);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)       // write-through stores (sc1): no release fence
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v[j]), rs, (t * TILE_F + 4 * (threadIdx.x + 256 * j)) * 4, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains ...
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(myflag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before ONE lane publishes
    } else {                          // ---- reference: every workgroup writes its own slab tile (no dependency)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v[j]), rs, (t * TILE_F + 4 * (threadIdx.x + 256 * j)) * 4, 0, 0);
    }
    if (threadIdx.x == 0) { if (t == 0) t_first = __builtin_amdgcn_s_memrealtime(); t_last = __builtin_amdgcn_s_memrealtime(); }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = t_last - t_first;   // 10-ns ticks for tiles 1..96 of this link
  (void)lds;
}

int main() {
  const int Hmax = 128;
  float* tiles; unsigned* flags; unsigned long long* out; int* fail;
  hipMalloc(&tiles, (size_t)Hmax * NT * TILE_F * sizeof(float));
  hipMalloc(&flags, (size_t)Hmax * NKB * 64 * sizeof(unsigned));
  hipMalloc(&out, (size_t)Hmax * NKB * sizeof(unsigned long long));
  hipMalloc(&fail, sizeof(int));
  hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
  for (int work_ns : {2200, 1100}) {
    for (int H : {10, 128}) {
      for (int mode : {0, 1}) {
        hipMemset(tiles, 0, (size_t)Hmax * NT * TILE_F * sizeof(float));
        hipMemset(flags, 0, (size_t)Hmax * NKB * 64 * sizeof(unsigned));
        hipMemset(fail, 0, sizeof(int));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        chain_kernel<<<H * NKB, 256, 102400>>>(tiles, flags, work_ns, mode, out, fail);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> ho((size_t)H * NKB); int hf; std::vector<float> t((size_t)NT * TILE_F);
        hipMemcpy(ho.data(), out, ho.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipMemcpy(&hf, fail, sizeof(int), hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), tiles + (size_t)(H - 1) * NT * TILE_F, t.size() * sizeof(float), hipMemcpyDeviceToHost);
        bool good = true;
        const float want = mode == 1 ? NKB * (NKB + 1) / 2.f : -1.f;
        if (mode == 1) for (size_t j = 0; j < t.size(); ++j) good = good && t[j] == want;
        double last = 0, first = 0;
        for (int h = 0; h < H; ++h) { last += (double)ho[(size_t)h * NKB + NKB - 1]; first += (double)ho[(size_t)h * NKB]; }
        const double rounds = (double)(H * NKB) / 256.0;
        printf("tile work %4d ns, %3d heads x %d links (%4.1f rounds of 256 CUs), %s: kernel %8.1f us (no hand-off floor %7.1f us x rounds = %8.1f), "
               "per tile: first link %5.0f ns, last link %5.0f ns, sums %s, %d workgroups gave up\n",
               work_ns, H, NKB, rounds, mode ? "ordered hand-off" : "independent slabs", ms * 1e3, NT * work_ns * 1e-3, NT * work_ns * 1e-3 * rounds,
               first / H / (NT - 1) * 10, last / H / (NT - 1) * 10, mode ? (good ? "exact" : "WRONG") : "n/a", hf);
      }
    }
  }
  return 0;
}
