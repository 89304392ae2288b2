// What would an ORDERED HAND-OFF between concurrent workgroups cost on gfx950?  (DESIGN.md "What comes next" item 1: the workgroups of one
// attention head passing a running dQ tile along instead of writing per-chain slabs.)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_handoff.hip -o /tmp/ubench_handoff && /tmp/ubench_handoff
// A ring of R workgroups (R = 2, 8, 25) passes a 16-KiB fp32 tile around: workgroup i waits for flag == its turn (one lane polls with an
// agent-scope acquire load and s_sleep), reads the tile, adds 1 to every element, writes it, publishes the next turn with an agent-scope
// release store.  Measured: shader clocks per hop (s_memtime of workgroup 0 over all its turns / hops), for rings whose members sit on
// ONE XCD (block ids 8 apart: the dispatcher deals consecutive ids round-robin over the 8 XCDs) and for rings spread over all XCDs
// (consecutive ids), and the tile's checksum (every element must equal the number of hops).  Spins are bounded: a ring that does not
// advance gives up and reports it instead of hanging the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int TILE_F = 4096;          // floats per tile (16 KiB): 256 threads x 16
constexpr int MAX_SPIN = 1 << 20;

__global__ __launch_bounds__(256) void ring_kernel(float* tiles, int* flags, int ring, int stride, int rounds, unsigned long long* out, int* fail) {
  // ring r = the workgroups {base + stride * i}: stride 8 keeps a ring on one XCD, stride 1 spreads it
  const int nrings = gridDim.x / ring;
  int r, i;
  if (stride == 1) { r = blockIdx.x / ring; i = blockIdx.x % ring; }
  else { const int x = blockIdx.x % 8, q = blockIdx.x / 8; r = x + 8 * (q / ring); i = q % ring; }
  if (r >= nrings) return;
  float* tile = tiles + (size_t)r * TILE_F;
  int* flag = flags + 64 * r;                       // one cache line per ring
  __shared__ int ok;
  unsigned long long t0 = 0, t1 = 0;
  for (int k = 0; k < rounds; ++k) {
    const int turn = k * ring + i;
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != turn && ++spins < MAX_SPIN) __builtin_amdgcn_s_sleep(2);
      ok = spins < MAX_SPIN;
      if (k == 0) t0 = __builtin_readcyclecounter();
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
    // the tile: written by another workgroup (another CU, maybe another XCD) a moment ago; the acquire above has invalidated this CU's L1
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const f4*>(tile)[threadIdx.x + 256 * j];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) reinterpret_cast<f4*>(tile)[threadIdx.x + 256 * j] = v[j];
    __threadfence();                                 // every thread's stores are visible device-wide before ...
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, turn + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... the next turn is published
  }
  if (threadIdx.x == 0 && i == 0) { t1 = __builtin_readcyclecounter(); out[r] = t1 - t0; }
}

int main() {
  const int rounds = 200;
  float* tiles; int* flags; unsigned long long* out; int* fail;
  hipMalloc(&tiles, 256 * TILE_F * sizeof(float));
  hipMalloc(&flags, 256 * 64 * sizeof(int));
  hipMalloc(&out, 256 * sizeof(unsigned long long));
  hipMalloc(&fail, sizeof(int));
  for (int ring : {2, 8, 25}) {
    for (int stride : {8, 1}) {
      const int nrings = 8;                           // 8 rings: one per XCD when stride = 8
      const int grid = nrings * ring;                 // <= 200 workgroups of 256 threads: all resident at once (256 CUs)
      hipMemset(tiles, 0, 256 * TILE_F * sizeof(float));
      hipMemset(flags, 0, 256 * 64 * sizeof(int));
      hipMemset(out, 0, 256 * sizeof(unsigned long long));
      hipMemset(fail, 0, sizeof(int));
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      ring_kernel<<<grid, 256>>>(tiles, flags, ring, stride, rounds, out, fail);
      hipEventRecord(e1);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(nrings); int hf; std::vector<float> t(TILE_F);
      hipMemcpy(h.data(), out, nrings * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      hipMemcpy(&hf, fail, sizeof(int), hipMemcpyDeviceToHost);
      hipMemcpy(t.data(), tiles, TILE_F * sizeof(float), hipMemcpyDeviceToHost);
      bool good = true;
      for (int j = 0; j < TILE_F; ++j) good = good && t[j] == (float)(rounds * ring);
      double cyc = 0; for (auto c : h) cyc += (double)c; cyc /= nrings;
      const double hops = (double)rounds * ring - 1;
      printf("ring of %2d workgroups, %s: %8.0f ns per hop (kernel %.3f ms / %d hops), %6.0f s_memtime ticks per hop, tile %s, %d workgroups gave up\n",
             ring, stride == 8 ? "one XCD      " : "all XCDs     ", ms * 1e6 / (rounds * ring), ms, rounds * ring, cyc / hops, good ? "exact" : "WRONG", hf);
    }
  }
  return 0;
}
