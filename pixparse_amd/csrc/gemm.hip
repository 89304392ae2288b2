// bf16 MFMA GEMM with fused epilogues for gfx950 -- the Linear / conv-as-GEMM / LM-head forward,
// dgrad and wgrad of the Cruller step (include/crl.h: crl_gemm_bf16).
//
// Structure (guide §5 "minimum 2-phase"): 128x128 output tile per 256-thread workgroup (4 waves,
// 2x2, 64x64 per wave = 4x4 v_mfma_f32_16x16x32_bf16 accumulators), BK = 64 (32 for K % 64 != 0),
// two LDS stages filled by bounds-checked LDS-DMA (`buffer_load_dwordx4 ... lds`: rows past the
// end of a matrix arrive as zeros, which is what makes ragged M / ragged contraction exact),
// one barrier per K tile.  Two LDS images:
//   KM  [128 rows][BK k]   (operand stored with k contiguous) read with ds_read_b128,
//   TR  [BK k][128 cols]   (operand stored with k as the ROW index) read with ds_read_b64_tr_b16,
// both XOR-swizzled on the DMA *source* side (LDS destination stays lane-linear, guide rule 21).
// The MFMA A operand is the matrix-B fragment (n on the row index) and the MFMA B operand is the
// matrix-A fragment, so every lane ends up with 4 consecutive n of one output row -> 8/16-byte
// row-contiguous epilogue stores.
#include "common.h"

#include "gemm_common.h"

namespace {
using namespace gemmc;

constexpr int BM = 128, BN = 128, NT_THREADS = 256;

template <int LAYOUT, int EPI, int BK>
__global__ __launch_bounds__(NT_THREADS, 2) void gemm_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 128 * BK * 2];
  constexpr int TILE_BYTES = 128 * BK * 2;
  constexpr bool A_TR = (LAYOUT == CRL_TN);
  constexpr bool B_TR = (LAYOUT != CRL_NT);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = t / g.ntn, tn = t % g.ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  const __amdgpu_buffer_rsrc_t ra = make_rsrc(g.A, g.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(g.B, g.b_bytes);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (g.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * g.kchunk;
  const int nk = min(nk_all, kt0 + g.kchunk);

  auto stage = [&](int buf, int kt) {
    char* la = smem + buf * 2 * TILE_BYTES;
    char* lb = la + TILE_BYTES;
    if constexpr (A_TR) stage_tr<BK, NT_THREADS>(ra, la, kt * BK, m0, g.lda, tid, wave);
    else stage_km<BK, NT_THREADS>(ra, la, m0, kt * BK, g.lda, tid, wave);
    if constexpr (B_TR) stage_tr<BK, NT_THREADS>(rb, lb, kt * BK, n0, g.ldb, tid, wave);
    else stage_km<BK, NT_THREADS>(rb, lb, n0, kt * BK, g.ldb, tid, wave);
  };

  stage(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* la = smem + cur * 2 * TILE_BYTES;
    const char* lb = la + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (A_TR) fa[i] = frag_tr(la, wm * 64 + i * 16, ks, lane);
        else fa[i] = frag_km<BK>(la, wm * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (B_TR) fb[j] = frag_tr(lb, wn * 64 + j * 16, ks, lane);
        else fb[j] = frag_km<BK>(lb, wn * 64 + j * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane (i = lane&15, q = lane>>4) holds C[m = .. + i][n = .. + 4q + 0..3]
  const int li = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + lq * 4;
      if (n >= g.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if constexpr (EPI == CRL_EPI_BF16 || EPI == CRL_EPI_BF16_GELU || EPI == CRL_EPI_F32_RESID) {
        if (g.bias) {
          const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
          v[0] += round_bf(b.x); v[1] += round_bf(b.y); v[2] += round_bf(b.z); v[3] += round_bf(b.w);
        }
      }
      if constexpr (EPI == CRL_EPI_BF16) {
        const float f = n < g.colscale_cols ? g.colscale : 1.f;
        uint2 o{pack_bf2(v[0] * f, v[1] * f), pack_bf2(v[2] * f, v[3] * f)};
        *reinterpret_cast<uint2*>((u16*)g.C + (size_t)m * g.ldc + n) = o;
      } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
        float h[4], y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gelu_grad_f(round_bf(v[r]), y[r], h[r]);      // h := gelu'(bf16(v + b)), saved as fp16
        *reinterpret_cast<uint2*>((u16*)g.aux + (size_t)m * g.ldaux + n) = uint2{pack_h2(h[0], h[1]), pack_h2(h[2], h[3])};
        *reinterpret_cast<uint2*>((u16*)g.C + (size_t)m * g.ldc + n) = uint2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
      } else if constexpr (EPI == CRL_EPI_BF16_DGELU) {
        const uint2 hh = *reinterpret_cast<const uint2*>((const u16*)g.aux + (size_t)m * g.ldaux + n);
        const float y0 = round_bf(v[0]) * h2f_lo(hh.x), y1 = round_bf(v[1]) * h2f_hi(hh.x);
        const float y2 = round_bf(v[2]) * h2f_lo(hh.y), y3 = round_bf(v[3]) * h2f_hi(hh.y);
        *reinterpret_cast<uint2*>((u16*)g.C + (size_t)m * g.ldc + n) = uint2{pack_bf2(y0, y1), pack_bf2(y2, y3)};
      } else if constexpr (EPI == CRL_EPI_F32_RESID) {
        const float4 r = *reinterpret_cast<const float4*>(g.resid + (size_t)m * g.ldr + n);
        float4 o{r.x + round_bf(v[0]), r.y + round_bf(v[1]), r.z + round_bf(v[2]), r.w + round_bf(v[3])};
        *reinterpret_cast<float4*>((float*)g.C + (size_t)m * g.ldc + n) = o;
      } else if constexpr (EPI == CRL_EPI_F32) {
        *reinterpret_cast<float4*>((float*)g.C + blockIdx.y * g.slab_stride + (size_t)m * g.ldc + n) = float4{v[0], v[1], v[2], v[3]};
      } else {  // CRL_EPI_F32_ACC
        float4* p = reinterpret_cast<float4*>((float*)g.C + (size_t)m * g.ldc + n);
        float4 o = *p;
        o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
        *p = o;
      }
    }
  }
}

// out (+)= sum over slabs (deterministic order); slabs are dense [M][N] fp32.  The slabs are read exactly once: non-temporal loads, four
// of them in flight per thread (round 4; the one-load-per-iteration form ran at 5.1 TB/s)
__device__ __forceinline__ float4 nt_load_f4(const float* p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return float4{v[0], v[1], v[2], v[3]};
}
// (round 6) the blocks behind the first `nb_slabs` reduce the partial column sums the 4-wave weight-gradient kernel left behind the slabs (bias gradient,
// crl_gemm_bf16 CRL_TN with aux): one launch for both reductions instead of two
__device__ __forceinline__ void cs_reduce_body(const float* __restrict__ ws, int P, int M, float* __restrict__ out, int acc, int m) {
  if (m >= M) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // fixed order: deterministic
  int p = 0;
  for (; p + 4 <= P; p += 4) { s0 += ws[(size_t)p * M + m]; s1 += ws[(size_t)(p + 1) * M + m]; s2 += ws[(size_t)(p + 2) * M + m]; s3 += ws[(size_t)(p + 3) * M + m]; }
  for (; p < P; ++p) s0 += ws[(size_t)p * M + m];
  const float t = (s0 + s1) + (s2 + s3);
  out[m] = acc ? out[m] + t : t;
}
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, size_t slab, float* __restrict__ out, int M, int N,
                                     int ldc, int acc, unsigned nb_slabs = 0xffffffffu, const float* __restrict__ cs_ws = nullptr, int cs_P = 0,
                                     float* __restrict__ cs_out = nullptr, int cs_acc = 0) {
  if (blockIdx.x >= nb_slabs) {
    cs_reduce_body(cs_ws, cs_P, M, cs_out, cs_acc, (int)((blockIdx.x - nb_slabs) * blockDim.x + threadIdx.x));
    return;
  }
  const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= (size_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  float4 s = nt_load_f4(ws + idx);
  int k = 1;
  for (; k + 4 <= nsplit; k += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = nt_load_f4(ws + (size_t)(k + u) * slab + idx);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; k < nsplit; ++k) {
    const float4 v = nt_load_f4(ws + (size_t)k * slab + idx);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float4* p = reinterpret_cast<float4*>(out + (size_t)m * ldc + n);
  if (acc) { const float4 o = *p; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
  *p = s;
}

// the same slab sum followed by a forward / dgrad epilogue (bias + bf16 store, or bias + bf16 rounding + fp32 residual): used for the
// few remainder rows of the wave-quantisation split when their contraction is long (see crl_gemm_bf16)
template <int EPI>
__global__ void splitk_reduce_epi_kernel(const float* __restrict__ ws, int nsplit, size_t slab, int M, int N, const float* __restrict__ bias,
                                         void* __restrict__ C, int ldc, const float* __restrict__ resid, int ldr, float colscale = 1.f, int colscale_cols = 0) {
  const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= (size_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  float4 s = *reinterpret_cast<const float4*>(ws + idx);
  for (int k = 1; k < nsplit; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(ws + k * slab + idx);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (bias) {
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    s.x += round_bf(b.x); s.y += round_bf(b.y); s.z += round_bf(b.z); s.w += round_bf(b.w);
  }
  if constexpr (EPI == CRL_EPI_BF16) {
    const float f = n < colscale_cols ? colscale : 1.f;
    *reinterpret_cast<uint2*>((u16*)C + (size_t)m * ldc + n) = uint2{pack_bf2(s.x * f, s.y * f), pack_bf2(s.z * f, s.w * f)};
  } else {
    const float4 r = *reinterpret_cast<const float4*>(resid + (size_t)m * ldr + n);
    *reinterpret_cast<float4*>((float*)C + (size_t)m * ldc + n) = float4{r.x + round_bf(s.x), r.y + round_bf(s.y), r.z + round_bf(s.z), r.w + round_bf(s.w)};
  }
}

template <int LAYOUT, int EPI>
int launch_bk(const GemmArgs& a, int bk, int nsplit, hipStream_t s) {
  const dim3 grid(a.ntm * a.ntn, nsplit);
  if (bk == 64) gemm_kernel<LAYOUT, EPI, 64><<<grid, NT_THREADS, 0, s>>>(a);
  else gemm_kernel<LAYOUT, EPI, 32><<<grid, NT_THREADS, 0, s>>>(a);
  CRL_LAUNCH_CHECK("crl_gemm_bf16");
  return 0;
}

template <int LAYOUT>
int launch_epi(const GemmArgs& a, int epi, int bk, int nsplit, hipStream_t s) {
  switch (epi) {
    case CRL_EPI_BF16: return launch_bk<LAYOUT, CRL_EPI_BF16>(a, bk, nsplit, s);
    case CRL_EPI_BF16_GELU: return launch_bk<LAYOUT, CRL_EPI_BF16_GELU>(a, bk, nsplit, s);
    case CRL_EPI_BF16_DGELU: return launch_bk<LAYOUT, CRL_EPI_BF16_DGELU>(a, bk, nsplit, s);
    case CRL_EPI_F32_RESID: return launch_bk<LAYOUT, CRL_EPI_F32_RESID>(a, bk, nsplit, s);
    case CRL_EPI_F32: return launch_bk<LAYOUT, CRL_EPI_F32>(a, bk, nsplit, s);
    case CRL_EPI_F32_ACC: return launch_bk<LAYOUT, CRL_EPI_F32_ACC>(a, bk, nsplit, s);
  }
  crl_set_error("crl_gemm_bf16: bad epilogue %d", epi);
  return -1;
}

}  // namespace

int crl_gemm256_launch(int layout, int epi, const gemmc::GemmArgs& a, int nsplit, hipStream_t s);
int crl_gemm4w_launch(int layout, int epi, const gemmc::GemmArgs& a, int nsplit, hipStream_t s);
bool crl_gemm4w_overlaps(int layout, int epi, const gemmc::GemmArgs& a, int nsplit);   // the 4-wave kernel would run this launch with the epilogue of tile T inside the main loop of tile T + 1

// ---- launch geometry of the persistent kernels + the ticket-counter pool of the dynamic tile scheduler (gemm_common.h) ----
#include <atomic>
#include <algorithm>
#include <mutex>
static constexpr int CHIP_CUS = 256;
static int g_reserved_cus = 0;     // CUs left to other kernels (RCCL) by the persistent launches
static int g_dynamic = 1;          // 1 = resident workgroups pull tiles from ticket counters, 0 = static walk b, b + grid, ...
constexpr int CRL_SCHED_RINGS = 4;
__device__ uint32_t g_sched_state[CRL_SCHED_RINGS * gemmc::CRL_SCHED_SLOTS * gemmc::CRL_SCHED_WORDS];   // zero at module load; every launch leaves its slot zeroed
int crl_gemm_cus() { return CHIP_CUS - g_reserved_cus; }
bool crl_gemm_dynamic() { return g_dynamic != 0; }
// A ring of slots per STREAM: launch n + CRL_SCHED_SLOTS of a ring reuses the slot of launch n, which is safe only if the two are ordered -- i.e.
// issued on one stream (the last workgroup of a launch zeroes its slot before the kernel ends).  The first CRL_SCHED_RINGS streams that launch a
// persistent GEMM get a ring each (the default stream, the stream torch.cuda.graph captures on -- a captured launch keeps its slot, replays of the
// graph are ordered among themselves --, the RCCL-overlap side stream); a launch on any further stream gets *ok = true with no slot and walks its
// tiles statically (same results, no shared state) instead of racing on the counters.
uint32_t* crl_sched_slot(hipStream_t stream, bool* ok) {
  static uint32_t* base = nullptr;
  static std::mutex mu;
  static void* owner[CRL_SCHED_RINGS];
  static unsigned seq[CRL_SCHED_RINGS];
  static int nrings = 0;
  *ok = true;
  std::lock_guard<std::mutex> lock(mu);
  if (!base) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sched_state)) != hipSuccess || !p) { crl_set_error("crl_gemm_bf16: no scheduler state"); *ok = false; return nullptr; }
    base = (uint32_t*)p;
  }
  int r = 0;
  while (r < nrings && owner[r] != (void*)stream) ++r;
  if (r == nrings) {
    if (nrings == CRL_SCHED_RINGS) return nullptr;
    owner[nrings] = (void*)stream;
    seq[nrings++] = 0;
  }
  return base + ((size_t)r * gemmc::CRL_SCHED_SLOTS + (size_t)(seq[r]++ % gemmc::CRL_SCHED_SLOTS)) * gemmc::CRL_SCHED_WORDS;
}
extern "C" int crl_gemm_set_reserved_cus(int n) {
  if (n < 0 || n > CHIP_CUS - 32) { crl_set_error("crl_gemm_set_reserved_cus: %d is outside [0, %d]", n, CHIP_CUS - 32); return -1; }
  g_reserved_cus = n;
  return 0;
}
extern "C" int crl_gemm_set_schedule(int dynamic) {
  if (dynamic != 0 && dynamic != 1) { crl_set_error("crl_gemm_set_schedule: 0 = static, 1 = dynamic"); return -1; }
  g_dynamic = dynamic;
  return 0;
}

// Kernel / split-K plan.  big = 256x256 8-phase kernel (one workgroup per CU) when it fills the chip, else the
// 128x128 kernel.  The wgrad layout (few output tiles, very long contraction) cuts the contraction into nsplit
// chunks: partial tiles go to fp32 slabs in the caller's scratch, then one deterministic reduce pass.
#ifndef G_SMALL_SPLIT_MIN_NK
#define G_SMALL_SPLIT_MIN_NK 6     // 32 = the round-2 rule (A/B)
#endif
struct Plan { bool big; int nsplit; int64_t chunk; };
static int g_policy = 0;  // 0 auto, 1 force the 128x128 kernel, 2 force a 256x256 kernel (tests, A/B)
// which 256x256 kernel serves the "big" launches: 0 = gemm256.hip (8 waves, two per SIMD), 1 = gemm4w.hip (4 waves, one per SIMD: round 5),
// 2 = per launch (default).  Same-box table (profiles/r5_gemm4w_grouped.txt): the 4-wave main loop is 3-16 % faster wherever a workgroup walks
// >= 32 K tiles behind a store-only epilogue (every weight gradient: +12-16 %, the K = 3072 / 4096 dgrads: +3-6 %, fc2 plain: +3 %), equal at
// K = 1024 (both prologue / epilogue bound) and SLOWER behind the tile-reading and VALU-heavy epilogues (fp32 residual 0.52 vs 0.66 PF/s, GELU
// 0.85 vs 0.88): a wave alone on its SIMD has half the loads in flight and half the VALU issue rate of two waves.
#ifndef G_BIG_4W
#define G_BIG_4W 2
#endif
static int g_big4w = G_BIG_4W;
static bool big_is_4w(int layout, int epi, const gemmc::GemmArgs& a, int nsplit);
static int big_launch(int layout, int epi, const gemmc::GemmArgs& a, int nsplit, hipStream_t s) {
  return big_is_4w(layout, epi, a, nsplit) ? crl_gemm4w_launch(layout, epi, a, nsplit, s) : crl_gemm256_launch(layout, epi, a, nsplit, s);
}
static bool big_is_4w(int layout, int epi, const gemmc::GemmArgs& a, int nsplit) {
  const bool store_only = epi == CRL_EPI_BF16 || epi == CRL_EPI_F32 || epi == CRL_EPI_F32_ACC;
  const int nk_wg = nsplit > 1 ? a.kchunk : (a.K + 63) / 64;
  // round 6: the dGELU epilogue only multiplies by the derivative the forward saved, so a lone wave per SIMD no longer loses behind it: 417 us against
  // 433 for the 8-wave kernel at the fc2-dgrad shape, equal at the decoder's (profiles/r6_gemm4w_overlapped_dgelu.txt)
  return g_big4w == 1 || (g_big4w == 2 && ((store_only && nk_wg >= 32) || epi == CRL_EPI_BF16_DGELU || crl_gemm4w_overlaps(layout, epi, a, nsplit)));
}
extern "C" int crl_gemm_set_big_kernel(int which) {
  if (which < 0 || which > 2) { crl_set_error("crl_gemm_set_big_kernel: 0 = gemm256 (8 waves), 1 = gemm4w (4 waves), 2 = per launch"); return -1; }
  g_big4w = which;
  return 0;
}
static Plan plan_gemm(int layout, int epilogue, int64_t M, int64_t N, int64_t K, bool allow_split) {
  Plan p{false, 1, 0};
  const int64_t nk = (K + 63) / 64;
  const bool splittable = allow_split && layout == CRL_TN && (epilogue == CRL_EPI_F32 || epilogue == CRL_EPI_F32_ACC);
  const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256), t128 = ((M + 127) / 128) * ((N + 127) / 128);
  const bool k64 = (layout == CRL_TN) || (K % 64) == 0;
  // min_nk / min_chunk: the 256x256 kernel wants >= 8 K tiles per slab (its prologue stages six half-tiles); the 128x128 kernel with a
  // handful of workgroups is LATENCY-bound at ~1.4 us per K tile (round 3, cfg-1 trace: 9 workgroups x 25 K tiles = 36 us for the Swin
  // stage-3 weight gradients), so even short contractions are worth cutting into slabs of >= 3 K tiles there
  auto split_for = [&](int64_t tiles, int64_t target, int64_t min_nk, int64_t min_chunk) {
    int ns = 1;
    if (splittable && nk >= min_nk && tiles < target) {
      ns = (int)(target / tiles);
      if (ns > 16) ns = 16;
      while (ns > 1 && nk / ns < min_chunk) --ns;
    }
    return ns < 1 ? 1 : ns;
  };
  if (k64 && g_policy != 1 && ((M >= 256 && N >= 256) || g_policy == 2)) {
    const int ncu = crl_gemm_cus();
    const int ns = split_for(t256, ncu, 32, 8);   // one workgroup per CU: aim at one full wave of split tiles
    if (t256 * ns >= (3 * ncu) / 4 || g_policy == 2) { p.big = true; p.nsplit = ns; }
  }
  if (!p.big) p.nsplit = k64 ? split_for(t128, 3 * crl_gemm_cus(), G_SMALL_SPLIT_MIN_NK, 3) : 1;
  p.chunk = (nk + p.nsplit - 1) / p.nsplit;
  p.nsplit = (int)((nk + p.chunk - 1) / p.chunk);
  return p;
}

// Wave quantisation: the 256x256 kernel runs one workgroup per CU, so R*ntn tiles cost about ceil(R*ntn/256) tile times.  When trimming
// a few row tiles lands on a whole number of rounds, the big kernel takes the first 256*R rows and the remaining rows go to the 128x128
// kernel (its contraction cut into slabs when it is long: rem_split).  Whether that pays depends on the contraction: the cut costs a
// second launch + the small kernel (~25-35 us whatever K is, thanks to the slabs), and saves one round of 256x256 tiles (~27 us at K = 1024,
// ~95 us at K = 4096).  Measured on the cfg-3 encoder shapes (scripts/bench_kernels.py quant, round 3): K = 1024 never pays (qkv 265 uncut
// vs 286 us cut, fc1+GELU 457 vs 470), out-width 1024 with K >= 2048 always does (fc1 dgrad 378 -> 319 us, qkv dgrad 289 -> 255, fc2 403 ->
// 349, kv dgrad 204 -> 185); cutting MORE than the last partial round (the round-2 model priced a round of small tiles at 0.32 of a
// big one; it is 0.75) never does.  Returns R, or -1 for "no cut".
static float g_quant_cost = 1.0f;   // multiplies the modelled cost of the remainder launch; < 0: never cut (tuning aid)
// microseconds per round of 256x256 tiles = a + b K / 1024.  The defaults were fitted on one box (round 3); crl_gemm_calibrate() replaces
// them by what THIS device sustains (boxes of the pool differ by ~5 % in their GEMM rate).  The fixed costs of the
// model (a second launch 14 us, a reduce launch 8 us, the 3 us hysteresis) are dispatch-side and stay.
static float g_round_a = 5.0f, g_round_b = 22.0f;
static int g_calibrated = 0;
extern "C" int crl_gemm_set_quant_cost(float c) { g_quant_cost = c; return 0; }
static int rem_split(int epilogue, int64_t rem, int64_t N, int64_t K, int cap);
static int64_t quant_rows(int layout, int epilogue, int64_t M, int64_t N, int64_t K) {
  if (layout == CRL_TN || g_quant_cost < 0.f) return -1;      // (the cut is along the rows of a row-major A: never for the weight-gradient layout)
  const int64_t ntn = (N + 255) / 256, rmax = M / 256, ncu = crl_gemm_cus();
  const int64_t wmax = (((M + 255) / 256) * ntn + ncu - 1) / ncu;   // rounds of the uncut launch
  auto round_us = [](double k) { return (double)g_round_a + (double)g_round_b * k / 1024.0; }; // one round of 256x256 tiles (plain epilogue, sustained clock)
  // uncut: the tiles of the last partial round start while the stragglers of the round before still run (dynamic schedule)
  double best_cost = ((double)wmax - 0.2) * round_us((double)K);
  int64_t best_r = -1;
  for (int64_t w = wmax; w >= 1 && w >= wmax - 2; --w) {
    const int64_t r = (w * ncu) / ntn < rmax ? (w * ncu) / ntn : rmax;
    const int64_t rem = M - 256 * r;
    if (r < 1 || rem <= 0) continue;
    const int ns = rem_split(epilogue, rem, N, K, 8);
    const int64_t t128 = ((rem + 127) / 128) * ((N + 127) / 128) * ns, full = t128 / (2 * ncu), part = t128 - full * 2 * ncu;
    const double small = 14.0 + (ns > 1 ? 8.0 : 0.0) + round_us((double)K / ns) * (0.75 * (double)full + (part == 0 ? 0.0 : part <= ncu ? 0.5 : 0.75));
    const double cost = (double)((r * ntn + ncu - 1) / ncu) * round_us((double)K) + (double)g_quant_cost * small;
    if (cost < best_cost - 3.0) { best_cost = cost; best_r = r; }
  }
  return best_r;
}
// The remainder launch has few tiles (360 rows x N of the cfg-3 encoder: 24 to 96) and, for fc2 / the fc1 and qkv dgrads, a long
// contraction: 24 workgroups walking 64 K tiles leave 90 % of the chip idle for ~60 us.  Its contraction is then cut into up to 8
// chunks (fp32 slabs in the caller's scratch + a reduce that applies the epilogue); 1 = no split.
#ifndef G_REM_SPLIT
#define G_REM_SPLIT 1
#endif
static bool slab_epilogue(int epilogue);
static int rem_split(int epilogue, int64_t rem, int64_t N, int64_t K, int cap) {
  // round 4: the fp32 store / accumulate epilogues too -- the 360 remainder rows of the fused cross-attention K/V dgrad (contraction 20 480,
  // fp32 output) ran 0.4 ms on 24 CUs
  if (!G_REM_SPLIT || !slab_epilogue(epilogue) || (K % 64) != 0) return 1;
  const int64_t nk = K / 64, tiles = ((rem + 127) / 128) * ((N + 127) / 128);
  // (round 6: the K = 1024 launches too -- nk >= 16, slabs of >= 2 K tiles, a cheaper modelled launch -- cut proj + residual 160 -> 155 us and dgrad proj
  // 121 -> 115 us and left the step where it was: profiles/r6_gemm_rem_split_k1024.txt.  Not taken.)
  if (nk < 32 || tiles >= 128) return 1;
  int ns = (int)(384 / tiles);
  if (ns > cap) ns = cap;
  while (ns > 1 && nk / ns < 8) --ns;
  return ns;
}
// The same treatment for a WHOLE forward / dgrad GEMM with few output tiles and a very long contraction -- the LM-head dgrad of a small
// batch: [254, 50304] x [50304, 768] = 12 tiles of 128 x 128 walking 786 K tiles each on 12 of 256 CUs (cfg-1: 0.4 ms of a 7 ms step).
// Up to 32 chunks of the contraction as fp32 slabs + the reduce that applies the epilogue.  The same latency argument holds from 32 K tiles
// on (cfg-1 trace: the Swin stage-4 fc2 dgrad, 24 tiles x 48 K tiles, took 41 us on 24 CUs).
#ifndef G_FEW_TILES_MIN_NK
#define G_FEW_TILES_MIN_NK 16     // 128 = the first form of the rule, 32 the second (A/B)
#endif
static bool slab_epilogue(int epilogue) {
  return epilogue == CRL_EPI_BF16 || epilogue == CRL_EPI_F32_RESID || epilogue == CRL_EPI_F32 || epilogue == CRL_EPI_F32_ACC;
}
static int few_tiles_split(int layout, int epilogue, int64_t M, int64_t N, int64_t K) {
  if (layout == CRL_TN || !slab_epilogue(epilogue) || (K % 64) != 0) return 1;
  const int64_t nk = K / 64, tiles = ((M + 127) / 128) * ((N + 127) / 128);
  // round 4: 128 ... 255 tiles (a quarter to a half of the 512 workgroup slots) with a long contraction are cut in two or three as well
  // (cfg-2: the decoder's 4088-row GEMMs, 192 tiles)
  if (nk < G_FEW_TILES_MIN_NK || tiles >= 256 || (tiles >= 128 && nk < 64)) return 1;
  int ns = (int)((tiles >= 128 ? 512 : 384) / tiles);
  if (ns > 32) ns = 32;
  while (ns > 1 && nk / ns < 4) --ns;       // slabs of >= 4 K tiles
  return ns;
}

// Round 6: a forward / dgrad GEMM whose 256x256 tiles fill at most HALF the chip and whose contraction is long (>= 64 K tiles) -- the decoder's 8184-row
// fc2 and fc1 dgrad (128 tiles x 64 K tiles) and the LM-head dgrad (128 tiles x 786) -- used to run as 512 tiles of the 128x128 kernel at half the
// rate (the LM-head dgrad: 1.1 ms = 0.76 PF/s).  Now: CUs / tiles contraction slices of >= 32 K tiles on the 4-wave 256x256 kernel (fp32 slabs in the
// caller's scratch) + the slab reduce that applies the epilogue.  1 = no split.
static int half_chip_split(int layout, int epilogue, int64_t M, int64_t N, int64_t K) {
  if (layout == CRL_TN || !slab_epilogue(epilogue) || (K % 64) != 0 || M < 256 || N < 256) return 1;
  const int64_t nk = K / 64, t256 = ((M + 255) / 256) * ((N + 255) / 256), ncu = crl_gemm_cus();
  if (nk < 64 || t256 * 2 > ncu || t256 * 8 < ncu) return 1;
  int ns = (int)(ncu / t256);
  while (ns > 1 && nk / ns < 32) --ns;
  return ns;
}

namespace {
// bf16 values in (-1, 1) from a hash of the element index: calibration operands (constant data would run at a higher clock than real activations)
__global__ void calib_fill_kernel(uint32_t* __restrict__ p, size_t n2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  uint32_t x = (uint32_t)i * 0x9E3779B1u + 0x7F4A7C15u;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  const float a = (float)(int)(x & 0xffff) * (1.0f / 32768.0f) - 1.0f, b = (float)(int)(x >> 16) * (1.0f / 32768.0f) - 1.0f;
  p[i] = pack_bf2(a, b);
}
}  // namespace

extern "C" int crl_gemm_model(float* round_a_us, float* round_b_us, int* calibrated) {
  if (round_a_us) *round_a_us = g_round_a;
  if (round_b_us) *round_b_us = g_round_b;
  if (calibrated) *calibrated = g_calibrated;
  return 0;
}
extern "C" size_t crl_gemm_calibrate_ws_bytes() { return ((size_t)32768 * 4096 + (size_t)1024 * 4096 + (size_t)32768 * 1024) * 2; }
// One-off, SYNCHRONISING: times one and two rounds of 256x256 tiles (N = 1024: 256 / 512 tiles) at K = 1024 and K = 4096 on random bf16
// operands in `ws` and refits round_us(K) = a + b K / 1024 of the wave-quantisation model.  Returns 0 (refitted), 1 (measurement
// implausible: defaults kept), < 0 on error.
extern "C" int crl_gemm_calibrate(void* ws, size_t ws_bytes, void* stream) {
  CRL_CHECK(ws && ws_bytes >= crl_gemm_calibrate_ws_bytes() && ((uintptr_t)ws % 16) == 0, "crl_gemm_calibrate: needs %zu bytes of 16-byte aligned scratch", crl_gemm_calibrate_ws_bytes());
  hipStream_t s = as_stream(stream);
  u16* A = (u16*)ws;
  u16* B = A + (size_t)32768 * 4096;
  u16* C = B + (size_t)1024 * 4096;
  const size_t n2 = ((size_t)32768 * 4096 + (size_t)1024 * 4096) / 2;
  calib_fill_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, s>>>((uint32_t*)ws, n2);
  CRL_LAUNCH_CHECK("crl_gemm_calibrate(fill)");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { crl_set_error("crl_gemm_calibrate: cannot create events"); return -2; }
  const float saved_cost = g_quant_cost;
  const int saved_policy = g_policy;
  g_quant_cost = -1.f;      // the timed launches are never cut
  g_policy = 2;
  double r[2] = {0, 0};
  int rc = 0;
  for (int ki = 0; ki < 2 && rc == 0; ++ki) {
    const int64_t K = ki == 0 ? 1024 : 4096;
    double t[2] = {0, 0};
    for (int rounds = 1; rounds <= 2 && rc == 0; ++rounds) {
      const int64_t M = 16384 * rounds;
      auto run = [&]() { return crl_gemm_bf16(CRL_NT, CRL_EPI_BF16, M, 1024, K, A, 4096, B, 4096, nullptr, C, 1024, nullptr, 0, nullptr, 0, 1.f, 0, nullptr, 0, stream); };
      for (int w = 0; w < 3 && rc == 0; ++w) rc = run();
      hipEventRecord(e0, s);
      for (int i = 0; i < 8 && rc == 0; ++i) rc = run();
      hipEventRecord(e1, s);
      if (hipEventSynchronize(e1) != hipSuccess) rc = -2;
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      t[rounds - 1] = (double)ms * 1000.0 / 8.0;
    }
    r[ki] = t[1] - t[0];
  }
  g_quant_cost = saved_cost;
  g_policy = saved_policy;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (rc) return rc;
  const double b = (r[1] - r[0]) / 3.0, a = r[0] - b;
  // plausible: within a factor 2 of the fitted defaults and a non-negative fixed part
  if (!(b > 11.0 && b < 44.0 && a > -5.0 && a < 30.0)) return 1;
  g_round_a = (float)(a < 0.0 ? 0.0 : a);
  g_round_b = (float)b;
  g_calibrated = 1;
  return 0;
}

extern "C" int crl_gemm_set_policy(int policy) {
  if (policy < 0 || policy > 2) { crl_set_error("crl_gemm_set_policy: bad policy %d", policy); return -1; }
  g_policy = policy;
  return 0;
}

// Bias gradient with the weight gradient (CRL_TN with aux != NULL): partial column sums of A behind the split-K slabs -- up to 16 splits x 4 column
// tiles x 2 waves rows of M floats from the 4-wave kernel (gen_gemm4w.py colsum_block), or the 64 rows of the stand-alone column-sum kernel when
// another kernel serves the launch
static constexpr size_t CS_ROWS = 128;
static size_t tn_slab_bytes(const Plan& p, int64_t M, int64_t N) { return p.nsplit > 1 ? (((size_t)p.nsplit * M * N * sizeof(float) + 255) & ~(size_t)255) : 0; }
extern "C" int crl_colsum_bf16(const void* X, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, void* ws, void* stream);
namespace {
__global__ void cs_reduce_kernel(const float* __restrict__ ws, int P, int M, float* __restrict__ out, int acc) {
  cs_reduce_body(ws, P, M, out, acc, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}
}  // namespace

extern "C" size_t crl_gemm_ws_bytes(int layout, int epilogue, int64_t M, int64_t N, int64_t K) {
  const Plan p = plan_gemm(layout, epilogue, M, N, K, true);
  if (layout == CRL_TN) return tn_slab_bytes(p, M, N) + CS_ROWS * (size_t)M * sizeof(float);
  if (p.nsplit > 1) return (size_t)p.nsplit * M * N * sizeof(float);
  if (!p.big && g_policy == 0) {
    const int nh = half_chip_split(layout, epilogue, M, N, K);
    if (nh > 1) return (size_t)nh * M * N * sizeof(float);
    const int ns = few_tiles_split(layout, epilogue, M, N, K);
    if (ns > 1) return (size_t)ns * M * N * sizeof(float);
  }
  if (p.big) {
    const int64_t r = quant_rows(layout, epilogue, M, N, K);
    if (r > 0) {
      const int ns = rem_split(epilogue, M - 256 * r, N, K, 8);
      if (ns > 1) return (size_t)ns * (M - 256 * r) * N * sizeof(float);
    }
  }
  return 0;
}

extern "C" int crl_gemm_bf16(int layout, int epilogue, int64_t M, int64_t N, int64_t K,
                             const void* A, int64_t lda, const void* B, int64_t ldb,
                             const float* bias, void* C, int64_t ldc, void* aux, int64_t ldaux,
                             const float* resid, int64_t ldr, float colscale, int64_t colscale_cols, void* ws, size_t ws_bytes, void* stream) {
  CRL_CHECK(colscale_cols == 0 || (epilogue == CRL_EPI_BF16 && colscale_cols > 0 && colscale_cols <= N && (colscale_cols % 4) == 0),
            "crl_gemm_bf16: the column scale applies to the first colscale_cols (multiple of 4, <= N) columns of the plain bf16 epilogue");
  CRL_CHECK(M > 0 && N > 0 && K > 0, "crl_gemm_bf16: empty problem %lld x %lld x %lld", (long long)M, (long long)N, (long long)K);
  CRL_CHECK(A && B && C, "crl_gemm_bf16: null operand");
  CRL_CHECK((N % 4) == 0 && (ldc % 4) == 0, "crl_gemm_bf16: N (%lld) and ldc (%lld) must be multiples of 4", (long long)N, (long long)ldc);
  CRL_CHECK((lda % 8) == 0 && (ldb % 8) == 0, "crl_gemm_bf16: lda/ldb must be multiples of 8 (16-byte rows)");
  CRL_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0, "crl_gemm_bf16: operands must be 16-byte aligned");
  int bk = 64;
  int64_t a_rows, a_cols, b_rows, b_cols;
  if (layout == CRL_NT) { a_rows = M; a_cols = K; b_rows = N; b_cols = K; }
  else if (layout == CRL_NN) { a_rows = M; a_cols = K; b_rows = K; b_cols = N; }
  else if (layout == CRL_TN) { a_rows = K; a_cols = M; b_rows = K; b_cols = N; }
  else { crl_set_error("crl_gemm_bf16: bad layout %d", layout); return -1; }
  if (layout != CRL_TN) {
    CRL_CHECK((K % 32) == 0, "crl_gemm_bf16: K (%lld) must be a multiple of 32 for NT/NN", (long long)K);
    if (K % 64) bk = 32;
  }
  CRL_CHECK(lda >= a_cols && ldb >= b_cols, "crl_gemm_bf16: leading dimension smaller than the row");
  if (epilogue == CRL_EPI_BF16_GELU || epilogue == CRL_EPI_BF16_DGELU) CRL_CHECK(aux != nullptr && (ldaux % 4) == 0, "crl_gemm_bf16: aux required");
  if (epilogue == CRL_EPI_F32_RESID) CRL_CHECK(resid != nullptr && (ldr % 4) == 0, "crl_gemm_bf16: resid required");
  const uint64_t ab = (uint64_t)((a_rows - 1) * lda + a_cols) * 2, bb = (uint64_t)((b_rows - 1) * ldb + b_cols) * 2;
  // DMA offsets are 32-bit and tiles may overhang by up to 127 rows
  CRL_CHECK((uint64_t)(a_rows + 128) * lda * 2 < (1ull << 32) && (uint64_t)(b_rows + 128) * ldb * 2 < (1ull << 32),
            "crl_gemm_bf16: operand larger than 4 GiB");
  {
    // the epilogue addresses C / aux / resid with 32-bit byte offsets through bounds-checked buffer descriptors (gemm_epilogue.h)
    const uint64_t esz = epilogue >= CRL_EPI_F32_RESID ? 4 : 2;
    CRL_CHECK((uint64_t)(M + 256) * (uint64_t)ldc * esz < (1ull << 32), "crl_gemm_bf16: output larger than 4 GiB");
    if (aux) CRL_CHECK((uint64_t)(M + 256) * (uint64_t)ldaux * 2 < (1ull << 32), "crl_gemm_bf16: aux larger than 4 GiB");
    if (resid) CRL_CHECK((uint64_t)(M + 256) * (uint64_t)ldr * 4 < (1ull << 32), "crl_gemm_bf16: resid larger than 4 GiB");
  }
  GemmArgs a;
  a.A = (const u16*)A; a.B = (const u16*)B; a.bias = bias; a.C = C; a.aux = aux; a.resid = resid;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = (int)ldc; a.ldaux = (int)ldaux; a.ldr = (int)ldr;
  a.a_bytes = (uint32_t)ab; a.b_bytes = (uint32_t)bb;
  a.sched = nullptr;
  a.cs_ws = nullptr; a.cs_ntn = 0;
  a.colscale = colscale; a.colscale_cols = (int)colscale_cols;
  a.ntm = (int)((M + BM - 1) / BM); a.ntn = (int)((N + BN - 1) / BN);
  hipStream_t s = as_stream(stream);
  Plan p = plan_gemm(layout, epilogue, M, N, K, ws != nullptr);
  if (p.nsplit > 1 && ws_bytes < (size_t)p.nsplit * M * N * sizeof(float)) p = plan_gemm(layout, epilogue, M, N, K, false);
  // CRL_TN with aux: aux[m] (+)= sum_k A[k][m], the bias gradient of the Linear whose weight gradient this is (ldaux != 0: accumulate)
  float* const cs_out = layout == CRL_TN ? (float*)aux : nullptr;
  float* cs_part = nullptr;
  if (cs_out) {
    CRL_CHECK(epilogue == CRL_EPI_F32 || epilogue == CRL_EPI_F32_ACC, "crl_gemm_bf16: the column sums ride the fp32 weight-gradient epilogues only");
    const size_t off = tn_slab_bytes(p, M, N);
    CRL_CHECK(ws && ws_bytes >= off + CS_ROWS * (size_t)M * sizeof(float), "crl_gemm_bf16: the column sums need crl_gemm_ws_bytes() of scratch");
    cs_part = (float*)((char*)ws + off);
  }
  // after the GEMM: the partial rows the 4-wave kernel wrote -> aux; any other kernel: the stand-alone column-sum pass over A
  auto finish_colsum = [&](bool fused, int nsplit_run) -> int {
    if (!cs_out) return 0;
    if (!fused) return crl_colsum_bf16(A, K, M, lda, cs_out, ldaux != 0, cs_part, stream);
    cs_reduce_kernel<<<(unsigned)((M + 255) / 256), 256, 0, s>>>(cs_part, nsplit_run * a.cs_ntn * 2, (int)M, cs_out, ldaux != 0);
    CRL_LAUNCH_CHECK("crl_gemm_bf16(column-sum reduce)");
    return 0;
  };
  if (bk == 32) { p.big = false; p.nsplit = 1; p.chunk = (K + 31) / 32; }
  a.kchunk = (int)p.chunk;
  a.slab_stride = 0;
  if (p.big) { a.ntm = (int)((M + 255) / 256); a.ntn = (int)((N + 255) / 256); }
  bool cs_fused = false;
  if (cs_out && p.big) {
    GemmArgs probe = a;
    if (p.nsplit > 1) probe.kchunk = (int)p.chunk;
    if (big_is_4w(CRL_TN, p.nsplit > 1 ? CRL_EPI_F32 : epilogue, probe, p.nsplit)) {
      cs_fused = true;
      a.cs_ws = cs_part;
      a.cs_ntn = a.ntn >= 4 ? 4 : a.ntn >= 2 ? 2 : 1;
    }
  }
  if (p.nsplit > 1) {
    GemmArgs b = a;
    b.C = ws; b.ldc = (int)N; b.slab_stride = (size_t)M * N;
    if (int rc = p.big ? big_launch(CRL_TN, CRL_EPI_F32, b, p.nsplit, s) : launch_epi<CRL_TN>(b, CRL_EPI_F32, bk, p.nsplit, s)) return rc;
    const size_t n4 = (size_t)M * N / 4;
    const unsigned nb = (unsigned)((n4 + 255) / 256);
    if (cs_fused) {      // the slab reduce and the column-sum reduce in one launch
      splitk_reduce_kernel<<<nb + (unsigned)((M + 255) / 256), 256, 0, s>>>((const float*)ws, p.nsplit, (size_t)M * N, (float*)C, (int)M, (int)N, (int)ldc,
                                                                           epilogue == CRL_EPI_F32_ACC, nb, cs_part, p.nsplit * a.cs_ntn * 2, cs_out, ldaux != 0);
      CRL_LAUNCH_CHECK("crl_gemm_bf16(splitk + column-sum reduce)");
      return 0;
    }
    splitk_reduce_kernel<<<nb, 256, 0, s>>>((const float*)ws, p.nsplit, (size_t)M * N, (float*)C, (int)M, (int)N, (int)ldc, epilogue == CRL_EPI_F32_ACC);
    CRL_LAUNCH_CHECK("crl_gemm_bf16(splitk reduce)");
    return finish_colsum(false, p.nsplit);
  }
  if (cs_out) {      // weight gradient without a contraction split: one launch (never row-cut: quant_rows leaves CRL_TN alone), then the column sums
    if (int rc = p.big ? big_launch(layout, epilogue, a, 1, s) : launch_epi<CRL_TN>(a, epilogue, bk, 1, s)) return rc;
    return finish_colsum(cs_fused, 1);
  }
  // slabs of a cut contraction -> C with the epilogue applied (plain fp32 store / accumulate, or bias + bf16 / bias + residual)
  auto reduce_slabs = [&](const GemmArgs& full, int nsl, const char* what) -> int {
    const size_t n4 = (size_t)M * N / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (epilogue == CRL_EPI_F32 || epilogue == CRL_EPI_F32_ACC)
      splitk_reduce_kernel<<<blocks, 256, 0, s>>>((const float*)ws, nsl, (size_t)M * N, (float*)full.C, (int)M, (int)N, (int)ldc, epilogue == CRL_EPI_F32_ACC);
    else if (epilogue == CRL_EPI_BF16)
      splitk_reduce_epi_kernel<CRL_EPI_BF16><<<blocks, 256, 0, s>>>((const float*)ws, nsl, (size_t)M * N, (int)M, (int)N, full.bias, full.C, (int)ldc, nullptr, 0, full.colscale, full.colscale_cols);
    else
      splitk_reduce_epi_kernel<CRL_EPI_F32_RESID><<<blocks, 256, 0, s>>>((const float*)ws, nsl, (size_t)M * N, (int)M, (int)N, full.bias, full.C, (int)ldc, full.resid, (int)ldr);
    CRL_LAUNCH_CHECK(what);
    return 0;
  };
  if (p.big) {
    const int64_t best_r = quant_rows(layout, epilogue, M, N, K);
    if (best_r < 0) return big_launch(layout, epilogue, a, 1, s);
    const int64_t m1 = 256 * best_r;
    GemmArgs big = a;
    big.M = (int)m1; big.ntm = (int)best_r;
    GemmArgs rest = a;
    const bool c32 = epilogue >= CRL_EPI_F32_RESID;
    rest.A = a.A + m1 * lda;
    rest.C = (char*)a.C + m1 * ldc * (c32 ? 4 : 2);
    if (a.aux) rest.aux = (char*)a.aux + m1 * ldaux * 2;
    if (a.resid) rest.resid = a.resid + m1 * ldr;
    rest.M = (int)(M - m1);
    rest.a_bytes = (uint32_t)(((M - m1 - 1) * lda + K) * 2);
    rest.ntm = (int)((M - m1 + BM - 1) / BM); rest.ntn = (int)((N + BN - 1) / BN);
    rest.kchunk = (int)((K + 63) / 64);
    const int64_t rem = M - m1;
    auto launch_rest = [&]() -> int {
      const int ns = rem_split(epilogue, rem, N, K, 8);
      if (ns > 1 && ws && ws_bytes >= (size_t)ns * rem * N * sizeof(float)) {
        GemmArgs sl = rest;
        sl.bias = nullptr; sl.aux = nullptr; sl.resid = nullptr;
        sl.C = ws; sl.ldc = (int)N; sl.slab_stride = (size_t)rem * N;
        sl.kchunk = (int)((K / 64 + ns - 1) / ns);
        const int nsl = (int)((K / 64 + sl.kchunk - 1) / sl.kchunk);
        if (int rc = (layout == CRL_NT ? launch_epi<CRL_NT>(sl, CRL_EPI_F32, 64, nsl, s) : launch_epi<CRL_NN>(sl, CRL_EPI_F32, 64, nsl, s))) return rc;
        const unsigned blocks = (unsigned)(((size_t)rem * N / 4 + 255) / 256);
        if (epilogue == CRL_EPI_F32 || epilogue == CRL_EPI_F32_ACC)
          splitk_reduce_kernel<<<blocks, 256, 0, s>>>((const float*)ws, nsl, (size_t)rem * N, (float*)rest.C, (int)rem, (int)N, (int)ldc, epilogue == CRL_EPI_F32_ACC);
        else if (epilogue == CRL_EPI_BF16)
          splitk_reduce_epi_kernel<CRL_EPI_BF16><<<blocks, 256, 0, s>>>((const float*)ws, nsl, (size_t)rem * N, (int)rem, (int)N, rest.bias, rest.C, (int)ldc, nullptr, 0, a.colscale, a.colscale_cols);
        else
          splitk_reduce_epi_kernel<CRL_EPI_F32_RESID><<<blocks, 256, 0, s>>>((const float*)ws, nsl, (size_t)rem * N, (int)rem, (int)N, rest.bias, rest.C, (int)ldc, rest.resid, (int)ldr);
        CRL_LAUNCH_CHECK("crl_gemm_bf16(remainder reduce)");
        return 0;
      }
      switch (layout) {
        case CRL_NT: return launch_epi<CRL_NT>(rest, epilogue, 64, 1, s);
        default: return launch_epi<CRL_NN>(rest, epilogue, 64, 1, s);
      }
    };
    // G_REM_FIRST (default): the remainder launch (+ its reduce) is enqueued first -- its few workgroups start while the chip still drains the
    // previous kernel and the persistent launch ends the call at full width; 0 = the round-3 order (A/B: profiles/r6_gemm_rem_order.txt)
#ifndef G_REM_FIRST
#define G_REM_FIRST 1
#endif
    if (G_REM_FIRST) {
      if (int rc = launch_rest()) return rc;
      return big_launch(layout, epilogue, big, 1, s);
    }
    if (int rc = big_launch(layout, epilogue, big, 1, s)) return rc;
    return launch_rest();
  }
  if (g_policy == 0 && bk == 64) {
    const int nh = half_chip_split(layout, epilogue, M, N, K);
    if (nh > 1 && ws && ws_bytes >= (size_t)nh * M * N * sizeof(float)) {
      GemmArgs sl = a;
      sl.bias = nullptr; sl.aux = nullptr; sl.resid = nullptr;
      sl.C = ws; sl.ldc = (int)N; sl.slab_stride = (size_t)M * N;
      sl.ntm = (int)((M + 255) / 256); sl.ntn = (int)((N + 255) / 256);
      sl.kchunk = (int)((K / 64 + nh - 1) / nh);
      const int nsl = (int)((K / 64 + sl.kchunk - 1) / sl.kchunk);
      if (int rc = big_launch(layout, CRL_EPI_F32, sl, nsl, s)) return rc;
      return reduce_slabs(a, nsl, "crl_gemm_bf16(half-chip split reduce)");
    }
    const int ns = few_tiles_split(layout, epilogue, M, N, K);
    if (ns > 1 && ws && ws_bytes >= (size_t)ns * M * N * sizeof(float)) {
      GemmArgs sl = a;
      sl.bias = nullptr; sl.aux = nullptr; sl.resid = nullptr;
      sl.C = ws; sl.ldc = (int)N; sl.slab_stride = (size_t)M * N;
      sl.kchunk = (int)((K / 64 + ns - 1) / ns);
      const int nsl = (int)((K / 64 + sl.kchunk - 1) / sl.kchunk);
      if (int rc = (layout == CRL_NT ? launch_epi<CRL_NT>(sl, CRL_EPI_F32, 64, nsl, s) : launch_epi<CRL_NN>(sl, CRL_EPI_F32, 64, nsl, s))) return rc;
      return reduce_slabs(a, nsl, "crl_gemm_bf16(few-tiles split reduce)");
    }
  }
  switch (layout) {
    case CRL_NT: return launch_epi<CRL_NT>(a, epilogue, bk, 1, s);
    case CRL_NN: return launch_epi<CRL_NN>(a, epilogue, bk, 1, s);
    default: return launch_epi<CRL_TN>(a, epilogue, bk, 1, s);
  }
}
