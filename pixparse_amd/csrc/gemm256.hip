// 256x256x64 bf16 MFMA GEMM, 8 waves, 8-phase software pipeline (guide §5 "256² 8-phase template"):
// the large-shape path of crl_gemm_bf16 for all three layouts.
//
// Workgroup = 512 threads = 8 waves, one per CU (128 KiB LDS = 2 K-tile buffers x 4 half-tiles of
// 16 KiB: A rows 0-127, B cols 0-127, B cols 128-255, A rows 128-255).  Wave (wr, wc) = (wave>>2, wave&3)
// owns rows {64wr..+64} of BOTH A halves and cols {32wc..+32} of BOTH B halves, so quadrant
// Q(qm, qn) of its 128x64 output uses exactly half-tile A[qm] and B[qn] for every wave -- half-tiles
// are consumed in a fixed order (A0,B0 | B1 | A1 | -), which is what lets the LDS-DMA of later
// K tiles land in a buffer one phase after its last read.
//
// Per K-tile, four phases of 16 MFMAs (one quadrant x K=64):
//   phase 1: ds_read A0 (8 x b128) + B0 (4)   MFMA Q00      issue half-tile h+7
//   phase 2: ds_read B1 (4)                   MFMA Q01      ...
//   phase 3: ds_read A1 (8)                   MFMA Q11
//   phase 4: (B0 kept in registers)           MFMA Q10      s_waitcnt vmcnt(6): all but the 3 youngest
//                                                           half-tiles have landed -> next K-tile complete
// Each phase: {ds_reads, 2 x buffer_load..lds, [vmcnt(6)], s_barrier, MFMA cluster, s_barrier}.
// Persistent: at most one workgroup per CU (grid = min(tiles, 256) when there is no split-K); after the K loop of a tile
// the first six half-tiles of the NEXT tile are issued before the epilogue, which only touches registers and global memory.
// Hazards: RAW -- a buffer is read one phase after the vmcnt+barrier that retires its DMA;
// WAR -- a half-tile is re-staged at least one full phase (two barriers) after its last ds_read, whose
// data has been consumed by MFMAs before the closing barrier of that phase.
#include <type_traits>
#include "gemm_common.h"
#include "gemm_epilogue.h"

namespace {
using namespace gemmc;

constexpr int T256 = 512;
#define BAR() asm volatile("s_barrier" ::: "memory")

// -DG_TIMING=<workgroup>: wave 0 of that workgroup records s_memtime at six points of every tile iteration (debug builds only:
// scripts/gemm_timeline.py reads them through crl_gemm_debug_read)
#ifdef G_TIMING
__device__ unsigned long long g_tm[16][8];
#define TM(pt) do { if ((int)blockIdx.x == G_TIMING && wave == 0 && lane == 0 && tile_iter < 16) g_tm[tile_iter][pt] = __builtin_readcyclecounter(); } while (0)
#else
#define TM(pt) do { } while (0)
#endif

template <int LAYOUT, int EPI>
__global__ __launch_bounds__(T256, 2) void gemm256_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool A_TR = (LAYOUT == CRL_TN);
  constexpr bool B_TR = (LAYOUT != CRL_NT);
  constexpr int HT = 16384, BUF = 4 * HT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

#ifndef G_PERSIST
#define G_PERSIST 1   // one workgroup per CU walks the tile list; the next tile's first K tiles are in flight during the epilogue (same-box A/B: +3..5 % on the K=1024 NT shapes, +0.1 % on the whole cfg-3 step)
#endif
  const int ntiles = g.ntm * g.ntn;
  // position in the XCD-aware tile order.  Static schedule (split-K launches, launches with a tile or less per workgroup): blockIdx.x,
  // advanced by gridDim.x.  Dynamic schedule (g.sched, every persistent launch): pulled from the ticket counters by lane 0 of wave 0
  // and handed to the other waves through s_next -- the pull for the NEXT tile is issued when a tile starts and read after its K loop.
  __shared__ int s_next, s_list;      // s_list: the list (XCD) this workgroup pulls from (kept out of the registers: the epilogues are full)
  const bool dyn = g.sched != nullptr;
  uint32_t ticket = 0;
  int logical = blockIdx.x;
  if (dyn) {
    if (wave == 0 && lane == 0) {
      int my_list = sched_xcd();
      ticket = sched_pull(g.sched + my_list);
      s_next = sched_resolve(g.sched, my_list, ticket, ntiles);
      s_list = my_list;
    }
    __syncthreads();
    logical = __builtin_amdgcn_readfirstlane(s_next);
    if (logical < 0) {                // started after the resident workgroups had emptied the queue
      if (tid == 0) sched_leave(g.sched, gridDim.x);
      return;
    }
  }
  int m0, n0;
  auto set_tile = [&](int l) {
    const int t = xcd_remap(l, ntiles);
    int tm, tn;
    tile_coords(t, g.ntm, g.ntn, tm, tn);
    m0 = tm * 256;
    n0 = tn * 256;
  };
  set_tile(logical);

  const u32x4 ra = make_srd(g.A, g.a_bytes);
  const u32x4 rb = make_srd(g.B, g.b_bytes);
  const uint32_t smem_base = lds_addr_of(smem);
  const u32x4 rzero = make_srd(g.A, 0);   // zero records: every load through it returns zeros (K tiles past the end)

  const int nk_all = (g.K + 63) / 64;
  const int kt0 = blockIdx.y * g.kchunk;
  const int nk = min(nk_all, kt0 + g.kchunk) - kt0;   // K tiles of this split

  f32x4 acc[2][2][4][2];

  // LDS-DMA addressing: ONE per-thread byte offset per operand (VGPR) + a wave-uniform part (SGPR soffset) that
  // carries the chunk, the half-tile and the K advance -- keeps the loop free of per-site address registers.
  //   KM image chunk c = it*512 + tid: row = 64 it + (tid>>3), slot' = tid&7   (the swizzle does not depend on it)
  //   TR image chunk c = it*512 + tid: krow = 32 it + (tid>>4), chunk' = tid&15
  uint32_t voffA, voffB, stepA, stepB;   // stepX = byte distance between the two chunks a thread stages
  if constexpr (A_TR) { const int kr = tid >> 4; voffA = (uint32_t)kr * g.lda * 2u + (uint32_t)(((tid & 15) ^ tr_swz(kr)) * 16); stepA = 32u * g.lda * 2u; }
  else { const int r = tid >> 3; voffA = (uint32_t)r * g.lda * 2u + (uint32_t)(((tid & 7) ^ km_swz<64>(r)) * 16); stepA = 64u * g.lda * 2u; }
  if constexpr (B_TR) { const int kr = tid >> 4; voffB = (uint32_t)kr * g.ldb * 2u + (uint32_t)(((tid & 15) ^ tr_swz(kr)) * 16); stepB = 32u * g.ldb * 2u; }
  else { const int r = tid >> 3; voffB = (uint32_t)r * g.ldb * 2u + (uint32_t)(((tid & 7) ^ km_swz<64>(r)) * 16); stepB = 64u * g.ldb * 2u; }

  // half-tile h (global sequence): K tile h>>2, slot j = h&3 : 0 = A0, 1 = B0, 2 = B1, 3 = A1
  auto issue = [&](int tile, int j) {
    const uint32_t lds = smem_base + (uint32_t)((tile & 1) * BUF + j * HT) + (uint32_t)wave * 1024u;
    const uint32_t k0 = (uint32_t)(kt0 + tile) * 64u;
    const bool live = tile < nk;          // tiles past this split's range contribute zeros (odd tile counts, pipeline tail)
    if (j == 0 || j == 3) {
      const uint32_t r0 = (uint32_t)m0 + (j == 3 ? 128u : 0u);
      const uint32_t soff = A_TR ? (k0 * g.lda + r0) * 2u : (r0 * g.lda + k0) * 2u;
      const u32x4 r = live ? ra : rzero;
      dma16(r, lds, voffA, soff);
      dma16(r, lds + 8192u, voffA, soff + stepA);
    } else {
      const uint32_t c0 = (uint32_t)n0 + (j == 2 ? 128u : 0u);
      const uint32_t soff = B_TR ? (k0 * g.ldb + c0) * 2u : (c0 * g.ldb + k0) * 2u;
      const u32x4 r = live ? rb : rzero;
      dma16(r, lds, voffB, soff);
      dma16(r, lds + 8192u, voffB, soff + stepB);
    }
  };
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  auto readA = [&](int buf, int qm) {
    const char* l = smem + buf * BUF + (qm ? 3 : 0) * HT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (A_TR) fa[i][ks] = frag_tr(l, 64 * wr + 16 * i, ks, lane);
        else fa[i][ks] = frag_km<64>(l, 64 * wr + 16 * i, ks, lane);
      }
  };
  auto readB = [&](int buf, int qn, bf16x8 (&f)[2][2]) {
    const char* l = smem + buf * BUF + (qn ? 2 : 1) * HT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (B_TR) f[j][ks] = frag_tr(l, 32 * wc + 16 * j, ks, lane);
        else f[j][ks] = frag_km<64>(l, 32 * wc + 16 * j, ks, lane);
      }
  };
  auto mma = [&](f32x4 (&c)[4][2], const bf16x8 (&b)[2][2]) {
#ifndef G_PRIO
#define G_PRIO 1
#endif
#ifndef G_ORDER
#define G_ORDER 0
#endif
    __builtin_amdgcn_s_setprio(G_PRIO);
#if G_ORDER == 0
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i][j]) : "v"(b[j][ks]), "v"(fa[i][ks]));
#else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i][j]) : "v"(b[j][ks]), "v"(fa[i][ks]));
#endif
    __builtin_amdgcn_s_setprio(0);
  };

#ifndef G_CU_STAGGER
#define G_CU_STAGGER 0
#endif
#ifndef G_CUS_KT
#define G_CUS_KT 22      // s_sleep units (64 clocks) per K tile: about half of a K tile's 1.4 us
#endif
  // Every persistent workgroup walks equally long tiles, so all 256 CUs reach their epilogues TOGETHER: the chip alternates
  // between an MFMA-bound phase with HBM idle and an HBM-bound phase (tile-sized epilogue traffic of every CU at once) with
  // the matrix pipes idle.  Half of the workgroups therefore start half a tile period late, once; the two populations then
  // run their epilogues under each other's K loops for the rest of the tile walk.
  if (G_CU_STAGGER && G_PERSIST && gridDim.y == 1 && ntiles >= 2 * (int)gridDim.x && ((blockIdx.x >> 3) & 1)) {
    constexpr int base = EPI == CRL_EPI_BF16 ? 190 : EPI == CRL_EPI_BF16_GELU ? 340 : EPI == CRL_EPI_BF16_DGELU ? 310 : 480;
    const int n = nk_all * G_CUS_KT + base;
    for (int i = 0; i < n; i += 64) __builtin_amdgcn_s_sleep(64);
  }
  // prologue: half-tiles 0..5 in flight, first K tile (0..3) landed
  issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3);
  issue(1, 0); issue(1, 1);
  bool first_tile = true;
  int tile_iter = 0;
  for (;;) {   // output tiles of this workgroup (exactly one unless G_PERSIST)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  TM(0);
  if (first_tile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetched half-tiles AND the previous epilogue's stores
  BAR();
  TM(1);
  if (dyn && wave == 0 && lane == 0) ticket = sched_pull(g.sched + s_list);   // the answer arrives under the K loop
  // stagger (guide §5 template): the wr==1 waves run one barrier behind, so on every SIMD one wave is in its
  // MFMA cluster while its partner (the wave 4 slots away shares the SIMD) issues LDS reads and DMA.
#ifndef G_STAGGER
#define G_STAGGER 1
#endif
  if (G_STAGGER && wr == 1) BAR();

  // four phases of the K tile in LDS buffer BUFI.  Half-tile sequence h = 4*tile + j (j: A0,B0,B1,A1); the phase
  // p of tile t issues h = 4t + 5 + p, i.e. (t+1,B1) (t+1,A1) (t+2,A0) (t+2,B0): every slot is re-staged >= 2 phases
  // after its last ds_read (A0: ph1 -> ph3, B0: ph1 -> ph4, B1: ph2 -> next ph1, A1: ph3 -> next ph2).
  auto tile_phases = [&](auto bufc, int t1, int t2) {
    constexpr int BUFI = decltype(bufc)::value;
    // phase 1
    readA(BUFI, 0); readB(BUFI, 0, fb0);
    issue(t1, 2);
    BAR();
    mma(acc[0][0], fb0);
    BAR();
    // phase 2
    readB(BUFI, 1, fb1);
    issue(t1, 3);
    BAR();
    mma(acc[0][1], fb1);
    BAR();
    // phase 3
    readA(BUFI, 1);
    issue(t2, 0);
    BAR();
    mma(acc[1][1], fb1);
    BAR();
    // phase 4: all but the two youngest half-tiles landed -> K tile t1 is complete for the next four phases
    issue(t2, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    BAR();
    mma(acc[1][0], fb0);
    BAR();
    // hipcc does not model the asm MFMAs: settle the XDL pipe before anything it may place at the loop boundary
    asm volatile("s_nop 7" ::: "memory");
  };
  const int npairs = (nk + 1) >> 1;   // an odd tile count is padded with one all-zero K tile
#pragma nounroll
  for (int it = 0; it < npairs; ++it) {
    const int te = 2 * it;      // even K tile -> buffer 0, odd -> buffer 1
    tile_phases(std::integral_constant<int, 0>{}, te + 1, te + 2);
    tile_phases(std::integral_constant<int, 1>{}, te + 2, te + 3);
  }
  TM(2);
  if (dyn && wave == 0) {             // publish the next tile before the barrier that closes the K loop for every wave
    if (lane == 0) {
      int my_list = s_list;
      s_next = sched_resolve(g.sched, my_list, ticket, ntiles);
      s_list = my_list;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if ((G_STAGGER && wr == 0) || (dyn && !G_STAGGER)) BAR();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");   // no DMA may outlive the workgroup; MFMA results settle before VALU reads
  TM(3);

  // ---------------- epilogue: lane (li, lq) holds C[m = .. + li][n = .. + 4 lq + 0..3]
  // opaque copies: nothing of the epilogue's address arithmetic may be hoisted above the K loop (it would sit in
  // ~80 VGPRs across the loop and push the accumulators into scratch)
  int lane_e = lane, m0e = m0, n0e = n0;
  asm volatile("" : "+v"(lane_e), "+s"(m0e), "+s"(n0e));
  const int next_logical = dyn ? __builtin_amdgcn_readfirstlane(s_next) : logical + (int)gridDim.x;
  const bool has_next = G_PERSIST && gridDim.y == 1 && next_logical >= 0 && next_logical < ntiles;
  if (has_next) {   // LDS is idle from here on: stage the next tile's first six half-tiles under the epilogue
    set_tile(next_logical);
    issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3);
    issue(1, 0); issue(1, 1);
  }
  TM(4);
  epilogue_tile<EPI, 128>(g, acc, m0e, n0e, wr, wc, lane_e, (size_t)blockIdx.y * g.slab_stride);
  TM(5);
  ++tile_iter;
  if (!has_next) break;
  logical = next_logical;
  first_tile = false;
  }
  if (dyn && tid == 0) sched_leave(g.sched, gridDim.x);
}

template <int LAYOUT, int EPI>
int launch256_one(const GemmArgs& a, int nsplit, hipStream_t s) {
  if (int rc = crl_enable_lds(reinterpret_cast<const void*>(&gemm256_kernel<LAYOUT, EPI>), 131072, "crl_gemm_bf16(256x256)")) return rc;
  int grid_x = a.ntm * a.ntn;
  GemmArgs b = a;
  b.sched = nullptr;
#if G_PERSIST
  const int ncu = crl_gemm_cus();
  if (nsplit == 1 && grid_x > ncu) {   // one resident workgroup per CU pulls tiles from the launch's ticket counters
    grid_x = ncu;
    if (crl_gemm_dynamic()) { bool ok; b.sched = crl_sched_slot(s, &ok); if (!ok) return -2; }
  }
#endif
#ifdef G_GRID_CAP
  if (nsplit == 1 && grid_x > G_GRID_CAP) grid_x = G_GRID_CAP;   // debug: fewer CUs busy (is a phase chip-bound or CU-bound?)
#endif
  gemm256_kernel<LAYOUT, EPI><<<dim3(grid_x, nsplit), T256, 131072, s>>>(b);
  CRL_LAUNCH_CHECK("crl_gemm_bf16(256)");
  return 0;
}

template <int LAYOUT>
int launch256_epi(const GemmArgs& a, int epi, int nsplit, hipStream_t s) {
  switch (epi) {
    case CRL_EPI_BF16: return launch256_one<LAYOUT, CRL_EPI_BF16>(a, nsplit, s);
    case CRL_EPI_BF16_GELU: return launch256_one<LAYOUT, CRL_EPI_BF16_GELU>(a, nsplit, s);
    case CRL_EPI_BF16_DGELU: return launch256_one<LAYOUT, CRL_EPI_BF16_DGELU>(a, nsplit, s);
    case CRL_EPI_F32_RESID: return launch256_one<LAYOUT, CRL_EPI_F32_RESID>(a, nsplit, s);
    case CRL_EPI_F32: return launch256_one<LAYOUT, CRL_EPI_F32>(a, nsplit, s);
    case CRL_EPI_F32_ACC: return launch256_one<LAYOUT, CRL_EPI_F32_ACC>(a, nsplit, s);
  }
  crl_set_error("crl_gemm_bf16: bad epilogue %d", epi);
  return -1;
}

}  // namespace

#ifdef G_TIMING
extern "C" int crl_gemm_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tm), sizeof(g_tm)); }
#endif

// called by crl_gemm_bf16 (gemm.hip) for shapes where the big tile pays
int crl_gemm256_launch(int layout, int epi, const gemmc::GemmArgs& a, int nsplit, hipStream_t s) {
  switch (layout) {
    case CRL_NT: return launch256_epi<CRL_NT>(a, epi, nsplit, s);
    case CRL_NN: return launch256_epi<CRL_NN>(a, epi, nsplit, s);
    default: return launch256_epi<CRL_TN>(a, epi, nsplit, s);
  }
}
