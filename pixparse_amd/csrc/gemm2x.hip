// 256x128x64 bf16 MFMA GEMM, 4 waves, TWO workgroups per CU: the epilogue-overlap variant of gemm256.hip for the forward
// (NT) and dgrad (NN) GEMMs of the step.
//
// Why: gemm256.hip runs ONE 8-wave workgroup per CU (128 KiB LDS, 2 x 256 VGPRs per SIMD), so nothing executes while a
// workgroup is in its epilogue.  scripts/bench_epilogue.py measures that exposed part per 256x256 tile: 12 us (bf16
// store), 22 us (GELU: 2 stores + erf math), 31 us (fp32 residual read + write) against 22 us for the whole K = 1024
// main loop -- about 23 ms of the cfg-3 step.  Here a workgroup is half as big (256 x 128 tile, 4 waves, one per SIMD,
// 80 KiB LDS), two of them share a CU and are NOT synchronised with each other: while one runs its epilogue (VALU +
// global memory) the other one's MFMA loop owns the matrix pipe, and the two main loops interleave on a SIMD the way the
// staggered wave groups of gemm256.hip do.
//
// Per wave the arithmetic is that of gemm256.hip: a 128x64 output block as 2x2 quadrants of 64x32, one quadrant (16
// v_mfma_f32_16x16x32_bf16, K = 64) per phase, four phases per K tile; wave (wr, wc) = (wave >> 1, wave & 1) owns rows
// 64 wr + [0, 64) of both A halves and columns 32 wc + [0, 32) of both 64-column halves of B.
// LDS: a ring of FIVE 16-KiB units; a K tile is three units (A rows 0-127 | B 128 columns | A rows 128-255), unit
// u = 3 t + {0, 1, 2} lives in slot u mod 5, so the slots rotate with period five tiles and are addressed through scalar
// registers.  Schedule of K tile t  (phase = {ds_reads, LDS-DMA issue, [counted vmcnt], s_barrier, 16 MFMA, s_barrier}):
//   phase 1  read A0(t), B(t) lower half   issue B(t+1)  -> slot of A1(t-1)   MFMA Q00
//   phase 2  read B(t) upper half          vmcnt(8): A1(t) landed             MFMA Q01
//   phase 3  read A1(t)                    issue A1(t+1) -> slot of A0(t)     MFMA Q11
//   phase 4  (B lower kept in registers)   issue A0(t+2) -> slot of B(t)      MFMA Q10   vmcnt(8): A0(t+1), B(t+1) landed
// Every unit has four to five phases (one K tile of the pair of workgroups) to land.  Hazards as in gemm256.hip:
// RAW -- a unit is read one phase after the counted vmcnt + barrier that retires its DMA (each wave issues 4 DMA
// instructions per unit, vmcnt(8) = all but the two youngest units); WAR -- a slot is re-staged at least two barriers
// after the phase of its last ds_read.
#include <type_traits>
#include "gemm_common.h"
#include "gemm_epilogue.h"

namespace {
using namespace gemmc;

constexpr int T2 = 256;
constexpr uint32_t UNIT = 16384, NSLOT = 5, LDS2 = UNIT * NSLOT;
#define BAR() asm volatile("s_barrier" ::: "memory")

template <int LAYOUT, int EPI>
__global__ __launch_bounds__(T2, 2) void gemm2x_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool B_TR = (LAYOUT != CRL_NT);
  static_assert(LAYOUT != CRL_TN, "the wgrad layout stays on gemm256.hip (split-K, long contraction)");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int ntiles = g.ntm * g.ntn;
  int m0, n0;

  const u32x4 ra = make_srd(g.A, g.a_bytes);
  const u32x4 rb = make_srd(g.B, g.b_bytes);
  const u32x4 rzero = make_srd(g.A, 0);   // zero records: K tiles past the end load zeros
  const uint32_t smem_base = lds_addr_of(smem);
  const int nk = (g.K + 63) / 64;

  // LDS-DMA addressing (one VGPR offset per operand + scalar parts).  A unit = 1024 chunks of 16 B, 4 per thread:
  //   KM image chunk c = 256 it + tid: row = 32 it + (tid >> 3), physical slot tid & 7 (the swizzle does not depend on it)
  //   TR image chunk c = 256 it + tid: krow = 16 it + (tid >> 4), physical chunk tid & 15
  uint32_t voffA, voffB, stepA, stepB;
  { const int r = tid >> 3; voffA = (uint32_t)r * g.lda * 2u + (uint32_t)(((tid & 7) ^ km_swz<64>(r)) * 16); stepA = 32u * g.lda * 2u; }
  if constexpr (B_TR) { const int kr = tid >> 4; voffB = (uint32_t)kr * g.ldb * 2u + (uint32_t)(((tid & 15) ^ tr_swz(kr)) * 16); stepB = 16u * g.ldb * 2u; }
  else { const int r = tid >> 3; voffB = (uint32_t)r * g.ldb * 2u + (uint32_t)(((tid & 7) ^ km_swz<64>(r)) * 16); stepB = 32u * g.ldb * 2u; }

  // kind 0 = A rows 0..127, 1 = B, 2 = A rows 128..255 of K tile `tile`, into the LDS slot at byte offset `slot`
  auto issue = [&](int kind, int tile, uint32_t slot) {
    const uint32_t lds = smem_base + slot + (uint32_t)wave * 1024u;
    const uint32_t k0 = (uint32_t)tile * 64u;
    const bool live = tile < nk;
    if (kind != 1) {
      const uint32_t r0 = (uint32_t)m0 + (kind == 2 ? 128u : 0u);
      const uint32_t soff = (r0 * g.lda + k0) * 2u;
      const u32x4 r = live ? ra : rzero;
#pragma unroll
      for (uint32_t it = 0; it < 4; ++it) dma16(r, lds + it * 4096u, voffA, soff + it * stepA);
    } else {
      const uint32_t c0 = (uint32_t)n0;
      const uint32_t soff = B_TR ? (k0 * g.ldb + c0) * 2u : (c0 * g.ldb + k0) * 2u;
      const u32x4 r = live ? rb : rzero;
#pragma unroll
      for (uint32_t it = 0; it < 4; ++it) dma16(r, lds + it * 4096u, voffB, soff + it * stepB);
    }
  };

#ifndef G2X_STAGGER
#define G2X_STAGGER 1
#endif
#ifndef G2X_DELAY_KT
#define G2X_DELAY_KT 16     // s_sleep units (64 clocks each) per K tile: ~ half of a co-resident pair's K-tile time
#endif
#ifndef G2X_DELAY_BASE
#define G2X_DELAY_BASE 96   // + ~ half an epilogue
#endif
  // The two workgroups of a CU start together and every tile takes the same time, so left alone they reach their epilogues
  // TOGETHER and nothing overlaps.  The workgroup that sits in the second wave slot of its SIMDs (HW_ID.wave_id != 0: slots
  // are handed out lowest first) therefore starts half a tile period late, once; the offset then persists over the tile
  // walk.  Speed only: no correctness property depends on the slot guess.
  if (G2X_STAGGER && gridDim.x < (unsigned)ntiles) {
    const uint32_t hw_id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID[3:0] = wave slot on the SIMD
    if (hw_id & 1u) {
      const int n = nk * G2X_DELAY_KT + G2X_DELAY_BASE;
      for (int i = 0; i < n; i += 64) __builtin_amdgcn_s_sleep(64);
    }
  }

  // Tile order: static walk (blockIdx.x, + gridDim.x) or, with g.sched, tickets pulled by lane 0 of wave 0 (gemm_common.h TileSched).
  // All 80 KiB of LDS belong to the ring (two workgroups per CU), so the mailbox that hands a pulled position to the other waves is the
  // first word of the ring, used only between tiles when no DMA is in flight and no fragment read is pending.
  const bool dyn = g.sched != nullptr;
  int* const mailbox = reinterpret_cast<int*>(smem);
  int my_list = 0;
  uint32_t ticket = 0;
  int logical = blockIdx.x;
  if (dyn) {
    if (wave == 0 && lane == 0) {
      my_list = sched_xcd();
      ticket = sched_pull(g.sched + my_list);
      *mailbox = sched_resolve(g.sched, my_list, ticket, ntiles);
    }
    __syncthreads();
    logical = __builtin_amdgcn_readfirstlane(*mailbox);
    __syncthreads();
  }
  f32x4 acc[2][2][4][2];
  while (logical >= 0 && logical < ntiles) {   // persistent: two resident workgroups per CU walk the tiles
  {
    const int t_id = xcd_remap(logical, ntiles);
    int tm, tn;
    tile_coords(t_id, g.ntm, g.ntn, tm, tn);       // grouped order (gemm_common.h): 8 tile rows x a few columns share panels in one XCD's L2
    m0 = tm * 256; n0 = tn * 128;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto adv = [](uint32_t s) { s += 3 * UNIT; return s >= LDS2 ? s - LDS2 : s; };
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  auto readA = [&](uint32_t slot) {
    const char* l = smem + slot;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = frag_km<64>(l, 64 * wr + 16 * i, ks, lane);
  };
  auto readB = [&](uint32_t slot, int qn, bf16x8 (&f)[2][2]) {
    const char* l = smem + slot;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (B_TR) f[j][ks] = frag_tr(l, 64 * qn + 32 * wc + 16 * j, ks, lane);
        else f[j][ks] = frag_km<64>(l, 64 * qn + 32 * wc + 16 * j, ks, lane);
      }
  };
  auto mma = [&](f32x4 (&c)[4][2], const bf16x8 (&b)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i][j]) : "v"(b[j][ks]), "v"(fa[i][ks]));
    __builtin_amdgcn_s_setprio(0);
  };

  // slot offsets of the current K tile's units (scalar); next tile: + 3 units mod 5
  uint32_t sA0 = 0, sB = UNIT, sA1 = 2 * UNIT;

  // prologue: A0(0) B(0) A1(0) A0(1) in slots 0..3; the first two landed
  issue(0, 0, 0); issue(1, 0, UNIT); issue(2, 0, 2 * UNIT); issue(0, 1, 3 * UNIT);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  BAR();
  if (dyn && wave == 0 && lane == 0) ticket = sched_pull(g.sched + my_list);   // next tile: the answer arrives under the K loop

#pragma nounroll
  for (int t = 0; t < nk; ++t) {
    // phase 1
    readA(sA0); readB(sB, 0, fb0);
    issue(1, t + 1, sA0 >= UNIT ? sA0 - UNIT : sA0 + 4 * UNIT);
    BAR();
    mma(acc[0][0], fb0);
    BAR();
    // phase 2
    readB(sB, 1, fb1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    BAR();
    mma(acc[0][1], fb1);
    BAR();
    // phase 3
    readA(sA1);
    issue(2, t + 1, sA0);
    BAR();
    mma(acc[1][1], fb1);
    BAR();
    // phase 4
    issue(0, t + 2, sB);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    BAR();
    mma(acc[1][0], fb0);
    BAR();
    // hipcc does not model the asm MFMAs: settle the XDL pipe before anything it may place at the loop boundary
    asm volatile("s_nop 7" ::: "memory");
    sA0 = adv(sA0); sB = adv(sB); sA1 = adv(sA1);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");   // no DMA may outlive the workgroup; MFMA results settle before VALU reads
  int next_logical = logical + (int)gridDim.x;
  if (dyn) {
    if (wave == 0 && lane == 0) *mailbox = sched_resolve(g.sched, my_list, ticket, ntiles);
    __syncthreads();
    next_logical = __builtin_amdgcn_readfirstlane(*mailbox);
    __syncthreads();        // every wave has read the word before the next tile's prologue DMA lands on it
  }

  // opaque copies: nothing of the epilogue's address arithmetic may be hoisted above the K loop
  int lane_e = lane, m0e = m0, n0e = n0;
  asm volatile("" : "+v"(lane_e), "+s"(m0e), "+s"(n0e));
  epilogue_tile<EPI, 64>(g, acc, m0e, n0e, wr, wc, lane_e, 0);
  logical = next_logical;
  }
  if (dyn && tid == 0) sched_leave(g.sched, gridDim.x);
}

template <int LAYOUT, int EPI>
int launch2x_one(const GemmArgs& a, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2x_kernel<LAYOUT, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS2);
    if (e != hipSuccess) { crl_set_error("gemm2x: cannot enable 80 KiB LDS: %s", hipGetErrorString(e)); return -2; }
    configured = true;
  }
  int grid = a.ntm * a.ntn;
  GemmArgs b = a;
  b.sched = nullptr;
#ifndef G2X_PERSIST
#define G2X_PERSIST 1
#endif
  const int nres = 2 * crl_gemm_cus();         // two resident workgroups per CU
  if (G2X_PERSIST && grid > nres) {
    grid = nres;
    if (crl_gemm_dynamic()) { bool ok; b.sched = crl_sched_slot(s, &ok); if (!ok) return -2; }
  }
  gemm2x_kernel<LAYOUT, EPI><<<dim3(grid), T2, LDS2, s>>>(b);
  CRL_LAUNCH_CHECK("crl_gemm_bf16(256x128)");
  return 0;
}

template <int LAYOUT>
int launch2x_epi(const GemmArgs& a, int epi, hipStream_t s) {
  switch (epi) {
    case CRL_EPI_BF16: return launch2x_one<LAYOUT, CRL_EPI_BF16>(a, s);
    case CRL_EPI_BF16_GELU: return launch2x_one<LAYOUT, CRL_EPI_BF16_GELU>(a, s);
    case CRL_EPI_BF16_DGELU: return launch2x_one<LAYOUT, CRL_EPI_BF16_DGELU>(a, s);
    case CRL_EPI_F32_RESID: return launch2x_one<LAYOUT, CRL_EPI_F32_RESID>(a, s);
    case CRL_EPI_F32: return launch2x_one<LAYOUT, CRL_EPI_F32>(a, s);
    case CRL_EPI_F32_ACC: return launch2x_one<LAYOUT, CRL_EPI_F32_ACC>(a, s);
  }
  crl_set_error("crl_gemm_bf16: bad epilogue %d", epi);
  return -1;
}

}  // namespace

// called by crl_gemm_bf16 (gemm.hip): NT / NN shapes with K % 64 == 0; a.ntm / a.ntn count 256 x 128 tiles
int crl_gemm2x_launch(int layout, int epi, const gemmc::GemmArgs& a, hipStream_t s) {
  if (layout == CRL_NT) return launch2x_epi<CRL_NT>(a, epi, s);
  if (layout == CRL_NN) return launch2x_epi<CRL_NN>(a, epi, s);
  crl_set_error("crl_gemm_bf16(256x128): layout %d is not supported by this kernel", layout);
  return -1;
}
