// 256x256x64 bf16 MFMA GEMM, FOUR waves = ONE WAVE PER SIMD (round 5): the large-shape path of crl_gemm_bf16 for all three layouts.
//
// Why: the 8-wave kernel of gemm256.hip keeps the matrix pipe 92 % busy at the ~1.3 GHz the chip holds under it and its K loop keeps
// the LDS 75 % busy (24 ds_read_b128 per 64 MFMAs and wave, two waves per SIMD); on the same box the vendor's 4-wave 256x256x64 kernel
// sustains 1.64 PF/s against 1.24 (profiles/r5_yardstick.txt) -- the loop is held down by energy per FLOP, not by issue slots.  Here a
// wave owns a 128x128 quadrant of the tile (256 accumulator registers in the AGPR half of its 512-entry file) and reads one A and one B
// half-tile per K tile: 32 ds_read_b128 per 128 MFMAs, 1.5 x fewer LDS bytes per FLOP, no partner wave to arbitrate with, 2 barriers per
// K tile of 128 MFMAs instead of 8 per 64; all 160 KiB of LDS as a ring of five 32-KiB operand pairs, two K tiles of look-ahead.  The main loop is ONE generated asm statement (gen_gemm4w.py -> gemm4w_body_{nt,nn,tn}.inc:
// register map, schedule, hazards); this file is the persistent tile walk around it (dynamic ticket scheduler of gemm_common.h), the
// staging of the first two K tiles of every output tile (issued under the previous tile's epilogue) and the epilogue, which receives the
// accumulators as physical-register asm outputs and applies the arithmetic of gemm_epilogue.h operation for operation (results are
// bit-identical to gemm256.hip: same k order per accumulator, same epilogue math).
#include <type_traits>
#include "gemm_common.h"
#include "gemm_epilogue.h"

namespace {
using namespace gemmc;

constexpr int T4W = 256;

// ---- epilogue over the wave's 8 x 8 accumulator tiles: c[i][j] = rows 128 wr + 16 i, columns 128 wc + 16 j of the workgroup tile; lane
// (li = lane & 15, lq = lane >> 4) holds C[row li][col 4 lq + 0..3].  Branch-free bounds-checked 16-byte buffer accesses in the lane-transposed
// access layout (EpiLanes), batches of 16 accesses (two 16-row groups), loads of a batch issued before the stores of the previous one.
template <int EPI>
__device__ __forceinline__ void epilogue4w(const GemmArgs& g, f32x4 (&c)[8][8], int m0e, int n0e, int wr, int wc, int lane_e, size_t slab_off) {
  constexpr bool BIAS_EPI = (EPI == CRL_EPI_BF16 || EPI == CRL_EPI_BF16_GELU || EPI == CRL_EPI_F32_RESID);
  constexpr bool BF16_OUT = (EPI == CRL_EPI_BF16 || EPI == CRL_EPI_BF16_GELU || EPI == CRL_EPI_BF16_DGELU);
  constexpr uint32_t ES = BF16_OUT ? 2u : 4u;
  const int lq = lane_e >> 4;
  const EpiLanes<BF16_OUT> L(lane_e);
  const uint32_t Mu = (uint32_t)g.M, Nu = (uint32_t)g.N;
  const uint32_t nrecC = (uint32_t)(((size_t)(Mu - 1) * (uint32_t)g.ldc + Nu) * ES);
  const __amdgpu_buffer_rsrc_t rC = epi_rsrc((const char*)g.C + slab_off * 4, nrecC);
  const uint32_t mrow = (uint32_t)(m0e + 128 * wr + L.ar);             // access row of row group i = mrow + 16 i
  const uint32_t rstepC = 16u * (uint32_t)g.ldc * ES;
  const uint32_t rbaseC = mrow * (uint32_t)g.ldc * ES;
  const int ncol0 = n0e + 128 * wc;

  // the lane's 32 bias values (its accumulator columns, rounded to bf16 like autocast)
  float bw[8][4];
  if constexpr (BIAS_EPI) {
    const __amdgpu_buffer_rsrc_t rB = epi_rsrc(g.bias, g.bias ? Nu * 4u : 0u);
    epi_u4 braw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) braw[j] = epi_ld(rB, (uint32_t)(ncol0 + 16 * j + 4 * lq) * 4u);
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bw[j][r] = round_bf(__uint_as_float(braw[j][r]));
  }
  if constexpr (BF16_OUT) {
    // after swap_strips a lane owns 8 consecutive columns of a 32-column strip pair -> one 16-byte access per (i, pr)
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = (ncol0 + 16 * j + 4 * lq) < g.colscale_cols ? g.colscale : 1.f;
    uint32_t cC[4], cA[4];
    bool okc[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const uint32_t n = (uint32_t)(ncol0 + 32 * pr + 8 * L.ac);
      okc[pr] = n < Nu;
      cC[pr] = rbaseC + n * 2u;
      cA[pr] = mrow * (uint32_t)g.ldaux * 2u + n * 2u;
    }
    const uint32_t nrecA = (EPI == CRL_EPI_BF16) ? 0u : (uint32_t)(((size_t)(Mu - 1) * (uint32_t)g.ldaux + Nu) * 2u);
    const __amdgpu_buffer_rsrc_t rA = epi_rsrc(g.aux, nrecA);
    const uint32_t rstepA = 16u * (uint32_t)g.ldaux * 2u;
    auto offC = [&](int i, int pr) { return okc[pr] ? cC[pr] + (uint32_t)i * rstepC : nrecC; };
    auto offA = [&](int i, int pr) { return okc[pr] ? cA[pr] + (uint32_t)i * rstepA : nrecA; };
    // saved derivatives of the dGELU epilogue: ALL thirty-two 16-byte loads of the lane go out before the first store (a load behind a
    // store would wait for its write acknowledgement); hw[8 i' + t] belongs to row group i, strip pair pr with 4 i + pr = index
    epi_u4 hw[32];
    if constexpr (EPI == CRL_EPI_BF16_DGELU) {
#pragma unroll
      for (int u = 0; u < 32; ++u) hw[u] = epi_ld(rA, offA(u >> 2, u & 3));
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int i = u >> 2, pr = u & 3;
      f32x2 v[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 a4 = c[i][2 * pr + j];
        v[j][0] = f32x2{a4[0], a4[1]}; v[j][1] = f32x2{a4[2], a4[3]};
        if constexpr (BIAS_EPI) { v[j][0] += f32x2{bw[2 * pr + j][0], bw[2 * pr + j][1]}; v[j][1] += f32x2{bw[2 * pr + j][2], bw[2 * pr + j][3]}; }
        if constexpr (EPI == CRL_EPI_BF16) { v[j][0] *= cs[2 * pr + j]; v[j][1] *= cs[2 * pr + j]; }
      }
      uint32_t x0, x1, y0, y1;
      if constexpr (EPI == CRL_EPI_BF16) {
        x0 = pack_bf2v(v[0][0]); x1 = pack_bf2v(v[0][1]); y0 = pack_bf2v(v[1][0]); y1 = pack_bf2v(v[1][1]);
        swap_strips(x0, x1, y0, y1);
        epi_st(rC, offC(i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
      } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
        // h = bf16(v + b) (the rounding the reference's autocast applies) -> gelu(h) and, saved for the backward as fp16, gelu'(h)
        f32x2 ga[4], da[4];
        gelu_grad2(unpack_bf2(pack_bf2v(v[0][0])), ga[0], da[0]); gelu_grad2(unpack_bf2(pack_bf2v(v[0][1])), ga[1], da[1]);
        gelu_grad2(unpack_bf2(pack_bf2v(v[1][0])), ga[2], da[2]); gelu_grad2(unpack_bf2(pack_bf2v(v[1][1])), ga[3], da[3]);
        x0 = pack_h2v(da[0]); x1 = pack_h2v(da[1]); y0 = pack_h2v(da[2]); y1 = pack_h2v(da[3]);
        swap_strips(x0, x1, y0, y1);
        epi_st_saved(rA, offA(i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
        x0 = pack_bf2v(ga[0]); x1 = pack_bf2v(ga[1]); y0 = pack_bf2v(ga[2]); y1 = pack_bf2v(ga[3]);
        swap_strips(x0, x1, y0, y1);
        epi_st(rC, offC(i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
      } else {   // dGELU: the 16-byte load holds 8 consecutive saved derivatives (fp16); the exchange (an involution) returns this lane's own
        const epi_u4 hm = L.to_math(hw[u]);
        uint32_t a0 = hm[0], a1 = hm[1], b0 = hm[2], b1 = hm[3];
        swap_strips(a0, a1, b0, b1);
        x0 = pack_bf2v(unpack_bf2(pack_bf2v(v[0][0])) * unpack_h2(a0));
        x1 = pack_bf2v(unpack_bf2(pack_bf2v(v[0][1])) * unpack_h2(a1));
        y0 = pack_bf2v(unpack_bf2(pack_bf2v(v[1][0])) * unpack_h2(b0));
        y1 = pack_bf2v(unpack_bf2(pack_bf2v(v[1][1])) * unpack_h2(b1));
        swap_strips(x0, x1, y0, y1);
        epi_st(rC, offC(i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
      }
    }
  } else {
    // fp32 outputs: math lane (li, lq) owns columns 4 lq + 0..3 of each 16-column strip -> one 16-byte access per (i, j)
    uint32_t cC[8], cR[8];
    bool okc[8];
    constexpr bool READS = (EPI == CRL_EPI_F32_RESID || EPI == CRL_EPI_F32_ACC);
    const uint32_t ldr = (EPI == CRL_EPI_F32_RESID) ? (uint32_t)g.ldr : (uint32_t)g.ldc;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t n = (uint32_t)(ncol0 + 16 * j + 4 * L.ac);
      okc[j] = n < Nu;
      cC[j] = rbaseC + n * 4u;
      cR[j] = mrow * ldr * 4u + n * 4u;
    }
    const uint32_t nrecR = READS ? (uint32_t)(((size_t)(Mu - 1) * ldr + Nu) * 4u) : 0u;
    const __amdgpu_buffer_rsrc_t rR = epi_rsrc(EPI == CRL_EPI_F32_RESID ? (const void*)g.resid : (const void*)g.C, nrecR);
    const uint32_t rstepR = 16u * ldr * 4u;
    auto offC = [&](int i, int j) { return okc[j] ? cC[j] + (uint32_t)i * rstepC : nrecC; };
    auto offR = [&](int i, int j) { return okc[j] ? cR[j] + (uint32_t)i * rstepR : nrecR; };
    // four batches of two row groups (sixteen 16-byte accesses); issue order L(0) L(1) S(0) L(2) S(1) L(3) S(2) S(3): a load behind a store
    // would wait for its write acknowledgement (vmcnt retires in issue order)
    epi_u4 rv[2][16];
    auto issue = [&](int q) {
#pragma unroll
      for (int u = 0; u < 16; ++u) rv[q & 1][u] = epi_ld(rR, offR(2 * q + (u >> 3), u & 7));
    };
    auto batch = [&](int q) {
      epi_u4 res[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = 2 * q + (u >> 3), j = u & 7;
        const f32x4 a4 = c[i][j];
        epi_u4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = a4[r];
          if constexpr (BIAS_EPI) v += bw[j][r];
          if constexpr (EPI == CRL_EPI_F32_RESID) v = round_bf(v);
          w[r] = __float_as_uint(v);
        }
        w = L.to_access(w);
        if constexpr (READS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = __float_as_uint(__uint_as_float(rv[q & 1][u][r]) + __uint_as_float(w[r]));
        }
        res[u] = w;
      }
      if (READS && q + 2 < 4) issue(q + 2);
#pragma unroll
      for (int u = 0; u < 16; ++u) epi_st(rC, offC(2 * q + (u >> 3), u & 7), res[u]);
    };
    if constexpr (READS) { issue(0); issue(1); }
    batch(0); batch(1); batch(2); batch(3);
  }
}

template <int LAYOUT, int EPI>
__global__ __launch_bounds__(T4W, 1) void gemm4w_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool A_TR = (LAYOUT == CRL_TN);
  constexpr bool B_TR = (LAYOUT != CRL_NT);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int ntiles = g.ntm * g.ntn;
  // dynamic tile schedule (gemm_common.h): thread 0 pulls the tickets; the tile index travels to the other waves through the first word of the
  // LDS ring, which is idle at every output-tile boundary (all 160 KiB belong to the ring: there is no room for a static __shared__ word)
  const bool dyn = g.sched != nullptr;
  uint32_t ticket = 0;
  int my_list = 0;
  int logical = blockIdx.x;
  if (dyn) {
    if (tid == 0) {
      my_list = sched_xcd();
      ticket = sched_pull(g.sched + my_list);
      *reinterpret_cast<volatile int*>(smem) = sched_resolve(g.sched, my_list, ticket, ntiles);
    }
    __syncthreads();
    logical = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem));
    __syncthreads();
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
  const u32x4 rzero = make_srd(g.A, 0);
  const uint32_t smem_base = lds_addr_of(smem);

  const int nk_all = (g.K + 63) / 64;
  const int kt0 = blockIdx.y * g.kchunk;
  const int nk = min(nk_all, kt0 + g.kchunk) - kt0;   // K tiles of this split
  const uint32_t s_ldsw = smem_base + (uint32_t)wave * 1024u;

  // scalar steps of the staging addresses: piece (half h, it) of a K tile reads at s_off + h * s_half + it * s_it, a K tile advances by s_kt
  //   KM image chunk c = 256 it + tid: row = 32 it + (tid >> 3), slot tid & 7;   TR image: krow = 16 it + (tid >> 4), chunk tid & 15
  const uint32_t s_itA = A_TR ? 16u * g.lda * 2u : 32u * g.lda * 2u, s_halfA = A_TR ? 256u : 128u * g.lda * 2u, s_ktA = A_TR ? 64u * g.lda * 2u : 128u;
  const uint32_t s_itB = B_TR ? 16u * g.ldb * 2u : 32u * g.ldb * 2u, s_halfB = B_TR ? 256u : 128u * g.ldb * 2u, s_ktB = B_TR ? 64u * g.ldb * 2u : 128u;
  auto offA_of = [&](int tile) { const uint32_t k0 = (uint32_t)(kt0 + tile) * 64u; return A_TR ? (k0 * g.lda + (uint32_t)m0) * 2u : ((uint32_t)m0 * g.lda + k0) * 2u; };
  auto offB_of = [&](int tile) { const uint32_t k0 = (uint32_t)(kt0 + tile) * 64u; return B_TR ? (k0 * g.ldb + (uint32_t)n0) * 2u : ((uint32_t)n0 * g.ldb + k0) * 2u; };

  // K tile `tile` (0 or 1) of the current output tile -> LDS buffer `tile`: the 16 pieces of this wave, same order / destinations as the stream
  auto stage = [&](int tile) {
    // per-lane source offsets: derived afresh (they must not occupy registers across the stream)
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    uint32_t voffA, voffB;
    if constexpr (A_TR) { const int kr = t2 >> 4; voffA = (uint32_t)kr * g.lda * 2u + (uint32_t)(((t2 & 15) ^ tr_swz(kr)) * 16); }
    else { const int r = t2 >> 3; voffA = (uint32_t)r * g.lda * 2u + (uint32_t)(((t2 & 7) ^ km_swz<64>(r)) * 16); }
    if constexpr (B_TR) { const int kr = t2 >> 4; voffB = (uint32_t)kr * g.ldb * 2u + (uint32_t)(((t2 & 15) ^ tr_swz(kr)) * 16); }
    else { const int r = t2 >> 3; voffB = (uint32_t)r * g.ldb * 2u + (uint32_t)(((t2 & 7) ^ km_swz<64>(r)) * 16); }
    const bool live = tile < nk;
    const u32x4 sa = live ? ra : rzero, sb = live ? rb : rzero;
    const uint32_t oa = offA_of(tile), ob = offB_of(tile);
    // ring of five pair-slots of 32 KiB: A(t) -> slot 2 t mod 5, B(t) -> slot (2 t + 1) mod 5; half h at + 16 KiB, piece it at + 4 KiB
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        dma16(sa, s_ldsw + (uint32_t)(((2 * tile) % 5) * 32768 + h * 16384 + it * 4096), voffA, oa + h * s_halfA + it * s_itA);
        dma16(sb, s_ldsw + (uint32_t)(((2 * tile + 1) % 5) * 32768 + h * 16384 + it * 4096), voffB, ob + h * s_halfB + it * s_itB);
      }
  };

#ifndef G4_CU_STAGGER
#define G4_CU_STAGGER 0
#endif
#ifndef G4_CUS_KT
#define G4_CUS_KT 10      // s_sleep units (64 clocks) per K tile: about half of a K tile's ~1.3 us
#endif
  // experiment (default off): half of the persistent workgroups start half a tile period late, so that the tile-sized epilogue traffic of one
  // half of the chip falls under the K loops of the other half instead of every CU reaching its epilogue at the same time
  if (G4_CU_STAGGER && gridDim.y == 1 && ntiles >= 2 * (int)gridDim.x && ((blockIdx.x >> 3) & 1)) {
    constexpr int base = EPI == CRL_EPI_BF16 ? 60 : EPI == CRL_EPI_BF16_GELU ? 120 : EPI == CRL_EPI_BF16_DGELU ? 120 : 200;
    const int n = nk_all * G4_CUS_KT + base;
    for (int i = 0; i < n; i += 64) __builtin_amdgcn_s_sleep(64);
  }
  stage(0); stage(1);
  for (;;) {   // output tiles of this workgroup
    if (dyn && tid == 0) ticket = sched_pull(g.sched + my_list);   // the answer arrives under the K loop
    // ---- operands of the stream (per-lane values derived from an opaque copy of the thread index: nothing of this may be hoisted into
    // registers that live across the epilogue)
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int lane = t2 & 63, li = lane & 15, lq = lane >> 4;
    uint32_t voffA, voffB;
    if constexpr (A_TR) { const int kr = t2 >> 4; voffA = (uint32_t)kr * g.lda * 2u + (uint32_t)(((t2 & 15) ^ tr_swz(kr)) * 16); }
    else { const int r = t2 >> 3; voffA = (uint32_t)r * g.lda * 2u + (uint32_t)(((t2 & 7) ^ km_swz<64>(r)) * 16); }
    if constexpr (B_TR) { const int kr = t2 >> 4; voffB = (uint32_t)kr * g.ldb * 2u + (uint32_t)(((t2 & 15) ^ tr_swz(kr)) * 16); }
    else { const int r = t2 >> 3; voffB = (uint32_t)r * g.ldb * 2u + (uint32_t)(((t2 & 7) ^ km_swz<64>(r)) * 16); }
    // fragment read addresses at ring offset 0 (the stream adds the pair-slot, row group and k-step offsets): the wave's half of an A / B pair
    const uint32_t unitA = smem_base + (uint32_t)wr * 16384u, unitB = smem_base + (uint32_t)wc * 16384u;
    //   KM: row 16 i + li, 16-byte slot (4 ks + lq) ^ ((li >> 1) & 7)
    const uint32_t km_k0 = (uint32_t)(li * 128 + (((0 + lq) ^ ((li >> 1) & 7)) << 4)), km_k1 = (uint32_t)(li * 128 + (((4 + lq) ^ ((li >> 1) & 7)) << 4));
    //   TR: lane (g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3): k row 8 g + q (+ 32 ks, + 4 for the second read), column 16 j + 4 p
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tsw = tr_swz(8 * tg + tq);
    auto tr_addr = [&](int j) { return (uint32_t)((8 * tg + tq) * 256 + ((((2 * j) ^ tsw) + (tp >> 1)) << 4) + 8 * (tp & 1)); };
    u32x4 srdA = ra, srdB = rb;
    uint32_t s_offA = offA_of(2), s_offB = offB_of(2);
    uint32_t s_live = (uint32_t)(nk - 2), s_cnt = (uint32_t)nk, s_t;
    f32x4 c[8][8];
    if constexpr (LAYOUT == CRL_NT) {
      uint32_t arAk0 = unitA + km_k0, arAk1 = unitA + km_k1, arBk0 = unitB + km_k0, arBk1 = unitB + km_k1;
#include "gemm4w_body_nt.inc"
    } else if constexpr (LAYOUT == CRL_NN) {
      uint32_t arAk0 = unitA + km_k0, arAk1 = unitA + km_k1;
      uint32_t arBt0 = unitB + tr_addr(0), arBt1 = unitB + tr_addr(1), arBt2 = unitB + tr_addr(2), arBt3 = unitB + tr_addr(3),
               arBt4 = unitB + tr_addr(4), arBt5 = unitB + tr_addr(5), arBt6 = unitB + tr_addr(6), arBt7 = unitB + tr_addr(7);
#include "gemm4w_body_nn.inc"
    } else {
      uint32_t arAt0 = unitA + tr_addr(0), arAt1 = unitA + tr_addr(1), arAt2 = unitA + tr_addr(2), arAt3 = unitA + tr_addr(3),
               arAt4 = unitA + tr_addr(4), arAt5 = unitA + tr_addr(5), arAt6 = unitA + tr_addr(6), arAt7 = unitA + tr_addr(7);
      uint32_t arBt0 = unitB + tr_addr(0), arBt1 = unitB + tr_addr(1), arBt2 = unitB + tr_addr(2), arBt3 = unitB + tr_addr(3),
               arBt4 = unitB + tr_addr(4), arBt5 = unitB + tr_addr(5), arBt6 = unitB + tr_addr(6), arBt7 = unitB + tr_addr(7);
      const int tn_cur = n0 >> 8;
      if (g.cs_ws != nullptr && tn_cur < g.cs_ntn) {
        // the bias gradient rides the weight gradient: column sums of this wave's A half on the matrix pipe, on every (2 cs_ntn)-th K tile -- phase
        // 2 tn + wc, so that the waves / workgroups reading the same A panel cover the contraction between them (gen_gemm4w.py colsum_block)
        f32x4 cs[8];
        const uint32_t s_csmask = (uint32_t)(2 * g.cs_ntn - 1);
        uint32_t s_csk = ((uint32_t)(2 * tn_cur + wc) - (uint32_t)kt0) & s_csmask, s_csgo;
#include "gemm4w_body_tn_cs.inc"
        (void)s_csgo;
        if (lq == 0) {      // lane li holds the sum of row 16 i + li of the wave's A half (the same value in all four registers and all four lq groups)
          float* dst = g.cs_ws + ((size_t)((blockIdx.y * g.cs_ntn + tn_cur) * 2 + wc)) * (size_t)g.M + (size_t)(m0 + 128 * wr + li);
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (m0 + 128 * wr + 16 * i + li < g.M) dst[16 * i] = cs[i][0];
        }
      } else {
#include "gemm4w_body_tn.inc"
      }
    }
    // ---- next tile (the stream ends behind a barrier: the ring is idle): thread 0 publishes it through the first word of the LDS
    int next_logical = logical + (int)gridDim.x;
    if (dyn) {
      if (tid == 0) *reinterpret_cast<volatile int*>(smem) = sched_resolve(g.sched, my_list, ticket, ntiles);
      __syncthreads();
      next_logical = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem));
      __syncthreads();                // ... and everybody has read it before the staging below overwrites the word
    }
    int lane_e = threadIdx.x & 63, m0e = m0, n0e = n0;
    asm volatile("" : "+v"(lane_e), "+s"(m0e), "+s"(n0e));
    const bool has_next = gridDim.y == 1 && next_logical >= 0 && next_logical < ntiles;
    if (has_next) {   // LDS is idle from here on: the next tile's first two K tiles travel under the epilogue
      set_tile(next_logical);
      stage(0); stage(1);
    }
    epilogue4w<EPI>(g, c, m0e, n0e, wr, wc, lane_e, (size_t)blockIdx.y * g.slab_stride);
    if (!has_next) break;
    logical = next_logical;
  }
  if (dyn && tid == 0) sched_leave(g.sched, gridDim.x);
}

// ---- the OVERLAPPED form (plain bf16 epilogue: bias, column scale, bf16 rounding; NT / NN): the epilogue of output tile T runs inside the
// statement of tile T + 1 -- its accumulators are packed to bf16 (128 registers) at the statement's entry while the staging DMA of T + 1's
// first two K tiles is in flight, and its 32 stores per lane are spread over T + 1's first five K tiles (gen_gemm4w.py): no workgroup ever
// sits in an epilogue with its matrix pipes idle and the tile-sized store traffic of the 256 CUs no longer arrives in one burst.  The last
// tile of a workgroup is finished by a drain statement.  Same arithmetic as epilogue4w / gemm_epilogue.h (bit-identical results); the
// accumulators are carried from statement to statement as in-out physical-register operands (the compiler leaves them in place).
// (An fp32-residual variant of this form was built in round 5 and measured slower than the 8-wave kernel: profiles/r5_gemm4w_ovl_resid.txt,
// removed in round 6, diff in profiles/r6_experiments/.)
template <int LAYOUT>
__global__ __launch_bounds__(T4W, 1) void gemm4w_ovl_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool B_TR = (LAYOUT != CRL_NT);
  static_assert(LAYOUT != CRL_TN, "weight gradients have fp32 outputs: classic form");
  constexpr uint32_t ES = 2u;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntiles = g.ntm * g.ntn;
  const bool dyn = g.sched != nullptr;
  uint32_t ticket = 0;
  int my_list = 0;
  int logical = blockIdx.x;
  if (dyn) {
    if (tid == 0) {
      my_list = sched_xcd();
      ticket = sched_pull(g.sched + my_list);
      *reinterpret_cast<volatile int*>(smem) = sched_resolve(g.sched, my_list, ticket, ntiles);
    }
    __syncthreads();
    logical = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem));
    __syncthreads();
    if (logical < 0) {
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
  const int nk = (g.K + 63) / 64;
  const uint32_t s_ldsw = smem_base + (uint32_t)wave * 1024u;
  const uint32_t s_itA = 32u * g.lda * 2u, s_halfA = 128u * g.lda * 2u, s_ktA = 128u;
  const uint32_t s_itB = B_TR ? 16u * g.ldb * 2u : 32u * g.ldb * 2u, s_halfB = B_TR ? 256u : 128u * g.ldb * 2u, s_ktB = B_TR ? 64u * g.ldb * 2u : 128u;
  // epilogue constants
  const uint32_t nrecC = (uint32_t)(((size_t)(g.M - 1) * (uint32_t)g.ldc + (uint32_t)g.N) * ES);
  const u32x4 rc_live = make_srd(g.C, nrecC), rc_dead = make_srd(g.C, 0);       // no previous tile: every store falls off a descriptor without records
  const u32x4 srdBias = make_srd(g.bias, g.bias ? (uint32_t)g.N * 4u : 0u);
  const uint32_t s_cstep = 16u * (uint32_t)g.ldc * ES, s_cscols = (uint32_t)g.colscale_cols, s_cscale = __float_as_uint(g.colscale);

  f32x4 c[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) c[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool have_prev = false;
  int pm0 = 0, pn0 = 0;
  for (;;) {   // output tiles of this workgroup
    if (dyn && tid == 0) ticket = sched_pull(g.sched + my_list);
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int lane = t2 & 63, li = lane & 15, lq = lane >> 4;
    uint32_t voffA, voffB;
    { const int r = t2 >> 3; voffA = (uint32_t)r * g.lda * 2u + (uint32_t)(((t2 & 7) ^ km_swz<64>(r)) * 16); }
    if constexpr (B_TR) { const int kr = t2 >> 4; voffB = (uint32_t)kr * g.ldb * 2u + (uint32_t)(((t2 & 15) ^ tr_swz(kr)) * 16); }
    else { const int r = t2 >> 3; voffB = (uint32_t)r * g.ldb * 2u + (uint32_t)(((t2 & 7) ^ km_swz<64>(r)) * 16); }
    const uint32_t unitA = smem_base + (uint32_t)wr * 16384u, unitB = smem_base + (uint32_t)wc * 16384u;
    const uint32_t km_k0 = (uint32_t)(li * 128 + (((0 + lq) ^ ((li >> 1) & 7)) << 4)), km_k1 = (uint32_t)(li * 128 + (((4 + lq) ^ ((li >> 1) & 7)) << 4));
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tsw = tr_swz(8 * tg + tq);
    auto tr_addr = [&](int j) { return (uint32_t)((8 * tg + tq) * 256 + ((((2 * j) ^ tsw) + (tp >> 1)) << 4) + 8 * (tp & 1)); };
    // the previous tile's epilogue: lane (li, lq) stores row 128 wr + 16 i + li, the 8 columns 128 wc + 32 pr + 8 bitswap(lq) .. of every strip pair
    const uint32_t voffC = (uint32_t)(128 * wr + li) * (uint32_t)g.ldc * 2u + (uint32_t)(128 * wc + 8 * (((lq & 1) << 1) | (lq >> 1))) * 2u;
    const uint32_t colv = (uint32_t)(pn0 + 128 * wc + 4 * lq), voffBias = colv * 4u;
    const u32x4 srdC = have_prev ? rc_live : rc_dead;
    uint32_t s_crow = ((uint32_t)pm0 * (uint32_t)g.ldc + (uint32_t)pn0) * ES;
    u32x4 srdA = ra, srdB = rb;
    uint32_t s_offA = ((uint32_t)m0 * g.lda) * 2u, s_offB = B_TR ? (uint32_t)n0 * 2u : ((uint32_t)n0 * g.ldb) * 2u;
    uint32_t s_live = (uint32_t)nk, s_cnt = (uint32_t)nk, s_t;
    if constexpr (LAYOUT == CRL_NT) {
      uint32_t arAk0 = unitA + km_k0, arAk1 = unitA + km_k1, arBk0 = unitB + km_k0, arBk1 = unitB + km_k1;
#include "gemm4w_body_nt_ovl.inc"
    } else {
      uint32_t arAk0 = unitA + km_k0, arAk1 = unitA + km_k1;
      uint32_t arBt0 = unitB + tr_addr(0), arBt1 = unitB + tr_addr(1), arBt2 = unitB + tr_addr(2), arBt3 = unitB + tr_addr(3),
               arBt4 = unitB + tr_addr(4), arBt5 = unitB + tr_addr(5), arBt6 = unitB + tr_addr(6), arBt7 = unitB + tr_addr(7);
#include "gemm4w_body_nn_ovl.inc"
    }
    have_prev = true;
    pm0 = m0; pn0 = n0;
    int next_logical = logical + (int)gridDim.x;
    if (dyn) {
      if (tid == 0) *reinterpret_cast<volatile int*>(smem) = sched_resolve(g.sched, my_list, ticket, ntiles);
      __syncthreads();
      next_logical = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem));
      __syncthreads();
    }
    if (!(next_logical >= 0 && next_logical < ntiles)) break;
    logical = next_logical;
    set_tile(logical);
  }
  {   // the last tile of this workgroup: read-out + 32 stores, back to back
    int t2 = threadIdx.x;
    asm volatile("" : "+v"(t2));
    const int lane = t2 & 63, li = lane & 15, lq = lane >> 4;
    const uint32_t voffC = (uint32_t)(128 * wr + li) * (uint32_t)g.ldc * 2u + (uint32_t)(128 * wc + 8 * (((lq & 1) << 1) | (lq >> 1))) * 2u;
    const uint32_t colv = (uint32_t)(pn0 + 128 * wc + 4 * lq), voffBias = colv * 4u;
    const u32x4 srdC = rc_live;
    uint32_t s_crow = ((uint32_t)pm0 * (uint32_t)g.ldc + (uint32_t)pn0) * ES;
#include "gemm4w_drain_ovl.inc"
  }
  if (dyn && tid == 0) sched_leave(g.sched, gridDim.x);
}

// the overlapped form needs whole column tiles (its stores are not column-masked), a contraction long enough to hold the 32 stores of the
// previous tile (five peeled K tiles) and the plain bf16 epilogue
#ifndef G4_OVERLAP
#define G4_OVERLAP 1
#endif
static int g4_overlap = G4_OVERLAP;
bool crl_gemm4w_can_overlap(int layout, int epi, const GemmArgs& a, int nsplit) {
  if (!g4_overlap || nsplit != 1 || (a.N % 256) != 0 || (a.K % 64) != 0 || a.K / 64 < 8) return false;
  // a workgroup with one or two tiles pays the entry + drain statements for nothing (cfg-2, M = 19 208: 27.3 ms per step overlapped against
  // 27.05 classic, profiles/r5_cfg2_ab.txt): the overlapped form from three rounds of tiles on
  if ((int64_t)a.ntm * a.ntn < 3 * (int64_t)crl_gemm_cus() && g4_overlap != 7) return false;
  return epi == CRL_EPI_BF16 && layout != CRL_TN && (a.ldc % 8) == 0;
}

template <int LAYOUT>
int launch4w_ovl(const GemmArgs& a, hipStream_t s) {
  if (int rc = crl_enable_lds(reinterpret_cast<const void*>(&gemm4w_ovl_kernel<LAYOUT>), 163840, "crl_gemm_bf16(4w, overlapped epilogue)")) return rc;
  int grid_x = a.ntm * a.ntn;
  GemmArgs b = a;
  b.sched = nullptr;
  const int ncu = crl_gemm_cus();
  if (grid_x > ncu) {
    grid_x = ncu;
    if (crl_gemm_dynamic()) { bool ok; b.sched = crl_sched_slot(s, &ok); if (!ok) return -2; }
  }
  gemm4w_ovl_kernel<LAYOUT><<<dim3(grid_x, 1), T4W, 163840, s>>>(b);
  CRL_LAUNCH_CHECK("crl_gemm_bf16(4w, overlapped epilogue)");
  return 0;
}

template <int LAYOUT, int EPI>
int launch4w_one(const GemmArgs& a, int nsplit, hipStream_t s) {
  if (int rc = crl_enable_lds(reinterpret_cast<const void*>(&gemm4w_kernel<LAYOUT, EPI>), 163840, "crl_gemm_bf16(4w)")) return rc;
  int grid_x = a.ntm * a.ntn;
  GemmArgs b = a;
  b.sched = nullptr;
  const int ncu = crl_gemm_cus();
  if (nsplit == 1 && grid_x > ncu) {   // one resident workgroup per CU pulls tiles from the launch's ticket counters
    grid_x = ncu;
    if (crl_gemm_dynamic()) { bool ok; b.sched = crl_sched_slot(s, &ok); if (!ok) return -2; }
  }
  gemm4w_kernel<LAYOUT, EPI><<<dim3(grid_x, nsplit), T4W, 163840, s>>>(b);
  CRL_LAUNCH_CHECK("crl_gemm_bf16(4w)");
  return 0;
}

template <int LAYOUT>
int launch4w_epi(const GemmArgs& a, int epi, int nsplit, hipStream_t s) {
  switch (epi) {
    case CRL_EPI_BF16: return launch4w_one<LAYOUT, CRL_EPI_BF16>(a, nsplit, s);
    case CRL_EPI_BF16_GELU: return launch4w_one<LAYOUT, CRL_EPI_BF16_GELU>(a, nsplit, s);
    case CRL_EPI_BF16_DGELU: return launch4w_one<LAYOUT, CRL_EPI_BF16_DGELU>(a, nsplit, s);
    case CRL_EPI_F32_RESID: return launch4w_one<LAYOUT, CRL_EPI_F32_RESID>(a, nsplit, s);
    case CRL_EPI_F32: return launch4w_one<LAYOUT, CRL_EPI_F32>(a, nsplit, s);
    case CRL_EPI_F32_ACC: return launch4w_one<LAYOUT, CRL_EPI_F32_ACC>(a, nsplit, s);
  }
  crl_set_error("crl_gemm_bf16: bad epilogue %d", epi);
  return -1;
}

}  // namespace

// called by crl_gemm_bf16 (gemm.hip): same contract as crl_gemm256_launch
extern "C" int crl_gemm_set_overlap(int on) { g4_overlap = on & 7; return 0; }    // 0 off; 1 on (default); 7 = also for launches of fewer than three rounds of tiles (tests)
bool crl_gemm4w_overlaps(int layout, int epi, const gemmc::GemmArgs& a, int nsplit) { return crl_gemm4w_can_overlap(layout, epi, a, nsplit); }
int crl_gemm4w_launch(int layout, int epi, const gemmc::GemmArgs& a, int nsplit, hipStream_t s) {
  if (crl_gemm4w_can_overlap(layout, epi, a, nsplit)) {
    return layout == CRL_NT ? launch4w_ovl<CRL_NT>(a, s) : launch4w_ovl<CRL_NN>(a, s);
  }
  switch (layout) {
    case CRL_NT: return launch4w_epi<CRL_NT>(a, epi, nsplit, s);
    case CRL_NN: return launch4w_epi<CRL_NN>(a, epi, nsplit, s);
    default: return launch4w_epi<CRL_TN>(a, epi, nsplit, s);
  }
}
