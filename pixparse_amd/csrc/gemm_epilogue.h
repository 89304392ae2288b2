// Epilogue of the 256-row MFMA GEMM kernels (gemm256.hip: 256x256 tile, 8 waves; gemm4w.hip takes the arithmetic helpers).
// Both leave the same per-wave accumulator block: acc[qm][qn][i][j] = 16x16 tile at rows 128 qm + 64 wr + 16 i and columns
// QN qn + 32 wc + 16 j of the workgroup tile (QN = 128 / 64), lane (li = lane & 15, lq = lane >> 4) holding C[row li][col 4 lq + 0..3].
#pragma once
#include <type_traits>
#include "gemm_common.h"

namespace gemmc {

// Epilogue widening (guide T21).  After the MFMAs lane (li = lane & 15, lq = lane >> 4) holds, for each 32-column strip pair,
// columns 4 lq + 0..3 of the first 16-column strip (X) and of the second (Y): two 8-byte pieces 32 bytes apart in a bf16
// row.  v_permlane16_swap_b32 exchanges the odd 16-lane rows of X with the even rows of Y, which leaves every lane with
// EIGHT consecutive columns  8 (lq >> 1) + 16 (lq & 1) + 0..7  of the strip pair -- one 16-byte access instead of two
// 8-byte ones, 64 contiguous bytes per row and instruction instead of 32.  The exchange is an involution: applied to a
// 16-byte LOAD of those columns it returns the lane's own two pieces.
__device__ __forceinline__ void swap_strips(uint32_t& x0, uint32_t& x1, uint32_t& y0, uint32_t& y1) {
  const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
  x0 = r0[0]; y0 = r0[1]; x1 = r1[0]; y1 = r1[1];
}

// ---- Branch-free epilogue over bounds-checked buffer accesses (the default path) -------------------------------------------
// Every tile-sized access of the epilogue is a raw buffer load / store of 16 bytes per lane through a descriptor that ends
// at the last valid element: rows >= M fall past num_records by themselves, ragged columns are steered there with one
// v_cndmask per access, a missing bias is a zero-record descriptor.  Out-of-range loads return zeros and out-of-range
// stores are dropped by the hardware, so the whole epilogue is ONE basic block: hipcc schedules the loads of several row
// groups ahead of the arithmetic and waits with counted vmcnt.  (With `if (ok) store` predicates it cut the code into one
// block per store and put `global_load; s_waitcnt vmcnt(0)` chains between them: 4 dependent round trips for the bias, 8 to
// 32 for a residual / saved-activation tile -- 12 to 31 us per 256x256 tile against 22 us for the whole K = 1024 main loop.)
// All offsets are 32-bit: crl_gemm_bf16 checks that (M + 255) rows of C / aux / resid stay below 4 GiB.
typedef unsigned epi_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
#ifndef G_EPI_LD_AUX
#define G_EPI_LD_AUX 0    // cache-policy bits of the epilogue's tile loads / stores (gfx942+: 1 = sc0, 2 = nt, 16 = sc1): A/B knobs
#endif
#ifndef G_EPI_ST_AUX
#define G_EPI_ST_AUX 0
#endif
__device__ __forceinline__ epi_u4 epi_ld(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, G_EPI_LD_AUX); }
#ifndef G_EPI_SAVE_AUX
#define G_EPI_SAVE_AUX 0   // ... of the store of the GELU pre-activation (read again only by the backward, a whole forward + half a backward later)
#endif
__device__ __forceinline__ void epi_st(__amdgpu_buffer_rsrc_t r, uint32_t off, epi_u4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, G_EPI_ST_AUX); }
__device__ __forceinline__ void epi_st_saved(__amdgpu_buffer_rsrc_t r, uint32_t off, epi_u4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, G_EPI_SAVE_AUX); }

#ifndef G_EPI_XPOSE
#define G_EPI_XPOSE 1
#endif
// Lane transposition in front of every tile-sized access (G_EPI_XPOSE).  The MFMAs leave lane (li = lane & 15, lq = lane >> 4)
// with 16 bytes of ROW li: the 16 lanes of a quarter-wave touch 16 different rows, so a 1-KiB wave access reaches the
// texture-address unit as 64 separate 16-byte requests -- the s_memtime timeline of the persistent kernel (scripts/
// gemm_timeline.py) shows the epilogues moving 8-13 bytes per clock and CU, unchanged with 32 or 256 CUs active, i.e. bound by
// the CU's request rate, not by HBM.  Four ds_bpermute_b32 per 16 bytes (LDS crossbar only, no LDS storage) re-deal the data so
// that lane d owns chunk (d & 3) of row (d >> 2): four neighbouring lanes then cover 64 contiguous bytes and the same access is
// 16 requests.  Arithmetic stays in the MFMA ("math") layout; loads are issued in the access layout and dealt back.
template <bool BF16>
struct EpiLanes {
  int ar, ac;            // access role: row within the 16-row group, 16-byte chunk within the row segment
  int fwd, inv;          // ds_bpermute byte addresses: math -> access, access -> math
  __device__ __forceinline__ explicit EpiLanes(int lane) {
    const int li = lane & 15, lq = lane >> 4;
    auto bitswap = [](int c) { return ((c & 1) << 1) | (c >> 1); };   // bf16: chunk c of a strip pair lives in lane group lq = bitswap(c) after swap_strips
    if (G_EPI_XPOSE) {
      ar = lane >> 2; ac = lane & 3;
      fwd = 4 * (ar + 16 * (BF16 ? bitswap(ac) : ac));
      inv = 4 * (4 * li + (BF16 ? bitswap(lq) : lq));
    } else {
      ar = li; ac = BF16 ? bitswap(lq) : lq;
      fwd = inv = 4 * lane;
    }
  }
  __device__ __forceinline__ epi_u4 to_access(epi_u4 v) const {
    if (G_EPI_XPOSE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (unsigned)__builtin_amdgcn_ds_bpermute(fwd, (int)v[k]);
    }
    return v;
  }
  __device__ __forceinline__ epi_u4 to_math(epi_u4 v) const {
    if (G_EPI_XPOSE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (unsigned)__builtin_amdgcn_ds_bpermute(inv, (int)v[k]);
    }
    return v;
  }
};

template <int EPI, int QN>
__device__ __forceinline__ void epilogue_tile_buf(const GemmArgs& g, f32x4 (&acc)[2][2][4][2], int m0e, int n0e, int wr, int wc, int lane_e,
                                                  size_t slab_off) {
  constexpr bool BIAS_EPI = (EPI == CRL_EPI_BF16 || EPI == CRL_EPI_BF16_GELU || EPI == CRL_EPI_F32_RESID);
  constexpr bool BF16_OUT = (EPI == CRL_EPI_BF16 || EPI == CRL_EPI_BF16_GELU || EPI == CRL_EPI_BF16_DGELU);
  constexpr uint32_t ES = BF16_OUT ? 2u : 4u;
  const int lq = lane_e >> 4;
  const EpiLanes<BF16_OUT> L(lane_e);
  const uint32_t Mu = (uint32_t)g.M, Nu = (uint32_t)g.N;
  const uint32_t nrecC = (uint32_t)(((size_t)(Mu - 1) * (uint32_t)g.ldc + Nu) * ES);
  const __amdgpu_buffer_rsrc_t rC = epi_rsrc((const char*)g.C + slab_off * 4, nrecC);
  const uint32_t mrow = (uint32_t)(m0e + 64 * wr + L.ar);            // access row of (qm, i) = mrow + 128 qm + 16 i
  const uint32_t rstepC = 16u * (uint32_t)g.ldc * ES;                  // byte distance between row groups i, i + 1 (x 8 between qm)
  const uint32_t rbaseC = mrow * (uint32_t)g.ldc * ES;

  // the lane's 16 bias values (its accumulator columns, rounded to bf16 like autocast): four loads in flight together
  float bw[2][2][4];
  if constexpr (BIAS_EPI) {
    const __amdgpu_buffer_rsrc_t rB = epi_rsrc(g.bias, g.bias ? Nu * 4u : 0u);
    epi_u4 braw[2][2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int j = 0; j < 2; ++j) braw[pr][j] = epi_ld(rB, (uint32_t)(n0e + QN * pr + 32 * wc + 16 * j + 4 * lq) * 4u);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bw[pr][j][r] = round_bf(__uint_as_float(braw[pr][j][r]));
  }

  // column scale of the plain bf16 epilogue (q columns of a q|k|v projection): per 16-column strip of the lane's math layout
  float cs[2][2];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int j = 0; j < 2; ++j) cs[pr][j] = (n0e + QN * pr + 32 * wc + 16 * j + 4 * lq) < g.colscale_cols ? g.colscale : 1.f;
  if constexpr (BF16_OUT) {
    // bf16 outputs: after swap_strips a lane owns 8 consecutive columns of a 32-column strip pair -> one 16-byte access per (qm, i, pr)
    uint32_t cC[2], cA[2];
    bool okc[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const uint32_t n = (uint32_t)(n0e + QN * pr + 32 * wc + 8 * L.ac);
      okc[pr] = n < Nu;
      cC[pr] = rbaseC + n * 2u;
      cA[pr] = mrow * (uint32_t)g.ldaux * 2u + n * 2u;
    }
    const uint32_t nrecA = (EPI == CRL_EPI_BF16) ? 0u : (uint32_t)(((size_t)(Mu - 1) * (uint32_t)g.ldaux + Nu) * 2u);
    const __amdgpu_buffer_rsrc_t rA = epi_rsrc(g.aux, nrecA);
    const uint32_t rstepA = 16u * (uint32_t)g.ldaux * 2u;
    auto offC = [&](int qm, int i, int pr) { return okc[pr] ? cC[pr] + (uint32_t)(8 * qm + i) * rstepC : nrecC; };
    auto offA = [&](int qm, int i, int pr) { return okc[pr] ? cA[pr] + (uint32_t)(8 * qm + i) * rstepA : nrecA; };
    // saved derivatives of the dGELU epilogue: ALL sixteen 16-byte loads of the lane are issued before the first use (64 VGPRs; the
    // operand fragments of the K loop are dead here) -- one memory round trip per tile instead of one per batch.  Loads issued between
    // the stores would also be counted behind them (vmcnt retires in issue order), i.e. wait for write acknowledgements.
    epi_u4 hw[4][4];
    if constexpr (EPI == CRL_EPI_BF16_DGELU) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) hw[b][t] = epi_ld(rA, offA(b >> 1, 2 * (b & 1) + (t >> 1), t & 1));
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int qm = b >> 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 2 * (b & 1) + (t >> 1), pr = t & 1;
        f32x2 v[2][2];     // [strip j][column pair]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 a4 = acc[qm][pr][i][j];
          v[j][0] = f32x2{a4[0], a4[1]}; v[j][1] = f32x2{a4[2], a4[3]};
          if constexpr (BIAS_EPI) { v[j][0] += f32x2{bw[pr][j][0], bw[pr][j][1]}; v[j][1] += f32x2{bw[pr][j][2], bw[pr][j][3]}; }
          if constexpr (EPI == CRL_EPI_BF16) { v[j][0] *= cs[pr][j]; v[j][1] *= cs[pr][j]; }
        }
        uint32_t x0, x1, y0, y1;
        if constexpr (EPI == CRL_EPI_BF16) {
          x0 = pack_bf2v(v[0][0]); x1 = pack_bf2v(v[0][1]); y0 = pack_bf2v(v[1][0]); y1 = pack_bf2v(v[1][1]);
          swap_strips(x0, x1, y0, y1);
          epi_st(rC, offC(qm, i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
        } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
          // h = bf16(v + b) (the rounding the reference's autocast applies) is expanded back to fp32 for the activation; what is SAVED for the
          // backward is gelu'(h) as fp16 (one erf / gauss evaluation serves both), so that the dgrad epilogue only multiplies
          f32x2 ga[4], da[4];
          gelu_grad2(unpack_bf2(pack_bf2v(v[0][0])), ga[0], da[0]); gelu_grad2(unpack_bf2(pack_bf2v(v[0][1])), ga[1], da[1]);
          gelu_grad2(unpack_bf2(pack_bf2v(v[1][0])), ga[2], da[2]); gelu_grad2(unpack_bf2(pack_bf2v(v[1][1])), ga[3], da[3]);
          x0 = pack_h2v(da[0]); x1 = pack_h2v(da[1]); y0 = pack_h2v(da[2]); y1 = pack_h2v(da[3]);
          swap_strips(x0, x1, y0, y1);
          epi_st_saved(rA, offA(qm, i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
          x0 = pack_bf2v(ga[0]); x1 = pack_bf2v(ga[1]); y0 = pack_bf2v(ga[2]); y1 = pack_bf2v(ga[3]);
          swap_strips(x0, x1, y0, y1);
          epi_st(rC, offC(qm, i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
        } else {   // dGELU: the 16-byte load holds 8 consecutive saved derivatives (fp16); the exchange (an involution) returns this lane's own
          const epi_u4 hm = L.to_math(hw[b][t]);
          uint32_t a0 = hm[0], a1 = hm[1], b0 = hm[2], b1 = hm[3];
          swap_strips(a0, a1, b0, b1);
          x0 = pack_bf2v(unpack_bf2(pack_bf2v(v[0][0])) * unpack_h2(a0));
          x1 = pack_bf2v(unpack_bf2(pack_bf2v(v[0][1])) * unpack_h2(a1));
          y0 = pack_bf2v(unpack_bf2(pack_bf2v(v[1][0])) * unpack_h2(b0));
          y1 = pack_bf2v(unpack_bf2(pack_bf2v(v[1][1])) * unpack_h2(b1));
          swap_strips(x0, x1, y0, y1);
          epi_st(rC, offC(qm, i, pr), L.to_access(epi_u4{x0, x1, y0, y1}));
        }
      }
    }
  } else {
    // fp32 outputs: math lane (li, lq) owns columns 4 lq + 0..3 of each 16-column strip -> one 16-byte access per (qm, i, qn, j)
    uint32_t cC[4], cR[4];
    bool okc[4];
    constexpr bool READS = (EPI == CRL_EPI_F32_RESID || EPI == CRL_EPI_F32_ACC);
    const uint32_t ldr = (EPI == CRL_EPI_F32_RESID) ? (uint32_t)g.ldr : (uint32_t)g.ldc;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const uint32_t n = (uint32_t)(n0e + QN * (c4 >> 1) + 32 * wc + 16 * (c4 & 1) + 4 * L.ac);
      okc[c4] = n < Nu;
      cC[c4] = rbaseC + n * 4u;
      cR[c4] = mrow * ldr * 4u + n * 4u;
    }
    const uint32_t nrecR = READS ? (uint32_t)(((size_t)(Mu - 1) * ldr + Nu) * 4u) : 0u;
    const __amdgpu_buffer_rsrc_t rR = epi_rsrc(EPI == CRL_EPI_F32_RESID ? (const void*)g.resid : (const void*)g.C, nrecR);
    const uint32_t rstepR = 16u * ldr * 4u;
    auto offC = [&](int rg, int c4) { return okc[c4] ? cC[c4] + (uint32_t)rg * rstepC : nrecC; };
    auto offR = [&](int rg, int c4) { return okc[c4] ? cR[c4] + (uint32_t)rg * rstepR : nrecR; };
    // residual / accumulated-gradient tile in two halves (qm) of sixteen 16-byte loads; issue order L(0) L(1) S(0) S(1): the loads of
    // the second half go out BEFORE the stores of the first (vmcnt retires in issue order, a load behind a store would wait for its
    // write acknowledgement), and each half is one memory round trip instead of one per row group.
    epi_u4 rv[2][16];
    auto issue = [&](int h) {
#pragma unroll
      for (int u = 0; u < 16; ++u) rv[h][u] = epi_ld(rR, offR(8 * h + (u >> 2), u & 3));
    };
    auto half = [&](int qm) {
      epi_u4 res[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = u >> 2, c4 = u & 3;
        const f32x4 a4 = acc[qm][c4 >> 1][i][c4 & 1];
        epi_u4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = a4[r];
          if constexpr (BIAS_EPI) v += bw[c4 >> 1][c4 & 1][r];
          if constexpr (EPI == CRL_EPI_F32_RESID) v = round_bf(v);
          w[r] = __float_as_uint(v);
        }
        w = L.to_access(w);
        if constexpr (READS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = __float_as_uint(__uint_as_float(rv[qm][u][r]) + __uint_as_float(w[r]));
        }
        res[u] = w;
      }
      if (READS && qm == 0) issue(1);
#pragma unroll
      for (int u = 0; u < 16; ++u) epi_st(rC, offC(8 * qm + (u >> 2), u & 3), res[u]);
    };
    if constexpr (READS) issue(0);
    half(0);
    half(1);
  }
}

template <int EPI, int QN>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& g, f32x4 (&acc)[2][2][4][2], int m0e, int n0e, int wr, int wc, int lane_e,
                                              size_t slab_off) {
  const int li = lane_e & 15, lq = lane_e >> 4;
  // Branch-free loads: every epilogue operand (bias, residual, saved pre-activation, accumulated gradient) of a row's four
  // column groups is fetched from CLAMPED coordinates before anything is stored, so the loads of a row -- and, registers
  // permitting, of the next row -- are in flight together; only the stores are predicated. (With the bounds checks as
  // branches around each group, hipcc serialised load -> wait -> store 32 times per lane: ~45 us of latency per tile.)
  constexpr bool READS_TILE = (EPI == CRL_EPI_BF16_DGELU || EPI == CRL_EPI_F32_RESID || EPI == CRL_EPI_F32_ACC);
  auto epilogue = [&](auto has_bias_c) {
  constexpr bool HAS_BIAS = decltype(has_bias_c)::value;
  float bh[4][4];      // the lane's bias values (rounded to bf16 like autocast), loaded once per tile
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      int nb = n0e + QN * (c4 >> 1) + 32 * wc + 16 * (c4 & 1) + 4 * lq;
      nb = nb < g.N ? nb : g.N - 4;
      const float4 b = *reinterpret_cast<const float4*>(g.bias + nb);
      bh[c4][0] = round_bf(b.x); bh[c4][1] = round_bf(b.y); bh[c4][2] = round_bf(b.z); bh[c4][3] = round_bf(b.w);
    }
  }
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0e + 128 * qm + 64 * wr + 16 * i + li;
      const bool m_ok = m < g.M;
      const size_t mc = (size_t)(m_ok ? m : g.M - 1);
      int nn[4];
      bool ok[4];
      float4 rv[4];
      uint2 hv[4];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const int n = n0e + QN * (c4 >> 1) + 32 * wc + 16 * (c4 & 1) + 4 * lq;
        ok[c4] = m_ok && n < g.N;
        nn[c4] = n < g.N ? n : g.N - 4;
        if constexpr (EPI == CRL_EPI_BF16_DGELU) hv[c4] = *reinterpret_cast<const uint2*>((const u16*)g.aux + mc * g.ldaux + nn[c4]);
        if constexpr (EPI == CRL_EPI_F32_RESID) rv[c4] = *reinterpret_cast<const float4*>(g.resid + mc * g.ldr + nn[c4]);
        if constexpr (EPI == CRL_EPI_F32_ACC) rv[c4] = *reinterpret_cast<const float4*>((const float*)g.C + mc * g.ldc + nn[c4]);
      }
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const f32x4 a4 = acc[qm][c4 >> 1][i][c4 & 1];
        float v[4] = {a4[0], a4[1], a4[2], a4[3]};
        const size_t n = (size_t)nn[c4];
        if constexpr (HAS_BIAS) { v[0] += bh[c4][0]; v[1] += bh[c4][1]; v[2] += bh[c4][2]; v[3] += bh[c4][3]; }
        if constexpr (EPI == CRL_EPI_BF16) {
          if (ok[c4]) *reinterpret_cast<uint2*>((u16*)g.C + mc * g.ldc + n) = uint2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
          float h[4], y[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) gelu_grad_f(round_bf(v[r]), y[r], h[r]);      // h := gelu'(bf16(v + b)), saved as fp16
          if (ok[c4]) {
            *reinterpret_cast<uint2*>((u16*)g.aux + mc * g.ldaux + n) = uint2{pack_h2(h[0], h[1]), pack_h2(h[2], h[3])};
            *reinterpret_cast<uint2*>((u16*)g.C + mc * g.ldc + n) = uint2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
          }
        } else if constexpr (EPI == CRL_EPI_BF16_DGELU) {
          const uint2 hh = hv[c4];
          const float y0 = round_bf(v[0]) * h2f_lo(hh.x), y1 = round_bf(v[1]) * h2f_hi(hh.x);
          const float y2 = round_bf(v[2]) * h2f_lo(hh.y), y3 = round_bf(v[3]) * h2f_hi(hh.y);
          if (ok[c4]) *reinterpret_cast<uint2*>((u16*)g.C + mc * g.ldc + n) = uint2{pack_bf2(y0, y1), pack_bf2(y2, y3)};
        } else if constexpr (EPI == CRL_EPI_F32_RESID) {
          const float4 r = rv[c4];
          if (ok[c4]) *reinterpret_cast<float4*>((float*)g.C + mc * g.ldc + n) =
              float4{r.x + round_bf(v[0]), r.y + round_bf(v[1]), r.z + round_bf(v[2]), r.w + round_bf(v[3])};
        } else if constexpr (EPI == CRL_EPI_F32) {
          if (ok[c4]) *reinterpret_cast<float4*>((float*)g.C + slab_off + mc * g.ldc + n) = float4{v[0], v[1], v[2], v[3]};
        } else {
          const float4 o = rv[c4];
          if (ok[c4]) *reinterpret_cast<float4*>((float*)g.C + mc * g.ldc + n) = float4{o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]};
        }
      }
    }
  };
  constexpr bool BIAS_EPI = (EPI == CRL_EPI_BF16 || EPI == CRL_EPI_BF16_GELU || EPI == CRL_EPI_F32_RESID);
  constexpr bool BF16_OUT = (EPI == CRL_EPI_BF16 || EPI == CRL_EPI_BF16_GELU || EPI == CRL_EPI_BF16_DGELU);
#ifndef G_WIDE
#define G_WIDE 1
#endif
  bool wide = false;
  if constexpr (BF16_OUT && G_WIDE) wide = ((g.N | g.ldc | ((EPI != CRL_EPI_BF16) ? g.ldaux : 0)) & 7) == 0;
#ifndef G_EPI_BUF
#define G_EPI_BUF 1   // 1 = branch-free buffer-access epilogue (epilogue_tile_buf) wherever 16-byte accesses apply; 0 = the predicated forms below (A/B)
#endif
  if (G_EPI_BUF && (wide || !BF16_OUT)) {
    epilogue_tile_buf<EPI, QN>(g, acc, m0e, n0e, wr, wc, lane_e, slab_off);
    return;
  }
  if (wide) {
    // bf16 outputs, 16-byte accesses: no lane leaves the code before the lane exchanges (partners share li, i.e. the row);
    // loads use clamped coordinates, only the stores are predicated
    if constexpr (BF16_OUT) {
      const bool has_bias = BIAS_EPI && g.bias != nullptr;
      const int wcol = 8 * (lq >> 1) + 16 * (lq & 1);       // this lane's 8 columns inside a 32-column strip pair after the exchange
      // the lane's 16 bias values (its own columns, already rounded to bf16 like autocast) are loaded ONCE per tile, not per row
      float bw[2][2][4];
      if constexpr (BIAS_EPI) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            int nb = n0e + QN * pr + 32 * wc + 16 * j + 4 * lq;
            nb = nb < g.N ? nb : g.N - 4;
            float4 b{0.f, 0.f, 0.f, 0.f};
            if (has_bias) b = *reinterpret_cast<const float4*>(g.bias + nb);
            bw[pr][j][0] = round_bf(b.x); bw[pr][j][1] = round_bf(b.y); bw[pr][j][2] = round_bf(b.z); bw[pr][j][3] = round_bf(b.w);
          }
      }
#pragma unroll
      for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0e + 128 * qm + 64 * wr + 16 * i + li;
          const bool m_ok = m < g.M;
          const size_t mc = (size_t)(m_ok ? m : g.M - 1);
          int nw[2];
          bool ok[2];
          uint4 hw[2];
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int n = n0e + QN * pr + 32 * wc + wcol;
            ok[pr] = m_ok && n < g.N;
            nw[pr] = n < g.N ? n : g.N - 8;
            if constexpr (EPI == CRL_EPI_BF16_DGELU) hw[pr] = *reinterpret_cast<const uint4*>((const u16*)g.aux + mc * g.ldaux + nw[pr]);
          }
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            float v[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x4 a4 = acc[qm][pr][i][j];
              v[j][0] = a4[0]; v[j][1] = a4[1]; v[j][2] = a4[2]; v[j][3] = a4[3];
              if constexpr (BIAS_EPI) { v[j][0] += bw[pr][j][0]; v[j][1] += bw[pr][j][1]; v[j][2] += bw[pr][j][2]; v[j][3] += bw[pr][j][3]; }
            }
            uint32_t x0, x1, y0, y1;
            u16* crow = (u16*)g.C + mc * g.ldc + nw[pr];
            if constexpr (EPI == CRL_EPI_BF16) {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const float f = (n0e + QN * pr + 32 * wc + 16 * j + 4 * lq) < g.colscale_cols ? g.colscale : 1.f;
                v[j][0] *= f; v[j][1] *= f; v[j][2] *= f; v[j][3] *= f;
              }
              x0 = pack_bf2(v[0][0], v[0][1]); x1 = pack_bf2(v[0][2], v[0][3]);
              y0 = pack_bf2(v[1][0], v[1][1]); y1 = pack_bf2(v[1][2], v[1][3]);
              swap_strips(x0, x1, y0, y1);
              if (ok[pr]) *reinterpret_cast<uint4*>(crow) = uint4{x0, x1, y0, y1};
            } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
              float h[2][4], y[2][4];
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) gelu_grad_f(round_bf(v[j][r]), y[j][r], h[j][r]);      // h := gelu'(bf16(v + b)), saved as fp16
              x0 = pack_h2(h[0][0], h[0][1]); x1 = pack_h2(h[0][2], h[0][3]);
              y0 = pack_h2(h[1][0], h[1][1]); y1 = pack_h2(h[1][2], h[1][3]);
              swap_strips(x0, x1, y0, y1);
              if (ok[pr]) *reinterpret_cast<uint4*>((u16*)g.aux + mc * g.ldaux + nw[pr]) = uint4{x0, x1, y0, y1};
              x0 = pack_bf2(y[0][0], y[0][1]); x1 = pack_bf2(y[0][2], y[0][3]);
              y0 = pack_bf2(y[1][0], y[1][1]); y1 = pack_bf2(y[1][2], y[1][3]);
              swap_strips(x0, x1, y0, y1);
              if (ok[pr]) *reinterpret_cast<uint4*>(crow) = uint4{x0, x1, y0, y1};
            } else {   // CRL_EPI_BF16_DGELU: the 16-byte load holds 8 consecutive saved derivatives (fp16); the exchange returns this lane's own
              uint32_t a0 = hw[pr].x, a1 = hw[pr].y, b0 = hw[pr].z, b1 = hw[pr].w;
              swap_strips(a0, a1, b0, b1);
              const uint32_t hh[2][2] = {{a0, a1}, {b0, b1}};
              float y[2][4];
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                y[j][0] = round_bf(v[j][0]) * h2f_lo(hh[j][0]);
                y[j][1] = round_bf(v[j][1]) * h2f_hi(hh[j][0]);
                y[j][2] = round_bf(v[j][2]) * h2f_lo(hh[j][1]);
                y[j][3] = round_bf(v[j][3]) * h2f_hi(hh[j][1]);
              }
              x0 = pack_bf2(y[0][0], y[0][1]); x1 = pack_bf2(y[0][2], y[0][3]);
              y0 = pack_bf2(y[1][0], y[1][1]); y1 = pack_bf2(y[1][2], y[1][3]);
              swap_strips(x0, x1, y0, y1);
              if (ok[pr]) *reinterpret_cast<uint4*>(crow) = uint4{x0, x1, y0, y1};
            }
          }
        }
    }
  } else if constexpr (READS_TILE) {
    if (BIAS_EPI && g.bias) epilogue(std::true_type{}); else epilogue(std::false_type{});
  } else {
    // store-only epilogues (bias from L1, nothing tile-sized to read): group by group; same-box A/B has this form 1-3 %
    // ahead of the batched one for them
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0e + 128 * qm + 64 * wr + 16 * i + li;
        if (m >= g.M) continue;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const int n = n0e + QN * (c4 >> 1) + 32 * wc + 16 * (c4 & 1) + 4 * lq;
          if (n >= g.N) continue;
          const f32x4 a4 = acc[qm][c4 >> 1][i][c4 & 1];
          float v[4] = {a4[0], a4[1], a4[2], a4[3]};
          if constexpr (BIAS_EPI) {
            if (g.bias) {
              const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
              v[0] += round_bf(b.x); v[1] += round_bf(b.y); v[2] += round_bf(b.z); v[3] += round_bf(b.w);
            }
          }
          if constexpr (EPI == CRL_EPI_BF16) {
            const float f = n < g.colscale_cols ? g.colscale : 1.f;
            *reinterpret_cast<uint2*>((u16*)g.C + (size_t)m * g.ldc + n) = uint2{pack_bf2(v[0] * f, v[1] * f), pack_bf2(v[2] * f, v[3] * f)};
          } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
            float h[4], y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gelu_grad_f(round_bf(v[r]), y[r], h[r]);      // h := gelu'(bf16(v + b)), saved as fp16
            *reinterpret_cast<uint2*>((u16*)g.aux + (size_t)m * g.ldaux + n) = uint2{pack_h2(h[0], h[1]), pack_h2(h[2], h[3])};
            *reinterpret_cast<uint2*>((u16*)g.C + (size_t)m * g.ldc + n) = uint2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
          } else {   // CRL_EPI_F32: split-K slab
            *reinterpret_cast<float4*>((float*)g.C + slab_off + (size_t)m * g.ldc + n) = float4{v[0], v[1], v[2], v[3]};
          }
        }
      }
  }
}

}  // namespace gemmc
