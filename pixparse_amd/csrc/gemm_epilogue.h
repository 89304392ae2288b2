// Epilogue of the 256-row MFMA GEMM kernels (gemm256.hip: 256x256 tile, 8 waves; gemm2x.hip: 256x128 tile, 4 waves).
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
          for (int r = 0; r < 4; ++r) { h[r] = round_bf(v[r]); y[r] = gelu_f(h[r]); }
          if (ok[c4]) {
            *reinterpret_cast<uint2*>((u16*)g.aux + mc * g.ldaux + n) = uint2{pack_bf2(h[0], h[1]), pack_bf2(h[2], h[3])};
            *reinterpret_cast<uint2*>((u16*)g.C + mc * g.ldc + n) = uint2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
          }
        } else if constexpr (EPI == CRL_EPI_BF16_DGELU) {
          const uint2 hh = hv[c4];
          const float h0 = bf2f(hh.x & 0xffff), h1 = bf2f(hh.x >> 16), h2 = bf2f(hh.y & 0xffff), h3 = bf2f(hh.y >> 16);
          const float y0 = round_bf(v[0]) * dgelu_f(h0), y1 = round_bf(v[1]) * dgelu_f(h1);
          const float y2 = round_bf(v[2]) * dgelu_f(h2), y3 = round_bf(v[3]) * dgelu_f(h3);
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
              x0 = pack_bf2(v[0][0], v[0][1]); x1 = pack_bf2(v[0][2], v[0][3]);
              y0 = pack_bf2(v[1][0], v[1][1]); y1 = pack_bf2(v[1][2], v[1][3]);
              swap_strips(x0, x1, y0, y1);
              if (ok[pr]) *reinterpret_cast<uint4*>(crow) = uint4{x0, x1, y0, y1};
            } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
              float h[2][4], y[2][4];
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { h[j][r] = round_bf(v[j][r]); y[j][r] = gelu_f(h[j][r]); }
              x0 = pack_bf2(h[0][0], h[0][1]); x1 = pack_bf2(h[0][2], h[0][3]);
              y0 = pack_bf2(h[1][0], h[1][1]); y1 = pack_bf2(h[1][2], h[1][3]);
              swap_strips(x0, x1, y0, y1);
              if (ok[pr]) *reinterpret_cast<uint4*>((u16*)g.aux + mc * g.ldaux + nw[pr]) = uint4{x0, x1, y0, y1};
              x0 = pack_bf2(y[0][0], y[0][1]); x1 = pack_bf2(y[0][2], y[0][3]);
              y0 = pack_bf2(y[1][0], y[1][1]); y1 = pack_bf2(y[1][2], y[1][3]);
              swap_strips(x0, x1, y0, y1);
              if (ok[pr]) *reinterpret_cast<uint4*>(crow) = uint4{x0, x1, y0, y1};
            } else {   // CRL_EPI_BF16_DGELU: the 16-byte load holds 8 consecutive saved pre-activations; the exchange returns this lane's own
              uint32_t a0 = hw[pr].x, a1 = hw[pr].y, b0 = hw[pr].z, b1 = hw[pr].w;
              swap_strips(a0, a1, b0, b1);
              const uint32_t hh[2][2] = {{a0, a1}, {b0, b1}};
              float y[2][4];
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                y[j][0] = round_bf(v[j][0]) * dgelu_f(bf2f(hh[j][0] & 0xffff));
                y[j][1] = round_bf(v[j][1]) * dgelu_f(bf2f(hh[j][0] >> 16));
                y[j][2] = round_bf(v[j][2]) * dgelu_f(bf2f(hh[j][1] & 0xffff));
                y[j][3] = round_bf(v[j][3]) * dgelu_f(bf2f(hh[j][1] >> 16));
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
            *reinterpret_cast<uint2*>((u16*)g.C + (size_t)m * g.ldc + n) = uint2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
          } else if constexpr (EPI == CRL_EPI_BF16_GELU) {
            float h[4], y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { h[r] = round_bf(v[r]); y[r] = gelu_f(h[r]); }
            *reinterpret_cast<uint2*>((u16*)g.aux + (size_t)m * g.ldaux + n) = uint2{pack_bf2(h[0], h[1]), pack_bf2(h[2], h[3])};
            *reinterpret_cast<uint2*>((u16*)g.C + (size_t)m * g.ldc + n) = uint2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
          } else {   // CRL_EPI_F32: split-K slab
            *reinterpret_cast<float4*>((float*)g.C + slab_off + (size_t)m * g.ldc + n) = float4{v[0], v[1], v[2], v[3]};
          }
        }
      }
  }
}

}  // namespace gemmc
