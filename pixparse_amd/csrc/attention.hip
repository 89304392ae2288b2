// Flash attention forward / backward for head_dim 64 on gfx950 (ViT global MHSA, BART causal
// self-attention, encoder-decoder cross attention) -- include/crl.h: crl_attn_fwd / crl_attn_bwd.
//
// All three kernels use v_mfma_f32_32x32x16_bf16 with the *reduction index of the following
// product kept in the accumulator registers* so that no probability tile ever crosses LDS
// (guide §3 "An accumulator tile as the next MFMA's operand"):
//   fwd   : S^T = K.Q^T  (query on the lane, keys in registers) -> softmax per lane ->
//           O^T += V^T.P^T   (V^T fragments by ds_read_b64_tr_b16 from the row-major V tile)
//   dK/dV : S = Q.K^T, dP = dO.V^T (key on the lane, queries in registers) ->
//           dV^T += dO^T.P, dK^T += Q^T.dS   (Q^T/dO^T by transposed LDS reads)
//   dQ    : S^T = K.Q^T, dP^T = V.dO^T (query on the lane) -> dQ^T += K^T.dS^T
// dQ is a second pass that recomputes S and dP (7 products instead of 5): deterministic and free
// of float atomics -- at N = 6189 an atomic dQ would be bound by the ~1.3 TB/s chip-wide atomic
// rate (DESIGN.md "attention backward").
// Tiles of 64 rows x 64 bf16 (128-B rows) are staged by bounds-checked LDS-DMA (rows past the end
// of the sequence arrive as zeros) with one XOR swizzle that is conflict-free for BOTH the
// ds_read_b128 row reads and the transposed reads.
#include <algorithm>
#include <type_traits>
#include <mutex>
#include <vector>
#include "common.h"
#include "attn_frag.h"
#include "gemm_common.h"      // the ticket scheduler of the persistent kernels (sched_*: the single-pass attention backward pulls its chains the same way)

namespace {
using namespace attnf;

#ifndef PRIO_MFMA
#define PRIO_MFMA 1
#endif
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnArgs {
  const u16 *q, *k, *v, *o, *d_o;
  u16 *out, *dq, *dk, *dv;
  float* lse;          // fwd: written; bwd: read
  float* delta;
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
  int B, H, Nq, Nk, causal;
  float scale;
  int nqt, nkt;
  int fused_delta;     // dQ pass: compute delta = rowsum(dO o O) and -lse/scale itself (and store them for the dK/dV pass that follows)
  // attention-probability dropout (transformers BartAttention: dropout on the softmax output, p = config.attention_dropout): element
  // (b, h, q, k) is kept iff hash(((b H + h) Nq + q) Nk + k, drop_key) >> 8 >= drop_thr; kept probabilities are scaled by drop_scale.
  // The mask is never stored: the forward and both backward passes re-evaluate it (DROP template argument of the kernels).
  uint32_t drop_thr, drop_key;
  float drop_scale;
  float dq_mul;        // factor on dQ at the store: the softmax scale (with q_prescaled the kernels' own `scale` is ln 2, see crl_attn_bwd)
};

// keep decision of attention dropout: a 32-bit integer hash (multiply / xor-shift rounds, full avalanche) of the element index, keyed by
// (seed, step, site) through `key`; 24 bits are compared with the threshold
__host__ __device__ __forceinline__ uint32_t attn_drop_hash(uint32_t x, uint32_t key) {
  x ^= key;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x >> 8;
}
static inline uint32_t attn_drop_key(uint64_t seed, uint32_t step, uint32_t site) {
  uint32_t k = (uint32_t)seed * 0x9E3779B1u ^ (uint32_t)(seed >> 32) * 0x85EBCA77u ^ step * 0xC2B2AE3Du ^ site * 0x27D4EB2Fu;
  k ^= k >> 15; k *= 0x2c1b3c6du; k ^= k >> 12;
  return k;
}

// LDS-DMA staging of a 64-row x 64-bf16 tile: per-thread source offset (swizzled chunk of row tid>>3, rows +32 for
// the second chunk) in a VGPR, everything wave-uniform in the scalar offset.
struct StageOff { uint32_t v; };
__device__ __forceinline__ StageOff make_stage_off(int tid, int64_t rs) {
  const int row = tid >> 3, pc = tid & 7;
  StageOff s{(uint32_t)(((int64_t)row * rs + ((pc ^ swz64(row)) * 8)) * 2)};
  asm volatile("" : "+v"(s.v));
  return s;
}
__device__ __forceinline__ void stage64(u32x4 srd, uint32_t lds_tile, StageOff so, int row0, int64_t rs, int wave) {
  const uint32_t soff = (uint32_t)((int64_t)row0 * rs * 2);
  dma16(srd, lds_tile + (uint32_t)wave * 1024u, so.v, soff);
  dma16(srd, lds_tile + 4096u + (uint32_t)wave * 1024u, so.v, soff + (uint32_t)(32 * rs * 2));
}
__device__ __forceinline__ void dma4(u32x4 srd, uint32_t lds_base, uint32_t voff, uint32_t soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void block_to_bh_tile(int bid, int ntile, int BH, int& bh, int& tile) {
  if ((BH & 7) == 0) {  // heads of one (b,h) stay on one XCD (blocks b, b+8 share an XCD): K/V reuse in its L2
    const int x = bid & 7, r = bid >> 3;
    bh = x + 8 * (r / ntile);
    tile = r % ntile;
  } else {
    bh = bid / ntile;
    tile = bid % ntile;
  }
}

// ======================================================================================= forward
template <int V> using ic = std::integral_constant<int, V>;

#ifndef FWD_OCC
#define FWD_OCC 4
#endif
#ifndef FWD_ONES
#define FWD_ONES 0   // 1 = row sums of P on the MFMA pipe (constant ones operand) instead of 32 VALU adds per tile; A/B on one box: 3 % slower
#endif
template <bool CAUSAL, bool DROP = false>
__global__ __launch_bounds__(256, FWD_OCC) void attn_fwd_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 8192];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, hh = lane >> 5;
  int bh, qt;
  block_to_bh_tile(blockIdx.x, a.nqt, a.B * a.H, bh, qt);
  const int b = bh / a.H, h = bh % a.H;
  const int off = a.Nk - a.Nq;

  const u16* qp = a.q + b * a.q_bs + h * 64;
  const u32x4 rk = make_srd(a.k + b * a.k_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.k_rs + 64) * 2));
  const u32x4 rv = make_srd(a.v + b * a.v_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.v_rs + 64) * 2));
  const uint32_t sbase = lds_addr_of(smem);
  const LaneAddr la = make_lane_addr(lane);
  const StageOff sk = make_stage_off(tid, a.k_rs), sv = make_stage_off(tid, a.v_rs);

  const int q0 = qt * 128 + wave * 32;
  const bool wave_live = q0 < a.Nq;    // wave-uniform (N = 6189: the last of 49 query tiles has 45 valid rows -> two idle waves, -1 % of the launch)
  const int qrow = min(q0 + qi, a.Nq - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)qrow * a.q_rs + 16 * ks + 8 * hh);

  int kend = a.Nk;
  if (CAUSAL) kend = min(a.Nk, qt * 128 + 127 + off + 1);
  const int nt = kend > 0 ? (kend + 63) / 64 : 0;

  // o2 is a third dv block whose V^T operand is the constant [1,0,...,0]^T: row 0 of O2^T accumulates the row sums
  // l = sum_k bf16(p) on the MFMA pipe (which has slack at head_dim 64) instead of 32 VALU adds per tile
  f32x16 o0 = zero16(), o1 = zero16(), o2 = zero16();
  float m = -INFINITY;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)((lane & 31) == 0 ? 1.0f : 0.0f);
  const float c = a.scale * LOG2E;
  const int qabs = q0 + qi;

  auto stage = [&](int buf, int kt) {
    stage64(rk, sbase + buf * 16384, sk, kt * 64, a.k_rs, wave);
    stage64(rv, sbase + buf * 16384 + 8192, sv, kt * 64, a.v_rs, wave);
  };
  auto tile = [&](auto bufc, int kt) {
    constexpr int BUFI = decltype(bufc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nt) stage(BUFI ^ 1, kt + 1);
    if (!wave_live) return;     // all 32 queries of this wave lie past Nq (ragged last tile): it only stages and keeps the barriers
    const char* kl = smem + BUFI * 16384;
    const char* vl = kl + 8192;
    const int k0 = kt * 64;
    f32x16 s0 = zero16(), s1 = zero16();
    __builtin_amdgcn_s_setprio(PRIO_MFMA);     // waves with matrix work ready win the issue slot over waves in their softmax
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#ifdef ATT_TIMING_HALF_LDS   // timing-only: half of the K / V fragment reads (wrong results) -- how LDS-read-bound is the forward?
      const bf16x8 kfr = frag_row(kl, la, 0, ks);
      s0 = mfma32(kfr, qf[ks], s0);
      s1 = mfma32(kfr, qf[ks], s1);
#else
      s0 = mfma32(frag_row(kl, la, 0, ks), qf[ks], s0);
      s1 = mfma32(frag_row(kl, la, 32, ks), qf[ks], s1);
#endif
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- online softmax, query on the lane, this lane holds 2 x 16 of the tile's 64 keys
    const bool need_mask = (k0 + 64 > a.Nk) || (CAUSAL && (k0 + 63 > q0 + off));
    if (need_mask) {
      const int lim = CAUSAL ? min(a.Nk - 1, qabs + off) : a.Nk - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + acc_row(r, hh);
        if (key > lim) s0[r] = -INFINITY;
        if (key + 32 > lim) s1[r] = -INFINITY;
      }
    }
    // row max: 16 x v_max3_f32 (plain fmaxf makes hipcc canonicalise every MFMA output with an extra v_max), then
    // one v_permlane32_swap to combine the two lane halves that share a query
    float mx = max3f(s0[0], s1[0], -INFINITY);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = max3f(mx, s0[r], s1[r]);
    mx = max3f(mx, swap32(mx), -INFINITY);
    const float m_new = fmaxf(m, mx * c);          // scores scaled into the log2 domain (c > 0)
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    if (__any(m_new != m)) {                       // wave-uniform: most tiles leave every running max untouched
      const float alpha = __builtin_amdgcn_exp2f(m - m_use);
      o2[0] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      m = m_new;
    }
#if !FWD_ONES
    float ps = 0.f;
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], c, -m_use));
      s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], c, -m_use));
#if !FWD_ONES
      ps += s0[r] + s1[r];
#endif
    }
#if !FWD_ONES
    o2[0] += ps;   // per-lane partial row sum (both lane halves are added at the end)
#endif
    if constexpr (DROP) {      // the row sum above is of the UNdropped probabilities (softmax, then dropout); P.V uses the dropped ones
      const uint32_t xb = ((uint32_t)bh * (uint32_t)a.Nq + (uint32_t)qabs) * (uint32_t)a.Nk + (uint32_t)(k0 + 4 * hh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t x = xb + (uint32_t)((r & 3) + 8 * (r >> 2));
        s0[r] = attn_drop_hash(x, a.drop_key) >= a.drop_thr ? s0[r] * a.drop_scale : 0.f;
        s1[r] = attn_drop_hash(x + 32u, a.drop_key) >= a.drop_thr ? s1[r] * a.drop_scale : 0.f;
      }
    }
    // ---- O^T += V^T . P^T
    __builtin_amdgcn_s_setprio(PRIO_MFMA);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 p0 = acc_frag(s0, s), p1 = acc_frag(s1, s);
#ifdef ATT_TIMING_HALF_LDS
      const bf16x8 v0f = frag_tr(vl, la, 16 * s, 0), v1f = frag_tr(vl, la, 16 * s, 1);
      o0 = mfma32(v0f, p0, o0);
      o1 = mfma32(v1f, p0, o1);
      o0 = mfma32(v0f, p1, o0);
      o1 = mfma32(v1f, p1, o1);
#else
      o0 = mfma32(frag_tr(vl, la, 16 * s, 0), p0, o0);
      o1 = mfma32(frag_tr(vl, la, 16 * s, 1), p0, o1);
      o0 = mfma32(frag_tr(vl, la, 32 + 16 * s, 0), p1, o0);
      o1 = mfma32(frag_tr(vl, la, 32 + 16 * s, 1), p1, o1);
#endif
#if FWD_ONES
      o2 = mfma32(ones, p0, o2);
      o2 = mfma32(ones, p1, o2);
#endif
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if (nt > 0) stage(0, 0);
  for (int kt = 0; kt < nt; kt += 2) {
    tile(ic<0>{}, kt);
    if (kt + 1 < nt) tile(ic<1>{}, kt + 1);
  }
  float l = o2[0];            // FWD_ONES: row 0 of O2^T lives in register 0 of the hh == 0 lanes
#if FWD_ONES
  if (hh) l = 0.f;
#endif
  l += swap32(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (qabs < a.Nq) {
    u16* op = a.out + b * a.o_bs + (int64_t)qabs * a.o_rs + h * 64;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dv = 8 * g4 + 4 * hh;
      *reinterpret_cast<uint2*>(op + dv) = uint2{pack_bf2(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv), pack_bf2(o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv)};
      *reinterpret_cast<uint2*>(op + 32 + dv) = uint2{pack_bf2(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv), pack_bf2(o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv)};
    }
    if (hh == 0) a.lse[((int64_t)b * a.H + h) * a.Nq + qabs] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * LN2 : -INFINITY;
  }
}

// ======================================================================================= forward, q prescaled (base-2 logits)
// The query operand already carries softmax_scale * log2(e) (the q part of the q|k|v projection leaves the GEMM that way:
// crl_gemm_bf16 colscale), so S^T = K.Q^T comes out of the MFMAs in base-2 logits.  attn_fwd_kernel above spends, per 32-query x
// 64-key block, 768 issue cycles for 512 MFMA cycles (it is VALU-issue bound): 32 v_exp (256), 32 fma for s c - m (128), 32 adds for
// the row sum (128), 16 v_max3 (64), 16 conversions (64) + the MFMAs' own 128.  Here
//   * the running reference -m is the INITIAL ACCUMULATOR of the S chains (a 16-register tuple holding -m in every register: the query
//     sits on the lane, so one value serves all 16), i.e. the subtraction costs nothing;
//   * the reference is moved lazily (guide T13): a tile takes the fast path -- no row maximum at all -- and checks afterwards that its
//     row sums stayed below 2^30 (probabilities up to 2^30 instead of <= 1 are harmless in fp32 / bf16: same relative precision);
//     the first tile, and any tile that fails the check (Inf / NaN included), recomputes its scores from a zero accumulator, takes the
//     true row maximum and re-centres O, l and the seed exactly like the kernel above;
// which leaves 32 v_exp + 32 adds + 16 conversions per block on the fast path.  Same results up to the rounding of p (the reference
// only shifts every probability of a row and its sum by the same power of two).
// the body for the 128 queries qt * 128 .. of (batch, head) bh; `smem` = 32 KiB of LDS nobody else touches (entered and left by all four waves)
template <bool CAUSAL, bool DROP>
__device__ __forceinline__ void fwd_pre_block(const AttnArgs& a, char* smem, int bh, int qt) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, hh = lane >> 5;
  const int b = bh / a.H, h = bh % a.H;
  const int off = a.Nk - a.Nq;

  const u16* qp = a.q + b * a.q_bs + h * 64;
  const u32x4 rk = make_srd(a.k + b * a.k_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.k_rs + 64) * 2));
  const u32x4 rv = make_srd(a.v + b * a.v_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.v_rs + 64) * 2));
  const uint32_t sbase = lds_addr_of(smem);
  const LaneAddr la = make_lane_addr(lane);
  const StageOff sk = make_stage_off(tid, a.k_rs), sv = make_stage_off(tid, a.v_rs);

  const int q0 = qt * 128 + wave * 32;
  const bool wave_live = q0 < a.Nq;
  const int qrow = min(q0 + qi, a.Nq - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)qrow * a.q_rs + 16 * ks + 8 * hh);

  int kend = a.Nk;
  if (CAUSAL) kend = min(a.Nk, qt * 128 + 127 + off + 1);
  const int nt = kend > 0 ? (kend + 63) / 64 : 0;

  f32x16 o0 = zero16(), o1 = zero16(), seed = zero16();     // seed = -m in every register
  float m = -INFINITY, l = 0.f;                             // base-2 reference of this query row, per-lane partial row sum
  const int qabs = q0 + qi;
  constexpr float FAST_LIMIT = 1073741824.f;                // 2^30

  auto stage = [&](int buf, int kt) {
    stage64(rk, sbase + buf * 16384, sk, kt * 64, a.k_rs, wave);
    stage64(rv, sbase + buf * 16384 + 8192, sv, kt * 64, a.v_rs, wave);
  };
  auto tile = [&](auto bufc, int kt) {
    constexpr int BUFI = decltype(bufc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nt) stage(BUFI ^ 1, kt + 1);
    if (!wave_live) return;
    const char* kl = smem + BUFI * 16384;
    const char* vl = kl + 8192;
    const int k0 = kt * 64;
    const bool need_mask = (k0 + 64 > a.Nk) || (CAUSAL && (k0 + 63 > q0 + off));
    const int lim = CAUSAL ? min(a.Nk - 1, qabs + off) : a.Nk - 1;
    auto scores = [&](f32x16& s0, f32x16& s1, const f32x16& init) {
      __builtin_amdgcn_s_setprio(PRIO_MFMA);
      s0 = mfma32(frag_row(kl, la, 0, 0), qf[0], init);
      s1 = mfma32(frag_row(kl, la, 32, 0), qf[0], init);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        s0 = mfma32(frag_row(kl, la, 0, ks), qf[ks], s0);
        s1 = mfma32(frag_row(kl, la, 32, ks), qf[ks], s1);
      }
      __builtin_amdgcn_s_setprio(0);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + acc_row(r, hh);
          if (key > lim) s0[r] = -INFINITY;
          if (key + 32 > lim) s1[r] = -INFINITY;
        }
      }
    };
    f32x16 s0, s1;
    float ps = 0.f;
    bool slow = kt == 0;
    if (!slow) {
      scores(s0, s1, seed);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r]);
        s1[r] = __builtin_amdgcn_exp2f(s1[r]);
        ps += s0[r] + s1[r];
      }
      slow = __any(!(ps < FAST_LIMIT));      // also catches Inf and NaN
    }
    if (slow) {
      scores(s0, s1, zero16());
      float mx = max3f(s0[0], s1[0], -INFINITY);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = max3f(mx, s0[r], s1[r]);
      mx = max3f(mx, swap32(mx), -INFINITY);
      const float m_new = fmaxf(m, mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      if (__any(m_new != m)) {
        const float alpha = __builtin_amdgcn_exp2f(m - m_use);
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; seed[r] = -m_use; }
        m = m_new;
      }
      ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_use);
        s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_use);
        ps += s0[r] + s1[r];
      }
    }
    l += ps;
    if constexpr (DROP) {      // the row sum is of the UNdropped probabilities (softmax, then dropout); P.V uses the dropped ones
      const uint32_t xb = ((uint32_t)bh * (uint32_t)a.Nq + (uint32_t)qabs) * (uint32_t)a.Nk + (uint32_t)(k0 + 4 * hh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t x = xb + (uint32_t)((r & 3) + 8 * (r >> 2));
        s0[r] = attn_drop_hash(x, a.drop_key) >= a.drop_thr ? s0[r] * a.drop_scale : 0.f;
        s1[r] = attn_drop_hash(x + 32u, a.drop_key) >= a.drop_thr ? s1[r] * a.drop_scale : 0.f;
      }
    }
    __builtin_amdgcn_s_setprio(PRIO_MFMA);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 p0 = acc_frag(s0, s), p1 = acc_frag(s1, s);
      o0 = mfma32(frag_tr(vl, la, 16 * s, 0), p0, o0);
      o1 = mfma32(frag_tr(vl, la, 16 * s, 1), p0, o1);
      o0 = mfma32(frag_tr(vl, la, 32 + 16 * s, 0), p1, o0);
      o1 = mfma32(frag_tr(vl, la, 32 + 16 * s, 1), p1, o1);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if (nt > 0) stage(0, 0);
  for (int kt = 0; kt < nt; kt += 2) {
    tile(ic<0>{}, kt);
    if (kt + 1 < nt) tile(ic<1>{}, kt + 1);
  }
  l += swap32(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (qabs < a.Nq) {
    u16* op = a.out + b * a.o_bs + (int64_t)qabs * a.o_rs + h * 64;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int dv = 8 * g4 + 4 * hh;
      *reinterpret_cast<uint2*>(op + dv) = uint2{pack_bf2(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv), pack_bf2(o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv)};
      *reinterpret_cast<uint2*>(op + 32 + dv) = uint2{pack_bf2(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv), pack_bf2(o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv)};
    }
    if (hh == 0) a.lse[((int64_t)b * a.H + h) * a.Nq + qabs] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * LN2 : -INFINITY;
  }
}

template <bool CAUSAL, bool DROP = false>
__global__ __launch_bounds__(256, 3) void attn_fwd_pre_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 8192];
  int bh, qt;
  block_to_bh_tile(blockIdx.x, a.nqt, a.B * a.H, bh, qt);
  fwd_pre_block<CAUSAL, DROP>(a, smem, bh, qt);
}

// ======================================================================================= forward, one wave per SIMD (hand-placed stream)
// 256 queries per workgroup, 64 per wave (two 32-query blocks: every K / V^T fragment read from LDS feeds two MFMAs), the main loop is the
// generated statement attn_fwd4w_body.inc (gen_attn_fwd4w.py: software pipeline over quarter tiles, two v_exp + one conversion per MFMA gap, row
// sums on the matrix pipe, K / V by LDS-DMA two tiles ahead into a ring of four slots).  Non-causal, no dropout, q prescaled, Nk >= 128.
// The softmax reference of a row is the exact maximum of its scores over the FIRST key tile and never moves (probabilities up to 2^127 keep their
// relative precision in bf16 / fp32); a block in which any row ends with a non-finite sum or output is run again by the whole workgroup with the
// moving-maximum body above (fwd_pre_block) -- results then equal attn_fwd_pre_kernel's.
#ifndef F4W_STAMPS
#define F4W_STAMPS 0
#endif
#ifndef F4W_OCC2
#define F4W_OCC2 1      // 0: build without the two-waves-per-SIMD form (its register budget is exact: 128 + 128 under LLVM's even split at 2 waves per SIMD)
#endif
constexpr int F4W_LDS = 4 * 16384;
#if F4W_STAMPS      // diagnostic builds (scripts/ab_f4w.sh): cycles of the stream statement per workgroup (wave 0), s_memtime
__device__ unsigned long long* g_f4w_stamps = nullptr;
#endif
// OCC = 1: one wave per SIMD, 512 registers per wave (attn_fwd4w_body.inc); OCC = 2: the same pipeline in 256 registers (single fragment sets re-read
// behind their last use, MFMA-only operands in the accumulator half) so that TWO workgroups share a CU: a lone wave issues its v_exp, conversions, LDS
// reads and MFMAs strictly one after the other (~1500 issue cycles per key tile against 1152 of matrix-pipe time), two waves per SIMD overlap them.
template <int OCC>
// sched != nullptr: persistent -- the resident workgroups pull their query blocks from the per-XCD ticket lists of gemm_common.h (see attn_bwd_spx_kernel)
__global__ __launch_bounds__(256, OCC) void attn_fwd4w_kernel(const AttnArgs a, int force_fallback, uint32_t* __restrict__ sched, int nitems) {
  extern __shared__ __attribute__((aligned(16))) char smem4[];
  __shared__ int next_item;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, hh = lane >> 5;
  const int nqb = (a.Nq + 255) / 256;
  int my_list = 0;
  int item = blockIdx.x;
  if (sched) {
    if (tid == 0) {
      my_list = gemmc::sched_xcd();
      next_item = gemmc::sched_resolve(sched, my_list, gemmc::sched_pull(sched + my_list), nitems);
    }
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(next_item);
  }
 for (; item >= 0 && item < nitems;) {
#if F4W_STAMPS
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  int bh, qt;
  block_to_bh_tile(item, nqb, a.B * a.H, bh, qt);
  const int b = bh / a.H, h = bh % a.H;

  const u16* qp = a.q + b * a.q_bs + h * 64;
  const u32x4 srdK = make_srd(a.k + b * a.k_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.k_rs + 64) * 2));
  const u32x4 srdV = make_srd(a.v + b * a.v_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.v_rs + 64) * 2));
  const uint32_t sbase = lds_addr_of(smem4);
  const LaneAddr la = make_lane_addr(lane);
  const StageOff sk = make_stage_off(tid, a.k_rs), sv = make_stage_off(tid, a.v_rs);
  const int q0 = qt * 256 + wave * 64;
  const int nt = (a.Nk + 63) / 64;

  // tiles 0 and 1 -> slots 0 and 1
  for (int t = 0; t < 2; ++t) {
    stage64(srdK, sbase + t * 16384, sk, t * 64, a.k_rs, wave);
    stage64(srdV, sbase + t * 16384 + 8192, sv, t * 64, a.v_rs, wave);
  }
  bf16x8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = min(q0 + 32 * qb + qi, a.Nq - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qp + (int64_t)qrow * a.q_rs + 16 * ks + 8 * hh);
  }
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // tile 0 (this wave's pieces; the Q loads are older)
  __syncthreads();
  // exact row maxima over the first key tile (Nk >= 64: no mask)
  float m[2];
  f32x16 seed0, seed1;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    f32x16 s0 = zero16(), s1 = zero16();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s0 = mfma32(frag_row(smem4, la, 0, ks), qf[qb][ks], s0);
      s1 = mfma32(frag_row(smem4, la, 32, ks), qf[qb][ks], s1);
    }
    float mx = max3f(s0[0], s1[0], -INFINITY);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = max3f(mx, s0[r], s1[r]);
    mx = max3f(mx, swap32(mx), -INFINITY);
    m[qb] = mx;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { seed0[r] = -m[0]; seed1[r] = -m[1]; }
  const float negm0 = -m[0], negm1 = -m[1], neginf = -INFINITY;
  bf16x8 ones;
  {
    const int mrow = lane & 15, g = lane >> 4;
    const float one = ((mrow == 1 && (g & 1) == 0) || (mrow == 2 && (g & 1) == 1)) ? 1.0f : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)one;
  }
  f32x16 o00 = zero16(), o01 = zero16(), o10 = zero16(), o11 = zero16();
  f32x4 lsum0 = f32x4{0.f, 0.f, 0.f, 0.f}, lsum1 = f32x4{0.f, 0.f, 0.f, 0.f};      // row sums on the matrix pipe ...
  float ps00 = 0.f, ps01 = 0.f, ps10 = 0.f, ps11 = 0.f;                              // ... or as VALU adds (generator option lsum=valu)
  // a wave whose 64 queries all lie past Nq (N = 6189: three of the four waves of a head's last block) only keeps the workgroup's staging going: its
  // four LDS-DMA pieces per tile and the tile's barrier, exactly as the stream issues them -- no matrix work, no energy
  const bool wave_live = __builtin_amdgcn_readfirstlane(q0 < a.Nq ? 1 : 0) != 0;
  if (!wave_live) {
    for (int t = 0; t + 1 < nt; ++t) {
      const uint32_t slot = sbase + (uint32_t)((t + 2) & 3) * 16384u;
      stage64(srdK, slot, sk, (t + 2) * 64, a.k_rs, wave);
      stage64(srdV, slot + 8192, sv, (t + 2) * 64, a.v_rs, wave);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // tile t + 1 has landed (this wave's pieces)
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    const int limlane = (a.Nk - (nt - 1) * 64) - 4 * hh;      // last tile: key row 32 kh + (r & 3) + 8 (r >> 2) of this lane half is valid iff < limlane
    const uint32_t voffK = sk.v, voffV = sv.v;
    const uint32_t s_ldsw = sbase + (uint32_t)wave * 1024u;
    const uint32_t s_k32 = (uint32_t)(32 * a.k_rs * 2), s_v32 = (uint32_t)(32 * a.v_rs * 2);
    const uint32_t s_kstep = (uint32_t)(64 * a.k_rs * 2), s_vstep = (uint32_t)(64 * a.v_rs * 2);
    uint32_t s_koff = 2u * s_kstep, s_voff = 2u * s_vstep, s_cnt = (uint32_t)(nt - 1), s_t, s_slot;
    uint32_t akr0 = sbase + la.row[0], akr1 = sbase + la.row[1], akr2 = sbase + la.row[2], akr3 = sbase + la.row[3];
    uint32_t avt00 = sbase + la.tr[0][0], avt01 = sbase + la.tr[0][1], avt10 = sbase + la.tr[1][0], avt11 = sbase + la.tr[1][1];
#if F4W_STAMPS
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (OCC == 1) {
      const bf16x8 q00 = qf[0][0], q01 = qf[0][1], q02 = qf[0][2], q03 = qf[0][3], q10 = qf[1][0], q11 = qf[1][1], q12 = qf[1][2], q13 = qf[1][3];
#include "attn_fwd4w_body.inc"
    } else {
#if F4W_OCC2
      // the stream loads the Q fragments itself (into accumulator registers) and builds its seed / ONES tuples from one register each
      const u32x4 srdQ = make_srd(qp, (uint32_t)(((int64_t)(a.Nq - 1) * a.q_rs + 64) * 2));
      const uint32_t voffQ0 = (uint32_t)(((int64_t)min(q0 + qi, a.Nq - 1) * a.q_rs + 8 * hh) * 2);
      const uint32_t voffQ1 = (uint32_t)(((int64_t)min(q0 + 32 + qi, a.Nq - 1) * a.q_rs + 8 * hh) * 2);
#include "attn_fwd2x_body.inc"
#endif
    }
#if F4W_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    if (g_f4w_stamps && tid == 0) {      // [0] stream cycles, [1] / [2] entry of the kernel / end of the stream in 10-ns ticks (s_memrealtime), [3] HW_ID | XCC_ID << 32
      g_f4w_stamps[4 * item] = st1 - st0;
      g_f4w_stamps[4 * item + 1] = rt0;
      g_f4w_stamps[4 * item + 2] = __builtin_amdgcn_s_memrealtime();
      g_f4w_stamps[4 * item + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                         ((rt1 - rt0) << 40);      // bits 40..: ticks from kernel entry to the stream's first instruction (prologue)
    }
#endif
  }
  // ---- epilogue: l of query (lane & 31) sits in lanes 0..15, register 1 (queries 0..15) / 2 (queries 16..31) of the row-sum accumulators
  float chk = 0.f;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#if F4W_LSUM_VALU
    float l = qb ? ps10 + ps11 : ps00 + ps01;
    l += swap32(l);
#elif F4W_LSUM4      // every lane summed its own probabilities (all four registers of the 4x4x4 accumulator hold that sum); the two lane halves share a query
    float l = qb ? lsum1[0] : lsum0[0];
    l += swap32(l);
#else
    const f32x4 ls = qb ? lsum1 : lsum0;
    const float a1 = __shfl(ls[1], lane & 15, 64), a2 = __shfl(ls[2], lane & 15, 64);
    const float l = (lane & 16) ? a2 : a1;
#endif
    const float inv = 1.f / l;
    chk = __builtin_fmaf(l, 0.f, chk);
    const f32x16& oa = qb ? o10 : o00;
    const f32x16& ob = qb ? o11 : o01;
    const int qabs = q0 + 32 * qb + qi;
    float va[16], vb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      va[r] = oa[r] * inv; vb[r] = ob[r] * inv;
      chk = __builtin_fmaf(va[r], 0.f, chk);
      chk = __builtin_fmaf(vb[r], 0.f, chk);
    }
    if (qabs < a.Nq) {
      u16* op = a.out + b * a.o_bs + (int64_t)qabs * a.o_rs + h * 64;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dv = 8 * g4 + 4 * hh;
        *reinterpret_cast<uint2*>(op + dv) = uint2{pack_bf2(va[4 * g4], va[4 * g4 + 1]), pack_bf2(va[4 * g4 + 2], va[4 * g4 + 3])};
        *reinterpret_cast<uint2*>(op + 32 + dv) = uint2{pack_bf2(vb[4 * g4], vb[4 * g4 + 1]), pack_bf2(vb[4 * g4 + 2], vb[4 * g4 + 3])};
      }
      if (hh == 0) a.lse[((int64_t)b * a.H + h) * a.Nq + qabs] = (m[qb] + __builtin_amdgcn_logf(l)) * LN2;
    }
  }
  if (!wave_live) chk = 0.f;
  // any non-finite l or output (NaN * 0 = NaN) -> the workgroup runs its two 128-query halves again with the moving maximum
#ifdef F4W_NO_FALLBACK      // timing-only variants of the stream (scripts/ab_f4w.sh) produce garbage on purpose
  chk = 0.f;
#endif
  if (__syncthreads_or((!(chk == 0.f)) || force_fallback)) {
    fwd_pre_block<false, false>(a, smem4, bh, 2 * qt);
    __syncthreads();
    if ((2 * qt + 1) * 128 < a.Nq) fwd_pre_block<false, false>(a, smem4, bh, 2 * qt + 1);
    __syncthreads();
  }
  if (sched) {
    if (tid == 0) next_item = gemmc::sched_resolve(sched, my_list, gemmc::sched_pull(sched + my_list), nitems);
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(next_item);
    __syncthreads();      // next_item may be rewritten only after every wave has read it
  } else {
    item += gridDim.x;
  }
 }
  if (sched && tid == 0) gemmc::sched_leave(sched, gridDim.x);
}

// ======================================================================================= delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(gid & 7);
  const int64_t r = gid >> 3;  // over B*H*Nq
  const int64_t total = (int64_t)a.B * a.H * a.Nq;
  float s = 0.f;
  if (r < total) {
    const int qn = (int)(r % a.Nq);
    const int bh = (int)(r / a.Nq);
    const int b = bh / a.H, h = bh % a.H;
    const uint4 x = *reinterpret_cast<const uint4*>(a.o + b * a.o_bs + (int64_t)qn * a.o_rs + h * 64 + sub * 8);
    const uint4 y = *reinterpret_cast<const uint4*>(a.d_o + b * a.do_bs + (int64_t)qn * a.do_rs + h * 64 + sub * 8);
    const uint32_t xw[4] = {x.x, x.y, x.z, x.w}, yw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) s += bf2f(xw[j] & 0xffff) * bf2f(yw[j] & 0xffff) + bf2f(xw[j] >> 16) * bf2f(yw[j] >> 16);
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (r < total && sub == 0) {
    a.delta[r] = -s;                              // accumulator init of dP:  dP - delta
    a.delta[total + r] = -a.lse[r] / a.scale;     // accumulator init of S:   (S - lse/scale) * scale*log2e = log2 P
  }
}

// ======================================================================================= dK, dV
// workgroup = 128 keys of one (b,h) (32 per wave, K/V fragments in registers), sweeps query tiles of 64.
// PRE: q carries scale * log2(e) (crl_attn_bwd q_prescaled): the logits come out of the MFMAs in base 2 and the multiply by c = 1 goes
template <bool CAUSAL, bool DROP = false, bool PRE = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const AttnArgs a) {
  constexpr int BUF = 2 * 8192 + 512;  // [Q | dO | lse(64 f32) | delta(64 f32)]
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ki = lane & 31, hh = lane >> 5;
  int bh, ktile;
  block_to_bh_tile(blockIdx.x, a.nkt, a.B * a.H, bh, ktile);
  const int b = bh / a.H, h = bh % a.H;
  const int off = a.Nk - a.Nq;

  const u32x4 rq = make_srd(a.q + b * a.q_bs + h * 64, (uint32_t)(((int64_t)(a.Nq - 1) * a.q_rs + 64) * 2));
  const u32x4 rdo = make_srd(a.d_o + b * a.do_bs + h * 64, (uint32_t)(((int64_t)(a.Nq - 1) * a.do_rs + 64) * 2));
  const int64_t nrows = (int64_t)a.B * a.H * a.Nq;
  const u32x4 rl = make_srd(a.delta + nrows + ((int64_t)b * a.H + h) * a.Nq, (uint32_t)a.Nq * 4u);   // -lse/scale
  const u32x4 rd = make_srd(a.delta + ((int64_t)b * a.H + h) * a.Nq, (uint32_t)a.Nq * 4u);          // -delta
  const uint32_t sbase = lds_addr_of(smem);
  const LaneAddr la = make_lane_addr(lane);
  const StageOff sq = make_stage_off(tid, a.q_rs), sdo = make_stage_off(tid, a.do_rs);

  const int key0 = ktile * 128 + wave * 32;
  const bool wave_live = key0 < a.Nk;
  const int kabs = key0 + ki;
  const int krow = min(kabs, a.Nk - 1);
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(a.k + b * a.k_bs + (int64_t)krow * a.k_rs + h * 64 + 16 * ks + 8 * hh);
    vf[ks] = *reinterpret_cast<const bf16x8*>(a.v + b * a.v_bs + (int64_t)krow * a.v_rs + h * 64 + 16 * ks + 8 * hh);
  }
  // causal: key j is seen by queries i >= j - off
  int qstart_tile = 0;
  if (CAUSAL) qstart_tile = max(0, ktile * 128 - off) / 64;
  const int nqt64 = (a.Nq + 63) / 64;

  f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
  const float c = a.scale * LOG2E;

  auto stage = [&](int buf, int t) {
    const uint32_t base = sbase + buf * BUF;
    stage64(rq, base, sq, t * 64, a.q_rs, wave);
    stage64(rdo, base + 8192, sdo, t * 64, a.do_rs, wave);
    // 64 lse + 64 delta values of the tile by 4-byte LDS-DMA (rows past Nq arrive as 0; their Q and dO rows are 0 too)
    if (wave == 0) dma4(rl, base + 16384, (uint32_t)lane * 4u, (uint32_t)t * 256u);
    else if (wave == 1) dma4(rd, base + 16384 + 256, (uint32_t)lane * 4u, (uint32_t)t * 256u);
  };
  auto tile = [&](auto bufc, int t) {
    constexpr int BUFI = decltype(bufc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nqt64) stage(BUFI ^ 1, t + 1);
    if (!wave_live) return;     // all 32 keys of this wave lie past Nk
    const char* ql = smem + BUFI * BUF;
    const char* dol = ql + 8192;
    const float* lse_s = reinterpret_cast<const float*>(ql + 16384);
    const float* del_s = lse_s + 64;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // accumulators start from the per-query row constants (-lse/scale, -delta) read straight from LDS, so the
      // MFMA chains leave S - lse/scale and dP - delta: no subtract / rescale VALU work per element
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 L = *reinterpret_cast<const float4*>(lse_s + 32 * qb + 8 * g4 + 4 * hh);
        const float4 Dl = *reinterpret_cast<const float4*>(del_s + 32 * qb + 8 * g4 + 4 * hh);
        s[4 * g4] = L.x; s[4 * g4 + 1] = L.y; s[4 * g4 + 2] = L.z; s[4 * g4 + 3] = L.w;
        if constexpr (DROP) { dp[4 * g4] = 0.f; dp[4 * g4 + 1] = 0.f; dp[4 * g4 + 2] = 0.f; dp[4 * g4 + 3] = 0.f; }   // -delta joins after the mask
        else { dp[4 * g4] = Dl.x; dp[4 * g4 + 1] = Dl.y; dp[4 * g4 + 2] = Dl.z; dp[4 * g4 + 3] = Dl.w; }
      }
      __builtin_amdgcn_s_setprio(PRIO_MFMA);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#ifdef ATT_TIMING_HALF_LDS_BWD   // timing-only (wrong results): half of the LDS fragment reads -- how LDS-read-bound is the dK/dV pass?
        const bf16x8 fr = frag_row(ql, la, 32 * qb, ks);
        s = mfma32(fr, kf[ks], s);
        dp = mfma32(fr, vf[ks], dp);
#else
        s = mfma32(frag_row(ql, la, 32 * qb, ks), kf[ks], s);
        dp = mfma32(frag_row(dol, la, 32 * qb, ks), vf[ks], dp);
#endif
      }
      __builtin_amdgcn_s_setprio(0);
      const int qbase = t * 64 + 32 * qb;
      const bool need_mask = CAUSAL && (key0 + 31 > qbase + off);
      if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kabs > qbase + acc_row(r, hh) + off) s[r] = -INFINITY;
      }
      if constexpr (DROP) {
        // dV takes the dropped probabilities, dP = mask o scale o (dO.V^T), dS = P o (dP - delta)
        const uint32_t xb = ((uint32_t)bh * (uint32_t)a.Nq + (uint32_t)(qbase + 4 * hh)) * (uint32_t)a.Nk + (uint32_t)kabs;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r] * c);
          const bool keep = attn_drop_hash(xb + (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)a.Nk, a.drop_key) >= a.drop_thr;
          const float nd = del_s[32 * qb + (r & 3) + 8 * (r >> 2) + 4 * hh];
          s[r] = keep ? p * a.drop_scale : 0.f;
          dp[r] = p * ((keep ? dp[r] * a.drop_scale : 0.f) + nd);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(PRE ? s[r] : s[r] * c);
          s[r] = p;
          dp[r] = p * dp[r];
        }
      }
      __builtin_amdgcn_s_setprio(PRIO_MFMA);
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = acc_frag(s, ss), dsf = acc_frag(dp, ss);
#ifdef ATT_TIMING_HALF_LDS_BWD
        const bf16x8 t0 = frag_tr(dol, la, 32 * qb + 16 * ss, 0), t1 = frag_tr(dol, la, 32 * qb + 16 * ss, 1);
        dv0 = mfma32(t0, pf, dv0);
        dv1 = mfma32(t1, pf, dv1);
        dk0 = mfma32(t0, dsf, dk0);
        dk1 = mfma32(t1, dsf, dk1);
#else
        dv0 = mfma32(frag_tr(dol, la, 32 * qb + 16 * ss, 0), pf, dv0);
        dv1 = mfma32(frag_tr(dol, la, 32 * qb + 16 * ss, 1), pf, dv1);
        dk0 = mfma32(frag_tr(ql, la, 32 * qb + 16 * ss, 0), dsf, dk0);
        dk1 = mfma32(frag_tr(ql, la, 32 * qb + 16 * ss, 1), dsf, dk1);
#endif
      }
      __builtin_amdgcn_s_setprio(0);
    }
  };
  if (qstart_tile < nqt64) stage(0, qstart_tile);
  for (int t = qstart_tile; t < nqt64; t += 2) {
    tile(ic<0>{}, t);
    if (t + 1 < nqt64) tile(ic<1>{}, t + 1);
  }
  if (kabs < a.Nk) {
    u16* dkp = a.dk + b * a.dk_bs + (int64_t)kabs * a.dk_rs + h * 64;
    u16* dvp = a.dv + b * a.dv_bs + (int64_t)kabs * a.dv_rs + h * 64;
    const float sc = a.scale;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = 8 * g4 + 4 * hh;
      *reinterpret_cast<uint2*>(dkp + d) = uint2{pack_bf2(dk0[4 * g4] * sc, dk0[4 * g4 + 1] * sc), pack_bf2(dk0[4 * g4 + 2] * sc, dk0[4 * g4 + 3] * sc)};
      *reinterpret_cast<uint2*>(dkp + 32 + d) = uint2{pack_bf2(dk1[4 * g4] * sc, dk1[4 * g4 + 1] * sc), pack_bf2(dk1[4 * g4 + 2] * sc, dk1[4 * g4 + 3] * sc)};
      *reinterpret_cast<uint2*>(dvp + d) = uint2{pack_bf2(dv0[4 * g4], dv0[4 * g4 + 1]), pack_bf2(dv0[4 * g4 + 2], dv0[4 * g4 + 3])};
      *reinterpret_cast<uint2*>(dvp + 32 + d) = uint2{pack_bf2(dv1[4 * g4], dv1[4 * g4 + 1]), pack_bf2(dv1[4 * g4 + 2], dv1[4 * g4 + 3])};
    }
  }
}

// ======================================================================================= dQ
// workgroup = 128 queries (32 per wave, Q/dO fragments in registers), sweeps key tiles of 64.
template <bool CAUSAL, bool DROP = false, bool PRE = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 8192];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, hh = lane >> 5;
  int bh, qt;
  block_to_bh_tile(blockIdx.x, a.nqt, a.B * a.H, bh, qt);
  const int b = bh / a.H, h = bh % a.H;
  const int off = a.Nk - a.Nq;

  const u32x4 rk = make_srd(a.k + b * a.k_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.k_rs + 64) * 2));
  const u32x4 rv = make_srd(a.v + b * a.v_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.v_rs + 64) * 2));
  const uint32_t sbase = lds_addr_of(smem);
  const LaneAddr la = make_lane_addr(lane);
  const StageOff sk = make_stage_off(tid, a.k_rs), sv = make_stage_off(tid, a.v_rs);

  const int q0 = qt * 128 + wave * 32;
  const bool wave_live = q0 < a.Nq;
  const int qabs = q0 + qi;
  const int qrow = min(qabs, a.Nq - 1);
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(a.q + b * a.q_bs + (int64_t)qrow * a.q_rs + h * 64 + 16 * ks + 8 * hh);
    dof[ks] = *reinterpret_cast<const bf16x8*>(a.d_o + b * a.do_bs + (int64_t)qrow * a.do_rs + h * 64 + 16 * ks + 8 * hh);
  }
  // per-lane row constants as persistent accumulator seeds (query on the lane)
  const int64_t nrows = (int64_t)a.B * a.H * a.Nq;
  const int64_t drow = ((int64_t)b * a.H + h) * a.Nq + qrow;
  float nL, nD;
  if (a.fused_delta) {
    // delta = sum_d dO[q][d] O[q][d]: the dO row is in registers already, the O row is one more 64-byte read per lane; the two lanes of
    // a query hold 32 channels each.  Written out for the dK/dV pass, which runs after this one (replaces attn_delta_kernel).
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.o + b * a.o_bs + (int64_t)qrow * a.o_rs + h * 64 + 16 * ks + 8 * hh);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl = __builtin_fmaf((float)dof[ks][j], (float)of[j], dl);
    }
    dl += swap32(dl);
    nD = -dl;
    nL = -a.lse[drow] / a.scale;
    if (hh == 0 && qabs < a.Nq) { a.delta[drow] = nD; a.delta[nrows + drow] = nL; }
  } else {
    nL = a.delta[nrows + drow];   // -lse/scale
    nD = a.delta[drow];           // -delta
  }
  f32x16 seedS, seedD;
#pragma unroll
  for (int r = 0; r < 16; ++r) { seedS[r] = nL; seedD[r] = DROP ? 0.f : nD; }     // with dropout -delta joins after the mask

  int kend = a.Nk;
  if (CAUSAL) kend = min(a.Nk, qt * 128 + 127 + off + 1);
  const int nt = kend > 0 ? (kend + 63) / 64 : 0;
  f32x16 dq0 = zero16(), dq1 = zero16();
  const float c = a.scale * LOG2E;

  auto stage = [&](int buf, int kt) {
    stage64(rk, sbase + buf * 16384, sk, kt * 64, a.k_rs, wave);
    stage64(rv, sbase + buf * 16384 + 8192, sv, kt * 64, a.v_rs, wave);
  };
  auto tile = [&](auto bufc, int kt) {
    constexpr int BUFI = decltype(bufc)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nt) stage(BUFI ^ 1, kt + 1);
    if (!wave_live) return;     // all 32 queries of this wave lie past Nq
    const char* kl = smem + BUFI * 16384;
    const char* vl = kl + 8192;
    const int k0 = kt * 64;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = seedS, dp = seedD;
      __builtin_amdgcn_s_setprio(PRIO_MFMA);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = mfma32(frag_row(kl, la, 32 * kb, ks), qf[ks], s);
        dp = mfma32(frag_row(vl, la, 32 * kb, ks), dof[ks], dp);
      }
      __builtin_amdgcn_s_setprio(0);
      const bool need_mask = (k0 + 32 * kb + 32 > a.Nk) || (CAUSAL && (k0 + 32 * kb + 31 > q0 + off));
      const int lim = CAUSAL ? min(a.Nk - 1, qabs + off) : a.Nk - 1;
      if (need_mask) {   // wave-uniform branch: only ragged / diagonal tiles pay for the compares
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + 32 * kb + acc_row(r, hh) > lim) s[r] = -INFINITY;   // also keeps exp(-lse) of zero-filled keys out
      }
      if constexpr (DROP) {
        const uint32_t xb = ((uint32_t)bh * (uint32_t)a.Nq + (uint32_t)qabs) * (uint32_t)a.Nk + (uint32_t)(k0 + 32 * kb + 4 * hh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool keep = attn_drop_hash(xb + (uint32_t)((r & 3) + 8 * (r >> 2)), a.drop_key) >= a.drop_thr;
          dp[r] = __builtin_amdgcn_exp2f(s[r] * c) * ((keep ? dp[r] * a.drop_scale : 0.f) + nD);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[r] = __builtin_amdgcn_exp2f(PRE ? s[r] : s[r] * c) * dp[r];
      }
      __builtin_amdgcn_s_setprio(PRIO_MFMA);
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 dsf = acc_frag(dp, ss);
        dq0 = mfma32(frag_tr(kl, la, 32 * kb + 16 * ss, 0), dsf, dq0);
        dq1 = mfma32(frag_tr(kl, la, 32 * kb + 16 * ss, 1), dsf, dq1);
      }
      __builtin_amdgcn_s_setprio(0);
    }
  };
  if (nt > 0) stage(0, 0);
  for (int kt = 0; kt < nt; kt += 2) {
    tile(ic<0>{}, kt);
    if (kt + 1 < nt) tile(ic<1>{}, kt + 1);
  }
  if (qabs < a.Nq) {
    u16* dqp = a.dq + b * a.dq_bs + (int64_t)qabs * a.dq_rs + h * 64;
    const float sc = a.dq_mul;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = 8 * g4 + 4 * hh;
      *reinterpret_cast<uint2*>(dqp + d) = uint2{pack_bf2(dq0[4 * g4] * sc, dq0[4 * g4 + 1] * sc), pack_bf2(dq0[4 * g4 + 2] * sc, dq0[4 * g4 + 3] * sc)};
      *reinterpret_cast<uint2*>(dqp + 32 + d) = uint2{pack_bf2(dq1[4 * g4] * sc, dq1[4 * g4 + 1] * sc), pack_bf2(dq1[4 * g4 + 2] * sc, dq1[4 * g4 + 3] * sc)};
    }
  }
}

// ======================================================================================= single-pass backward
// dK, dV AND dQ from ONE recomputation of S and dP: 5 MFMA products per (query, key) tile instead of the 7 of the two-pass form above
// (guide Appendix B "Attention backward").  The price is a sum of dQ over the workgroups that share a (b, h): it is paid with plain
// stores + a deterministic reduce pass (no float atomics: at N = 6189 they would run at the chip's 1.3 TB/s atomic rate, and the sum
// would depend on arrival order).
//
// Workgroup = 4 waves = 256 keys of one (b, h), ONE workgroup per CU: every wave is alone on its SIMD and owns the whole 512-entry
// register file.  Wave w owns keys 64 w .. 64 w + 63: dK^T and dV^T of its keys (128 accumulator registers), the K and V ROW fragments of
// its keys (64 registers: the B operands of S and dP never touch LDS) and the K^T fragments of the workgroup's 256 keys for its dQ block
// (64 registers).  The workgroup sweeps the query tiles of 64 rows; per tile and wave:
//   S = Q.K^T, dP = dO.V^T (key on the lane, seeded with -lse/scale and -delta)  ->  P = exp2(c S'), dS = P o dP'   [32 MFMAs; every Q / dO
//                                                                                  row fragment feeds the MFMAs of BOTH key blocks]
//   dV^T += dO^T.P, dK^T += Q^T.dS   (accumulators as the next operand; every transposed Q^T / dO^T fragment feeds both key blocks) [32]
//   dS (bf16) -> LDS as a [key][query] tile, one per wave, double-buffered;  ONE barrier per tile
//   dQ^T block [32 d x 32 q] of the PREVIOUS tile += K^T.dS^T over the workgroup's 256 keys (dS^T by transposed reads)            [16]
// so only dS crosses LDS.  The partial dQ of key block kb goes to the bf16 [B, Nq, H*64] slab number kb; attn_dq_reduce_kernel adds the
// ceil(Nk / 256) slabs in slab order in fp32, applies the softmax scale and rounds once more to the bf16 dQ the following GEMMs read.
// (A partial is rounded to bf16 before the sum: with 25 slabs of comparable magnitude that adds about one more bf16 rounding to dQ --
// 3.0e-3 instead of 2.4e-3 relative L2 against fp32 -- well inside the tolerance of every parity test.)
// LDS: ring of 2 x [Q tile | dO tile | -lse/scale, -delta of the tile] | 2 x dS [4 waves][64 keys][64 q]  (K is staged once through the
// first dS buffer for the K^T fragments).
// vmcnt discipline (loads and stores count together, in issue order): per tile a wave issues its five LDS-DMA loads FIRST (right after the
// barrier that frees the ring slot) and then exactly four slab stores; the wait at the top of the next tile is vmcnt(4): the loads have
// landed, the stores may still be in flight.
constexpr int SP_SLOT = 2 * 8192 + 512;
constexpr int SP_DS = 2 * SP_SLOT;
constexpr int SP_DUMMY = SP_DS + 2 * 32768;
constexpr int SP_LDS = SP_DUMMY + 256;

template <bool PRE>
__global__ __launch_bounds__(256, 1) void attn_bwd_sp_kernel(const AttnArgs a, u16* __restrict__ slabs, int64_t slab_stride, int chain, int nfull) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ki = lane & 31, hh = lane >> 5;
  // a chain of `chain` consecutive key blocks per workgroup, full chains first (see attn_bwd_spx_kernel)
  int bh, cidx;
  const int nfull_wg = nfull * a.B * a.H;
  if ((int)blockIdx.x < nfull_wg) block_to_bh_tile(blockIdx.x, nfull, a.B * a.H, bh, cidx);
  else { block_to_bh_tile(blockIdx.x - nfull_wg, 1, a.B * a.H, bh, cidx); cidx = nfull; }
  const int kb_first = cidx * chain, kb_end = min(a.nkt, kb_first + chain);
  const int b = bh / a.H, h = bh % a.H;

  const u32x4 rq = make_srd(a.q + b * a.q_bs + h * 64, (uint32_t)(((int64_t)(a.Nq - 1) * a.q_rs + 64) * 2));
  const u32x4 rdo = make_srd(a.d_o + b * a.do_bs + h * 64, (uint32_t)(((int64_t)(a.Nq - 1) * a.do_rs + 64) * 2));
  const u32x4 rk = make_srd(a.k + b * a.k_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.k_rs + 64) * 2));
  const int64_t nrows = (int64_t)a.B * a.H * a.Nq;
  const u32x4 rl = make_srd(a.delta + nrows + ((int64_t)b * a.H + h) * a.Nq, (uint32_t)a.Nq * 4u);   // -lse/scale
  const u32x4 rd = make_srd(a.delta + ((int64_t)b * a.H + h) * a.Nq, (uint32_t)a.Nq * 4u);          // -delta
  const uint32_t sbase = lds_addr_of(smem);
  const LaneAddr la = make_lane_addr(lane);
  const StageOff sq = make_stage_off(tid, a.q_rs), sdo = make_stage_off(tid, a.do_rs), sk = make_stage_off(tid, a.k_rs);

  for (int kblk = kb_first; kblk < kb_end; ++kblk) {
  if (kblk != kb_first) __syncthreads();           // every wave is done with the previous key block's LDS
  const int key_wg = kblk * 256;
  const int key0 = key_wg + wave * 64;            // this wave's 64 keys
  const bool wave_live = key0 < a.Nk;
  const bool kmask = key0 + 64 > a.Nk;            // some of them lie past Nk
  // K of the whole workgroup -> LDS (rows past Nk arrive as zeros: their dS contributes nothing to dQ)
#pragma unroll
  for (int j = 0; j < 4; ++j) stage64(rk, sbase + SP_DS + j * 8192, sk, key_wg + 64 * j, a.k_rs, wave);
  bf16x8 kf[2][4], vf[2][4];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int krow = min(key0 + 32 * kb + ki, a.Nk - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[kb][ks] = *reinterpret_cast<const bf16x8*>(a.k + b * a.k_bs + (int64_t)krow * a.k_rs + h * 64 + 16 * ks + 8 * hh);
      vf[kb][ks] = *reinterpret_cast<const bf16x8*>(a.v + b * a.v_bs + (int64_t)krow * a.v_rs + h * 64 + 16 * ks + 8 * hh);
    }
  }
  // a key past Nk is switched off through the seed of S (exp2(-inf) = 0: P = dS = 0 whatever the clamped K / V rows hold)
  const float kneg0 = (kmask && key0 + ki >= a.Nk) ? -INFINITY : 0.f, kneg1 = (kmask && key0 + 32 + ki >= a.Nk) ? -INFINITY : 0.f;
  f32x16 dv[2][2], dk[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int db = 0; db < 2; ++db) { dv[kb][db] = zero16(); dk[kb][db] = zero16(); }
  const float c = a.scale * LOG2E;
  const int nqt64 = (a.Nq + 63) / 64;

  // five DMA instructions per wave and tile: two pieces of the Q tile, two of the dO tile, one row-constant vector (waves 2 / 3: a dummy)
  auto stage = [&](int slot, int t) {
    const uint32_t base = sbase + slot * SP_SLOT;
    stage64(rq, base, sq, t * 64, a.q_rs, wave);
    stage64(rdo, base + 8192, sdo, t * 64, a.do_rs, wave);
    const uint32_t dst = wave == 0 ? base + 16384 : wave == 1 ? base + 16384 + 256 : sbase + SP_DUMMY;
    dma4(wave == 1 ? rd : rl, dst, (uint32_t)lane * 4u, (uint32_t)t * 256u);
  };
  const int qh = wave >> 1, dbq = wave & 1;       // dQ phase: this wave's block = d rows 32 dbq.., q columns 32 qh..
  // the transposed-read addresses of that block, selected ONCE (indexing la.tr with a wave-dependent index inside the loop would put the
  // array into scratch memory)
  const uint32_t trk0 = dbq ? la.tr[1][0] : la.tr[0][0], trk1 = dbq ? la.tr[1][1] : la.tr[0][1];   // K^T operand (d block dbq)
  const uint32_t trs0 = qh ? la.tr[1][0] : la.tr[0][0], trs1 = qh ? la.tr[1][1] : la.tr[0][1];     // dS^T operand (q block qh)
  auto tr_read = [&](const char* tile, uint32_t o0, uint32_t o1, int kbase) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + o0 + kbase * 128));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + o1 + kbase * 128));
    typedef short short8v __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  // running partial dQ of this chain: rows q of batch b in slab cidx; bounds-checked stores drop the rows past Nq.  The first key block
  // of a chain starts from zero (a descriptor without records), the others from what the block before them stored.
  const __amdgpu_buffer_rsrc_t rslab = __builtin_amdgcn_make_buffer_rsrc(slabs + (int64_t)cidx * slab_stride + (int64_t)b * a.Nq * (a.H * 64), 0,
                                                                        (int)((int64_t)a.Nq * (a.H * 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rprev = __builtin_amdgcn_make_buffer_rsrc(slabs + (int64_t)cidx * slab_stride + (int64_t)b * a.Nq * (a.H * 64), 0,
                                                                        kblk == kb_first ? 0 : (int)((int64_t)a.Nq * (a.H * 64) * 2), 0x00020000);
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 ka[16];                                   // K^T [32 d of block dbq][256 keys]: the A operands of this wave's dQ block
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) ka[kk] = tr_read(smem + SP_DS + (kk >> 2) * 8192, trk0, trk1, 16 * (kk & 3));

  auto phase_a = [&](auto slotc, int t) {
    constexpr int SLOT = decltype(slotc)::value;
    const char* ql = smem + SLOT * SP_SLOT;
    const char* dol = ql + 8192;
    const float* lse_s = reinterpret_cast<const float*>(ql + 16384);
    const float* del_s = lse_s + 64;
    char* my_ds = smem + SP_DS + SLOT * 32768 + wave * 8192;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s[2], dp[2];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 L = *reinterpret_cast<const float4*>(lse_s + 32 * qb + 8 * g4 + 4 * hh);
        const float4 Dl = *reinterpret_cast<const float4*>(del_s + 32 * qb + 8 * g4 + 4 * hh);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          s[kb][4 * g4] = L.x; s[kb][4 * g4 + 1] = L.y; s[kb][4 * g4 + 2] = L.z; s[kb][4 * g4 + 3] = L.w;
          dp[kb][4 * g4] = Dl.x; dp[kb][4 * g4 + 1] = Dl.y; dp[kb][4 * g4 + 2] = Dl.z; dp[kb][4 * g4 + 3] = Dl.w;
        }
      }
      if (kmask) {   // wave-uniform: only the last key block of a head pays
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] += kneg0; s[1][r] += kneg1; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 fq = frag_row(ql, la, 32 * qb, ks), fdo = frag_row(dol, la, 32 * qb, ks);
        s[0] = mfma32(fq, kf[0][ks], s[0]);
        s[1] = mfma32(fq, kf[1][ks], s[1]);
        dp[0] = mfma32(fdo, vf[0][ks], dp[0]);
        dp[1] = mfma32(fdo, vf[1][ks], dp[1]);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(PRE ? s[kb][r] : s[kb][r] * c);
          s[kb][r] = p;
          dp[kb][r] = p * dp[kb][r];
        }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 tdo0 = frag_tr(dol, la, 32 * qb + 16 * ss, 0), tdo1 = frag_tr(dol, la, 32 * qb + 16 * ss, 1);
        const bf16x8 tq0 = frag_tr(ql, la, 32 * qb + 16 * ss, 0), tq1 = frag_tr(ql, la, 32 * qb + 16 * ss, 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const bf16x8 pf = acc_frag(s[kb], ss), dsf = acc_frag(dp[kb], ss);
          mfma32_acc_a(tdo0, pf, dv[kb][0]);
          mfma32_acc_a(tdo1, pf, dv[kb][1]);
          mfma32_acc_a(tq0, dsf, dk[kb][0]);
          mfma32_acc_a(tq1, dsf, dk[kb][1]);
          // the dS block (32 keys x 32 queries) -> this wave's [key][query] tile: the operand registers ARE the bf16 pairs of query rows
          // 8 (2 ss) + 4 hh + 0..3 and 8 (2 ss + 1) + 4 hh + 0..3: two 8-byte stores per lane
          const uint4 dw = __builtin_bit_cast(uint4, dsf);
          const int row = 32 * kb + ki, c0 = 4 * qb + 2 * ss;
          *reinterpret_cast<uint2*>(my_ds + row * 128 + ((c0 ^ swz64(row)) << 4) + 8 * hh) = uint2{dw.x, dw.y};
          *reinterpret_cast<uint2*>(my_ds + row * 128 + (((c0 + 1) ^ swz64(row)) << 4) + 8 * hh) = uint2{dw.z, dw.w};
        }
      }
    }
  };
  // dQ^T block [32 d x 32 q] of this wave for query tile t over the workgroup's 256 keys (K^T from registers, dS^T by transposed reads),
  // then always four stores per wave (rows past Nq -- and everything when t < 0 -- fall off the descriptor)
  auto phase_b = [&](auto slotc, int t) {
    constexpr int SLOT = decltype(slotc)::value;
    const char* dsr = smem + SP_DS + SLOT * 32768;
    const uint32_t row_off = t >= 0 ? (uint32_t)(t * 64 + 32 * qh + ki) * (uint32_t)(a.H * 64 * 2) + (uint32_t)((h * 64 + 32 * dbq + 4 * hh) * 2) : 0xfffffff0u;
    f32x16 dq;                            // starts from the chain's running partial (bf16 pairs -> fp32: the C operand of the first MFMA)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(rprev, row_off + (t >= 0 ? 16 * g4 : 0), 0, 0);
      dq[4 * g4] = __uint_as_float(w2[0] << 16); dq[4 * g4 + 1] = __uint_as_float(w2[0] & 0xffff0000u);
      dq[4 * g4 + 2] = __uint_as_float(w2[1] << 16); dq[4 * g4 + 3] = __uint_as_float(w2[1] & 0xffff0000u);
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (key_wg + 64 * w < a.Nk) {       // wave-uniform: tiles of waves without keys hold no dS
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dq = mfma32(ka[4 * w + kk], tr_read(dsr + w * 8192, trs0, trs1, 16 * kk), dq);
      }
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      u32x2 w2;
      w2[0] = pack_bf2(dq[4 * g4], dq[4 * g4 + 1]);
      w2[1] = pack_bf2(dq[4 * g4 + 2], dq[4 * g4 + 3]);
      __builtin_amdgcn_raw_buffer_store_b64(w2, rslab, row_off + (t >= 0 ? 16 * g4 : 0), 0, 0);
    }
  };
  auto iter = [&](auto slotc, int t) {
    constexpr int SLOT = decltype(slotc)::value;
    // tile t landed -- the four slab stores of the previous iteration may still be in flight -- and, behind the barrier, every wave
    // is done with tile t - 1 (its ring slot and the dS buffer of tile t - 2 are free, dS of tile t - 1 is complete)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    stage(SLOT ^ 1, min(t + 1, nqt64 - 1));
    if (t < nqt64 && wave_live) phase_a(slotc, t);
    phase_b(ic<(SLOT ^ 1)>{}, t - 1);
  };
  for (int t = 0; t <= nqt64; t += 2) {
    iter(ic<0>{}, t);
    if (t + 1 <= nqt64) iter(ic<1>{}, t + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  const float sc = a.scale;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int kabs = key0 + 32 * kb + ki;
    if (kabs < a.Nk) {
      u16* dkp = a.dk + b * a.dk_bs + (int64_t)kabs * a.dk_rs + h * 64;
      u16* dvp = a.dv + b * a.dv_bs + (int64_t)kabs * a.dv_rs + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d = 32 * db + 8 * g4 + 4 * hh;
          *reinterpret_cast<uint2*>(dkp + d) = uint2{pack_bf2(dk[kb][db][4 * g4] * sc, dk[kb][db][4 * g4 + 1] * sc), pack_bf2(dk[kb][db][4 * g4 + 2] * sc, dk[kb][db][4 * g4 + 3] * sc)};
          *reinterpret_cast<uint2*>(dvp + d) = uint2{pack_bf2(dv[kb][db][4 * g4], dv[kb][db][4 * g4 + 1]), pack_bf2(dv[kb][db][4 * g4 + 2], dv[kb][db][4 * g4 + 3])};
        }
    }
  }
  }   // key blocks of the chain
}

// The same algorithm as ONE hand-placed instruction stream (gen_attn_bwd_sp.py -> attn_bwd_sp_body.inc; register map, schedule and the
// hazard rules are documented there).  This wrapper computes the per-lane LDS / global offsets and the scalar state, stages K and the
// first two query tiles, and hands everything to the generated asm statement, which owns v[0:223] and a[0:255].  q must be prescaled
// (base-2 logits straight from the MFMAs: no multiply per element).
#ifndef SPX_FIRST_VARIANT
#define SPX_FIRST_VARIANT 1      // 0: every block through the general form (a descriptor without records feeds zeros to the first block of a chain: rounds 4 / 5)
#endif
constexpr int SPX_SLOT = 16384 + 1024;
constexpr int SPX_DS = 65536;
constexpr int SPX_PART = SPX_DS + 2 * 32768;       // running partial dQ tiles: ring of 3 buffers x 4 waves x 2 KiB
constexpr int SPX_LDS = SPX_PART + 3 * 8192;

#ifdef SPX_STAMPS
__device__ unsigned long long spx_dbg[2 * 4096];    // diagnostic build: (shader cycles, 100 MHz ticks) of the asm statement per workgroup
constexpr int SPX_TR_WG = 64, SPX_TR_PASSES = 128;  // per-pass trace of the first 64 workgroups: [stamp k][workgroup][pass]; the others share a dump slot
__device__ unsigned long long spx_trace[6 * (SPX_TR_WG + 1) * SPX_TR_PASSES];
#endif
// One workgroup = a CHAIN of `chain` consecutive 256-key blocks of one (batch, head), walked one after the other: the partial dQ of a key
// block is added to what the blocks before it in the chain left in the slab (read back tile by tile through LDS-DMA, as the C operand of
// the tile's first dQ MFMA), so a chain leaves ONE slab and attn_dq_reduce_kernel adds ceil(nkt / chain) of them instead of nkt.
// QUERY SPLIT of the remainder chains (qsplit): when the nkt % chain key blocks every head has left over would occupy only part of the chip for a
// whole block time (cfg-3: 3200 key blocks over 256 CUs = 12.5 -- half the CUs walk 13), each remainder chain is run by TWO workgroups that take half
// of the query tiles each: dQ rows are disjoint (same slab), the second half writes its dK / dV to a compact scratch [B H][rows][64] behind the slabs
// and attn_bwd_addkv_kernel adds it to the first half's (one more bf16 rounding on those key rows).
// PERSISTENT form (sched != nullptr): one workgroup per CU pulls its items -- chains, in the index order above -- from the ticket lists of
// gemm_common.h: list x = the items a static launch would have placed on XCD x (x, x + 8, ...: the heads of an XCD stay on it, its L2 keeps their
// Q / dO), a workgroup pulls from the list of the XCD it runs on and steals from the next ones once that is empty.  The XCDs of one chip differ by
// +-3 % in speed under this kernel (profiles/r5_attn_bwd_timeline.txt): with a fixed eighth of the items each the launch ends with the slowest.
__global__ __launch_bounds__(256, 1) void attn_bwd_spx_kernel(const AttnArgs a, u16* __restrict__ slabs, int64_t slab_stride, int chain, int nfull, int qsplit,
                                                                u16* __restrict__ tmpkv, uint32_t* __restrict__ sched, int nitems) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int next_item;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int my_list = 0;
  int item = blockIdx.x;
  if (sched) {
    if (threadIdx.x == 0) {
      my_list = gemmc::sched_xcd();
      next_item = gemmc::sched_resolve(sched, my_list, gemmc::sched_pull(sched + my_list), nitems);
    }
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(next_item);
  }
 bool sched_started = false;      // a second item of this workgroup: the LDS is still the previous stream's until every wave has left it
 for (; item >= 0 && item < nitems;) {
#if F4W_STAMPS      // diagnostic builds: entry / exit of every item in 10-ns ticks + where it ran (scripts/bench_attn_fwd.py bwd timeline)
  const unsigned long long dbg_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // longest first: the nfull = nkt / chain full chains of every (batch, head), then the remainders (items start in index order, so
  // the short ones fill the last round)
  int bh, cidx, qhalf = -1;
  const int nfull_wg = nfull * a.B * a.H;
  if (item < nfull_wg) block_to_bh_tile(item, nfull, a.B * a.H, bh, cidx);
  else if (!qsplit) { block_to_bh_tile(item - nfull_wg, 1, a.B * a.H, bh, cidx); cidx = nfull; }
  else { block_to_bh_tile(item - nfull_wg, 2, a.B * a.H, bh, qhalf); cidx = nfull; }
  const int b = bh / a.H, h = bh % a.H;
  const int HD2 = a.H * 64 * 2;                    // bytes per slab row
  // the query rows this workgroup walks: all of them, or one half of the 64-query tiles (query split of a remainder chain)
  int q_row0 = 0, nq = a.Nq;
  if (qhalf >= 0) {
    const int h0 = (((a.Nq + 63) / 64 + 1) / 2) * 64;
    if (qhalf == 0) nq = h0; else { q_row0 = h0; nq = a.Nq - h0; }
  }

  const u32x4 rq = make_srd(a.q + b * a.q_bs + (int64_t)q_row0 * a.q_rs + h * 64, (uint32_t)(((int64_t)(nq - 1) * a.q_rs + 64) * 2));
  const u32x4 rdo = make_srd(a.d_o + b * a.do_bs + (int64_t)q_row0 * a.do_rs + h * 64, (uint32_t)(((int64_t)(nq - 1) * a.do_rs + 64) * 2));
  const u32x4 rk = make_srd(a.k + b * a.k_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.k_rs + 64) * 2));
  const u32x4 rv = make_srd(a.v + b * a.v_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.v_rs + 64) * 2));
  // dK / dV: the outputs, or -- second half of a query split -- compact rows [key - key_base][64] of this (batch, head) in the scratch
  const bool to_tmp = qhalf == 1;
  const int key_base = nfull * chain * 256, tmp_rows = a.Nk - key_base;
  const int64_t dk_rs = to_tmp ? 64 : a.dk_rs, dv_rs = to_tmp ? 64 : a.dv_rs;
  const u16* tdk = tmpkv + (int64_t)bh * tmp_rows * 64 - (int64_t)key_base * 64;
  const u32x4 rdk = to_tmp ? make_srd(tdk, (uint32_t)((int64_t)a.Nk * 128)) : make_srd(a.dk + b * a.dk_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.dk_rs + 64) * 2));
  const u32x4 rdv = to_tmp ? make_srd(tdk + (int64_t)a.B * a.H * tmp_rows * 64, (uint32_t)((int64_t)a.Nk * 128))
                           : make_srd(a.dv + b * a.dv_bs + h * 64, (uint32_t)(((int64_t)(a.Nk - 1) * a.dv_rs + 64) * 2));
  const u16* slab = slabs + (int64_t)cidx * slab_stride + ((int64_t)b * a.Nq + q_row0) * (a.H * 64);
  const u32x4 rslab = make_srd(slab, (uint32_t)((int64_t)nq * HD2));
  const int64_t nrows = (int64_t)a.B * a.H * a.Nq;
  // row constants: wave 1 fetches -delta, the others -lse/scale (waves 2 / 3 into spare vectors of the slot: every wave issues the same five pieces)
  const float* rcp = (wave == 1 ? a.delta + ((int64_t)b * a.H + h) * a.Nq : a.delta + nrows + ((int64_t)b * a.H + h) * a.Nq) + q_row0;
  const u32x4 rrc = make_srd(rcp, (uint32_t)nq * 4u);
  const uint32_t sbase = lds_addr_of(smem);
  const int qh = wave >> 1, dbq = wave & 1;        // dQ phase: this wave's block = d rows 32 dbq.., q columns 32 qh..
  const int nqt64 = (nq + 63) / 64;
  const uint32_t s_qstep = (uint32_t)(64 * a.q_rs * 2), s_dostep = (uint32_t)(64 * a.do_rs * 2);
  const uint32_t s_q32 = (uint32_t)(32 * a.q_rs * 2), s_do32 = (uint32_t)(32 * a.do_rs * 2);
  const uint32_t s_slabstep = (uint32_t)(64 * HD2), s_slab3 = 3u * s_slabstep, s_dk32 = (uint32_t)(32 * dk_rs * 2), s_dv32 = (uint32_t)(32 * dv_rs * 2);
  const float s_dkscale = a.scale;
  const uint32_t s_iters = (uint32_t)nqt64;        // + the drain behind the loop: dV / dK of the last block, dQ of the last tile
  const uint32_t s_m0q = sbase + wave * 1024, s_m0rc = sbase + 16384 + wave * 256;     // + ring slot + piece: immediates of the unrolled passes
  const uint32_t s_m0p = sbase + SPX_PART + wave * 2048;                                // + partial buffer + piece
  const uint32_t s_krs2 = (uint32_t)(a.k_rs * 2), s_vrs2 = (uint32_t)(a.v_rs * 2), s_dkrs2 = (uint32_t)(dk_rs * 2), s_dvrs2 = (uint32_t)(dv_rs * 2);

  const int kb_end = min(a.nkt, (cidx + 1) * chain);
  for (int kblk = cidx * chain; kblk < kb_end; ++kblk) {
    // the first key block of a chain starts from zero (its own stream variant; with SPX_FIRST_VARIANT = 0: a descriptor without records returns zeros)
    const u32x4 rprev = make_srd(slab, kblk == cidx * chain ? 0u : (uint32_t)((int64_t)nq * HD2));
    // every per-lane value is derived afresh from an OPAQUE copy of the thread index: hoisted out of the key-block loop it would have to
    // live across the stream, which leaves the compiler 32 vector registers (26 of them its operands) -- i.e. in scratch memory
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, ki = tid & 31, hh = (tid >> 5) & 1;
    const LaneAddr la = make_lane_addr(lane);
    const int key_wg = kblk * 256, key0 = key_wg + wave * 64;
    uint32_t arow0 = sbase + la.row[0], arow1 = sbase + la.row[1], arow2 = sbase + la.row[2], arow3 = sbase + la.row[3];
    uint32_t atr0 = sbase + la.tr[0][0], atr1 = sbase + la.tr[0][1], atr2 = sbase + la.tr[1][0], atr3 = sbase + la.tr[1][1];
    uint32_t aseed = sbase + 16384 + 16 * hh;
    const uint32_t dsw = sbase + SPX_DS + wave * 8192 + ki * 128 + 8 * hh;
    const int sw = swz64(ki);
    uint32_t adsw0 = dsw + ((0 ^ sw) << 4), adsw1 = dsw + ((1 ^ sw) << 4), adsw2 = dsw + ((2 ^ sw) << 4), adsw3 = dsw + ((3 ^ sw) << 4),
             adsw4 = dsw + ((4 ^ sw) << 4), adsw5 = dsw + ((5 ^ sw) << 4), adsw6 = dsw + ((6 ^ sw) << 4), adsw7 = dsw + ((7 ^ sw) << 4);
    uint32_t atrs0 = sbase + SPX_DS + (qh ? la.tr[1][0] : la.tr[0][0]), atrs1 = sbase + SPX_DS + (qh ? la.tr[1][1] : la.tr[0][1]);
    uint32_t atrk0 = sbase + SPX_DS + (dbq ? la.tr[1][0] : la.tr[0][0]), atrk1 = sbase + SPX_DS + (dbq ? la.tr[1][1] : la.tr[0][1]);
    const StageOff sq = make_stage_off(tid, a.q_rs), sdo = make_stage_off(tid, a.do_rs), sk = make_stage_off(tid, a.k_rs);
    uint32_t sqv = sq.v, sdov = sdo.v, rcv = (uint32_t)lane * 4u;
    uint32_t slabv = (uint32_t)(32 * qh + ki) * (uint32_t)HD2 + (uint32_t)((h * 64 + 32 * dbq + 8 * hh) * 2);   // 16 bytes per lane (the stream swaps lane halves)
    // read-back address of the running partial (undoes the lane-half exchange of the store: gen_attn_bwd_sp.py load_part)
    uint32_t apart = sbase + SPX_PART + wave * 2048 + ki * 16 + 8 * hh;
    // (the K / V fragment and dK / dV store offsets -- row key0 + ki, clamped to Nk - 1 for the loads, 16 bytes at channel 8 hh -- are
    // computed inside the stream from rcv and these scalars)
    const uint32_t s_key0 = (uint32_t)key0, s_nkm1 = (uint32_t)(a.Nk - 1);
    // running state: the DMA of pass t fetches tile t + 2; the slab offset advances before the stores of a pass (tile t - 1)
    uint32_t s_qoff = 2u * s_qstep, s_dooff = 2u * s_dostep, s_rcoff = 2u * 256u, s_slaboff = (uint32_t)(-2 * (int)s_slabstep);
    uint32_t s_tmp1, s_cnt;

    // every wave has left the previous key block's stream (its last LDS reads are behind a full wait): the LDS is free
    if (kblk != cidx * chain || sched_started) __syncthreads();
    // K of the whole workgroup -> LDS (rows past Nk arrive as zeros: whatever dS the clamped fragments of such keys produce, it meets a
    // zero K^T row in dQ), query tiles 0 and 1 -> ring slots 0 and 1, the running partials of tiles 0 and 1 -> partial buffers 0 and 1
#pragma unroll
    for (int j = 0; j < 4; ++j) stage64(rk, sbase + SPX_DS + j * 8192, sk, key_wg + 64 * j, a.k_rs, wave);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t base = sbase + t * SPX_SLOT;
      stage64(rq, base, sq, t * 64, a.q_rs, wave);
      stage64(rdo, base + 8192, sdo, t * 64, a.do_rs, wave);
      dma4(rrc, base + 16384 + wave * 256, rcv, (uint32_t)t * 256u);
    }
    const bool first_of_chain = kblk == cidx * chain;      // its running partial is zero: the stream variant without the running-tile machinery
    if (!(SPX_FIRST_VARIANT && first_of_chain)) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dma16(rprev, s_m0p + t * 8192u, slabv, (uint32_t)t * s_slabstep);
        dma16(rprev, s_m0p + t * 8192u + 1024u, slabv, (uint32_t)t * s_slabstep + 32u);
      }
    }
#ifdef SPX_STAMPS
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t s_dbgoff = 0;
    unsigned long long st0, st1, st2, st3, st4, st5;
    const int trw = wave == 0 && blockIdx.x < SPX_TR_WG && nqt64 + 1 < SPX_TR_PASSES ? (int)blockIdx.x : SPX_TR_WG;
    unsigned long long* const dbg0 = spx_trace + (0 * (SPX_TR_WG + 1) + trw) * SPX_TR_PASSES;
    unsigned long long* const dbg1 = spx_trace + (1 * (SPX_TR_WG + 1) + trw) * SPX_TR_PASSES;
    unsigned long long* const dbg2 = spx_trace + (2 * (SPX_TR_WG + 1) + trw) * SPX_TR_PASSES;
    unsigned long long* const dbg3 = spx_trace + (3 * (SPX_TR_WG + 1) + trw) * SPX_TR_PASSES;
    unsigned long long* const dbg4 = spx_trace + (4 * (SPX_TR_WG + 1) + trw) * SPX_TR_PASSES;
    unsigned long long* const dbg5 = spx_trace + (5 * (SPX_TR_WG + 1) + trw) * SPX_TR_PASSES;
#endif
    // ONE statement holding both variants of the stream (general | first key block of a chain, selected by s_first): see gen_attn_bwd_sp.py FIRST
    const uint32_t s_first = (uint32_t)__builtin_amdgcn_readfirstlane((SPX_FIRST_VARIANT && first_of_chain) ? 1 : 0);
#include "attn_bwd_sp_body.inc"
#ifdef SPX_STAMPS
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 4096) { spx_dbg[2 * blockIdx.x] = c1 - c0; spx_dbg[2 * blockIdx.x + 1] = r1 - r0; }
#endif
  }
#if F4W_STAMPS
  if (g_f4w_stamps && threadIdx.x == 0) {
    g_f4w_stamps[4 * item] = (unsigned long long)(kb_end - cidx * chain);      // key blocks of this chain
    g_f4w_stamps[4 * item + 1] = dbg_rt0;
    g_f4w_stamps[4 * item + 2] = __builtin_amdgcn_s_memrealtime();
    g_f4w_stamps[4 * item + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
#endif
  sched_started = true;
  if (sched) {
    __syncthreads();      // every wave has read next_item of the previous round (and left the stream)
    if (threadIdx.x == 0) next_item = gemmc::sched_resolve(sched, my_list, gemmc::sched_pull(sched + my_list), nitems);
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(next_item);
  } else {
    item += gridDim.x;
  }
 }
  if (sched && threadIdx.x == 0) gemmc::sched_leave(sched, gridDim.x);
}

// query split of the remainder chains: dk / dv rows key_base .. Nk - 1 (+)= the second half's compact partials; 16 bytes per thread and array
__global__ __launch_bounds__(256) void attn_bwd_addkv_kernel(const AttnArgs a, const u16* __restrict__ tmpkv, int key_base) {
  const int tmp_rows = a.Nk - key_base;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;                 // over B H x tmp_rows x 8 chunks
  if (i >= (int64_t)a.B * a.H * tmp_rows * 8) return;
  const int c8 = (int)(i & 7);
  const int r = (int)((i >> 3) % tmp_rows);
  const int bh = (int)((i >> 3) / tmp_rows);
  const int b = bh / a.H, h = bh % a.H;
  const u16* t = tmpkv + ((int64_t)bh * tmp_rows + r) * 64 + c8 * 8;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    u16* dst = (which ? a.dv + b * a.dv_bs + (int64_t)(key_base + r) * a.dv_rs : a.dk + b * a.dk_bs + (int64_t)(key_base + r) * a.dk_rs) + h * 64 + c8 * 8;
    const uint4 x = *reinterpret_cast<const uint4*>(dst), y = *reinterpret_cast<const uint4*>(t + (which ? (int64_t)a.B * a.H * tmp_rows * 64 : 0));
    const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf2(bf2f(xs[j] & 0xffff) + bf2f(ys[j] & 0xffff), bf2f(xs[j] >> 16) + bf2f(ys[j] >> 16));
    *reinterpret_cast<uint4*>(dst) = uint4{o[0], o[1], o[2], o[3]};
  }
}

// dq[b, q, h*64 + d] = bf16(scale * sum over slabs (fp32, in slab order) of slab[s][b][q][h*64 + d]); 16 bytes per thread and slab
#ifndef RED_UNROLL
#define RED_UNROLL 8
#endif
__device__ __forceinline__ uint4 nt_load16(const u16* p) {
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const u32x4v v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(p));
  return uint4{v[0], v[1], v[2], v[3]};
}
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(const u16* __restrict__ slabs, int64_t slab_stride, int nslab, u16* __restrict__ dq,
                                                             int64_t dq_bs, int64_t dq_rs, int B, int Nq, int HD, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over B * Nq * HD / 8
  const int per_row = HD / 8;
  if (i >= (int64_t)B * Nq * per_row) return;
  const int c8 = (int)(i % per_row);
  const int64_t row = i / per_row;                                // b * Nq + q
  const u16* src = slabs + row * HD + c8 * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  int sidx = 0;
  // eight independent 16-byte loads in flight per thread; the slabs are read exactly once: non-temporal (round 4: 0.52 -> see profiles/)
  for (; sidx + RED_UNROLL <= nslab; sidx += RED_UNROLL) {
    uint4 v[RED_UNROLL];
#pragma unroll
    for (int u = 0; u < RED_UNROLL; ++u) v[u] = nt_load16(src + (int64_t)(sidx + u) * slab_stride);
#pragma unroll
    for (int u = 0; u < RED_UNROLL; ++u) {
      const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += bf2f(w[j] & 0xffff); acc[2 * j + 1] += bf2f(w[j] >> 16); }
    }
  }
  for (; sidx < nslab; ++sidx) {
    const uint4 v = nt_load16(src + (int64_t)sidx * slab_stride);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[2 * j] += bf2f(w[j] & 0xffff); acc[2 * j + 1] += bf2f(w[j] >> 16); }
  }
  const int bq = (int)(row / Nq), qn = (int)(row % Nq);
  uint4 o;
  o.x = pack_bf2(acc[0] * scale, acc[1] * scale); o.y = pack_bf2(acc[2] * scale, acc[3] * scale);
  o.z = pack_bf2(acc[4] * scale, acc[5] * scale); o.w = pack_bf2(acc[6] * scale, acc[7] * scale);
  *reinterpret_cast<uint4*>(dq + bq * dq_bs + (int64_t)qn * dq_rs + c8 * 8) = o;
}

int set_drop(const char* who, AttnArgs& a, float p, uint64_t seed, uint32_t step, uint32_t site) {
  CRL_CHECK(p >= 0.f && p < 1.f, "%s: attention dropout p = %g outside [0, 1)", who, p);
  a.drop_thr = (uint32_t)(p * 16777216.f + 0.5f);        // 24-bit threshold; 0 = off
  a.drop_scale = 1.f / (1.f - p);
  a.drop_key = attn_drop_key(seed, step, site);
  return 0;
}

int check_common(const char* who, int B, int H, int Nq, int Nk, int64_t rs_min) {
  CRL_CHECK(B > 0 && H > 0 && Nq > 0 && Nk > 0, "%s: empty problem", who);
  CRL_CHECK(rs_min >= (int64_t)H * 64, "%s: head_dim is 64 (head h at channel 64 h): a row must hold H * 64 = %d channels, row stride is %lld",
            who, H * 64, (long long)rs_min);
  return 0;
}
#define CHK_STRIDE(name, p, bs, rs)                                                                         \
  CRL_CHECK(((uintptr_t)(p) % 16) == 0 && ((bs) % 8) == 0 && ((rs) % 8) == 0, "%s: " name " must be 16-byte aligned with strides multiple of 8", who)
#define CHK_EXTENT(name, n, rs) CRL_CHECK((uint64_t)((n) + 64) * (uint64_t)(rs) * 2 < (1ull << 32), "%s: " name " per-batch extent exceeds 4 GiB", who)

}  // namespace

#if F4W_STAMPS
extern "C" int crl_debug_f4w_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_f4w_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
static int g_fwd_persist = 1;      // the forward stream as a persistent launch pulling query blocks from ticket lists (0: one workgroup per block)
extern "C" int crl_attn_fwd_set_persistent(int on) { g_fwd_persist = on != 0; return 0; }
static int g_fwd_mode = 0;      // 0 auto (hand-placed stream where it applies), 1 the 32-queries-per-wave kernels only, 2 stream + forced fallback, 3 / 4 stream with one / two waves per SIMD
extern "C" int crl_attn_fwd_set_mode(int mode) {
  if (mode < 0 || mode > 4) { crl_set_error("crl_attn_fwd_set_mode: 0 auto, 1 compiler-scheduled kernels only, 2 hand-placed stream with its fallback forced (tests), 3 / 4 the stream with one / two waves per SIMD"); return -1; }
  g_fwd_mode = mode;
  return 0;
}

extern "C" int crl_attn_fwd(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                            const void* v, int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs,
                            float* lse, int B, int H, int Nq, int Nk, float scale, int causal, int q_prescaled,
                            float drop_p, uint64_t drop_seed, uint32_t drop_step, uint32_t drop_site, void* stream) {
  const char* who = "crl_attn_fwd";
  {
    int64_t rs_min = q_rs;
    for (int64_t r : {k_rs, v_rs, o_rs}) rs_min = r < rs_min ? r : rs_min;
    if (check_common(who, B, H, Nq, Nk, rs_min)) return -1;
  }
  CRL_CHECK(q && k && v && o && lse, "%s: null pointer", who);
  CHK_STRIDE("q", q, q_bs, q_rs); CHK_STRIDE("k", k, k_bs, k_rs); CHK_STRIDE("v", v, v_bs, v_rs); CHK_STRIDE("o", o, o_bs, o_rs);
  CHK_EXTENT("k", Nk, k_rs); CHK_EXTENT("v", Nk, v_rs);
  AttnArgs a{};
  a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.out = (u16*)o; a.lse = lse;
  a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_rs = o_rs;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.causal = causal; a.scale = scale;
  a.nqt = (Nq + 127) / 128;
  if (set_drop(who, a, drop_p, drop_seed, drop_step, drop_site)) return -1;
  const bool drop = a.drop_thr != 0;
  const unsigned grid = (unsigned)a.nqt * B * H;
  // algorithmic FLOPs: QK^T and PV, 2 x 2 x Nq x Nk x 64 per head (a causal mask halves them when Nq == Nk)
  const double pairs = causal ? (double)Nq * (Nk - Nq) + 0.5 * (double)Nq * (Nq + 1) : (double)Nq * Nk;
  CRL_PROF_START(CRL_K_ATTN_FWD + (causal ? 1 : 0), stream, 4.0 * 64 * pairs * B * H);
  // (the 256-register form loads its Q fragments through a buffer descriptor: 32-bit offsets)
  if (q_prescaled && !drop && !causal && g_fwd_mode != 1 && Nk >= 128 && (uint64_t)(Nq + 64) * (uint64_t)q_rs * 2 < (1ull << 32)) {
    // the hand-placed stream: 256 queries per workgroup, two workgroups per CU (mode 3: the 512-register form, one per CU; mode 2: every block also
    // runs its moving-maximum fallback -- tests)
    for (const void* f : {reinterpret_cast<const void*>(&attn_fwd4w_kernel<1>), reinterpret_cast<const void*>(&attn_fwd4w_kernel<F4W_OCC2 ? 2 : 1>)})
      if (int rc = crl_enable_lds(f, F4W_LDS, who)) return rc;
    const unsigned g4 = (unsigned)((Nq + 255) / 256) * B * H;
    const bool one_per_cu = g_fwd_mode == 3 || !F4W_OCC2;
    // persistent when there are more query blocks than workgroup slots: the resident workgroups pull blocks from the per-XCD ticket lists
    uint32_t* sched = nullptr;
    unsigned launch = g4;
    const unsigned slots = (unsigned)crl_gemm_cus() * (one_per_cu ? 1u : 2u);
    // (worth the ticket round trips from four blocks per slot on: at cfg-2's 1.9 the persistent form measured 18 us per launch slower)
    if (crl_gemm_dynamic() && g_fwd_persist && g4 >= 4 * slots) {
      bool ok;
      sched = crl_sched_slot(as_stream(stream), &ok);
      if (!ok) return -2;
      if (sched) launch = slots;
    }
    if (one_per_cu) attn_fwd4w_kernel<1><<<launch, 256, F4W_LDS, as_stream(stream)>>>(a, g_fwd_mode == 2, sched, (int)g4);
    else attn_fwd4w_kernel<F4W_OCC2 ? 2 : 1><<<launch, 256, F4W_LDS, as_stream(stream)>>>(a, g_fwd_mode == 2, sched, (int)g4);
  } else if (q_prescaled) {       // base-2 logits straight from the MFMAs: the seeded / lazy-maximum kernel (`scale` is not used)
    if (drop) { if (causal) attn_fwd_pre_kernel<true, true><<<grid, 256, 0, as_stream(stream)>>>(a); else attn_fwd_pre_kernel<false, true><<<grid, 256, 0, as_stream(stream)>>>(a); }
    else if (causal) attn_fwd_pre_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(a);
    else attn_fwd_pre_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(a);
  } else if (drop) {
    if (causal) attn_fwd_kernel<true, true><<<grid, 256, 0, as_stream(stream)>>>(a);
    else attn_fwd_kernel<false, true><<<grid, 256, 0, as_stream(stream)>>>(a);
  } else if (causal) attn_fwd_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(a);
  else attn_fwd_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(a);
  CRL_PROF_STOP(CRL_K_ATTN_FWD + (causal ? 1 : 0), stream);
  CRL_LAUNCH_CHECK(who);
  return 0;
}

static int g_bwd_parts = 7;
static int g_bwd_mode = 0;      // 0 auto (single pass for long non-causal sequences with a prescaled q), 1 two-pass, 2 single pass whenever legal, 3 its C++ form
extern "C" int crl_attn_bwd_set_mode(int mode) {
  if (mode < 0 || mode > 3) { crl_set_error("crl_attn_bwd_set_mode: 0 auto, 1 two-pass, 2 single pass, 3 single pass (C++ reference form)"); return -1; }
  g_bwd_mode = mode;
  return 0;
}
// auto: the single pass pays from about sixteen query tiles per workgroup on (its prologue / epilogue -- K^T fragments, 128 accumulators
// in and out -- cost as much as ~5 tile passes): the ViT encoders (N = 2401 ... 24 935) and the decoder's cross-attention at cfg-3
// (Nq = 1023: 0.63 against 0.67 ms per layer), not short target sequences
static int g_bwd_chain = 0;     // key blocks per workgroup of the hand-placed single pass: 0 = auto
extern "C" int crl_attn_bwd_set_chain(int chain) {
  if (chain < 0) { crl_set_error("crl_attn_bwd_set_chain: 0 auto, n >= 1 key blocks per workgroup"); return -1; }
  g_bwd_chain = chain;
  return 0;
}
int crl_gemm_cus();              // gemm.hip: CUs not set aside for RCCL (crl_gemm_set_reserved_cus)
// Key blocks per workgroup.  A chain of c blocks leaves one slab instead of c (the reduce reads ceil(nkt / c) slabs: ~0.1 of a key block's
// time each at 128 heads, both proportional to Nq) but its workgroups are c times longer (fewer of them to balance over the CUs) and fewer workgroups
// of a head run side by side (the query tiles they share come from L2 only while they do).
// The workgroups start longest first -- nfull = nkt / c full chains per head, then the remainders --; the makespan of that order on the
// available CUs is simulated once per (nkt, heads, CUs) and the cheapest c kept.  Same-box A/B at cfg-3 (25 key blocks, 128 heads):
// c = 3 / 4 / 6 within noise of each other, -2.5 ms per step against c = 1, c = 12 half of that (profiles/r4_attn_chain.txt).
static int g_bwd_persist = 1;      // the hand-placed single pass as a persistent launch that pulls its chains from ticket lists (0: one workgroup per chain)
extern "C" int crl_attn_bwd_set_persistent(int on) { g_bwd_persist = on != 0; return 0; }
static int g_bwd_qsplit = -1;      // query split of the remainder chains: -1 auto (with the automatic chain only), 0 off, 1 whenever legal
extern "C" int crl_attn_bwd_set_qsplit(int mode) {
  if (mode < -1 || mode > 1) { crl_set_error("crl_attn_bwd_set_qsplit: -1 auto, 0 off, 1 whenever legal"); return -1; }
  g_bwd_qsplit = mode;
  return 0;
}
static int bwd_chain_length(int nkt, int BH, bool stream, int* qsplit = nullptr) {
  if (qsplit) *qsplit = g_bwd_qsplit == 1;
  if (g_bwd_chain > 0) return g_bwd_chain < nkt ? g_bwd_chain : nkt;      // forced: both forms (same arithmetic at the same chain length)
  if (!stream) return 1;
  // memo: the last few (key blocks, heads, CUs) -- encoder and cross-attention shapes, each with and without the CUs a data-parallel run reserves
  // for RCCL while a bucket is in flight (a single entry missed twice per step there and re-simulated ~15 k heap operations)
  struct Memo { int nkt, bh, cu, c, split; };
  static Memo memo[8];
  static int memo_n = 0, memo_next = 0;
  static std::mutex memo_mu;
  std::lock_guard<std::mutex> memo_lock(memo_mu);
  const int ncu = crl_gemm_cus();
  for (int i = 0; i < memo_n; ++i)
    if (memo[i].nkt == nkt && memo[i].bh == BH && memo[i].cu == ncu) {
      if (qsplit && g_bwd_qsplit == -1) *qsplit = memo[i].split;
      return memo[i].c;
    }
  // in units of one key block's time on one CU (~ Nq): the stream slows by ~0.6 % per link (fewer workgroups of a head side by side: c = 12
  // against 6 at cfg-3), a slab costs the reduce (chip-wide, HBM-bound: bytes ~ heads x Nq) 0.1 at 128 heads, and a link is charged 0.03
  // more for the bf16 rounding it adds to the running sum (ties go to the shorter chain)
  auto price = [&](double makespan, int c, int nslab) { return makespan * (1.0 + 0.006 * (c - 1)) + 0.1 * (BH / 128.0) * nslab + 0.03 * (c - 1); };
  double best = 1e30;
  int best_c = 1;
  std::vector<int> busy((size_t)ncu);
  for (int c = 1; c <= nkt && c <= 32; ++c) {
    const int nfull = nkt / c, rem = nkt % c;
    if ((int64_t)nfull * BH > 16 * (int64_t)ncu) {      // many rounds: the last one hardly matters, and the simulation would take long
      const double cost = price((double)nkt * BH / ncu + c, c, nfull + (rem ? 1 : 0));      // + c: one workgroup of slack
      if (cost < best - 1e-9) { best = cost; best_c = c; }
      continue;
    }
    // workgroups start in index order on the CU that frees up first (a min-heap over the CUs' finish times)
    std::fill(busy.begin(), busy.end(), 0);
    auto later = [](int x, int y) { return x > y; };
    int makespan = 0;
    auto place = [&](int n, int len) {
      for (int i = 0; i < n && len > 0; ++i) {
        std::pop_heap(busy.begin(), busy.end(), later);
        busy.back() += len;
        makespan = busy.back() > makespan ? busy.back() : makespan;
        std::push_heap(busy.begin(), busy.end(), later);
      }
    };
    place(nfull * BH, c);
    place(BH, rem);
    const double cost = price(makespan, c, nfull + (rem ? 1 : 0));
    if (cost < best - 1e-9) { best = cost; best_c = c; }
  }
  // query split of the remainder chains (two workgroups of half the query tiles each instead of one): worth it when it shortens the simulated
  // makespan by at least half a key block (the second half's dK / dV cost one more small kernel and a bf16 rounding)
  int best_split = 0;
  {
    const int c = best_c, nfull = nkt / c, rem = nkt % c;
    if (rem > 0 && (int64_t)(nfull + 2) * BH <= 64 * (int64_t)ncu) {
      int span[2];
      for (int sp = 0; sp < 2; ++sp) {          // lengths in half key blocks
        std::fill(busy.begin(), busy.end(), 0);
        auto later = [](int x, int y) { return x > y; };
        int makespan = 0;
        auto place = [&](int n, int len) {
          for (int i = 0; i < n && len > 0; ++i) {
            std::pop_heap(busy.begin(), busy.end(), later);
            busy.back() += len;
            makespan = busy.back() > makespan ? busy.back() : makespan;
            std::push_heap(busy.begin(), busy.end(), later);
          }
        };
        place(nfull * BH, 2 * c);
        place(sp ? 2 * BH : BH, sp ? rem : 2 * rem);
        span[sp] = makespan;
      }
      best_split = span[1] + 1 <= span[0];
    }
  }
  if (qsplit && g_bwd_qsplit == -1) *qsplit = best_split;
  memo[memo_next] = Memo{nkt, BH, ncu, best_c, best_split};
  memo_next = (memo_next + 1) % 8;
  memo_n = memo_n < 8 ? memo_n + 1 : 8;
  return best_c;
}
// whether crl_attn_bwd would split the remainder chains of that problem between two workgroups by query halves (pure host arithmetic)
extern "C" int crl_attn_bwd_qsplit_for(int Nk, int BH) {
  if (Nk <= 0 || BH <= 0) { crl_set_error("crl_attn_bwd_qsplit_for: empty problem"); return -1; }
  int split = 0;
  const int nkt = (Nk + 255) / 256, c = bwd_chain_length(nkt, BH, true, &split);
  return (split && nkt % c != 0) ? 1 : 0;
}
// the chain length crl_attn_bwd would use for Nk keys and B * H heads right now (pure host arithmetic: no GPU needed)
extern "C" int crl_attn_bwd_chain_for(int Nk, int BH) {
  if (Nk <= 0 || BH <= 0) { crl_set_error("crl_attn_bwd_chain_for: empty problem"); return -1; }
  return bwd_chain_length((Nk + 255) / 256, BH, true);
}
static bool bwd_fused_wanted(int Nq, int Nk, int causal) {
  if (causal || g_bwd_parts != 7) return false;
  return g_bwd_mode >= 2 || (g_bwd_mode == 0 && Nq >= 1000 && Nk >= 1024);
}
// bf16 partial-dQ slabs of the fused backward: ceil(Nk / 256) x [B, Nq, H * 64]; 0 = the two-pass form runs (no workspace needed)
extern "C" size_t crl_attn_bwd_ws_bytes(int B, int H, int Nq, int Nk, int causal) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || !bwd_fused_wanted(Nq, Nk, causal)) return 0;
  return (size_t)((Nk + 255) / 256) * (size_t)B * (size_t)Nq * (size_t)H * 64 * sizeof(u16);
}

// measurement hook: which of the three backward launches crl_attn_bwd issues (bit0 delta, bit1 dK/dV, bit2 dQ); default all
extern "C" int crl_attn_bwd_set_parts(int parts) {
  if (parts < 1 || parts > 7) { crl_set_error("crl_attn_bwd_set_parts: bad mask %d", parts); return -1; }
  g_bwd_parts = parts;
  return 0;
}

extern "C" int crl_attn_bwd(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                            const void* v, int64_t v_bs, int64_t v_rs, const void* o, int64_t o_bs, int64_t o_rs,
                            const void* d_o, int64_t do_bs, int64_t do_rs, const float* lse, float* delta,
                            void* dq, int64_t dq_bs, int64_t dq_rs, void* dk, int64_t dk_bs, int64_t dk_rs,
                            void* dv, int64_t dv_bs, int64_t dv_rs, int B, int H, int Nq, int Nk, float scale, int causal, int q_prescaled,
                            float drop_p, uint64_t drop_seed, uint32_t drop_step, uint32_t drop_site,
                            void* ws, size_t ws_bytes, void* stream) {
  const char* who = "crl_attn_bwd";
  {
    int64_t rs_min = q_rs;
    for (int64_t r : {k_rs, v_rs, o_rs, do_rs, dq_rs, dk_rs, dv_rs}) rs_min = r < rs_min ? r : rs_min;
    if (check_common(who, B, H, Nq, Nk, rs_min)) return -1;
  }
  CRL_CHECK(q && k && v && o && d_o && lse && delta && dq && dk && dv, "%s: null pointer", who);
  CHK_STRIDE("q", q, q_bs, q_rs); CHK_STRIDE("k", k, k_bs, k_rs); CHK_STRIDE("v", v, v_bs, v_rs); CHK_STRIDE("o", o, o_bs, o_rs);
  CHK_STRIDE("do", d_o, do_bs, do_rs); CHK_STRIDE("dq", dq, dq_bs, dq_rs); CHK_STRIDE("dk", dk, dk_bs, dk_rs); CHK_STRIDE("dv", dv, dv_bs, dv_rs);
  CHK_EXTENT("k", Nk, k_rs); CHK_EXTENT("v", Nk, v_rs); CHK_EXTENT("q", Nq, q_rs); CHK_EXTENT("do", Nq, do_rs);
  AttnArgs a{};
  a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.o = (const u16*)o; a.d_o = (const u16*)d_o;
  a.dq = (u16*)dq; a.dk = (u16*)dk; a.dv = (u16*)dv; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_rs = o_rs;
  a.do_bs = do_bs; a.do_rs = do_rs; a.dq_bs = dq_bs; a.dq_rs = dq_rs; a.dk_bs = dk_bs; a.dk_rs = dk_rs; a.dv_bs = dv_bs; a.dv_rs = dv_rs;
  // q_prescaled: q holds q * scale * log2(e).  The kernels recompute  S' = q'.k (base-2 logits), so their own scale is ln 2:
  //   c = scale_k log2(e) = 1 (P = exp2(S' - lse log2e)), the S seed -lse / scale_k = -lse log2(e), dK = ln 2 (dS^T q') = scale (dS^T q)
  // (d logit' = ln 2 dS); dQ is stored as the gradient of the UNscaled projection output, scale dS.K, exactly as without prescaling
  // (the GEMM epilogue's factor is part of the layer: d(x W + b) = scale log2(e) dq' = scale dS.K).
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.causal = causal; a.scale = q_prescaled ? LN2 : scale;
  a.dq_mul = scale;
  a.nqt = (Nq + 127) / 128; a.nkt = (Nk + 127) / 128;
  if (set_drop(who, a, drop_p, drop_seed, drop_step, drop_site)) return -1;
  const bool drop = a.drop_thr != 0;
  hipStream_t s = as_stream(stream);
  const int64_t rows = (int64_t)B * H * Nq;
  // normal operation (all parts): the dQ pass runs FIRST and produces the per-query row constants (delta, -lse/scale) as a by-product;
  // the measurement hook's partial runs keep the separate delta launch so that each pass can be timed alone
#ifndef BWD_FUSED_DELTA
#define BWD_FUSED_DELTA 1
#endif
  const bool fused = BWD_FUSED_DELTA && g_bwd_parts == 7;
  a.fused_delta = fused ? 1 : 0;
  if ((g_bwd_parts & 1) && !fused) {
    attn_delta_kernel<<<(unsigned)((rows * 8 + 255) / 256), 256, 0, s>>>(a);
    CRL_LAUNCH_CHECK("crl_attn_bwd(delta)");
  }
  // the single-pass form has no dropout variant; in auto mode it is the hand-placed stream or nothing (that one wants a prescaled q)
  const size_t need = (drop || (g_bwd_mode == 0 && !q_prescaled)) ? 0 : crl_attn_bwd_ws_bytes(B, H, Nq, Nk, causal);
  if (need && ws && ws_bytes >= need) {
    // ---- fused single pass: row constants, then dK / dV / partial dQ slabs from one recomputation, then the slab reduce
    CRL_CHECK(((uintptr_t)ws % 16) == 0 && (dq_rs % 8) == 0, "%s: workspace / dq must be 16-byte aligned", who);
    for (const void* f : {reinterpret_cast<const void*>(&attn_bwd_sp_kernel<false>), reinterpret_cast<const void*>(&attn_bwd_sp_kernel<true>),
                          reinterpret_cast<const void*>(&attn_bwd_spx_kernel)})
      if (int rc = crl_enable_lds(f, f == reinterpret_cast<const void*>(&attn_bwd_spx_kernel) ? SPX_LDS : SP_LDS, who)) return rc;
    a.fused_delta = 0;
    attn_delta_kernel<<<(unsigned)((rows * 8 + 255) / 256), 256, 0, s>>>(a);
    CRL_LAUNCH_CHECK("crl_attn_bwd(delta)");
    a.nkt = (Nk + 255) / 256;
    const bool spx = q_prescaled && g_bwd_mode != 3;
    int want_split = 0;
    const int chain = bwd_chain_length(a.nkt, B * H, spx, &want_split);
    const int nfull = a.nkt / chain, nchain = (a.nkt + chain - 1) / chain;
    const int nslab = nchain;
    const int64_t slab_stride = (int64_t)B * Nq * H * 64;
    // query split of the remainder chains: the hand-placed stream only, at least two query tiles per half, scratch behind the slabs in use
    const int key_base = nfull * chain * 256;
    const int64_t tmp_elems = 2 * (int64_t)B * H * (Nk - key_base) * 64;
    const bool qsplit = spx && want_split && nchain > nfull && (Nq + 63) / 64 >= 4 &&
                        (uint64_t)(nchain * slab_stride + tmp_elems) * 2 <= (uint64_t)ws_bytes && (uint64_t)Nk * 128 < (1ull << 32);
    u16* const tmpkv = (u16*)ws + (int64_t)nchain * slab_stride;
    const unsigned grid_spx = (unsigned)(nfull + (nchain > nfull ? (qsplit ? 2 : 1) : 0)) * B * H;
    const double pairs_f = (double)Nq * Nk;
    CRL_PROF_START(CRL_K_ATTN_BWD_FUSED, stream, 8.0 * 64 * pairs_f * B * H);      // the WHOLE algorithmic backward (dV, dP, dK, dQ)
    if (spx) {
      // persistent from three items per CU on (fewer: nothing to balance, only ticket round trips to pay) when the tile tickets are on
      // (crl_gemm_set_schedule); the slot ring of this stream
      uint32_t* sched = nullptr;
      unsigned launch = grid_spx;
      if (crl_gemm_dynamic() && g_bwd_persist && (int)grid_spx >= 3 * crl_gemm_cus()) {
        bool ok;
        sched = crl_sched_slot(s, &ok);
        if (!ok) return -2;
        if (sched) launch = (unsigned)crl_gemm_cus();
      }
      attn_bwd_spx_kernel<<<launch, 256, SPX_LDS, s>>>(a, (u16*)ws, slab_stride, chain, nfull, qsplit ? 1 : 0, tmpkv, sched, (int)grid_spx);
    }
    else if (q_prescaled) attn_bwd_sp_kernel<true><<<(unsigned)nchain * B * H, 256, SP_LDS, s>>>(a, (u16*)ws, slab_stride, chain, nfull);
    else attn_bwd_sp_kernel<false><<<(unsigned)nchain * B * H, 256, SP_LDS, s>>>(a, (u16*)ws, slab_stride, chain, nfull);
    CRL_PROF_STOP(CRL_K_ATTN_BWD_FUSED, stream);
    CRL_LAUNCH_CHECK("crl_attn_bwd(fused)");
    if (qsplit) {
      const int64_t nadd = (int64_t)B * H * (Nk - key_base) * 8;
      attn_bwd_addkv_kernel<<<(unsigned)((nadd + 255) / 256), 256, 0, s>>>(a, tmpkv, key_base);
      CRL_LAUNCH_CHECK("crl_attn_bwd(dK / dV of the query split)");
    }
    const int64_t n8 = (int64_t)B * Nq * (H * 64 / 8);
    CRL_PROF_START(CRL_K_ATTN_DQ_REDUCE, stream, 0.0);
    attn_dq_reduce_kernel<<<(unsigned)((n8 + 255) / 256), 256, 0, s>>>((const u16*)ws, slab_stride, nslab, (u16*)dq, dq_bs, dq_rs, B, Nq, H * 64, a.dq_mul);
    CRL_PROF_STOP(CRL_K_ATTN_DQ_REDUCE, stream);
    CRL_LAUNCH_CHECK("crl_attn_bwd(dq reduce)");
    return 0;
  }
  const unsigned gk = (unsigned)a.nkt * B * H, gq = (unsigned)a.nqt * B * H;
  // algorithmic backward = dV, dP, dK (dK/dV pass) + dQ (dQ pass): 3 + 1 products of 2 x Nq x Nk x 64 per head; the
  // recomputed S (both passes) and dP (dQ pass) are executed but not counted
  const double pairs = causal ? (double)Nq * (Nk - Nq) + 0.5 * (double)Nq * (Nq + 1) : (double)Nq * Nk;
  auto run_dkdv = [&]() -> int {
    if (g_bwd_parts & 2) {
      CRL_PROF_START(CRL_K_ATTN_BWD_DKDV + (causal ? 1 : 0), stream, 6.0 * 64 * pairs * B * H);
      if (drop) { if (causal) attn_bwd_dkdv_kernel<true, true><<<gk, 256, 0, s>>>(a); else attn_bwd_dkdv_kernel<false, true><<<gk, 256, 0, s>>>(a); }
      else if (q_prescaled) { if (causal) attn_bwd_dkdv_kernel<true, false, true><<<gk, 256, 0, s>>>(a); else attn_bwd_dkdv_kernel<false, false, true><<<gk, 256, 0, s>>>(a); }
      else if (causal) attn_bwd_dkdv_kernel<true><<<gk, 256, 0, s>>>(a); else attn_bwd_dkdv_kernel<false><<<gk, 256, 0, s>>>(a);
      CRL_PROF_STOP(CRL_K_ATTN_BWD_DKDV + (causal ? 1 : 0), stream);
    }
    CRL_LAUNCH_CHECK("crl_attn_bwd(dkdv)");
    return 0;
  };
  auto run_dq = [&]() -> int {
    if (g_bwd_parts & 4) {
      CRL_PROF_START(CRL_K_ATTN_BWD_DQ + (causal ? 1 : 0), stream, 2.0 * 64 * pairs * B * H);
      if (drop) { if (causal) attn_bwd_dq_kernel<true, true><<<gq, 256, 0, s>>>(a); else attn_bwd_dq_kernel<false, true><<<gq, 256, 0, s>>>(a); }
      else if (q_prescaled) { if (causal) attn_bwd_dq_kernel<true, false, true><<<gq, 256, 0, s>>>(a); else attn_bwd_dq_kernel<false, false, true><<<gq, 256, 0, s>>>(a); }
      else if (causal) attn_bwd_dq_kernel<true><<<gq, 256, 0, s>>>(a); else attn_bwd_dq_kernel<false><<<gq, 256, 0, s>>>(a);
      CRL_PROF_STOP(CRL_K_ATTN_BWD_DQ + (causal ? 1 : 0), stream);
    }
    CRL_LAUNCH_CHECK("crl_attn_bwd(dq)");
    return 0;
  };
  if (fused) { if (int rc = run_dq()) return rc; if (int rc = run_dkdv()) return rc; }
  else { if (int rc = run_dkdv()) return rc; if (int rc = run_dq()) return rc; }
  return 0;
}

#ifdef SPX_STAMPS
extern "C" int crl_debug_spx_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(spx_dbg), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
extern "C" int crl_debug_spx_trace(unsigned long long* host) {      // 6 x 65 x 128
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(spx_trace), sizeof(spx_trace)) == hipSuccess ? 0 : -1;
}
#endif

// keep mask of the attention dropout as bytes [B, H, Nq, Nk] (1 = kept): lets the CPU oracle apply the very mask the kernels regenerate
namespace {
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(uint8_t* __restrict__ keep, uint64_t n, uint32_t thr, uint32_t key) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = attn_drop_hash((uint32_t)i, key) >= thr ? 1 : 0;
}
}  // namespace
extern "C" int crl_attn_dropout_mask(void* keep_u8, int B, int H, int Nq, int Nk, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream) {
  CRL_CHECK(keep_u8 && B > 0 && H > 0 && Nq > 0 && Nk > 0 && p >= 0.f && p < 1.f, "crl_attn_dropout_mask: bad arguments");
  const uint64_t n = (uint64_t)B * H * Nq * Nk;
  attn_dropout_mask_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>((uint8_t*)keep_u8, n, (uint32_t)(p * 16777216.f + 0.5f), attn_drop_key(seed, step, site));
  CRL_LAUNCH_CHECK("crl_attn_dropout_mask");
  return 0;
}
