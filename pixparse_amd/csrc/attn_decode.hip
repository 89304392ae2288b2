// Single-query attention for generation with a KV cache (include/crl.h: crl_attn_decode), head_dim 64.
// One new token per sequence attends to Nk cached keys: 256 B of K/V per key and 256 FLOP -- purely HBM-bound, no
// matrix cores. The key range of every (batch, head) is split over `nsplit` workgroups so that a handful of sequences
// still covers the chip ("flash decoding"); eight lanes own a key (16 bytes of its K and V rows each: whole rows per load
// instruction), take partial dot products with their 8 query channels, sum them over the 8 lanes, keep the online-softmax state
// of their key group and a private fp32 P.V accumulator of their 8 channels. Key groups, waves, then splits are merged by rescaling with 2^(m - m_max) in a
// fixed order (deterministic); a key range of up to 1024 keys (the decoder's self-attention cache) is one workgroup per
// (b, h) that normalises and stores directly. P is rounded to bf16 before P.V like crl_attn_fwd, the row sum stays fp32.
#include "common.h"

namespace {

constexpr float LOG2E_D = 1.4426950408889634f;

struct DecArgs {
  const u16 *q, *k, *v;
  u16* o;
  float* ws;              // [B*H][nsplit][66] : m, l, o[64]
  int64_t q_bs, k_bs, k_rs, v_bs, v_rs, o_bs;
  int B, H, Nk, nsplit, chunk;
  float scale;
  const int* nk_m1;       // optional: the valid prefix is *nk_m1 + 1 keys (Nk is then the cache capacity the splits are planned for)
  const int* q_row; int64_t q_row_stride;   // optional: q += *q_row * q_row_stride (the query lives in cache row `step`)
};

__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// sum over the 8 lanes that share a key (lane & 7 = which 16-byte part of the row): two quad permutes and a half-row mirror, on the VALU's
// DPP path (no LDS crossbar)
__device__ __forceinline__ float sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  return v;
}

// Eight lanes share a key: lane & 7 owns 16 bytes (8 channels) of its 128-byte K and V rows, lane >> 3 picks the key among the 8 a
// load instruction covers, so every load of a wave fetches 8 whole rows (round 3: the lane-per-key form issued 64 different 128-byte
// lines per instruction and re-fetched each of them 8 times through a 32 KiB L1 that cannot hold a wave's 16 KiB x 12 waves: 3.0 TB/s).
// All 16 loads of a 64-key batch are in flight before the first use.  The score is a partial dot product over the lane's 8 channels
// summed over the 8 lanes; the online-softmax state (m, l) is replicated in the 8 lanes of a key group and each lane accumulates its
// own 8 output channels.
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecArgs a) {
  __shared__ float red[4][66];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int part = lane & 7, kg = lane >> 3;
  const int bh = blockIdx.x, split = blockIdx.y;
  const int b = bh / a.H, h = bh % a.H;
  const int nk = a.nk_m1 ? min(a.Nk, *a.nk_m1 + 1) : a.Nk;
  const int k_lo = split * a.chunk, k_hi = min(nk, k_lo + a.chunk);
  const float c = a.scale * LOG2E_D;

  float q[8];
  {
    const uint4 qv = *reinterpret_cast<const uint4*>(a.q + (a.q_row ? (int64_t)(*a.q_row) * a.q_row_stride : 0) + b * a.q_bs + h * 64 + 8 * part);
    unpack8(qv, q);
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] *= c;                       // scores directly in the log2 domain
  }
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const u16* kb = a.k + b * a.k_bs + h * 64 + 8 * part;
  const u16* vb = a.v + b * a.v_bs + h * 64 + 8 * part;

  for (int base = k_lo + wave * 64; base < k_hi; base += 256) {
    uint4 kr[8], vr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = min(base + 8 * j + kg, k_hi - 1);          // clamped: rows past the range are loaded twice and skipped below
      kr[j] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * a.k_rs);
      vr[j] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * a.v_rs);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f[8];
      unpack8(kr[j], f);
      float s0 = f[0] * q[0], s1 = f[1] * q[1];
#pragma unroll
      for (int e = 2; e < 8; e += 2) { s0 = __builtin_fmaf(f[e], q[e], s0); s1 = __builtin_fmaf(f[e + 1], q[e + 1], s1); }
      const float s = sum8(s0 + s1);
      if (base + 8 * j + kg < k_hi) {                            // uniform over the 8 lanes of a key
        const float m_new = fmaxf(m, s);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);   // first key: exp2(-inf) = 0
        const float p = __builtin_amdgcn_exp2f(s - m_new);
        const float pb = round_bf(p);
        l = l * alpha + p;
        m = m_new;
        unpack8(vr[j], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(acc[e], alpha, pb * f[e]);
      }
    }
  }
  // key groups -> wave: the 8 groups of a wave hold different references
  const float m_w = wave_max(m);
  const float r = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_w);
  float l_w = l * r;
  l_w += __shfl_xor(l_w, 8, 64); l_w += __shfl_xor(l_w, 16, 64); l_w += __shfl_xor(l_w, 32, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = acc[e] * r;
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    acc[e] = v;
  }
  if (kg == 0) {
    if (part == 0) { red[wave][0] = m_w; red[wave][1] = l_w; }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][2 + 8 * part + e] = acc[e];
  }
  __syncthreads();
  // waves -> workgroup partial (thread d < 64 owns output channel d)
  if (threadIdx.x < 64) {
    const int d = threadIdx.x;
    const float mb = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    float lb = 0.f, ob = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float rw = (red[w][0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(red[w][0] - mb);
      lb += red[w][1] * rw;
      ob += red[w][2 + d] * rw;
    }
    if (a.nsplit == 1) {   // whole key range in one workgroup: normalise and store, no merge pass
      a.o[b * a.o_bs + h * 64 + d] = f2bf(lb > 0.f ? ob / lb : 0.f);
      return;
    }
    float* dst = a.ws + ((int64_t)bh * a.nsplit + split) * 66;
    if (d == 0) { dst[0] = mb; dst[1] = lb; }
    dst[2 + d] = ob;
  }
}

// splits -> output: one wave per (b, h), partials combined in split order (deterministic). A separate launch on purpose: a
// "last arriving split merges" variant needs agent-scope fences, i.e. L2 write-back / invalidate on every one of the 8 XCDs
// per workgroup -- measured 0.24 ms per decode step slower than this 64-thread kernel.
__global__ __launch_bounds__(64) void attn_decode_merge(const DecArgs a) {
  const int bh = blockIdx.x, d = threadIdx.x;
  const int b = bh / a.H, h = bh % a.H;
  const float* src = a.ws + (int64_t)bh * a.nsplit * 66;
  float mg = -INFINITY;
  for (int s = 0; s < a.nsplit; ++s) mg = fmaxf(mg, src[s * 66]);
  float l = 0.f, o = 0.f;
  for (int s = 0; s < a.nsplit; ++s) {
    const float ms = src[s * 66];
    const float r = (ms == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(ms - mg);
    l += src[s * 66 + 1] * r;
    o += src[s * 66 + 2 + d] * r;
  }
  a.o[b * a.o_bs + h * 64 + d] = f2bf(l > 0.f ? o / l : 0.f);
}

int plan_splits(int BH, int Nk, int* chunk) {
  if (Nk <= 1024) { *chunk = 1024; return 1; }       // decoder self-attention: one workgroup per (b, h), direct store
  int nsplit = (1024 + BH - 1) / BH;                 // ~4 workgroups per CU when the key range allows it
  const int max_split = (Nk + 255) / 256;            // at least one 256-key pass per workgroup
  nsplit = nsplit < 1 ? 1 : (nsplit > max_split ? max_split : nsplit);
  if (nsplit > 64) nsplit = 64;
  int ch = (Nk + nsplit - 1) / nsplit;
  ch = (ch + 255) / 256 * 256;
  *chunk = ch;
  return (Nk + ch - 1) / ch;
}

}  // namespace

extern "C" size_t crl_attn_decode_ws_bytes(int B, int H, int Nk) {
  int chunk;
  return (size_t)B * H * plan_splits(B * H, Nk, &chunk) * 66 * sizeof(float);
}

extern "C" int crl_attn_decode(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_rs, const void* v, int64_t v_bs,
                               int64_t v_rs, void* o, int64_t o_bs, int B, int H, int Nk, float scale, const int* nk_minus1_dev,
                               const int* q_row_dev, int64_t q_row_stride, void* ws, size_t ws_bytes, void* stream) {
  const char* who = "crl_attn_decode";
  CRL_CHECK(q && k && v && o && ws, "%s: null pointer", who);
  CRL_CHECK(!q_row_dev || (q_row_stride % 8) == 0, "%s: q_row_stride must be a multiple of 8 elements", who);
  CRL_CHECK(B > 0 && H > 0 && Nk > 0, "%s: empty problem", who);
  CRL_CHECK(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && (q_bs % 8) == 0 && (k_bs % 8) == 0 &&
                (k_rs % 8) == 0 && (v_bs % 8) == 0 && (v_rs % 8) == 0 && k_rs >= (int64_t)H * 64 && v_rs >= (int64_t)H * 64,
            "%s: operands must be 16-byte aligned with strides multiple of 8 elements", who);
  DecArgs a{};
  a.q = (const u16*)q; a.k = (const u16*)k; a.v = (const u16*)v; a.o = (u16*)o; a.ws = (float*)ws;
  a.q_bs = q_bs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs;
  a.B = B; a.H = H; a.Nk = Nk; a.scale = scale; a.nk_m1 = nk_minus1_dev; a.q_row = q_row_dev; a.q_row_stride = q_row_stride;
  a.nsplit = plan_splits(B * H, Nk, &a.chunk);
  CRL_CHECK(ws_bytes >= (size_t)B * H * a.nsplit * 66 * sizeof(float), "%s: workspace too small (%zu bytes)", who, ws_bytes);
  hipStream_t s = as_stream(stream);
  attn_decode_kernel<<<dim3((unsigned)(B * H), (unsigned)a.nsplit), 256, 0, s>>>(a);
  CRL_LAUNCH_CHECK(who);
  if (a.nsplit > 1) {
    attn_decode_merge<<<(unsigned)(B * H), 64, 0, s>>>(a);
    CRL_LAUNCH_CHECK("crl_attn_decode(merge)");
  }
  return 0;
}
