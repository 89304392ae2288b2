// Single-query attention for generation with a KV cache (include/crl.h: crl_attn_decode), head_dim 64.
// One new token per sequence attends to Nk cached keys: 256 B of K/V per key and 256 FLOP -- purely HBM-bound, no
// matrix cores. The key range of every (batch, head) is split over `nsplit` workgroups so that a handful of sequences
// still covers the chip ("flash decoding"); a lane owns a key: it reads that key's 128-B K row, takes the dot product
// with the query (held in registers by every lane), keeps a private online-softmax state and a private fp32 P.V
// accumulator over the keys it visits. Lanes, then waves, then splits are merged by rescaling with 2^(m - m_max) in a
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

__global__ __launch_bounds__(256) void attn_decode_kernel(const DecArgs a) {
  __shared__ float red[4][66];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x, split = blockIdx.y;
  const int b = bh / a.H, h = bh % a.H;
  const int nk = a.nk_m1 ? min(a.Nk, *a.nk_m1 + 1) : a.Nk;
  const int k_lo = split * a.chunk, k_hi = min(nk, k_lo + a.chunk);
  const float c = a.scale * LOG2E_D;

  float q[64];
  {
    const uint4* qp = reinterpret_cast<const uint4*>(a.q + (a.q_row ? (int64_t)(*a.q_row) * a.q_row_stride : 0) + b * a.q_bs + h * 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f[8];
      unpack8(qp[j], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) q[8 * j + e] = f[e] * c;     // scores directly in the log2 domain
    }
  }
  float m = -INFINITY, l = 0.f, acc[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) acc[d] = 0.f;

  for (int key = k_lo + wave * 64 + lane; key < k_hi; key += 256) {
    const uint4* kp = reinterpret_cast<const uint4*>(a.k + b * a.k_bs + (int64_t)key * a.k_rs + h * 64);
    const uint4* vp = reinterpret_cast<const uint4*>(a.v + b * a.v_bs + (int64_t)key * a.v_rs + h * 64);
    uint4 kr[8], vr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kr[j] = kp[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) vr[j] = vp[j];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f[8];
      unpack8(kr[j], f);
#pragma unroll
      for (int e = 0; e < 8; e += 2) { s0 = __builtin_fmaf(f[e], q[8 * j + e], s0); s1 = __builtin_fmaf(f[e + 1], q[8 * j + e + 1], s1); }
    }
    const float s = s0 + s1;
    const float m_new = fmaxf(m, s);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);        // first key: exp2(-inf) = 0
    const float p = __builtin_amdgcn_exp2f(s - m_new);
    const float pb = round_bf(p);
    l = l * alpha + p;
    m = m_new;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f[8];
      unpack8(vr[j], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[8 * j + e] = __builtin_fmaf(acc[8 * j + e], alpha, pb * f[e]);
    }
  }
  // lanes -> wave
  const float m_w = wave_max(m);
  const float r = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_w);
  const float l_w = wave_sum(l * r);
#pragma unroll
  for (int d = 0; d < 64; ++d) acc[d] = wave_sum(acc[d] * r);
  if (lane == 0) {
    red[wave][0] = m_w; red[wave][1] = l_w;
#pragma unroll
    for (int d = 0; d < 64; ++d) red[wave][2 + d] = acc[d];
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
