// Swin shifted-window attention (head_dim 32, window <= 8x8) forward / backward for gfx950.
// One 64-lane wavefront per (image, window, head): the whole window (<= 64 tokens x 32) lives in
// LDS as fp32, lane i owns query row i.  The cyclic shift (torch.roll), window partition /
// reverse and the 0 / -100 shift-region mask are index arithmetic on the natural NHWC token
// order, so the qkv Linear and the projection run on un-permuted activations.
// (cfg-1 `cruller_small` is the plumbing config: these kernels are latency-, not MFMA-bound.)
#include "common.h"

namespace {

constexpr int HD = 32, MAXT = 64, LDP = HD + 1;

struct SwinArgs {
  const u16* qkv; const float* table; const u16* d_out; u16* out; u16* dqkv; float* dtable;
  int B, Hf, Wf, heads, w, shift, nWx, nWy;
  float scale;
};

__device__ __forceinline__ int region(int s, int size, int w, int shift) { return (s >= size - w) + (s >= size - shift); }

struct Tok { int t; int reg; int iy, ix; };
__device__ __forceinline__ Tok token_of(const SwinArgs& a, int b, int wy, int wx, int i) {
  Tok r;
  r.iy = i / a.w; r.ix = i - r.iy * a.w;
  const int sy = wy * a.w + r.iy, sx = wx * a.w + r.ix;       // coordinates in the rolled map
  const int y = (sy + a.shift) % a.Hf, x = (sx + a.shift) % a.Wf;  // roll(-shift): rolled[s] = orig[(s + shift) % size]
  r.t = (b * a.Hf + y) * a.Wf + x;
  r.reg = a.shift ? region(sy, a.Hf, a.w, a.shift) * 3 + region(sx, a.Wf, a.w, a.shift) : 0;
  return r;
}

template <bool BWD>
__global__ __launch_bounds__(64) void swin_attn_kernel(const SwinArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* qs = reinterpret_cast<float*>(smem_raw);   // [MAXT][LDP]
  float* ks = qs + MAXT * LDP;
  float* vs = ks + MAXT * LDP;
  int* regs = reinterpret_cast<int*>(vs + MAXT * LDP);  // [MAXT] region id, [MAXT] iy*w+ix is i itself
  float* dos = reinterpret_cast<float*>(regs + MAXT);   // BWD: [MAXT][LDP]
  const int lane = threadIdx.x;
  const int n = a.w * a.w;
  float* Ps = dos + MAXT * LDP;                          // BWD: [n][n+1]
  float* dSs = Ps + n * (n + 1);                         // BWD: [n][n+1]
  float* tbl = dSs + n * (n + 1);                        // BWD: [(2w-1)^2]

  int bid = blockIdx.x;
  const int head = bid % a.heads; bid /= a.heads;
  const int wx = bid % a.nWx; bid /= a.nWx;
  const int wy = bid % a.nWy;
  const int b = bid / a.nWy;
  const int C = a.heads * HD;
  const int nt = (2 * a.w - 1) * (2 * a.w - 1);

  Tok me{};
  if (lane < n) {
    me = token_of(a, b, wy, wx, lane);
    const u16* base = a.qkv + (size_t)me.t * 3 * C + head * HD;
#pragma unroll
    for (int c8 = 0; c8 < HD / 8; ++c8) {
      const uint4 qv = *reinterpret_cast<const uint4*>(base + c8 * 8);
      const uint4 kv = *reinterpret_cast<const uint4*>(base + C + c8 * 8);
      const uint4 vv = *reinterpret_cast<const uint4*>(base + 2 * C + c8 * 8);
      const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w}, kw[4] = {kv.x, kv.y, kv.z, kv.w}, vw[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qs[lane * LDP + c8 * 8 + 2 * j] = bf2f(qw[j] & 0xffff); qs[lane * LDP + c8 * 8 + 2 * j + 1] = bf2f(qw[j] >> 16);
        ks[lane * LDP + c8 * 8 + 2 * j] = bf2f(kw[j] & 0xffff); ks[lane * LDP + c8 * 8 + 2 * j + 1] = bf2f(kw[j] >> 16);
        vs[lane * LDP + c8 * 8 + 2 * j] = bf2f(vw[j] & 0xffff); vs[lane * LDP + c8 * 8 + 2 * j + 1] = bf2f(vw[j] >> 16);
      }
    }
    regs[lane] = me.reg;
    if (BWD) {
      const u16* dob = a.d_out + (size_t)me.t * C + head * HD;
#pragma unroll
      for (int c8 = 0; c8 < HD / 8; ++c8) {
        const uint4 dv = *reinterpret_cast<const uint4*>(dob + c8 * 8);
        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { dos[lane * LDP + c8 * 8 + 2 * j] = bf2f(dw[j] & 0xffff); dos[lane * LDP + c8 * 8 + 2 * j + 1] = bf2f(dw[j] >> 16); }
      }
    }
  }
  if (BWD) for (int i = lane; i < nt; i += 64) tbl[i] = 0.f;
  __syncthreads();

  float q[HD];
  float srow[MAXT];
  float mx = -INFINITY, sum = 0.f;
  if (lane < n) {
#pragma unroll
    for (int c = 0; c < HD; ++c) q[c] = qs[lane * LDP + c];
    for (int j = 0; j < n; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) s += q[c] * ks[j * LDP + c];
      const int jy = j / a.w, jx = j - jy * a.w;
      const int ridx = (me.iy - jy + a.w - 1) * (2 * a.w - 1) + (me.ix - jx + a.w - 1);
      s = s * a.scale + a.table[ridx * a.heads + head];
      if (a.shift && regs[j] != me.reg) s += -100.f;
      srow[j] = s;
      mx = fmaxf(mx, s);
    }
    for (int j = 0; j < n; ++j) { srow[j] = __expf(srow[j] - mx); sum += srow[j]; }
    const float inv = 1.f / sum;
    for (int j = 0; j < n; ++j) srow[j] *= inv;
  }

  if (!BWD) {
    if (lane < n) {
      float o[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = 0.f;
      for (int j = 0; j < n; ++j) {
        const float p = round_bf(srow[j]);  // probabilities enter P.V as bf16, like the flash kernels
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] += p * vs[j * LDP + c];
      }
      u16* ob = a.out + (size_t)me.t * C + head * HD;
#pragma unroll
      for (int c8 = 0; c8 < HD / 8; ++c8)
        *reinterpret_cast<uint4*>(ob + c8 * 8) = uint4{pack_bf2(o[c8 * 8], o[c8 * 8 + 1]), pack_bf2(o[c8 * 8 + 2], o[c8 * 8 + 3]),
                                                       pack_bf2(o[c8 * 8 + 4], o[c8 * 8 + 5]), pack_bf2(o[c8 * 8 + 6], o[c8 * 8 + 7])};
    }
    return;
  }

  // ---------------- backward
  const int LDS_N = n + 1;
  if (lane < n) {
    float dO[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dO[c] = dos[lane * LDP + c];
    float delta = 0.f;
    float dprow[MAXT];
    for (int j = 0; j < n; ++j) {
      float dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) dp += dO[c] * vs[j * LDP + c];
      dprow[j] = dp;
      delta += srow[j] * dp;
    }
    float dq[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dq[c] = 0.f;
    for (int j = 0; j < n; ++j) {
      const float ds = srow[j] * (dprow[j] - delta);
      Ps[lane * LDS_N + j] = round_bf(srow[j]);
      dSs[lane * LDS_N + j] = ds;
      const int jy = j / a.w, jx = j - jy * a.w;
      const int ridx = (me.iy - jy + a.w - 1) * (2 * a.w - 1) + (me.ix - jx + a.w - 1);
      atomicAdd(&tbl[ridx], ds);
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] += ds * ks[j * LDP + c];
    }
    u16* dqb = a.dqkv + (size_t)me.t * 3 * C + head * HD;
#pragma unroll
    for (int c8 = 0; c8 < HD / 8; ++c8)
      *reinterpret_cast<uint4*>(dqb + c8 * 8) = uint4{pack_bf2(dq[c8 * 8] * a.scale, dq[c8 * 8 + 1] * a.scale), pack_bf2(dq[c8 * 8 + 2] * a.scale, dq[c8 * 8 + 3] * a.scale),
                                                      pack_bf2(dq[c8 * 8 + 4] * a.scale, dq[c8 * 8 + 5] * a.scale), pack_bf2(dq[c8 * 8 + 6] * a.scale, dq[c8 * 8 + 7] * a.scale)};
  }
  __syncthreads();
  if (lane < n) {  // lane now plays key/value row j
    float dk[HD], dv[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
    for (int i = 0; i < n; ++i) {
      const float p = Ps[i * LDS_N + lane], ds = dSs[i * LDS_N + lane];
#pragma unroll
      for (int c = 0; c < HD; ++c) { dk[c] += ds * qs[i * LDP + c]; dv[c] += p * dos[i * LDP + c]; }
    }
    u16* dkb = a.dqkv + (size_t)me.t * 3 * C + C + head * HD;
    u16* dvb = dkb + C;
#pragma unroll
    for (int c8 = 0; c8 < HD / 8; ++c8) {
      *reinterpret_cast<uint4*>(dkb + c8 * 8) = uint4{pack_bf2(dk[c8 * 8] * a.scale, dk[c8 * 8 + 1] * a.scale), pack_bf2(dk[c8 * 8 + 2] * a.scale, dk[c8 * 8 + 3] * a.scale),
                                                      pack_bf2(dk[c8 * 8 + 4] * a.scale, dk[c8 * 8 + 5] * a.scale), pack_bf2(dk[c8 * 8 + 6] * a.scale, dk[c8 * 8 + 7] * a.scale)};
      *reinterpret_cast<uint4*>(dvb + c8 * 8) = uint4{pack_bf2(dv[c8 * 8], dv[c8 * 8 + 1]), pack_bf2(dv[c8 * 8 + 2], dv[c8 * 8 + 3]),
                                                      pack_bf2(dv[c8 * 8 + 4], dv[c8 * 8 + 5]), pack_bf2(dv[c8 * 8 + 6], dv[c8 * 8 + 7])};
    }
  }
  for (int i = lane; i < nt; i += 64) atomicAdd(&a.dtable[i * a.heads + head], tbl[i]);
}

size_t swin_lds_bytes(bool bwd, int w) {
  const size_t n = (size_t)w * w;
  size_t f = 3 * MAXT * LDP + MAXT;  // q,k,v + region ids
  if (bwd) f += MAXT * LDP + 2 * n * (n + 1) + (size_t)(2 * w - 1) * (2 * w - 1);
  return f * 4;
}

int swin_check(const char* who, int B, int Hf, int Wf, int heads, int w, int shift) {
  CRL_CHECK(B > 0 && heads > 0 && w > 0 && w * w <= MAXT, "%s: window %d not supported (w*w <= 64)", who, w);
  CRL_CHECK((Hf % w) == 0 && (Wf % w) == 0, "%s: feature map %dx%d not divisible by window %d", who, Hf, Wf, w);
  CRL_CHECK(shift >= 0 && shift < w, "%s: bad shift %d", who, shift);
  return 0;
}

}  // namespace

extern "C" int crl_swin_attn_fwd(const void* qkv, const float* table, void* out, int B, int Hf, int Wf, int heads, int w,
                                 int shift, float scale, void* stream) {
  if (swin_check("crl_swin_attn_fwd", B, Hf, Wf, heads, w, shift)) return -1;
  CRL_CHECK(qkv && table && out, "crl_swin_attn_fwd: null pointer");
  SwinArgs a{};
  a.qkv = (const u16*)qkv; a.table = table; a.out = (u16*)out;
  a.B = B; a.Hf = Hf; a.Wf = Wf; a.heads = heads; a.w = w; a.shift = shift; a.nWx = Wf / w; a.nWy = Hf / w; a.scale = scale;
  swin_attn_kernel<false><<<(unsigned)(B * a.nWy * a.nWx * heads), 64, swin_lds_bytes(false, w), as_stream(stream)>>>(a);
  CRL_LAUNCH_CHECK("crl_swin_attn_fwd");
  return 0;
}

extern "C" int crl_swin_attn_bwd(const void* qkv, const float* table, const void* d_out, void* dqkv, float* dtable, int B,
                                 int Hf, int Wf, int heads, int w, int shift, float scale, void* stream) {
  if (swin_check("crl_swin_attn_bwd", B, Hf, Wf, heads, w, shift)) return -1;
  CRL_CHECK(qkv && table && d_out && dqkv && dtable, "crl_swin_attn_bwd: null pointer");
  SwinArgs a{};
  a.qkv = (const u16*)qkv; a.table = table; a.d_out = (const u16*)d_out; a.dqkv = (u16*)dqkv; a.dtable = dtable;
  a.B = B; a.Hf = Hf; a.Wf = Wf; a.heads = heads; a.w = w; a.shift = shift; a.nWx = Wf / w; a.nWy = Hf / w; a.scale = scale;
  const size_t lds = swin_lds_bytes(true, w);
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&swin_attn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CRL_CHECK(e == hipSuccess, "crl_swin_attn_bwd: cannot raise dynamic LDS to %zu bytes", lds);
  }
  swin_attn_kernel<true><<<(unsigned)(B * a.nWy * a.nWx * heads), 64, lds, as_stream(stream)>>>(a);
  CRL_LAUNCH_CHECK("crl_swin_attn_bwd");
  return 0;
}
