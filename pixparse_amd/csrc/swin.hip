// Swin shifted-window attention (head_dim 32, window <= 8x8) forward / backward for gfx950 on the matrix cores
// (include/crl.h: crl_swin_attn_fwd / crl_swin_attn_bwd; replaces timm WindowAttention + roll / window_partition /
// window_reverse, timm/models/swin_transformer.py as built at models/image_encoder_timm.py:13-20 of the reference).
//
// One wavefront per (image, window, head); a workgroup = NW windows of ONE head (4 forward, 2 backward), so the head's
// relative-position bias is expanded once per workgroup into a [64][64] fp32 LDS matrix.  The window (49 tokens for w = 7,
// padded to 64 with zero rows) is staged in LDS as bf16 tiles of the flash-attention format (attn_frag.h: 64 rows x 128 B,
// XOR swizzle conflict-free for row and transposed reads; head_dim 32 fills the first half of a row) and all products run
// as v_mfma_f32_32x32x16_bf16 with the query on the lane:
//   fwd   S^T = K.Q^T (2x2 tiles, K = 32) -> + bias, shift mask, padding -> softmax per lane (+ one v_permlane32_swap)
//         -> O^T += V^T.P^T with the probability accumulators used directly as the next operand
//   bwd   S^T, dP^T = V.dO^T -> dS^T = P^T o (dP^T - delta) in registers -> dQ^T += K^T.dS^T;  P and dS are written once to
//         LDS as bf16 tiles [query][key] and read back transposed: dV^T += dO^T.P, dK^T += Q^T.dS;  the bias gradient is
//         accumulated per workgroup in LDS and added to the table with one atomic per entry.
// The cyclic shift (torch.roll), window partition / reverse and the 0 / -100 shift-region mask are index arithmetic on the
// natural NHWC token order, so the qkv Linear and the projection run on un-permuted activations.
#include <type_traits>
#include "common.h"
#include "attn_frag.h"

namespace {
using namespace attnf;

constexpr int HD = 32, MAXT = 64;
constexpr int TILE = 8192;          // one 64 x 64 bf16 tile
constexpr int BPITCH = 68;          // bias matrix row pitch in floats (272 B: 16-byte aligned rows, conflict-free float4 reads per lane)
constexpr float LOG2E = 1.4426950408889634f;

struct SwinArgs {
  const u16* qkv; const float* table; const u16* d_out; u16* out; u16* dqkv; float* dtable;
  int B, Hf, Wf, heads, w, shift, nWx, nWy, nwin;
  float scale;
};

__device__ __forceinline__ int region(int s, int size, int w, int shift) { return (s >= size - w) + (s >= size - shift); }

struct Tok { int t; int reg; };
// token i of window (b, wy, wx): index in the natural NHWC order and its shift region
__device__ __forceinline__ Tok token_of(const SwinArgs& a, int b, int wy, int wx, int i) {
  Tok r;
  const int iy = i / a.w, ix = i - iy * a.w;
  const int sy = wy * a.w + iy, sx = wx * a.w + ix;             // coordinates in the rolled map
  const int y = (sy + a.shift) % a.Hf, x = (sx + a.shift) % a.Wf;  // roll(-shift): rolled[s] = orig[(s + shift) % size]
  r.t = (b * a.Hf + y) * a.Wf + x;
  r.reg = a.shift ? region(sy, a.Hf, a.w, a.shift) * 3 + region(sx, a.Wf, a.w, a.shift) : 0;
  return r;
}

// row `row` of a tile <- 32 bf16 from global memory (or zeros): logical 16-byte slot s lives at physical slot s ^ swz64(row)
__device__ __forceinline__ void stage_row(char* tile, int row, const u16* src, bool valid) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint4 v{0u, 0u, 0u, 0u};
    if (valid) v = *reinterpret_cast<const uint4*>(src + 8 * s);
    *reinterpret_cast<uint4*>(tile + row * 128 + ((s ^ swz64(row)) << 4)) = v;
  }
}
// accumulator block (query on the lane, 16 rows of the other index in the registers) -> bf16 tile [lane row][32 cb + row index]
__device__ __forceinline__ void store_acc_tile(char* tile, int row, int cb, int hh, const f32x16& x) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int col = 32 * cb + 8 * g4 + 4 * hh;
    *reinterpret_cast<uint2*>(tile + row * 128 + (((col >> 3) ^ swz64(row)) << 4) + (col & 7) * 2) =
        uint2{pack_bf2(x[4 * g4], x[4 * g4 + 1]), pack_bf2(x[4 * g4 + 2], x[4 * g4 + 3])};
  }
}

template <bool BWD, int NW>
__global__ __launch_bounds__(64 * NW) void swin_attn_kernel(const SwinArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTILE = BWD ? 6 : 2;                 // per wave: K, V (+ Q, dO, P, dS)
  float* biasT = reinterpret_cast<float*>(smem);     // [64 queries][BPITCH] bias of this head, log2 domain
  int* kofs = reinterpret_cast<int*>(biasT + MAXT * BPITCH);   // [64] jy (2w-1) + jx of key j
  float* tbl = reinterpret_cast<float*>(kofs + MAXT);          // BWD: [(2w-1)^2] bias-gradient partial of this workgroup
  const int nt = (2 * a.w - 1) * (2 * a.w - 1);
  char* tiles0 = reinterpret_cast<char*>(tbl + ((BWD ? nt : 0) + 3) / 4 * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = lane & 31, hh = lane >> 5;
  const int n = a.w * a.w;
  const int head = blockIdx.x % a.heads;
  const int gw = (blockIdx.x / a.heads) * NW + wave;     // global window index of this wave
  const bool live = gw < a.nwin;
  const int C = a.heads * HD;

  // ---- per-workgroup tables: bias^T[i][j] = table[rel(i, j)][head] * log2(e), key offsets, zeroed gradient partial
  for (int e = tid; e < MAXT * MAXT; e += 64 * NW) {
    const int i = e >> 6, j = e & 63;
    float v = 0.f;
    if (i < n && j < n) {
      const int iy = i / a.w, ix = i - iy * a.w, jy = j / a.w, jx = j - jy * a.w;
      v = a.table[((iy - jy + a.w - 1) * (2 * a.w - 1) + (ix - jx + a.w - 1)) * a.heads + head] * LOG2E;
    }
    biasT[i * BPITCH + j] = v;
  }
  if (tid < MAXT) { const int jy = tid / a.w; kofs[tid] = jy * (2 * a.w - 1) + (tid - jy * a.w); }
  if (BWD) for (int e = tid; e < nt; e += 64 * NW) tbl[e] = 0.f;

  // ---- stage the window: lane i = token i
  char* kl = tiles0 + wave * NTILE * TILE;
  char* vl = kl + TILE;
  char* ql = vl + TILE;      // BWD
  char* dol = ql + TILE;     // BWD
  char* pl = dol + TILE;     // BWD: P  [query][key]
  char* dsl = pl + TILE;     // BWD: dS [query][key]
  int b = 0, wy = 0, wx = 0;
  if (live) { wx = gw % a.nWx; const int r = gw / a.nWx; wy = r % a.nWy; b = r / a.nWy; }
  const bool tok_ok = live && lane < n;
  Tok me{0, 0};
  if (tok_ok) me = token_of(a, b, wy, wx, lane);
  {
    const u16* base = a.qkv + (size_t)me.t * 3 * C + head * HD;
    stage_row(kl, lane, base + C, tok_ok);
    stage_row(vl, lane, base + 2 * C, tok_ok);
    if (BWD) {
      stage_row(ql, lane, base, tok_ok);
      stage_row(dol, lane, a.d_out + (size_t)me.t * C + head * HD, tok_ok);
    }
  }
  __syncthreads();   // tables + tiles visible (the only workgroup barrier)

  const LaneAddr la = make_lane_addr(lane);
  // the two query rows of this lane (query blocks 0 / 1): token, region, natural-order Q (and dO) fragments
  int qtok[2], qreg[2];
  bf16x8 qf[2][2], dof[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = 32 * qb + qi;
    qtok[qb] = __shfl(me.t, q, 64);
    qreg[qb] = __shfl(me.reg, q, 64);
    if (BWD) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { qf[qb][ks] = frag_row(ql, la, 32 * qb, ks); dof[qb][ks] = frag_row(dol, la, 32 * qb, ks); }
    } else {
      const bool ok = live && q < n;
      const u16* qp = a.qkv + (size_t)qtok[qb] * 3 * C + head * HD;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
        qf[qb][ks] = ok ? *reinterpret_cast<const bf16x8*>(qp + 16 * ks + 8 * hh) : z;
      }
    }
  }
  // region ids of the 32 keys this lane sees per key block (keys 32 kb + 8 g4 + 4 hh + 0..3)
  const float c = a.scale * LOG2E;

  // ---- S^T = K.Q^T (+ bias + mask), softmax over the keys of each query
  f32x16 p[2][2];   // [key block][query block]
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s = zero16();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) s = mfma32(frag_row(kl, la, 32 * kb, ks), qf[qb][ks], s);
      const int q = 32 * qb + qi;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int k0 = 32 * kb + 8 * g4 + 4 * hh;
        const float4 bq = *reinterpret_cast<const float4*>(biasT + q * BPITCH + k0);
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int key = k0 + t;
          float v = __builtin_fmaf(s[4 * g4 + t], c, bb[t]);
          if (a.shift) { if (__shfl(me.reg, key, 64) != qreg[qb]) v += -100.f * LOG2E; }
          if (key >= n) v = -INFINITY;
          s[4 * g4 + t] = v;
        }
      }
      p[kb][qb] = s;
    }
  float inv[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = max3f(mx, p[0][qb][r], p[1][qb][r]);
    mx = fmaxf(mx, swap32(mx));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(p[kb][qb][r] - mx); p[kb][qb][r] = e; sum += e; }
    sum += swap32(sum);
    inv[qb] = 1.f / sum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) p[kb][qb][r] *= inv[qb];
  }

  if (!BWD) {
    // ---- O^T = V^T.P^T: probabilities enter the product as bf16 (like the flash kernels)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 o = zero16();
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) o = mfma32(frag_tr(vl, la, 32 * kb + 16 * s, 0), acc_frag(p[kb][qb], s), o);
      const int q = 32 * qb + qi;
      if (live && q < n) {
        u16* ob = a.out + (size_t)qtok[qb] * C + head * HD;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<uint2*>(ob + 8 * g4 + 4 * hh) = uint2{pack_bf2(o[4 * g4], o[4 * g4 + 1]), pack_bf2(o[4 * g4 + 2], o[4 * g4 + 3])};
      }
    }
    return;
  }

  // ---------------- backward
  // dP^T = V.dO^T, delta = sum_keys P dP, dS^T = P (dP - delta)
  f32x16 ds[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 d = zero16();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) d = mfma32(frag_row(vl, la, 32 * kb, ks), dof[qb][ks], d);
      ds[kb][qb] = d;
    }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float delta = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) delta += p[kb][qb][r] * ds[kb][qb][r];
    delta += swap32(delta);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[kb][qb][r] = p[kb][qb][r] * (ds[kb][qb][r] - delta);
  }
  // P and dS as bf16 tiles [query][key] for the key-on-the-lane products; bias gradient (dS in logit units) into the LDS partial
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = 32 * qb + qi;
    const int qy = q / a.w, qx = q - qy * a.w;
    const int qbase = (qy + a.w - 1) * (2 * a.w - 1) + (qx + a.w - 1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      store_acc_tile(pl, q, kb, hh, p[kb][qb]);
      store_acc_tile(dsl, q, kb, hh, ds[kb][qb]);
      if (live && q < n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * kb + acc_row(r, hh);
          if (key < n) atomicAdd(&tbl[qbase - kofs[key]], ds[kb][qb][r]);
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's tile writes are complete before its transposed reads (no other wave touches them)
  // dQ^T = K^T.dS^T
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    f32x16 dq = zero16();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) dq = mfma32(frag_tr(kl, la, 32 * kb + 16 * s, 0), acc_frag(ds[kb][qb], s), dq);
    const int q = 32 * qb + qi;
    if (live && q < n) {
      u16* dqb = a.dqkv + (size_t)qtok[qb] * 3 * C + head * HD;
      const float sc = a.scale;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<uint2*>(dqb + 8 * g4 + 4 * hh) = uint2{pack_bf2(dq[4 * g4] * sc, dq[4 * g4 + 1] * sc), pack_bf2(dq[4 * g4 + 2] * sc, dq[4 * g4 + 3] * sc)};
    }
  }
  // dV^T = dO^T.P, dK^T = Q^T.dS (key on the lane): both operands by transposed reads, contraction over the 64 queries
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    f32x16 dv = zero16(), dk = zero16();
#pragma unroll
    for (int qs = 0; qs < 4; ++qs) {
      dv = mfma32(frag_tr(dol, la, 16 * qs, 0), frag_tr(pl, la, 16 * qs, kb), dv);
      dk = mfma32(frag_tr(ql, la, 16 * qs, 0), frag_tr(dsl, la, 16 * qs, kb), dk);
    }
    const int key = 32 * kb + qi;
    if (live && key < n) {
      const int kt = __shfl(me.t, key, 64);
      u16* dkb = a.dqkv + (size_t)kt * 3 * C + C + head * HD;
      u16* dvb = dkb + C;
      const float sc = a.scale;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *reinterpret_cast<uint2*>(dkb + 8 * g4 + 4 * hh) = uint2{pack_bf2(dk[4 * g4] * sc, dk[4 * g4 + 1] * sc), pack_bf2(dk[4 * g4 + 2] * sc, dk[4 * g4 + 3] * sc)};
        *reinterpret_cast<uint2*>(dvb + 8 * g4 + 4 * hh) = uint2{pack_bf2(dv[4 * g4], dv[4 * g4 + 1]), pack_bf2(dv[4 * g4 + 2], dv[4 * g4 + 3])};
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < nt; e += 64 * NW) atomicAdd(&a.dtable[e * a.heads + head], tbl[e]);
}

template <bool BWD> constexpr int waves_per_wg() { return BWD ? 2 : 4; }

size_t swin_lds_bytes(bool bwd, int w) {
  const size_t nt = (size_t)(2 * w - 1) * (2 * w - 1);
  size_t f = (size_t)MAXT * BPITCH + MAXT + ((bwd ? nt : 0) + 3) / 4 * 4;
  return f * 4 + (size_t)(bwd ? 2 * 6 : 4 * 2) * TILE;
}

int swin_check(const char* who, int B, int Hf, int Wf, int heads, int w, int shift) {
  CRL_CHECK(B > 0 && heads > 0 && w > 0 && w * w <= MAXT, "%s: window %d not supported (w*w <= 64)", who, w);
  CRL_CHECK((Hf % w) == 0 && (Wf % w) == 0, "%s: feature map %dx%d not divisible by window %d", who, Hf, Wf, w);
  CRL_CHECK(shift >= 0 && shift < w, "%s: bad shift %d", who, shift);
  return 0;
}

template <bool BWD>
int swin_launch(const char* who, SwinArgs& a, hipStream_t s) {
  constexpr int NW = waves_per_wg<BWD>();
  a.nWx = a.Wf / a.w; a.nWy = a.Hf / a.w; a.nwin = a.B * a.nWy * a.nWx;
  const size_t lds = swin_lds_bytes(BWD, a.w);
  if (int rc = crl_enable_lds(reinterpret_cast<const void*>(&swin_attn_kernel<BWD, NW>), 160 * 1024, who)) return rc;
  const unsigned grid = (unsigned)(((a.nwin + NW - 1) / NW) * a.heads);
  swin_attn_kernel<BWD, NW><<<grid, 64 * NW, lds, s>>>(a);
  CRL_LAUNCH_CHECK(who);
  return 0;
}

}  // namespace

extern "C" int crl_swin_attn_fwd(const void* qkv, const float* table, void* out, int B, int Hf, int Wf, int heads, int w,
                                 int shift, float scale, void* stream) {
  if (swin_check("crl_swin_attn_fwd", B, Hf, Wf, heads, w, shift)) return -1;
  CRL_CHECK(qkv && table && out, "crl_swin_attn_fwd: null pointer");
  SwinArgs a{};
  a.qkv = (const u16*)qkv; a.table = table; a.out = (u16*)out;
  a.B = B; a.Hf = Hf; a.Wf = Wf; a.heads = heads; a.w = w; a.shift = shift; a.scale = scale;
  return swin_launch<false>("crl_swin_attn_fwd", a, as_stream(stream));
}

extern "C" int crl_swin_attn_bwd(const void* qkv, const float* table, const void* d_out, void* dqkv, float* dtable, int B,
                                 int Hf, int Wf, int heads, int w, int shift, float scale, void* stream) {
  if (swin_check("crl_swin_attn_bwd", B, Hf, Wf, heads, w, shift)) return -1;
  CRL_CHECK(qkv && table && d_out && dqkv && dtable, "crl_swin_attn_bwd: null pointer");
  SwinArgs a{};
  a.qkv = (const u16*)qkv; a.table = table; a.d_out = (const u16*)d_out; a.dqkv = (u16*)dqkv; a.dtable = dtable;
  a.B = B; a.Hf = Hf; a.Wf = Wf; a.heads = heads; a.w = w; a.shift = shift; a.scale = scale;
  return swin_launch<true>("crl_swin_attn_bwd", a, as_stream(stream));
}
