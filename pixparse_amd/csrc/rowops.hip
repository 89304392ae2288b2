// HBM-bound row kernels of the Cruller step: LayerNorm fwd/bwd (one wavefront per row, fp32
// statistics, no LDS in the forward), bias-gradient column sums, decoder embedding, ViT token
// assembly, im2row for the patch conv, Swin patch-merge permutation, casts.  All accesses are
// 8/16 bytes per lane and row-contiguous (guide G13); reductions over rows are two-stage and
// deterministic (per-block partials in a caller-provided workspace, no float atomics), the
// token-embedding scatter included (counting rank + segmented sums).
#include "common.h"

namespace {

constexpr int LN_MAXV = 8;  // float4 per lane kept in registers -> D <= 2048 single read

// ------------------------------------------------------------------ LayerNorm forward
// NV = float4 slots per lane; EXACT: D == NV * 256, i.e. every slot of every lane is live (the D = 1024 rows of the step: NV = 4) -- no bounds
// branches, so a row's NV loads and the gamma / beta loads are issued together before any arithmetic (round 5: with the `c < D` branches hipcc
// kept load -> wait -> use per slot at 8 slots, the pattern round 2 removed from the backward: 61 us per 49512 x 1024 launch = 5.0 TB/s).
// Same operations in the same order as the generic form: bit-identical.
template <int NV, bool EXACT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int M, int D,
                                                     float* __restrict__ y32, u16* __restrict__ y16,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  float4 xv[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (EXACT || c < D) xv[i] = *reinterpret_cast<const float4*>(xr + c);
  }
  float4 gm[NV], bt[NV];
  if constexpr (EXACT) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      gm[i] = *reinterpret_cast<const float4*>(gamma + c);
      bt[i] = *reinterpret_cast<const float4*>(beta + c);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (EXACT || c < D) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
  }
  if constexpr (!EXACT) {
    for (int c = (NV * 64 + lane) * 4; c < D; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      s += (v.x + v.y) + (v.z + v.w);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (EXACT || c < D) {
      const float a = xv[i].x - mean, b = xv[i].y - mean, cc = xv[i].z - mean, d = xv[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  if constexpr (!EXACT) {
    for (int c = (NV * 64 + lane) * 4; c < D; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  auto emit = [&](int c, const float4& v, const float4& g4, const float4& b4) {
    float4 o;
    o.x = (v.x - mean) * rstd * g4.x + b4.x; o.y = (v.y - mean) * rstd * g4.y + b4.y;
    o.z = (v.z - mean) * rstd * g4.z + b4.z; o.w = (v.w - mean) * rstd * g4.w + b4.w;
    if (y32) *reinterpret_cast<float4*>(y32 + (size_t)row * D + c) = o;
    if (y16) *reinterpret_cast<uint2*>(y16 + (size_t)row * D + c) = uint2{pack_bf2(o.x, o.y), pack_bf2(o.z, o.w)};
  };
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if constexpr (EXACT) emit(c, xv[i], gm[i], bt[i]);
    else if (c < D) emit(c, xv[i], *reinterpret_cast<const float4*>(gamma + c), *reinterpret_cast<const float4*>(beta + c));
  }
  if constexpr (!EXACT) {
    for (int c = (NV * 64 + lane) * 4; c < D; c += 256)
      emit(c, *reinterpret_cast<const float4*>(xr + c), *reinterpret_cast<const float4*>(gamma + c), *reinterpret_cast<const float4*>(beta + c));
  }
}

// ------------------------------------------------------------------ LayerNorm backward
// each wave walks rows (grid-stride); dgamma/dbeta partials stay in registers, are combined over
// the block's 4 waves through LDS and written to ws[block][2][D]; ln_bwd_reduce sums the blocks.
#ifndef LNB_MAXBLK_N
#define LNB_MAXBLK_N 512
#endif
constexpr int LNB_MAXBLK = LNB_MAXBLK_N;

// NV = float4 slots per lane (4: D <= 1024, 8: D <= 2048), COLSUM = also emit the column sums of dx16.  Both are compile-time so the
// D = 1024 rows of the step keep ~110 VGPRs (4 waves / SIMD); with 8 slots and a run-time colsum flag the kernel sat at 2 waves /
// SIMD and 48 KiB of LDS and took 148 us instead of 104 us per 49512 x 1024 launch (rocprofv3, r2 baseline).
template <int NV, bool COLSUM>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy32, const u16* __restrict__ dy16,
                                                     const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     int M, int D, float* __restrict__ dx32, int dx_acc,
                                                     u16* __restrict__ dx16, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);  // [4 waves][D]: the NP partial kinds are combined one after the other
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = gridDim.x * 4;
  constexpr int NP = COLSUM ? 3 : 2;
  // ao: column sums of the bf16 gradient this kernel EMITS (dx16) = the bias gradient of the Linear whose output
  // gradient it is -- the rows are in registers anyway, so the separate colsum pass over dx16 disappears
  float4 ag[NV], ab[NV], ao[COLSUM ? NV : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) { ag[i] = float4{0, 0, 0, 0}; ab[i] = float4{0, 0, 0, 0}; if constexpr (COLSUM) ao[i] = float4{0, 0, 0, 0}; }
  for (int row = blockIdx.x * 4 + wave; row < M; row += nw) {
    const float mean = mean_i[row], rstd = rstd_i[row];
    const size_t ro = (size_t)row * D;
    // All loads of a row are issued kind by kind BEFORE the arithmetic (fp32 dy, bf16 dy, x; gamma stays cached): the
    // wave then waits for one round trip per kind instead of one per 1-KB slot (the per-slot form measured 160 us for
    // 49512 x 1024 -- latency-bound at 2 waves per SIMD although it looked like 5 TB/s).
    float4 gv[NV], xh[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) gv[i] = float4{0, 0, 0, 0};
    if (dy32) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) gv[i] = *reinterpret_cast<const float4*>(dy32 + ro + c);
      }
    }
    uint2 hb[NV];
    if (dy16) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        hb[i] = uint2{0u, 0u};
        if (c < D) hb[i] = *reinterpret_cast<const uint2*>(dy16 + ro + c);
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      xh[i] = float4{mean, mean, mean, mean};
      if (c < D) xh[i] = *reinterpret_cast<const float4*>(x + ro + c);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        float4 d = gv[i];
        if (dy16) {
          const uint2 h = hb[i];
          d.x += bf2f(h.x & 0xffff); d.y += bf2f(h.x >> 16); d.z += bf2f(h.y & 0xffff); d.w += bf2f(h.y >> 16);
        }
        const float4 xv = xh[i];
        const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
        float4 h{(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
        ag[i].x += d.x * h.x; ag[i].y += d.y * h.y; ag[i].z += d.z * h.z; ag[i].w += d.w * h.w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        float4 g{d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w};
        s1 += (g.x + g.y) + (g.z + g.w);
        s2 += (g.x * h.x + g.y * h.y) + (g.z * h.z + g.w * h.w);
        gv[i] = g; xh[i] = h;
      }
    }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
    float4 old[NV];
    if (dx32 && dx_acc) {   // the accumulated residual gradient: one batch of loads, not a read-modify-write per slot
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        old[i] = float4{0, 0, 0, 0};
        if (c < D) old[i] = *reinterpret_cast<const float4*>(dx32 + ro + c);
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        float4 o{rstd * (gv[i].x - c1 - xh[i].x * c2), rstd * (gv[i].y - c1 - xh[i].y * c2),
                 rstd * (gv[i].z - c1 - xh[i].z * c2), rstd * (gv[i].w - c1 - xh[i].w * c2)};
        if (dx32) {
          if (dx_acc) { o.x += old[i].x; o.y += old[i].y; o.z += old[i].z; o.w += old[i].w; }
          *reinterpret_cast<float4*>(dx32 + ro + c) = o;
        }
        if (dx16) *reinterpret_cast<uint2*>(dx16 + ro + c) = uint2{pack_bf2(o.x, o.y), pack_bf2(o.z, o.w)};
        if constexpr (COLSUM) { ao[i].x += round_bf(o.x); ao[i].y += round_bf(o.y); ao[i].z += round_bf(o.z); ao[i].w += round_bf(o.w); }
      }
    }
  }
  // block combine: one partial kind at a time through a [4 waves][D] LDS buffer (16 KiB at D = 1024)
#pragma unroll
  for (int which = 0; which < NP; ++which) {
    if (which) __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        float4 v = ag[i];
        if (which == 1) v = ab[i];
        if constexpr (COLSUM) { if (which == 2) v = ao[i]; }
        *reinterpret_cast<float4*>(red + wave * D + c) = v;
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256)
      ws[((size_t)blockIdx.x * NP + which) * D + c] = (red[c] + red[D + c]) + (red[2 * D + c] + red[3 * D + c]);
  }
}

// sum of the per-block partials: 16 columns per workgroup x 16 groups of partial blocks (128+ workgroups for D = 1024
// instead of 32), four independent accumulators per thread so the strided loads overlap, fixed-order LDS combine
__global__ __launch_bounds__(256) void ln_bwd_reduce(const float* __restrict__ ws, int nblk, int D, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int acc, float* __restrict__ dcol, int NP) {
  __shared__ float red[16][16];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < NP * D) {
    const float* p = ws + idx;
    const size_t st = (size_t)NP * D;
    int b = rg;
    for (; b + 48 < nblk; b += 64) {
      s0 += p[(size_t)b * st]; s1 += p[(size_t)(b + 16) * st]; s2 += p[(size_t)(b + 32) * st]; s3 += p[(size_t)(b + 48) * st];
    }
    for (; b < nblk; b += 16) s0 += p[(size_t)b * st];
  }
  red[rg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg != 0 || idx >= NP * D) return;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += red[r][c];
  const int which = idx / D;
  float* base = which == 0 ? dgamma : (which == 1 ? dbeta : dcol);
  if (!base) return;
  float* dst = base + (idx - which * D);
  *dst = acc ? *dst + s : s;
}

// ------------------------------------------------------------------ column sums (bias grads)
constexpr int CS_SPLIT = 64;
__global__ __launch_bounds__(256) void colsum_kernel(const u16* __restrict__ X, int M, int N, int ldx, float* __restrict__ ws) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < N) {
    auto add = [&](const uint4 v) {
      a[0] += bf2f(v.x & 0xffff); a[1] += bf2f(v.x >> 16); a[2] += bf2f(v.y & 0xffff); a[3] += bf2f(v.y >> 16);
      a[4] += bf2f(v.z & 0xffff); a[5] += bf2f(v.z >> 16); a[6] += bf2f(v.w & 0xffff); a[7] += bf2f(v.w >> 16);
    };
    constexpr int S = CS_SPLIT * 8;
    const u16* p = X + c0;
    int r = blockIdx.y * 8 + rl;
    for (; r + 3 * S < M; r += 4 * S) {   // four independent 16-B loads in flight per thread (one per iteration left the kernel latency-bound)
      const uint4 v0 = *reinterpret_cast<const uint4*>(p + (size_t)r * ldx);
      const uint4 v1 = *reinterpret_cast<const uint4*>(p + (size_t)(r + S) * ldx);
      const uint4 v2 = *reinterpret_cast<const uint4*>(p + (size_t)(r + 2 * S) * ldx);
      const uint4 v3 = *reinterpret_cast<const uint4*>(p + (size_t)(r + 3 * S) * ldx);
      add(v0); add(v1); add(v2); add(v3);
    }
    for (; r < M; r += S) add(*reinterpret_cast<const uint4*>(p + (size_t)r * ldx));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = a[j];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < N) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][c];
    ws[(size_t)blockIdx.y * N + blockIdx.x * 256 + c] = s;
  }
}
__global__ void colsum_reduce(const float* __restrict__ ws, int N, float* __restrict__ out, int acc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int b = 0; b < CS_SPLIT; ++b) s += ws[(size_t)b * N + c];
  out[c] = acc ? out[c] + s : s;
}

// ------------------------------------------------------------------ decoder embedding
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                        const float* __restrict__ pos, float* __restrict__ out,
                                                        int BT, int T, int D, int off, const int* __restrict__ step, int vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BT) return;
  const int64_t id = ids[row];
  const int t = row % T + (step ? *step : 0);      // generation: the position comes from the device-side step counter
  if (id < 0 || id >= vocab) {                     // torch raises a device assert: here the row turns NaN (visible in the loss), nothing is read
    const float nanv = __builtin_nanf("");
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<float4*>(out + (size_t)row * D + c) = float4{nanv, nanv, nanv, nanv};
    return;
  }
  const float* tr = tok + (size_t)id * D;
  const float* pr = pos + (size_t)(t + off) * D;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(tr + c);
    const float4 b = *reinterpret_cast<const float4*>(pr + c);
    *reinterpret_cast<float4*>(out + (size_t)row * D + c) = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
  }
}
// Token-embedding gradient: dtok[ids[r]] += dt[r] for every row r -- DETERMINISTIC, no float atomics (the fp32
// atomic scatter made the tied embedding / LM-head gradient depend on the order in which duplicate tokens arrived: every
// sequence starts with the same prompt token, so (g + a) + b vs (g + b) + a differed in the last bit from run to run).
//   1. embed_rank_kernel: counting rank.  Row r gets its position in the stable sort of the ids,
//        pos = #{j : ids[j] < ids[r]} + #{j < r : ids[j] == ids[r]},  plus its offset inside its run of equal ids and the
//        run length (n^2 / 256 compares per workgroup over the L2-resident id list: microseconds for n = 8184).
//   2. embed_seg_partial_kernel: the row at offset 0, 32, 64, ... of a run sums up to 32 rows of the run in sorted (=
//        original) order -- independent loads, ordered adds.  Runs of <= 32 rows are added to dtok directly (one writer
//        per id); longer runs (padding: thousands of rows) leave one partial row per chunk in the workspace.
//   3. embed_seg_final_kernel: the first row of a long run adds its chunk partials in chunk order.
// Bounded serial work per workgroup (32 rows), no dependence on arrival order, out-of-vocabulary ids skipped.
constexpr int EMB_CHUNK = 32;
__global__ __launch_bounds__(256) void embed_rank_kernel(const int64_t* __restrict__ ids, int n, int vocab, int* __restrict__ sorted_row,
                                                         int* __restrict__ run_off, int* __restrict__ run_len, int* __restrict__ sorted_id) {
  __shared__ int sh[3][4];
  const int r = blockIdx.x;
  const int64_t id = ids[r];
  int less = 0, eq_before = 0, eq = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const int64_t v = ids[j];
    less += v < id;
    eq += v == id;
    eq_before += (v == id) && (j < r);
  }
  less = (int)wave_sum((float)less); eq = (int)wave_sum((float)eq); eq_before = (int)wave_sum((float)eq_before);   // counts < 2^24: exact in fp32
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = less; sh[1][w] = eq; sh[2][w] = eq_before; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int L = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3], E = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    const int Eb = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    const int pos = L + Eb;
    sorted_row[pos] = r;
    run_off[pos] = Eb;
    run_len[pos] = E;
    sorted_id[pos] = (id >= 0 && id < vocab) ? (int)id : -1;
  }
}
__global__ __launch_bounds__(256) void embed_seg_partial_kernel(const float* __restrict__ dt, float* __restrict__ dtok, int n, int D,
                                                                const int* __restrict__ sorted_row, const int* __restrict__ run_off,
                                                                const int* __restrict__ run_len, const int* __restrict__ sorted_id,
                                                                float* __restrict__ partial) {
  const int p = blockIdx.x;
  const int off = run_off[p], len = run_len[p], id = sorted_id[p];
  if (id < 0 || (off % EMB_CHUNK) != 0) return;
  const int cnt = min(EMB_CHUNK, len - off);
  for (int c = threadIdx.x * 4; c < D; c += 1024) {
    float4 s{0.f, 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < cnt; q0 += 8) {          // 8 independent 16-byte loads in flight, added in order
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = float4{0.f, 0.f, 0.f, 0.f};
        if (q0 + u < cnt) v[u] = *reinterpret_cast<const float4*>(dt + (size_t)sorted_row[p + q0 + u] * D + c);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    if (len <= EMB_CHUNK) {
      float4* d = reinterpret_cast<float4*>(dtok + (size_t)id * D + c);
      const float4 o = *d;
      *d = float4{o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w};
    } else {
      *reinterpret_cast<float4*>(partial + (size_t)p * D + c) = s;
    }
  }
}
__global__ __launch_bounds__(256) void embed_seg_final_kernel(float* __restrict__ dtok, int D, const int* __restrict__ run_off,
                                                              const int* __restrict__ run_len, const int* __restrict__ sorted_id,
                                                              const float* __restrict__ partial) {
  const int p = blockIdx.x;
  const int len = run_len[p], id = sorted_id[p];
  if (id < 0 || run_off[p] != 0 || len <= EMB_CHUNK) return;
  for (int c = threadIdx.x * 4; c < D; c += 1024) {
    float4* d = reinterpret_cast<float4*>(dtok + (size_t)id * D + c);
    float4 s = *d;
    for (int q = 0; q < len; q += EMB_CHUNK) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)(p + q) * D + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *d = s;
  }
}
__global__ void embed_bwd_pos_kernel(const float* __restrict__ dt, float* __restrict__ dpos, int acc, int B, int T, int D, int off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over T*D
  if (idx >= T * D) return;
  const int t = idx / D, c = idx - t * D;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dt[((size_t)b * T + t) * D + c];
  float* p = dpos + (size_t)(t + off) * D + c;
  *p = acc ? *p + s : s;
}

// ------------------------------------------------------------------ ViT token assembly
__global__ void vit_tokens_fwd_kernel(const u16* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                      float* __restrict__ x, int B, int Np, int D) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*(Np+1)*D/4
  const int D4 = D / 4;
  const size_t total = (size_t)B * (Np + 1) * D4;
  if (idx >= total) return;
  const int c = (int)(idx % D4) * 4;
  const size_t rt = idx / D4;
  const int tkn = (int)(rt % (Np + 1));
  const int b = (int)(rt / (Np + 1));
  const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)tkn * D + c);
  float4 v;
  if (tkn == 0) {
    v = *reinterpret_cast<const float4*>(cls + c);
  } else {
    const uint2 h = *reinterpret_cast<const uint2*>(patch + ((size_t)b * Np + tkn - 1) * D + c);
    v = float4{bf2f(h.x & 0xffff), bf2f(h.x >> 16), bf2f(h.y & 0xffff), bf2f(h.y >> 16)};
  }
  *reinterpret_cast<float4*>(x + rt * D + c) = float4{v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w};
}
__global__ void vit_tokens_bwd_kernel(const float* __restrict__ dx, u16* __restrict__ dpatch, float* __restrict__ dcls,
                                      float* __restrict__ dpos, int acc, int B, int Np, int D) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over (Np+1)*D/4
  const int D4 = D / 4;
  if (idx >= (size_t)(Np + 1) * D4) return;
  const int c = (int)(idx % D4) * 4;
  const int tkn = (int)(idx / D4);
  float4 s{0, 0, 0, 0};
  for (int b = 0; b < B; ++b) {
    const float4 v = *reinterpret_cast<const float4*>(dx + ((size_t)b * (Np + 1) + tkn) * D + c);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    if (tkn > 0)
      *reinterpret_cast<uint2*>(dpatch + ((size_t)b * Np + tkn - 1) * D + c) = uint2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
  }
  float4* pp = reinterpret_cast<float4*>(dpos + (size_t)tkn * D + c);
  if (acc) { const float4 o = *pp; *pp = float4{o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w}; } else *pp = s;
  if (tkn == 0) {
    float4* pc = reinterpret_cast<float4*>(dcls + c);
    if (acc) { const float4 o = *pc; *pc = float4{o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w}; } else *pc = s;
  }
}

// ------------------------------------------------------------------ im2row (stride == kernel conv)
__global__ void im2row_kernel(const float* __restrict__ img, u16* __restrict__ out, int B, int C, int H, int W, int P,
                              int gh, int gw, int Kp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over rows * Kp/8
  const int K8 = Kp / 8;
  const size_t total = (size_t)B * gh * gw * K8;
  if (idx >= total) return;
  const int k0 = (int)(idx % K8) * 8;
  const size_t row = idx / K8;
  const int gx = (int)(row % gw), gy = (int)((row / gw) % gh), b = (int)(row / ((size_t)gw * gh));
  const int Kreal = C * P * P;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    if (k < Kreal) {
      const int c = k / (P * P), r = k - c * P * P, ph = r / P, pw = r - ph * P;
      v[j] = img[(((size_t)b * C + c) * H + gy * P + ph) * W + gx * P + pw];
    } else v[j] = 0.f;
  }
  *reinterpret_cast<uint4*>(out + row * Kp + k0) = uint4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}

// ------------------------------------------------------------------ Swin patch merging permutation
template <bool FWD>
__global__ void patch_merge_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int Hf, int Wf, int C) {
  // merged element (b, i, j, [dw][dh][c]) <-> x[b, 2i+dh, 2j+dw, c]; concat order (0,0),(1,0),(0,1),(1,1) over (dh,dw)
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*Hf*Wf*C/4 (elements of x)
  const int C4 = C / 4;
  const size_t total = (size_t)B * Hf * Wf * C4;
  if (idx >= total) return;
  const int c = (int)(idx % C4) * 4;
  size_t r = idx / C4;
  const int xw = (int)(r % Wf); r /= Wf;
  const int xh = (int)(r % Hf);
  const int b = (int)(r / Hf);
  const int i = xh >> 1, dh = xh & 1, j = xw >> 1, dw = xw & 1;
  const size_t xo = (((size_t)b * Hf + xh) * Wf + xw) * C + c;
  const size_t yo = ((((size_t)b * (Hf / 2) + i) * (Wf / 2) + j) * 4 + (dw * 2 + dh)) * C + c;
  if (FWD) *reinterpret_cast<float4*>(dst + yo) = *reinterpret_cast<const float4*>(src + xo);
  else *reinterpret_cast<float4*>(dst + xo) = *reinterpret_cast<const float4*>(src + yo);
}

// ------------------------------------------------------------------ casts
__global__ void cast_kernel(const float* __restrict__ s, u16* __restrict__ d, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(s + i);
    *reinterpret_cast<uint2*>(d + i) = uint2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
  } else {
    for (size_t j = i; j < n; ++j) d[j] = f2bf(s[j]);
  }
}
__global__ void cast_pad_kernel(const float* __restrict__ s, u16* __restrict__ d, size_t R, int C, int Cp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * (size_t)Cp) return;
  const int c = (int)(idx % Cp);
  const size_t r = idx / Cp;
  d[idx] = c < C ? f2bf(s[r * C + c]) : (u16)0;
}
__global__ void add_bf16_kernel(const u16* __restrict__ x, float* __restrict__ y, size_t n, int acc) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const uint2 h = *reinterpret_cast<const uint2*>(x + i);
    float4 v{bf2f(h.x & 0xffff), bf2f(h.x >> 16), bf2f(h.y & 0xffff), bf2f(h.y >> 16)};
    float4* p = reinterpret_cast<float4*>(y + i);
    if (acc) { const float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *p = v;
  } else {
    for (size_t j = i; j < n; ++j) y[j] = (acc ? y[j] : 0.f) + bf2f(x[j]);
  }
}

inline unsigned blocks_for(size_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int crl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int64_t M, int64_t D,
                                 float* y_f32, void* y_bf16, float* mean, float* rstd, void* stream) {
  CRL_CHECK(M > 0 && D > 0 && (D % 4) == 0, "crl_layernorm_fwd: bad shape %lld x %lld (D %% 4)", (long long)M, (long long)D);
  CRL_CHECK(x && gamma && beta && mean && rstd && (y_f32 || y_bf16), "crl_layernorm_fwd: null pointer");
  if (D == 1024) ln_fwd_kernel<4, true><<<blocks_for(M, 4), 256, 0, as_stream(stream)>>>(x, gamma, beta, eps, (int)M, (int)D, y_f32, (u16*)y_bf16, mean, rstd);
  else ln_fwd_kernel<LN_MAXV, false><<<blocks_for(M, 4), 256, 0, as_stream(stream)>>>(x, gamma, beta, eps, (int)M, (int)D, y_f32, (u16*)y_bf16, mean, rstd);
  CRL_LAUNCH_CHECK("crl_layernorm_fwd");
  return 0;
}

extern "C" size_t crl_layernorm_bwd_ws_bytes(int64_t D) { return (size_t)LNB_MAXBLK * 3 * D * sizeof(float); }

extern "C" int crl_layernorm_bwd(const float* dy_f32, const void* dy_bf16, const float* x, const float* gamma,
                                 const float* mean, const float* rstd, int64_t M, int64_t D, float* dx_f32,
                                 int dx_accumulate, void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, int acc_wgrad,
                                 void* ws, void* stream) {
  CRL_CHECK(M > 0 && D > 0 && (D % 4) == 0 && D <= LN_MAXV * 256, "crl_layernorm_bwd: bad shape %lld x %lld (D %% 4, D <= 2048)", (long long)M, (long long)D);
  CRL_CHECK((dy_f32 || dy_bf16) && x && gamma && mean && rstd && ws && (dx_f32 || dx_bf16), "crl_layernorm_bwd: null pointer");
  const int nblk = (int)(blocks_for(M, 4) < (unsigned)LNB_MAXBLK ? blocks_for(M, 4) : LNB_MAXBLK);
  hipStream_t s = as_stream(stream);
  const int np = dx_colsum ? 3 : 2;
  const size_t lds = 4 * D * sizeof(float);
#define LNB_LAUNCH(NV, CS) ln_bwd_kernel<NV, CS><<<nblk, 256, lds, s>>>(dy_f32, (const u16*)dy_bf16, x, gamma, mean, rstd, (int)M, (int)D, \
                                                                     dx_f32, dx_accumulate, (u16*)dx_bf16, (float*)ws)
  if (D <= 1024) { if (dx_colsum) LNB_LAUNCH(4, true); else LNB_LAUNCH(4, false); }
  else { if (dx_colsum) LNB_LAUNCH(8, true); else LNB_LAUNCH(8, false); }
#undef LNB_LAUNCH
  CRL_LAUNCH_CHECK("crl_layernorm_bwd");
  if (dgamma || dbeta || dx_colsum) {
    ln_bwd_reduce<<<blocks_for(np * D, 16), 256, 0, s>>>((const float*)ws, nblk, (int)D, dgamma, dbeta, acc_wgrad, dx_colsum, np);
    CRL_LAUNCH_CHECK("crl_layernorm_bwd(reduce)");
  }
  return 0;
}

extern "C" size_t crl_colsum_ws_bytes(int64_t N) { return (size_t)CS_SPLIT * N * sizeof(float); }

extern "C" int crl_colsum_bf16(const void* X, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate, void* ws, void* stream) {
  CRL_CHECK(M > 0 && N > 0 && (N % 8) == 0 && (ldx % 8) == 0, "crl_colsum_bf16: bad shape");
  CRL_CHECK(X && out && ws, "crl_colsum_bf16: null pointer");
  hipStream_t s = as_stream(stream);
  colsum_kernel<<<dim3(blocks_for(N, 256), CS_SPLIT), 256, 0, s>>>((const u16*)X, (int)M, (int)N, (int)ldx, (float*)ws);
  CRL_LAUNCH_CHECK("crl_colsum_bf16");
  colsum_reduce<<<blocks_for(N, 256), 256, 0, s>>>((const float*)ws, (int)N, out, accumulate);
  CRL_LAUNCH_CHECK("crl_colsum_bf16(reduce)");
  return 0;
}

extern "C" int crl_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* out, int B, int T, int D,
                             int pos_offset, int vocab, void* stream) {
  CRL_CHECK(B > 0 && T > 0 && D > 0 && (D % 4) == 0 && vocab > 0, "crl_embed_fwd: bad shape");
  embed_fwd_kernel<<<blocks_for((size_t)B * T, 4), 256, 0, as_stream(stream)>>>(ids, tok, pos, out, B * T, T, D, pos_offset, nullptr, vocab);
  CRL_LAUNCH_CHECK("crl_embed_fwd");
  return 0;
}
extern "C" int crl_embed_decode(const int64_t* ids, const float* tok, const float* pos, float* out, int B, int D, int pos_offset,
                                int vocab, const int* step_dev, void* stream) {
  CRL_CHECK(B > 0 && D > 0 && (D % 4) == 0 && vocab > 0 && step_dev, "crl_embed_decode: bad shape / null step counter");
  embed_fwd_kernel<<<blocks_for((size_t)B, 4), 256, 0, as_stream(stream)>>>(ids, tok, pos, out, B, 1, D, pos_offset, step_dev, vocab);
  CRL_LAUNCH_CHECK("crl_embed_decode");
  return 0;
}
extern "C" size_t crl_embed_bwd_ws_bytes(int B, int T, int D) {
  const size_t n = (size_t)B * T;
  return ((4 * n * sizeof(int) + 255) / 256) * 256 + n * (size_t)D * sizeof(float);   // sort bookkeeping + one partial row per position
}

extern "C" int crl_embed_bwd(const int64_t* ids, const float* dt, float* dtok, float* dpos, int acc_pos, int B, int T, int D,
                             int pos_offset, int vocab, void* ws, size_t ws_bytes, void* stream) {
  CRL_CHECK(B > 0 && T > 0 && D > 0 && (D % 4) == 0 && vocab > 0, "crl_embed_bwd: bad shape");
  CRL_CHECK(ws && ws_bytes >= crl_embed_bwd_ws_bytes(B, T, D), "crl_embed_bwd: workspace too small (need %zu bytes)", crl_embed_bwd_ws_bytes(B, T, D));
  CRL_CHECK(((uintptr_t)ws % 16) == 0 && ((uintptr_t)dt % 16) == 0 && ((uintptr_t)dtok % 16) == 0, "crl_embed_bwd: 16-byte alignment required");
  hipStream_t s = as_stream(stream);
  const int n = B * T;
  int* sorted_row = (int*)ws;
  int* run_off = sorted_row + n;
  int* run_len = run_off + n;
  int* sorted_id = run_len + n;
  float* partial = (float*)((char*)ws + ((4 * (size_t)n * sizeof(int) + 255) / 256) * 256);
  embed_rank_kernel<<<n, 256, 0, s>>>(ids, n, vocab, sorted_row, run_off, run_len, sorted_id);
  CRL_LAUNCH_CHECK("crl_embed_bwd(rank)");
  embed_seg_partial_kernel<<<n, 256, 0, s>>>(dt, dtok, n, D, sorted_row, run_off, run_len, sorted_id, partial);
  CRL_LAUNCH_CHECK("crl_embed_bwd(partial)");
  embed_seg_final_kernel<<<n, 256, 0, s>>>(dtok, D, run_off, run_len, sorted_id, partial);
  CRL_LAUNCH_CHECK("crl_embed_bwd(final)");
  embed_bwd_pos_kernel<<<blocks_for((size_t)T * D, 256), 256, 0, s>>>(dt, dpos, acc_pos, B, T, D, pos_offset);
  CRL_LAUNCH_CHECK("crl_embed_bwd(pos)");
  return 0;
}

extern "C" int crl_vit_tokens_fwd(const void* patch_bf16, const float* cls, const float* pos, float* x, int B, int Np, int D, void* stream) {
  CRL_CHECK(B > 0 && Np > 0 && (D % 4) == 0, "crl_vit_tokens_fwd: bad shape");
  vit_tokens_fwd_kernel<<<blocks_for((size_t)B * (Np + 1) * (D / 4), 256), 256, 0, as_stream(stream)>>>((const u16*)patch_bf16, cls, pos, x, B, Np, D);
  CRL_LAUNCH_CHECK("crl_vit_tokens_fwd");
  return 0;
}
extern "C" int crl_vit_tokens_bwd(const float* dx, void* dpatch_bf16, float* dcls, float* dpos, int acc, int B, int Np, int D, void* stream) {
  CRL_CHECK(B > 0 && Np > 0 && (D % 4) == 0, "crl_vit_tokens_bwd: bad shape");
  vit_tokens_bwd_kernel<<<blocks_for((size_t)(Np + 1) * (D / 4), 256), 256, 0, as_stream(stream)>>>(dx, (u16*)dpatch_bf16, dcls, dpos, acc, B, Np, D);
  CRL_LAUNCH_CHECK("crl_vit_tokens_bwd");
  return 0;
}

extern "C" int crl_im2row(const float* image, void* patches, int B, int C, int H, int W, int P, int gh, int gw, int Kp, void* stream) {
  CRL_CHECK(B > 0 && C > 0 && P > 0 && gh * P <= H && gw * P <= W && (Kp % 8) == 0 && Kp >= C * P * P, "crl_im2row: bad shape");
  im2row_kernel<<<blocks_for((size_t)B * gh * gw * (Kp / 8), 256), 256, 0, as_stream(stream)>>>(image, (u16*)patches, B, C, H, W, P, gh, gw, Kp);
  CRL_LAUNCH_CHECK("crl_im2row");
  return 0;
}

extern "C" int crl_patch_merge_fwd(const float* x, float* y, int B, int Hf, int Wf, int C, void* stream) {
  CRL_CHECK((Hf % 2) == 0 && (Wf % 2) == 0 && (C % 4) == 0, "crl_patch_merge_fwd: bad shape");
  patch_merge_kernel<true><<<blocks_for((size_t)B * Hf * Wf * (C / 4), 256), 256, 0, as_stream(stream)>>>(x, y, B, Hf, Wf, C);
  CRL_LAUNCH_CHECK("crl_patch_merge_fwd");
  return 0;
}
extern "C" int crl_patch_merge_bwd(const float* dy, float* dx, int B, int Hf, int Wf, int C, void* stream) {
  CRL_CHECK((Hf % 2) == 0 && (Wf % 2) == 0 && (C % 4) == 0, "crl_patch_merge_bwd: bad shape");
  patch_merge_kernel<false><<<blocks_for((size_t)B * Hf * Wf * (C / 4), 256), 256, 0, as_stream(stream)>>>(dy, dx, B, Hf, Wf, C);
  CRL_LAUNCH_CHECK("crl_patch_merge_bwd");
  return 0;
}

extern "C" int crl_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  CRL_CHECK(n > 0 && src && dst, "crl_cast_bf16: bad args");
  cast_kernel<<<blocks_for((size_t)(n + 3) / 4, 256), 256, 0, as_stream(stream)>>>(src, (u16*)dst, (size_t)n);
  CRL_LAUNCH_CHECK("crl_cast_bf16");
  return 0;
}
extern "C" int crl_cast_pad_bf16(const float* src, void* dst, int64_t R, int64_t C, int64_t Cp, void* stream) {
  CRL_CHECK(R > 0 && C > 0 && Cp >= C, "crl_cast_pad_bf16: bad args");
  cast_pad_kernel<<<blocks_for((size_t)R * Cp, 256), 256, 0, as_stream(stream)>>>(src, (u16*)dst, (size_t)R, (int)C, (int)Cp);
  CRL_LAUNCH_CHECK("crl_cast_pad_bf16");
  return 0;
}
extern "C" int crl_add_bf16_to_f32(const void* x_bf16, float* y, int64_t n, int accumulate, void* stream) {
  CRL_CHECK(n > 0 && x_bf16 && y, "crl_add_bf16_to_f32: bad args");
  add_bf16_kernel<<<blocks_for((size_t)(n + 3) / 4, 256), 256, 0, as_stream(stream)>>>((const u16*)x_bf16, y, (size_t)n, accumulate);
  CRL_LAUNCH_CHECK("crl_add_bf16_to_f32");
  return 0;
}
