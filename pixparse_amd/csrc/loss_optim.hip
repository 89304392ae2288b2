// Shifted-token cross-entropy (fp32 log-softmax over bf16 logits, ignore_index = -100, mean over
// valid rows, gradient written in place) and the optimiser tail over the flat fp32 arenas:
// GradScaler unscale + inf check + global-norm clip coefficient (device-resident, no host sync)
// and AdamW fused with zero_grad and the bf16 weight-shadow refresh.  All HBM-bound:
// 16-byte accesses, deterministic two-stage reductions.
#include "common.h"

namespace {

__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, sh[i]);
  return t;
}

__global__ __launch_bounds__(256) void ce_count_kernel(const int64_t* __restrict__ target, int M, int32_t* __restrict__ n_valid) {
  __shared__ float sh[4];
  float c = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) c += (target[i] != -100) ? 1.f : 0.f;   // like torch: every non-ignored row counts
  const float t = block_reduce_sum(c, sh);
  if (threadIdx.x == 0) *n_valid = (int32_t)(t + 0.5f);
}

// one workgroup per row: pass 1 online (max, sum exp) over the bf16 logits, pass 2 writes the gradient
__global__ __launch_bounds__(256) void ce_row_kernel(const u16* __restrict__ logits, int ldl, const int64_t* __restrict__ target,
                                                     int V, float grad_mul, const float* __restrict__ grad_mul_dev,
                                                     const int32_t* __restrict__ n_valid,
                                                     float* __restrict__ row_loss, u16* __restrict__ dlogits) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  const int64_t tgt = target[row];
  const u16* lr = logits + (size_t)row * ldl;
  u16* dr = dlogits + (size_t)row * ldl;
  const int nch = ldl / 8;
  // a target outside [0, V) that is not the ignore index is a caller bug (torch raises a device assert): the row's loss
  // becomes NaN -- the step's loss and the optimiser's inf/nan check both show it -- and nothing is read out of bounds
  const bool invalid = tgt != -100 && (tgt < 0 || tgt >= V);
  if (tgt == -100 || invalid) {
    if (threadIdx.x == 0) row_loss[row] = invalid ? __builtin_nanf("") : 0.f;
    for (int ch = threadIdx.x; ch < nch; ch += 256) *reinterpret_cast<uint4*>(dr + ch * 8) = uint4{0, 0, 0, 0};
    return;
  }
  const float tgt_logit = bf2f(lr[tgt]);  // read before any thread can overwrite it (dlogits may alias logits)
  float mx = -INFINITY, sm = 0.f;
  for (int ch = threadIdx.x; ch < nch; ch += 256) {
    const uint4 v = *reinterpret_cast<const uint4*>(lr + ch * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
    float cm = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = bf2f((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffff));
      f[j] = (ch * 8 + j < V) ? x : -INFINITY;
      cm = fmaxf(cm, f[j]);
    }
    if (cm == -INFINITY) continue;  // chunk entirely in the padded columns
    if (cm > mx) { sm *= __expf(mx - cm); mx = cm; }
#pragma unroll
    for (int j = 0; j < 8; ++j) sm += __expf(f[j] - mx);
  }
  const float gmx = block_reduce_max(mx, sh);
  sm *= (mx == -INFINITY) ? 0.f : __expf(mx - gmx);
  const float gsm = block_reduce_sum(sm, sh);
  const float lse = gmx + __logf(gsm);
  if (threadIdx.x == 0) row_loss[row] = lse - tgt_logit;
  const float gm = grad_mul * (grad_mul_dev ? *grad_mul_dev : 1.f) / (float)(*n_valid);
  for (int ch = threadIdx.x; ch < nch; ch += 256) {
    const uint4 v = *reinterpret_cast<const uint4*>(lr + ch * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = ch * 8 + j;
      const float x = bf2f((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffff));
      float p = (col < V) ? __expf(x - lse) : 0.f;
      if (col == (int)tgt) p -= 1.f;
      o[j] = p * gm;
    }
    *reinterpret_cast<uint4*>(dr + ch * 8) = uint4{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
  }
}

__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ row_loss, int M, const int32_t* __restrict__ n_valid,
                                                        float loss_mul, float* __restrict__ loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) s += row_loss[i];
  const float t = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) *loss = t / (float)(*n_valid) * loss_mul;  // 0/0 = NaN like torch when every target is ignored
}

// ------------------------------------------------------------------ grad norm / clip coefficient
constexpr int GN_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)GN_BLOCKS * 256) {
    const float4 v = *reinterpret_cast<const float4*>(g + i * 4);
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  const float t = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
// SCALED: the loss scale lives in state[4] (GradScaler on the device): inv_scale = 1 / (state[4] * grad_divisor), and after
// the inf check the scale / growth tracker are updated exactly like torch's _amp_update_scale_ -- the next step's
// cross-entropy reads the new scale from the same word, so an overflow costs exactly one skipped step and no host sync.
template <bool SCALED>
__global__ __launch_bounds__(256) void gradnorm_finish_kernel(const float* __restrict__ partial, float max_norm, float inv_scale_or_div,
                                                              float growth, float backoff, float interval, float* __restrict__ state) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < GN_BLOCKS; i += 256) s += partial[i];
  const float t = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) {
    const bool bad = !(t == t) || t == INFINITY;
    const float inv_scale = SCALED ? 1.f / (state[4] * inv_scale_or_div) : inv_scale_or_div;
    const float norm = sqrtf(t) * inv_scale;
    float coef = inv_scale;
    if (max_norm > 0.f) coef *= fminf(1.f, max_norm / (norm + 1e-6f));
    state[0] = norm;
    state[1] = bad ? 0.f : coef;
    state[2] = bad ? 1.f : 0.f;
    if (!bad) {                         // optimiser steps actually taken (torch: state['step'] is not advanced on a skipped step)
      state[3] += 1.f;                  // fp32 mirror (exact below 2^24)
      reinterpret_cast<uint32_t*>(state)[11] += 1u;   // the exact count the bias corrections are computed from
    }
    if (SCALED) {
      if (bad) { state[4] *= backoff; state[5] = 0.f; }
      else {
        state[5] += 1.f;
        if (state[5] >= interval) { state[4] *= growth; state[5] = 0.f; }
      }
    }
  }
}

// One thread: what the host used to compute per step and pass as launch arguments (learning rate, bias corrections), computed on the
// device so that every launch of a train step has the same arguments from step to step (the step can be replayed from a hipGraph):
//   state[8]  = learning rate of this update = timm CosineLRScheduler closed form at u = updates attempted so far (state[7]):
//               u < warmup_t: warmup_lr_init + u (base - warmup_lr_init) / warmup_t;  u < t_initial: lr_min + (base - lr_min)(1 + cos(pi u / t_initial)) / 2;
//               else lr_min;   t_initial <= 0: constant base_lr
//   state[9]  = 1 - beta1^t,  state[10] = 1 / sqrt(1 - beta2^t)   with t = state[3] = steps taken including this one (double precision:
//               1 - beta2^t cancels badly in fp32 for small t)
//   state[7] += 1
// The counts themselves are the integer words 11 (steps taken) and 12 (updates attempted) of the state vector; words 3 and 7 are their
// fp32 mirrors for host-side readers.
__global__ void optim_prepare_kernel(float* __restrict__ state, float base_lr, float warmup_lr_init, float lr_min, int warmup_t, int t_initial,
                                     float beta1, float beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t* counts = reinterpret_cast<uint32_t*>(state);     // words 11 / 12: exact integer counts (the fp32 mirrors 3 / 7 stop at 2^24)
  const double u = (double)counts[12];
  double lr = (double)base_lr;
  if (t_initial > 0) {
    if (u < (double)warmup_t) lr = (double)warmup_lr_init + u * ((double)base_lr - (double)warmup_lr_init) / (double)warmup_t;
    else if (u < (double)t_initial) lr = (double)lr_min + 0.5 * ((double)base_lr - (double)lr_min) * (1.0 + cos(3.14159265358979323846 * u / (double)t_initial));
    else lr = (double)lr_min;
  }
  const double t = fmax((double)counts[11], 1.0);
  state[8] = (float)lr;
  state[9] = (float)(1.0 - pow((double)beta1, t));
  state[10] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
  state[7] = (float)(u + 1.0);
  counts[12] += 1u;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, u16* __restrict__ pb, size_t n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float rsqrt_bc2,
                                                    const float* __restrict__ state, int zero_grad, int dev_step) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float coef = state ? state[1] : 1.f;
  const bool skip = state ? (state[2] != 0.f) : false;
  // clip-by-value (timm dispatch_clip_grad mode 'value' = torch clip_grad_value_): |g| <= state[6] after unscaling; only with the
  // 8-float device state of the task's optimiser (dev_step), 0 = off
  const float cv = (dev_step && state[6] > 0.f) ? state[6] : INFINITY;
  if (dev_step) {                        // bias corrections (and, for lr < 0, the learning rate) prepared on the device by crl_optim_prepare
    bc1 = state[9];
    rsqrt_bc2 = state[10];
    if (lr < 0.f) lr = state[8];
  }
  if (i + 3 < n) {
    float4 pv = *reinterpret_cast<float4*>(p + i);
    if (!skip) {
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = fminf(fmaxf(gp[j] * coef, -cv), cv);
        if (wd != 0.f) pp[j] *= (1.f - lr * wd);
        mp[j] = b1 * mp[j] + (1.f - b1) * gj;
        vp[j] = b2 * vp[j] + (1.f - b2) * gj * gj;
        const float denom = sqrtf(vp[j]) * rsqrt_bc2 + eps;
        pp[j] -= (lr / bc1) * (mp[j] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = pv;
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
    }
    if (pb) *reinterpret_cast<uint2*>(pb + i) = uint2{pack_bf2(pv.x, pv.y), pack_bf2(pv.z, pv.w)};
    if (zero_grad) *reinterpret_cast<float4*>(g + i) = float4{0, 0, 0, 0};
  } else {
    for (size_t j = i; j < n; ++j) {
      if (!skip) {
        const float gj = fminf(fmaxf(g[j] * coef, -cv), cv);
        if (wd != 0.f) p[j] *= (1.f - lr * wd);
        m[j] = b1 * m[j] + (1.f - b1) * gj;
        v[j] = b2 * v[j] + (1.f - b2) * gj * gj;
        p[j] -= (lr / bc1) * (m[j] / (sqrtf(v[j]) * rsqrt_bc2 + eps));
      }
      if (pb) pb[j] = f2bf(p[j]);
      if (zero_grad) g[j] = 0.f;
    }
  }
}

}  // namespace

extern "C" int crl_cross_entropy(const void* logits, int64_t ldl, const int64_t* target, int64_t M, int V, float loss_mul,
                                 float grad_mul, const float* grad_mul_dev, float* loss, int32_t* n_valid, float* row_loss,
                                 void* dlogits, void* stream) {
  CRL_CHECK(M > 0 && V > 0 && ldl >= V && (ldl % 8) == 0, "crl_cross_entropy: bad shape M=%lld V=%d ldl=%lld", (long long)M, V, (long long)ldl);
  CRL_CHECK(logits && target && loss && n_valid && row_loss && dlogits, "crl_cross_entropy: null pointer");
  hipStream_t s = as_stream(stream);
  ce_count_kernel<<<1, 256, 0, s>>>(target, (int)M, n_valid);
  CRL_LAUNCH_CHECK("crl_cross_entropy(count)");
  ce_row_kernel<<<(unsigned)M, 256, 0, s>>>((const u16*)logits, (int)ldl, target, V, grad_mul, grad_mul_dev, n_valid, row_loss, (u16*)dlogits);
  CRL_LAUNCH_CHECK("crl_cross_entropy(rows)");
  ce_finish_kernel<<<1, 256, 0, s>>>(row_loss, (int)M, n_valid, loss_mul, loss);
  CRL_LAUNCH_CHECK("crl_cross_entropy(finish)");
  return 0;
}

extern "C" size_t crl_grad_norm_ws_bytes(void) { return GN_BLOCKS * sizeof(float); }

extern "C" int crl_grad_norm(const float* g, int64_t n, float max_norm, float inv_scale, float* state, void* ws, void* stream) {
  CRL_CHECK(n > 0 && g && state && ws, "crl_grad_norm: bad args");
  CRL_CHECK(((uintptr_t)g % 16) == 0, "crl_grad_norm: grad arena must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  sumsq_kernel<<<GN_BLOCKS, 256, 0, s>>>(g, (size_t)n, (float*)ws);
  CRL_LAUNCH_CHECK("crl_grad_norm(sumsq)");
  gradnorm_finish_kernel<false><<<1, 256, 0, s>>>((const float*)ws, max_norm, inv_scale, 0.f, 0.f, 0.f, state);
  CRL_LAUNCH_CHECK("crl_grad_norm(finish)");
  return 0;
}

extern "C" int crl_grad_norm_scaled(const float* g, int64_t n, float max_norm, float grad_divisor, float growth_factor,
                                    float backoff_factor, int growth_interval, float* state, void* ws, void* stream) {
  CRL_CHECK(n > 0 && g && state && ws, "crl_grad_norm_scaled: bad args");
  CRL_CHECK(((uintptr_t)g % 16) == 0, "crl_grad_norm_scaled: grad arena must be 16-byte aligned");
  CRL_CHECK(grad_divisor > 0.f && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0,
            "crl_grad_norm_scaled: bad scaler parameters");
  hipStream_t s = as_stream(stream);
  sumsq_kernel<<<GN_BLOCKS, 256, 0, s>>>(g, (size_t)n, (float*)ws);
  CRL_LAUNCH_CHECK("crl_grad_norm_scaled(sumsq)");
  gradnorm_finish_kernel<true><<<1, 256, 0, s>>>((const float*)ws, max_norm, grad_divisor, growth_factor, backoff_factor,
                                                 (float)growth_interval, state);
  CRL_LAUNCH_CHECK("crl_grad_norm_scaled(finish)");
  return 0;
}

extern "C" int crl_optim_prepare(float* state, float base_lr, float warmup_lr_init, float lr_min, int warmup_t, int t_initial,
                                 float beta1, float beta2, void* stream) {
  CRL_CHECK(state && base_lr >= 0.f && warmup_t >= 0 && beta1 > 0.f && beta1 < 1.f && beta2 > 0.f && beta2 < 1.f, "crl_optim_prepare: bad arguments");
  optim_prepare_kernel<<<1, 64, 0, as_stream(stream)>>>(state, base_lr, warmup_lr_init, lr_min, warmup_t, t_initial, beta1, beta2);
  CRL_LAUNCH_CHECK("crl_optim_prepare");
  return 0;
}

extern "C" int crl_adamw(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, const float* state, int zero_grad, void* stream) {
  CRL_CHECK(n > 0 && p && g && m && v && (step >= 1 || (step == 0 && state)), "crl_adamw: bad args (step >= 1, or step == 0 with the device-side state prepared by crl_optim_prepare)");
  CRL_CHECK(lr >= 0.f || step == 0, "crl_adamw: lr < 0 (= read it from state[8]) needs the device-side mode (step == 0)");
  CRL_CHECK(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0, "crl_adamw: arenas must be 16-byte aligned");
  const int hs = step >= 1 ? step : 1;
  const double bc1 = 1.0 - pow((double)beta1, hs), bc2 = 1.0 - pow((double)beta2, hs);
  const unsigned blocks = (unsigned)(((size_t)n + 1023) / 1024);
  adamw_kernel<<<blocks, 256, 0, as_stream(stream)>>>(p, g, m, v, (u16*)p_bf16, (size_t)n, lr, beta1, beta2, eps, weight_decay,
                                                      (float)bc1, (float)(1.0 / sqrt(bc2)), state, zero_grad, step == 0 ? 1 : 0);
  CRL_LAUNCH_CHECK("crl_adamw");
  return 0;
}
