// Image preprocessing on the GPU: uint8 HWC page -> ToTensor (/255) -> separable bicubic ANTIALIASED resize ->
// Normalize(mean, std) -> fp32 CHW.  Replaces the torchvision Compose the reference builds per task
// (ref: task/task_cruller_pretrain.py:132-143: ToTensor, Resize(image_size, BICUBIC, antialias=True), Normalize), i.e.
// aten's upsample_bicubic2d_aa on a float tensor: per output index the cubic (a = -0.5) filter is stretched by the
// down-scale factor, evaluated at (j + xmin - center + 0.5) / max(scale, 1), and normalised; width first, then height.
// The (xmin, xsize, weights) tables depend only on (in, out) sizes and are built on the host (pixparse_amd/data/gpu_preprocess.py).
// HBM-bound: one read of the uint8 page, one write + read of the [C, H_in, W_out] intermediate, one write of the result.
#include "common.h"

namespace {

__global__ void resize_h_kernel(const uint8_t* __restrict__ img, int H, int W, int C, const int32_t* __restrict__ xmin,
                                const int32_t* __restrict__ xsize, const float* __restrict__ wts, int kmax, float* __restrict__ tmp, int Wo) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over C * H * Wo, xo fastest
  if (idx >= (size_t)C * H * Wo) return;
  const int xo = (int)(idx % Wo);
  const int y = (int)((idx / Wo) % H);
  const int c = (int)(idx / ((size_t)Wo * H));
  const int x0 = xmin[xo], n = xsize[xo];
  const float* w = wts + (size_t)xo * kmax;
  const uint8_t* row = img + ((size_t)y * W + x0) * C + c;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) acc += w[j] * ((float)row[(size_t)j * C] * (1.0f / 255.0f));
  tmp[idx] = acc;
}

__global__ void resize_v_kernel(const float* __restrict__ tmp, int H, int Wo, int C, const int32_t* __restrict__ ymin,
                                const int32_t* __restrict__ ysize, const float* __restrict__ wts, int kmax, const float* __restrict__ mean,
                                const float* __restrict__ stdv, float* __restrict__ out, int Ho) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over C * Ho * Wo
  if (idx >= (size_t)C * Ho * Wo) return;
  const int xo = (int)(idx % Wo);
  const int yo = (int)((idx / Wo) % Ho);
  const int c = (int)(idx / ((size_t)Wo * Ho));
  const int y0 = ymin[yo], n = ysize[yo];
  const float* w = wts + (size_t)yo * kmax;
  const float* col = tmp + ((size_t)c * H + y0) * Wo + xo;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) acc += w[j] * col[(size_t)j * Wo];
  out[idx] = (acc - mean[c]) / stdv[c];
}

}  // namespace

extern "C" int crl_image_preprocess_u8(const void* img_hwc_u8, int H, int W, int C, const int32_t* xmin, const int32_t* xsize,
                                       const float* xw, int xk, const int32_t* ymin, const int32_t* ysize, const float* yw, int yk,
                                       const float* mean, const float* stdv, float* tmp, float* out_chw, int Ho, int Wo, void* stream) {
  CRL_CHECK(H > 0 && W > 0 && C > 0 && C <= 4 && Ho > 0 && Wo > 0 && xk > 0 && yk > 0, "crl_image_preprocess_u8: bad shape");
  CRL_CHECK(img_hwc_u8 && xmin && xsize && xw && ymin && ysize && yw && mean && stdv && tmp && out_chw, "crl_image_preprocess_u8: null pointer");
  hipStream_t s = as_stream(stream);
  const size_t n1 = (size_t)C * H * Wo, n2 = (size_t)C * Ho * Wo;
  resize_h_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, s>>>((const uint8_t*)img_hwc_u8, H, W, C, xmin, xsize, xw, xk, tmp, Wo);
  CRL_LAUNCH_CHECK("crl_image_preprocess_u8(h)");
  resize_v_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, s>>>(tmp, H, Wo, C, ymin, ysize, yw, yk, mean, stdv, out_chw, Ho);
  CRL_LAUNCH_CHECK("crl_image_preprocess_u8(v)");
  return 0;
}
