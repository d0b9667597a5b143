// Sigmoid focal loss forward / backward for gfx950.
// Replaces SigmoidFocalLossForward / Backward
// (mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu:24-59, 62-97).  The reference's
// expressions mix float libm calls with double literals; the same promotions are kept so results
// differ from it only by libm ulps.  2100 x 13 logits per call: launch-latency bound, so the
// kernel is a plain grid-stride elementwise pass.
#include <float.h>

#include "common.h"

namespace kgdet {

namespace {
__device__ __forceinline__ double neg_softplus(float x) {  // -x*[x>=0] - log(1 + exp(x - 2x*[x>=0]))
  const int ge = x >= 0;
  return -1. * x * ge - logf((float)(1. + expf((float)(x - 2. * x * ge))));
}
}  // namespace

__global__ __launch_bounds__(256) void focal_forward(const float *__restrict__ logits,
                                                     const long long *__restrict__ targets, long long total,
                                                     int num_classes, float gamma, float alpha,
                                                     float *__restrict__ losses) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / num_classes;
    const int d = (int)(i - n * num_classes);
    const int t = (int)targets[n];
    const float c1 = (t == (d + 1));
    const float c2 = ((t >= 0) & (t != (d + 1)));
    const float zn = (float)(1.0 - alpha), zp = alpha;
    const float x = logits[i];
    const float p = (float)(1. / (1. + expf(-x)));
    const float term1 = powf((float)(1. - p), gamma) * logf(fmaxf(p, FLT_MIN));
    const float term2 = (float)(powf(p, gamma) * neg_softplus(x));
    float l = 0.0f;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}

__global__ __launch_bounds__(256) void focal_backward(const float *__restrict__ logits,
                                                      const long long *__restrict__ targets,
                                                      const float *__restrict__ d_losses, long long total,
                                                      int num_classes, float gamma, float alpha,
                                                      float *__restrict__ d_logits) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / num_classes;
    const int d = (int)(i - n * num_classes);
    const int t = (int)targets[n];
    const float c1 = (t == (d + 1));
    const float c2 = ((t >= 0) & (t != (d + 1)));
    const float zn = (float)(1.0 - alpha), zp = alpha;
    const float x = logits[i];
    const float p = (float)(1. / (1. + expf(-x)));
    const float term1 = (float)(powf((float)(1. - p), gamma) * (1. - p - (p * gamma * logf(fmaxf(p, FLT_MIN)))));
    const float term2 = (float)(powf(p, gamma) * (neg_softplus(x) * (1. - p) * gamma - p));
    float g = 0.0f;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    d_logits[i] = g * d_losses[i];
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" {

int kgdet_sigmoid_focal_loss_forward(const float *logits, const int64_t *targets, int64_t num,
                                     int32_t num_classes, float gamma, float alpha, float *losses, void *stream) {
  KGDET_CHECK_SHAPE(num >= 0 && num_classes > 0, "bad sizes");
  if (num == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(logits && targets && losses, "null pointer");
  const long long total = (long long)num * num_classes;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(focal_forward, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits,
                     (const long long *)targets, total, num_classes, gamma, alpha, losses);
  KGDET_CHECK_LAUNCH("focal_forward");
  return KGDET_OK;
}

int kgdet_sigmoid_focal_loss_backward(const float *logits, const int64_t *targets, const float *d_losses,
                                      int64_t num, int32_t num_classes, float gamma, float alpha, float *d_logits,
                                      void *stream) {
  KGDET_CHECK_SHAPE(num >= 0 && num_classes > 0, "bad sizes");
  if (num == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(logits && targets && d_losses && d_logits, "null pointer");
  const long long total = (long long)num * num_classes;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(focal_backward, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits,
                     (const long long *)targets, d_losses, total, num_classes, gamma, alpha, d_logits);
  KGDET_CHECK_LAUNCH("focal_backward");
  return KGDET_OK;
}

}  // extern "C"
