// Moment bounding box ("points2bbox", transform_method='moment') forward / backward for gfx950.
// Fuses what the reference does with ~10 separate torch ops per call
// (mmdet/models/anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py:373-388): mean and
// unbiased std of the n points' x and y at every location, half extents = std * exp(transfer),
// box = mean -/+ half extent.  One thread per (image, location); the 2n channel reads of
// neighbouring threads are contiguous, so the pass is a coalesced stream over pts.
#include "common.h"

namespace kgdet {

namespace {

struct Moments {
  float mean, stdv;
};

// mean of v_i, then the unbiased std of (v_i - mean) exactly as torch.std(pts - mean) computes it
__device__ __forceinline__ Moments moments(const float *__restrict__ base, int n, long long stride) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += base[(long long)i * stride];
  const float mean = s / (float)n;
  float s2 = 0.f;
  for (int i = 0; i < n; ++i) s2 += base[(long long)i * stride] - mean;
  const float m2 = s2 / (float)n;
  float q = 0.f;
  for (int i = 0; i < n; ++i) {
    const float d = (base[(long long)i * stride] - mean) - m2;
    q += d * d;
  }
  Moments r;
  r.mean = mean;
  r.stdv = sqrtf(q / (float)(n - 1));
  return r;
}

}  // namespace

__global__ __launch_bounds__(256) void moment_bbox_forward(const float *__restrict__ pts,
                                                           const float *__restrict__ transfer, int B, int n, int HW,
                                                           int y_first, float *__restrict__ bbox) {
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= (long long)B * HW) return;
  const int b = (int)(idx / HW), hw = (int)(idx - (long long)b * HW);
  const float *p = pts + (long long)b * 2 * n * HW + hw;
  const float *py = p + (y_first ? 0 : HW), *px = p + (y_first ? HW : 0);
  const Moments my = moments(py, n, 2LL * HW), mx = moments(px, n, 2LL * HW);
  const float half_w = mx.stdv * expf(transfer[0]);
  const float half_h = my.stdv * expf(transfer[1]);
  float *o = bbox + (long long)b * 4 * HW + hw;
  o[0] = mx.mean - half_w;
  o[HW] = my.mean - half_h;
  o[2LL * HW] = mx.mean + half_w;
  o[3LL * HW] = my.mean + half_h;
}

__global__ __launch_bounds__(256) void moment_bbox_backward(const float *__restrict__ pts,
                                                            const float *__restrict__ transfer,
                                                            const float *__restrict__ grad_bbox, int B, int n, int HW,
                                                            int y_first, float *__restrict__ grad_pts,
                                                            float *__restrict__ grad_transfer) {
  __shared__ float red[2][4];
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  float gt0 = 0.f, gt1 = 0.f;
  if (idx < (long long)B * HW) {
    const int b = (int)(idx / HW), hw = (int)(idx - (long long)b * HW);
    const float *p = pts + (long long)b * 2 * n * HW + hw;
    float *gp = grad_pts + (long long)b * 2 * n * HW + hw;
    const int oy = y_first ? 0 : HW, ox = y_first ? HW : 0;
    const Moments my = moments(p + oy, n, 2LL * HW), mx = moments(p + ox, n, 2LL * HW);
    const float *g = grad_bbox + (long long)b * 4 * HW + hw;
    const float g0 = g[0], g1 = g[HW], g2 = g[2LL * HW], g3 = g[3LL * HW];
    const float e0 = expf(transfer[0]), e1 = expf(transfer[1]);
    const float d_half_w = g2 - g0, d_half_h = g3 - g1;
    gt0 = d_half_w * mx.stdv * e0;
    gt1 = d_half_h * my.stdv * e1;
    const float cx = mx.stdv > 0.f ? d_half_w * e0 / ((float)(n - 1) * mx.stdv) : 0.f;
    const float cy = my.stdv > 0.f ? d_half_h * e1 / ((float)(n - 1) * my.stdv) : 0.f;
    const float mxg = (g0 + g2) / (float)n, myg = (g1 + g3) / (float)n;
    for (int i = 0; i < n; ++i) {
      const long long o = 2LL * i * HW;
      gp[o + ox] = mxg + cx * (p[o + ox] - mx.mean);
      gp[o + oy] = myg + cy * (p[o + oy] - my.mean);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    gt0 += __shfl_xor(gt0, d);
    gt1 += __shfl_xor(gt1, d);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = gt0; red[1][wave] = gt1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(grad_transfer, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(grad_transfer + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" {

int kgdet_moment_bbox_forward(const float *pts, const float *moment_transfer, int32_t B, int32_t n_pts,
                              int32_t HW, int32_t y_first, float *bbox, void *stream) {
  KGDET_CHECK_SHAPE(B >= 0 && HW >= 0, "bad sizes");
  KGDET_CHECK_SHAPE(n_pts >= 2, "moment bbox needs at least 2 points (unbiased std)");
  if ((long long)B * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(pts && moment_transfer && bbox, "null pointer");
  const int grid = (int)(((long long)B * HW + 255) / 256);
  hipLaunchKernelGGL(moment_bbox_forward, dim3(grid), dim3(256), 0, (hipStream_t)stream, pts, moment_transfer, B,
                     n_pts, HW, y_first, bbox);
  KGDET_CHECK_LAUNCH("moment_bbox_forward");
  return KGDET_OK;
}

int kgdet_moment_bbox_backward(const float *pts, const float *moment_transfer, const float *grad_bbox, int32_t B,
                               int32_t n_pts, int32_t HW, int32_t y_first, float *grad_pts, float *grad_transfer,
                               void *stream) {
  KGDET_CHECK_SHAPE(B >= 0 && HW >= 0, "bad sizes");
  KGDET_CHECK_SHAPE(n_pts >= 2, "moment bbox needs at least 2 points (unbiased std)");
  if ((long long)B * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(pts && moment_transfer && grad_bbox && grad_pts && grad_transfer, "null pointer");
  const int grid = (int)(((long long)B * HW + 255) / 256);
  hipLaunchKernelGGL(moment_bbox_backward, dim3(grid), dim3(256), 0, (hipStream_t)stream, pts, moment_transfer,
                     grad_bbox, B, n_pts, HW, y_first, grad_pts, grad_transfer);
  KGDET_CHECK_LAUNCH("moment_bbox_backward");
  return KGDET_OK;
}

}  // extern "C"
