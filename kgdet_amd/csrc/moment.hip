// Moment bounding box ("points2bbox", transform_method='moment') forward / backward for gfx950.
// Fuses what the reference does with ~10 separate torch ops per call
// (mmdet/models/anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py:373-388): mean and
// unbiased std of the n points' x and y at every location, half extents = std * exp(transfer),
// box = mean -/+ half extent.
// Only B*H*W = 2100 locations exist, so one thread per location (83 dependent strided loads, three passes)
// is pure latency: 80-107 us per call.  Here a 1024-thread block owns 64 consecutive locations and its sixteen
// waves split the n points (wave w takes points w, w+16, ...): every load instruction is a coalesced 256-byte
// row of one channel, a thread has ~5 independent loads in flight per pass, and the three reductions
// (sum, residual of the mean, squared deviations) are combined across the waves through LDS.
#include "common.h"

namespace kgdet {

namespace {

constexpr int kLoc = 64;    // locations per block
constexpr int kSplit = 16;  // waves sharing a location's points (83 points: ~5 per wave and pass; 4 waves took 24-32 us per call)

struct Moments {
  float mean, stdv;
};

// sum over the waves' partials of this location (all threads call it)
__device__ __forceinline__ float block_sum(float v, float (*red)[kLoc], int w, int l) {
  __syncthreads();
  red[w][l] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kSplit; ++k) s += red[k][l];
  return s;
}

// mean of v_i, then the unbiased std of (v_i - mean) as torch.std(pts - mean) computes it (deviations from the
// mean OF the centred values, which removes the rounding residual of the first mean)
__device__ __forceinline__ Moments moments(const float *__restrict__ base, int n, long long stride, bool live,
                                           float (*red)[kLoc], int w, int l) {
  float s = 0.f;
  if (live)
    for (int i = w; i < n; i += kSplit) s += base[(long long)i * stride];
  const float mean = block_sum(s, red, w, l) / (float)n;
  float s2 = 0.f;
  if (live)
    for (int i = w; i < n; i += kSplit) s2 += base[(long long)i * stride] - mean;
  const float m2 = block_sum(s2, red, w, l) / (float)n;
  float q = 0.f;
  if (live)
    for (int i = w; i < n; i += kSplit) {
      const float d = (base[(long long)i * stride] - mean) - m2;
      q += d * d;
    }
  Moments r;
  r.mean = mean;
  r.stdv = sqrtf(block_sum(q, red, w, l) / (float)(n - 1));
  return r;
}

}  // namespace

__global__ __launch_bounds__(kLoc * kSplit) void moment_bbox_forward(const float *__restrict__ pts,
                                                           const float *__restrict__ transfer, int B, int n, int HW,
                                                           int y_first, float *__restrict__ bbox) {
  __shared__ float red[kSplit][kLoc];
  const int l = threadIdx.x & (kLoc - 1), w = threadIdx.x >> 6;
  const long long idx = blockIdx.x * (long long)kLoc + l;
  const bool live = idx < (long long)B * HW;
  const int b = live ? (int)(idx / HW) : 0, hw = live ? (int)(idx - (long long)b * HW) : 0;
  const float *p = pts + (long long)b * 2 * n * HW + hw;
  const float *py = p + (y_first ? 0 : HW), *px = p + (y_first ? HW : 0);
  const Moments my = moments(py, n, 2LL * HW, live, red, w, l);
  const Moments mx = moments(px, n, 2LL * HW, live, red, w, l);
  if (!live || w != 0) return;
  const float half_w = mx.stdv * expf(transfer[0]);
  const float half_h = my.stdv * expf(transfer[1]);
  float *o = bbox + (long long)b * 4 * HW + hw;
  o[0] = mx.mean - half_w;
  o[HW] = my.mean - half_h;
  o[2LL * HW] = mx.mean + half_w;
  o[3LL * HW] = my.mean + half_h;
}

__global__ __launch_bounds__(kLoc * kSplit) void moment_bbox_backward(const float *__restrict__ pts,
                                                            const float *__restrict__ transfer,
                                                            const float *__restrict__ grad_bbox, int B, int n, int HW,
                                                            int y_first, float *__restrict__ grad_pts,
                                                            float *__restrict__ partial /*[gridDim.x][2]*/) {
  __shared__ float red[kSplit][kLoc];
  const int l = threadIdx.x & (kLoc - 1), w = threadIdx.x >> 6;
  const long long idx = blockIdx.x * (long long)kLoc + l;
  const bool live = idx < (long long)B * HW;
  const int b = live ? (int)(idx / HW) : 0, hw = live ? (int)(idx - (long long)b * HW) : 0;
  const float *p = pts + (long long)b * 2 * n * HW + hw;
  float *gp = grad_pts + (long long)b * 2 * n * HW + hw;
  const int oy = y_first ? 0 : HW, ox = y_first ? HW : 0;
  const Moments my = moments(p + oy, n, 2LL * HW, live, red, w, l);
  const Moments mx = moments(p + ox, n, 2LL * HW, live, red, w, l);
  float gt0 = 0.f, gt1 = 0.f;
  if (live) {
    const float *g = grad_bbox + (long long)b * 4 * HW + hw;
    const float g0 = g[0], g1 = g[HW], g2 = g[2LL * HW], g3 = g[3LL * HW];
    const float e0 = expf(transfer[0]), e1 = expf(transfer[1]);
    const float d_half_w = g2 - g0, d_half_h = g3 - g1;
    if (w == 0) {  // one wave carries the location's share of the transfer gradient
      gt0 = d_half_w * mx.stdv * e0;
      gt1 = d_half_h * my.stdv * e1;
    }
    const float cx = mx.stdv > 0.f ? d_half_w * e0 / ((float)(n - 1) * mx.stdv) : 0.f;
    const float cy = my.stdv > 0.f ? d_half_h * e1 / ((float)(n - 1) * my.stdv) : 0.f;
    const float mxg = (g0 + g2) / (float)n, myg = (g1 + g3) / (float)n;
    for (int i = w; i < n; i += kSplit) {
      const long long o = 2LL * i * HW;
      gp[o + ox] = mxg + cx * (p[o + ox] - mx.mean);
      gp[o + oy] = myg + cy * (p[o + oy] - my.mean);
    }
  }
  // block total of the transfer gradient: wave 0 holds the only non-zero terms
  if (w == 0) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      gt0 += __shfl_xor(gt0, d);
      gt1 += __shfl_xor(gt1, d);
    }
    if (l == 0) {      // per-block partials, added in block order by moment_transfer_finish: no float atomics, bit-repeatable
      partial[2 * blockIdx.x] = gt0;
      partial[2 * blockIdx.x + 1] = gt1;
    }
  }
}

__global__ __launch_bounds__(64) void moment_transfer_finish(const float *__restrict__ partial, int blocks,
                                                             float *__restrict__ grad_transfer) {
  const int l = threadIdx.x;
  float a = 0.f, b = 0.f;
  for (int i = l; i < blocks; i += 64) { a += partial[2 * i]; b += partial[2 * i + 1]; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
  if (l == 0) { grad_transfer[0] = a; grad_transfer[1] = b; }
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
  const int grid = (int)(((long long)B * HW + 63) / 64);
  hipLaunchKernelGGL(moment_bbox_forward, dim3(grid), dim3(kLoc * kSplit), 0, (hipStream_t)stream, pts, moment_transfer, B,
                     n_pts, HW, y_first, bbox);
  KGDET_CHECK_LAUNCH("moment_bbox_forward");
  return KGDET_OK;
}

size_t kgdet_moment_bbox_backward_workspace_bytes(int32_t B, int32_t HW) {
  return (size_t)(((long long)(B > 0 ? B : 0) * (HW > 0 ? HW : 0) + 63) / 64) * 2 * sizeof(float) + 16;
}

int kgdet_moment_bbox_backward(const float *pts, const float *moment_transfer, const float *grad_bbox, int32_t B,
                               int32_t n_pts, int32_t HW, int32_t y_first, float *grad_pts, float *grad_transfer,
                               void *workspace, size_t workspace_bytes, void *stream) {
  KGDET_CHECK_SHAPE(B >= 0 && HW >= 0, "bad sizes");
  KGDET_CHECK_SHAPE(n_pts >= 2, "moment bbox needs at least 2 points (unbiased std)");
  if ((long long)B * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(pts && moment_transfer && grad_bbox && grad_pts && grad_transfer, "null pointer");
  KGDET_CHECK_SHAPE(workspace && workspace_bytes >= kgdet_moment_bbox_backward_workspace_bytes(B, HW),
                    "workspace too small (kgdet_moment_bbox_backward_workspace_bytes)");
  const int grid = (int)(((long long)B * HW + 63) / 64);
  hipLaunchKernelGGL(moment_bbox_backward, dim3(grid), dim3(kLoc * kSplit), 0, (hipStream_t)stream, pts, moment_transfer,
                     grad_bbox, B, n_pts, HW, y_first, grad_pts, (float *)workspace);
  hipLaunchKernelGGL(moment_transfer_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float *)workspace, grid,
                     grad_transfer);
  KGDET_CHECK_LAUNCH("moment_bbox_backward");
  return KGDET_OK;
}

}  // extern "C"
