// Weighted smooth-L1 loss of the KGDet head's box / keypoint branches as one pass each way (gfx950).
//
// The reference computes  sum(w * smooth_l1(pred / d - target / d)) / avg_factor  (KP3:621-665 with
// mmdet/models/losses/smooth_l1_loss.py:8-45 and utils.py:7-52) as a chain of element-wise torch kernels -- two divisions,
// subtract, abs, compare, the two branches, select, weight, sum: fourteen launches forward and about as many backward per
// loss, six losses per step, on tensors of 2100 x 588 (keypoints) or 2100 x 4 (boxes) elements.  Same expressions, same
// operation order per element; the sum runs over per-workgroup partials in fixed order (deterministic).
//   forward   s = sum_i w_i * l(|p_i / d - t_i / d|),  l(x) = x < beta ? 0.5 x x / beta : x - 0.5 beta
//   backward  grad_p_i = g * w_i * l'(.) / d,          l'   = x < beta ? (p_i / d - t_i / d) / beta : sign(.)
#include "common.h"

namespace kgdet {

namespace {
constexpr int kSl1Blocks = 256;   // partial sums of the forward pass

__device__ __forceinline__ float block_sum_256(float v, float *red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
}  // namespace

__global__ __launch_bounds__(256) void smooth_l1_forward(const float *__restrict__ pred, const float *__restrict__ target,
                                                         const float *__restrict__ weight, long long n, float beta,
                                                         float divisor, float *__restrict__ partial) {
  __shared__ float red[4];
  float s = 0.0f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    // The dense targets of a five-level head weight whole rows with zero (every point that is not a positive: > 99 % of a
    // [33600, 588] tensor): a wave whose 64 weights are all zero adds 0 without reading pred and target (for finite inputs the
    // same sum; the reference would turn an infinite prediction at a zero weight into NaN).
    float w = 1.0f;
    if (weight) {
      w = weight[i];
      if (__ballot(w != 0.0f) == 0ull) continue;
    }
    const float diff = fabsf(pred[i] / divisor - target[i] / divisor);
    const float l = diff < beta ? 0.5f * diff * diff / beta : diff - 0.5f * beta;
    s += weight ? l * w : l;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void smooth_l1_finish(const float *__restrict__ partial, int count, float *__restrict__ out) {
  __shared__ float red[4];
  const float s = block_sum_256((int)threadIdx.x < count ? partial[threadIdx.x] : 0.0f, red);
  if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void smooth_l1_backward(const float *__restrict__ pred, const float *__restrict__ target,
                                                          const float *__restrict__ weight, const float *__restrict__ grad_sum,
                                                          long long n, float beta, float divisor, float *__restrict__ grad_pred) {
  const float g = grad_sum[0];
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float w = 1.0f;
    if (weight) {
      w = weight[i];
      if (__ballot(w != 0.0f) == 0ull) {      // (as in the forward pass: nothing to read where the whole wave weighs zero)
        grad_pred[i] = 0.0f;
        continue;
      }
    }
    const float x = pred[i] / divisor - target[i] / divisor, diff = fabsf(x);
    // autograd of the reference chain: d|x| = sign(x) (0 at 0); quadratic branch diff / beta * sign(x) = x / beta
    const float dl = diff < beta ? x / beta : (x > 0.0f ? 1.0f : x < 0.0f ? -1.0f : 0.0f);
    grad_pred[i] = g * w * dl / divisor;
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" int kgdet_smooth_l1_sum_forward(const float *pred, const float *target, const float *weight, int64_t n, float beta,
                                           float divisor, float *partial, float *sum_out, void *stream) {
  KGDET_CHECK_SHAPE(n >= 0 && beta > 0.0f && divisor != 0.0f, "bad arguments");
  KGDET_CHECK_SHAPE(partial && sum_out && (n == 0 || (pred && target)), "null pointer");
  hipLaunchKernelGGL(smooth_l1_forward, dim3(kSl1Blocks), dim3(256), 0, (hipStream_t)stream, pred, target, weight,
                     (long long)n, beta, divisor, partial);
  hipLaunchKernelGGL(smooth_l1_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, kSl1Blocks, sum_out);
  KGDET_CHECK_LAUNCH("smooth_l1_forward");
  return KGDET_OK;
}

extern "C" int32_t kgdet_smooth_l1_partials(void) { return kSl1Blocks; }

extern "C" int kgdet_smooth_l1_sum_backward(const float *pred, const float *target, const float *weight, const float *grad_sum,
                                            int64_t n, float beta, float divisor, float *grad_pred, void *stream) {
  KGDET_CHECK_SHAPE(n >= 0 && beta > 0.0f && divisor != 0.0f, "bad arguments");
  if (n == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(pred && target && grad_sum && grad_pred, "null pointer");
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(smooth_l1_backward, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     pred, target, weight, grad_sum, (long long)n, beta, divisor, grad_pred);
  KGDET_CHECK_LAUNCH("smooth_l1_backward");
  return KGDET_OK;
}
