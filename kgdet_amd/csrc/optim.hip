// Gradient clipping + Adam of the training step as multi-tensor passes (gfx950).
//
// The reference's step is OptimizerHook.after_train_iter: clip_grad_norm_ (max_norm 35, L2) then optimizer.step()
// (mmdet/core/utils/dist_utils.py:44-58 -> mmcv OptimizerHook; the KGDet config trains with Adam).  torch runs that as
// three multi-tensor norm launches + a reduction, a chain of scalar ops, three multi-tensor scale launches and six fused-Adam
// launches over the ~50 M parameters: 0.57 ms of a 15 ms step, at about half the memory bandwidth.  Here:
//   multi_sqnorm    one pass over all gradients -> per-block partial sums -> ||g||^2 (fixed order: deterministic)
//   multi_clip_adam one pass: coef = min(1, max_norm / (||g|| + 1e-6)) read from the device scalar (no host round trip),
//                   g' = coef g (written back only when coef < 1: clip_grad_norm_ scales the gradients in place),
//                   then torch's Adam update expression by expression (torch/optim/adam.py _single_tensor_adam /
//                   fused_adam_utils.cuh adam_math, ADAM mode ORIGINAL: weight decay added to the gradient).
// Table row per tensor: {param, grad, exp_avg, exp_avg_sq, numel, first block}; a block covers kOptChunk elements.
#include "common.h"

namespace kgdet {

namespace {
constexpr int kOptChunk = 4096;   // elements per block (256 threads x 4 x float4)

__device__ __forceinline__ const long long *opt_row(const long long *__restrict__ table, int n, int block) {
  int lo = 0, hi = n - 1;   // uniform binary search on the first-block column
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)table[mid * 6 + 5] <= block) lo = mid; else hi = mid - 1;
  }
  return table + lo * 6;
}

__device__ __forceinline__ float opt_block_sum(float v, float *red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
}  // namespace

__global__ __launch_bounds__(256) void multi_sqnorm(const long long *__restrict__ table, int n, float *__restrict__ partial) {
  __shared__ float red[4];
  const long long *row = opt_row(table, n, blockIdx.x);
  const float *g = reinterpret_cast<const float *>(row[1]);
  const long long numel = row[4], base = (long long)((int)blockIdx.x - (int)row[5]) * kOptChunk;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + (k * 256 + threadIdx.x) * 4;
    if (i + 3 < numel && (reinterpret_cast<size_t>(g) & 15) == 0) {
      const float4 v = *reinterpret_cast<const float4 *>(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int e = 0; e < 4; ++e)
        if (i + e < numel) s += g[i + e] * g[i + e];
    }
  }
  s = opt_block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void multi_sqnorm_finish(const float *__restrict__ partial, int count, float *__restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < count; i += 1024) s += partial[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = sqrtf(t);      // the L2 norm itself (what clip_grad_norm_ returns)
  }
}

// The step's schedule in device memory (round 5: the whole training step as ONE captured HIP graph -- kernel arguments are
// frozen at capture time, so the step count, the two bias corrections and the learning rate cannot be arguments there):
//   sched[0] = learning rate   sched[1] = 1 - beta1^t   sched[2] = sqrt(1 - beta2^t)   sched[3] = t (steps taken, as a float)
// adam_schedule_step advances t by one, recomputes the corrections in double precision (the host path's expressions:
// 1.0 - beta1 ** t, math.sqrt(1.0 - beta2 ** t)) and takes this step's learning rate from slot t % ring of `lr_ring` -- page-locked
// host memory the device reads in place; the host writes slot k before it launches step k (kgdet_amd/optim.py).
__global__ void adam_schedule_step(float *__restrict__ sched, const float *__restrict__ lr_ring, int ring, double beta1, double beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float t = sched[3] + 1.0f;
  sched[3] = t;
  sched[1] = (float)(1.0 - pow(beta1, (double)t));
  sched[2] = (float)sqrt(1.0 - pow(beta2, (double)t));
  if (lr_ring) sched[0] = lr_ring[(long long)t % ring];
}

__global__ __launch_bounds__(256) void multi_clip_adam(const long long *__restrict__ table, int n, const float *__restrict__ norm,
                                                       float max_norm, float lr, float beta1, float beta2, float eps,
                                                       float weight_decay, float bias_correction1, float bias_correction2_sqrt,
                                                       float one_minus_beta1, float one_minus_beta2,
                                                       const float *__restrict__ sched) {
  if (sched) {   // (device schedule: the arguments above are capture-time values)
    lr = sched[0];
    bias_correction1 = sched[1];
    bias_correction2_sqrt = sched[2];
  }
  const long long *row = opt_row(table, n, blockIdx.x);
  float *p = reinterpret_cast<float *>(row[0]), *g = reinterpret_cast<float *>(row[1]);
  float *m = reinterpret_cast<float *>(row[2]), *v = reinterpret_cast<float *>(row[3]);
  const long long numel = row[4], base = (long long)((int)blockIdx.x - (int)row[5]) * kOptChunk;
  float coef = 1.0f;
  if (max_norm > 0.0f) {
    coef = max_norm / (norm[0] + 1e-6f);     // clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    coef = coef < 1.0f ? coef : 1.0f;
  }
  const bool scale = coef < 1.0f;
  const float step_size = lr / bias_correction1, w1 = one_minus_beta1;   // (1 - beta rounded from double, as torch passes them)
  for (int k = 0; k < 16; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i >= numel) break;
    float gv = g[i];
    if (scale) {
      gv *= coef;
      g[i] = gv;                              // the gradient stays scaled, as after clip_grad_norm_
    }
    float pv = p[i];
    if (weight_decay != 0.0f) gv += pv * weight_decay;
    float mv = m[i], vv = v[i];
    mv = w1 < 0.5f ? mv + w1 * (gv - mv) : gv - (gv - mv) * (1.0f - w1);      // lerp(exp_avg, grad, 1 - beta1)
    vv = beta2 * vv + one_minus_beta2 * gv * gv;
    const float denom = sqrtf(vv) / bias_correction2_sqrt + eps;
    pv -= step_size * mv / denom;
    p[i] = pv;
    m[i] = mv;
    v[i] = vv;
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" int32_t kgdet_optim_chunk(void) { return kOptChunk; }

extern "C" int kgdet_multi_grad_norm(const int64_t *table_dev, int32_t n, int64_t total_blocks, float *partial, float *norm_out,
                                     void *stream) {
  KGDET_CHECK_SHAPE(table_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31) && partial && norm_out, "bad arguments");
  hipLaunchKernelGGL(multi_sqnorm, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const long long *)table_dev, n,
                     partial);
  hipLaunchKernelGGL(multi_sqnorm_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, (int)total_blocks, norm_out);
  KGDET_CHECK_LAUNCH("multi_grad_norm");
  return KGDET_OK;
}

extern "C" int kgdet_multi_clip_adam(const int64_t *table_dev, int32_t n, int64_t total_blocks, const float *norm, float max_norm,
                                     float lr, double beta1_d, double beta2_d, float eps, float weight_decay,
                                     float bias_correction1, float bias_correction2_sqrt, void *stream) {
  const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
  KGDET_CHECK_SHAPE(table_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "bad arguments");
  KGDET_CHECK_SHAPE(max_norm <= 0.0f || norm, "a positive max_norm needs the norm");
  hipLaunchKernelGGL(multi_clip_adam, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const long long *)table_dev,
                     n, norm, max_norm, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt,
                     (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), (const float *)nullptr);
  KGDET_CHECK_LAUNCH("multi_clip_adam");
  return KGDET_OK;
}

// The same update with the step's schedule read from device memory (see adam_schedule_step): first the schedule advances
// (sched[3] = steps taken so far on entry), then the update uses it.  `lr_ring`: `ring` floats of page-locked host memory
// (device-readable), slot t % ring = the learning rate of step t (1-based), or NULL: sched[0] is left as it is.
extern "C" int kgdet_multi_clip_adam_dev(const int64_t *table_dev, int32_t n, int64_t total_blocks, const float *norm,
                                         float max_norm, float *sched, const float *lr_ring, int32_t ring, double beta1_d,
                                         double beta2_d, float eps, float weight_decay, void *stream) {
  const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
  KGDET_CHECK_SHAPE(table_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31) && sched, "bad arguments");
  KGDET_CHECK_SHAPE(max_norm <= 0.0f || norm, "a positive max_norm needs the norm");
  KGDET_CHECK_SHAPE(lr_ring == nullptr || ring > 0, "empty learning-rate ring");
  hipLaunchKernelGGL(adam_schedule_step, dim3(1), dim3(64), 0, (hipStream_t)stream, sched, lr_ring, (int)ring, beta1_d, beta2_d);
  hipLaunchKernelGGL(multi_clip_adam, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const long long *)table_dev,
                     n, norm, max_norm, 0.0f, beta1, beta2, eps, weight_decay, 1.0f, 1.0f,
                     (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), (const float *)sched);
  KGDET_CHECK_LAUNCH("multi_clip_adam_dev");
  return KGDET_OK;
}
