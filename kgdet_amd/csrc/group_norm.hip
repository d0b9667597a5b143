// GroupNorm + ReLU of the head towers / neck (ConvModule with norm_cfg = GN, 32 groups: mmdet/models/utils/conv_module.py:142-165
// calls norm then activate) as one pass forward and one backward (gfx950).
//
// ATen runs GroupNorm as RowwiseMoments + ComputeFusedParams + an element-wise pass, then ReLU, and backward as
// threshold_backward + ComputeInternalGradients + ComputeBackwardFusedParams + GammaBetaBackward + an element-wise pass: ten
// launches per layer, each a few microseconds on the head's 25 x 42 maps.  Here one workgroup owns one (image, group):
//   forward   mean, rstd over the group's D x HW elements (two passes: mean, then centred squares), y = [relu]((x - mean) rstd g + b)
//   backward  g' = gy [y > 0];  per channel ds = sum g' x, db = sum g';  dgamma_n = (ds - mean db) rstd, dbeta_n = db;
//             dx = g' gamma rstd + c2 x + c3,  c2 = (S1 mean - S2) rstd^3 / (D HW),  c3 = -c2 mean - S1 rstd / (D HW),
//             S1 = sum_c gamma db, S2 = sum_c gamma ds      (the standard GroupNorm gradient)
// dgamma / dbeta leave as per-image rows [N][C]; the caller adds the N rows.  Deterministic.
#include "common.h"

namespace kgdet {

namespace {
constexpr int kGnThreads = 1024;

__device__ __forceinline__ float gn_block_sum(float v, float *red) {   // all threads get the sum; red: 16 floats
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();                                  // (red may still be read from the previous call)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kGnThreads / 64; ++w) s += red[w];
  return s;
}
}  // namespace

__global__ __launch_bounds__(kGnThreads) void gn_act_forward(const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps, int relu,
                                                             float *__restrict__ y, float *__restrict__ mean_out,
                                                             float *__restrict__ rstd_out, int C, int G, int HW) {
  __shared__ float red[16];
  const int n = blockIdx.x / G, g = blockIdx.x % G, D = C / G;
  const long long base = ((long long)n * C + (long long)g * D) * HW;
  const int total = D * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < total; i += kGnThreads) s += x[base + i];
  const float mean = gn_block_sum(s, red) / (float)total;
  float q = 0.f;
  for (int i = threadIdx.x; i < total; i += kGnThreads) {
    const float d = x[base + i] - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(gn_block_sum(q, red) / (float)total + eps);
  if (threadIdx.x == 0) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = rstd;
  }
  for (int i = threadIdx.x; i < total; i += kGnThreads) {
    const int c = g * D + i / HW;
    float v = (x[base + i] - mean) * rstd * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f);
    if (relu) v = fmaxf(v, 0.0f);
    y[base + i] = v;
  }
}

// dgb: [2][N][C] (dgamma rows, then dbeta rows)
__global__ __launch_bounds__(kGnThreads) void gn_act_backward(const float *__restrict__ gy, const float *__restrict__ x,
                                                              const float *__restrict__ y, const float *__restrict__ gamma,
                                                              const float *__restrict__ mean_in,
                                                              const float *__restrict__ rstd_in, int relu,
                                                              float *__restrict__ gx, float *__restrict__ dgb, int N, int C,
                                                              int G, int HW) {
  __shared__ float part_ds[64][16], part_db[64][16];   // [channel][wave slice]
  __shared__ float ch_ds[64], ch_db[64];
  const int n = blockIdx.x / G, g = blockIdx.x % G, D = C / G;
  const long long base = ((long long)n * C + (long long)g * D) * HW;
  const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
  // per channel: ds = sum g' x, db = sum g'.  The sixteen waves split the channels: wpc waves per channel when D <= 16 (each
  // takes a slice of the pixels), one wave per channel in rounds of sixteen otherwise -- one barrier instead of two per channel
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpc = D <= 16 ? 16 / D : 1;
  for (int d0 = 0; d0 < D; d0 += 16 / wpc) {
    const int d = d0 + wave / wpc, sub = wave % wpc;
    if (d < D && wave < (16 / wpc) * wpc) {
      const long long cb = base + (long long)d * HW;
      float ds = 0.f, db = 0.f;
      for (int i = sub * 64 + lane; i < HW; i += wpc * 64) {
        float gv = gy[cb + i];
        if (relu && !(y[cb + i] > 0.0f)) gv = 0.0f;
        ds += gv * x[cb + i];
        db += gv;
      }
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) { ds += __shfl_xor(ds, s); db += __shfl_xor(db, s); }
      if (lane == 0) { part_ds[d][sub] = ds; part_db[d][sub] = db; }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < D) {
    float ds = 0.f, db = 0.f;
    for (int s = 0; s < wpc; ++s) { ds += part_ds[threadIdx.x][s]; db += part_db[threadIdx.x][s]; }
    ch_ds[threadIdx.x] = ds;
    ch_db[threadIdx.x] = db;
  }
  __syncthreads();
  float S1 = 0.f, S2 = 0.f;
  for (int d = 0; d < D; ++d) {
    const float gm = gamma ? gamma[g * D + d] : 1.0f;
    S1 += gm * ch_db[d];
    S2 += gm * ch_ds[d];
  }
  if ((int)threadIdx.x < D) {
    const int c = g * D + threadIdx.x;
    dgb[(long long)n * C + c] = (ch_ds[threadIdx.x] - mean * ch_db[threadIdx.x]) * rstd;
    dgb[((long long)N + n) * C + c] = ch_db[threadIdx.x];
  }
  if (!gx) return;
  const float inv = 1.0f / ((float)D * (float)HW);
  const float c2 = (S1 * mean - S2) * rstd * rstd * rstd * inv;
  const float c3 = -c2 * mean - S1 * rstd * inv;
  const int total = D * HW;
  for (int i = threadIdx.x; i < total; i += kGnThreads) {
    const int c = g * D + i / HW;
    float gv = gy[base + i];
    if (relu && !(y[base + i] > 0.0f)) gv = 0.0f;
    gx[base + i] = gv * (gamma ? gamma[c] : 1.0f) * rstd + c2 * x[base + i] + c3;
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" int kgdet_gn_act_forward(const float *x, const float *gamma, const float *beta, int32_t groups, float eps,
                                    int32_t relu, float *y, float *mean, float *rstd, int64_t N, int32_t C, int64_t HW,
                                    void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && groups > 0 && C % groups == 0 && HW >= 0 && (long long)(C / groups) * HW < (1LL << 31),
                    "bad sizes");
  KGDET_CHECK_SHAPE(C / groups <= 64 && N * groups < (1LL << 31), "at most 64 channels per group");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x && y && mean && rstd, "null pointer");
  hipLaunchKernelGGL(gn_act_forward, dim3((unsigned)(N * groups)), dim3(kGnThreads), 0, (hipStream_t)stream, x, gamma, beta, eps,
                     relu, y, mean, rstd, C, groups, (int)HW);
  KGDET_CHECK_LAUNCH("gn_act_forward");
  return KGDET_OK;
}

extern "C" int kgdet_gn_act_backward(const float *grad_y, const float *x, const float *y, const float *gamma, const float *mean,
                                     const float *rstd, int32_t groups, int32_t relu, float *grad_x, float *dgamma_dbeta,
                                     int64_t N, int32_t C, int64_t HW, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && groups > 0 && C % groups == 0 && HW >= 0 && (long long)(C / groups) * HW < (1LL << 31),
                    "bad sizes");
  KGDET_CHECK_SHAPE(C / groups <= 64 && N * groups < (1LL << 31), "at most 64 channels per group");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(grad_y && x && mean && rstd && dgamma_dbeta && (!relu || y), "null pointer");
  hipLaunchKernelGGL(gn_act_backward, dim3((unsigned)(N * groups)), dim3(kGnThreads), 0, (hipStream_t)stream, grad_y, x, y, gamma,
                     mean, rstd, relu, grad_x, dgamma_dbeta, (int)N, C, groups, (int)HW);
  KGDET_CHECK_LAUNCH("gn_act_backward");
  return KGDET_OK;
}
