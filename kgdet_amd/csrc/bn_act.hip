// Frozen-statistics BatchNorm + residual add + ReLU as ONE pass, forward and backward (training, fp32, NCHW).
//
// mmdet's ResNet trains with norm_eval=True (mmdet/models/backbones/resnet.py:518-525): BatchNorm uses its running
// statistics, i.e. it is the per-channel affine map  y = x * s + t,  s = gamma / sqrt(var + eps),  t = beta - mean * s,
// whose gamma / beta still receive gradients.  A bottleneck (resnet.py:240-262) runs it as BatchNorm, residual add
// and ReLU -- three passes over the activation forward, threshold + BatchNorm backward (4 reads, 2 writes) backward.
// Here:   forward   y = [relu](x * s + t [+ r])                                   1-2 reads, 1 write
//         backward  g' = g * [pre-activation > 0];  grad_x = g' * s;  grad_r = g'  2-3 reads, 1-2 writes
//                   grad_beta = sum g',  grad_gamma = invstd * sum g' * (x - mean)
// One workgroup owns a slice of one (image, channel) plane, so the channel constants are scalars; the two channel
// sums leave as per-workgroup partials [2][C][P] (deterministic; the caller adds the P partials).
#include "common.h"

namespace kgdet {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

struct ChannelAffine {
  float s, t, mean, invstd;
};

__device__ __forceinline__ ChannelAffine channel_affine(const float *gamma, const float *beta, const float *mean,
                                                        const float *var, float eps, int c) {
  ChannelAffine a;
  a.mean = mean[c];
  a.invstd = 1.0f / sqrtf(var[c] + eps);
  a.s = (gamma ? gamma[c] : 1.0f) * a.invstd;
  a.t = (beta ? beta[c] : 0.0f) - a.mean * a.s;
  return a;
}

}  // namespace

// grid (chunks, N*C); a chunk is a contiguous run of `per` elements of the plane (per % 4 == 0)
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const float *__restrict__ mean,
                                                         const float *__restrict__ var, float eps,
                                                         const float *__restrict__ res, float *__restrict__ y, int C,
                                                         int HW, int per) {
  const int plane = blockIdx.y, c = plane % C;
  const ChannelAffine a = channel_affine(gamma, beta, mean, var, eps, c);
  const long long base = (long long)plane * HW;
  const int lo = blockIdx.x * per, hi = min(HW, lo + per);
  if ((HW & 3) == 0) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      float4 v = *reinterpret_cast<const float4 *>(x + base + i);
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (RES) r = *reinterpret_cast<const float4 *>(res + base + i);
      v.x = v.x * a.s + a.t + r.x; v.y = v.y * a.s + a.t + r.y;
      v.z = v.z * a.s + a.t + r.z; v.w = v.w * a.s + a.t + r.w;
      if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4 *>(y + base + i) = v;
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      float v = x[base + i] * a.s + a.t;
      if (RES) v += res[base + i];
      if (RELU) v = fmaxf(v, 0.f);
      y[base + i] = v;
    }
  }
}

// y is read only when RES && RELU (the mask cannot be recomputed without the residual input);
// grad_res is written only when RES && RELU (otherwise the residual gradient IS grad_y)
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                         const float *__restrict__ y, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const float *__restrict__ mean,
                                                         const float *__restrict__ var, float eps,
                                                         float *__restrict__ gx, float *__restrict__ gres,
                                                         float *__restrict__ partial, int C, int HW, int per, int P) {
  __shared__ float red[2][4];
  const int plane = blockIdx.y, c = plane % C, n = plane / C;
  const ChannelAffine a = channel_affine(gamma, beta, mean, var, eps, c);
  const long long base = (long long)plane * HW;
  const int lo = blockIdx.x * per, hi = min(HW, lo + per);
  float s1 = 0.f, s2 = 0.f;
  auto one = [&](float g, float xv, float yv) -> float {
    if (RELU) {
      const bool on = RES ? (yv > 0.f) : (xv * a.s + a.t > 0.f);
      g = on ? g : 0.f;
    }
    s1 += g;
    s2 += g * (xv - a.mean);
    return g;
  };
  if ((HW & 3) == 0) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      float4 g = *reinterpret_cast<const float4 *>(gy + base + i);
      const float4 xv = *reinterpret_cast<const float4 *>(x + base + i);
      float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (RES && RELU) yv = *reinterpret_cast<const float4 *>(y + base + i);
      g.x = one(g.x, xv.x, yv.x); g.y = one(g.y, xv.y, yv.y); g.z = one(g.z, xv.z, yv.z); g.w = one(g.w, xv.w, yv.w);
      if (RES && RELU) *reinterpret_cast<float4 *>(gres + base + i) = g;
      if (gx) *reinterpret_cast<float4 *>(gx + base + i) = make_float4(g.x * a.s, g.y * a.s, g.z * a.s, g.w * a.s);
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      const float g = one(gy[base + i], x[base + i], (RES && RELU) ? y[base + i] : 0.f);
      if (RES && RELU) gres[base + i] = g;
      if (gx) gx[base + i] = g * a.s;
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int slot = n * gridDim.x + blockIdx.x;
    partial[(long long)c * P + slot] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[(long long)(C + c) * P + slot] = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * a.invstd;
  }
}

// sums[2][C] = sum over the P partials of every (row, channel), slot order (deterministic): one thread per (row, channel)
__global__ __launch_bounds__(256) void bn_partial_sum(const float *__restrict__ partial, float *__restrict__ sums, int rows,
                                                      int P) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const float *p = partial + (long long)i * P;
  float s = 0.f;
  for (int k = 0; k < P; ++k) s += p[k];
  sums[i] = s;
}

// ---- BatchNorm folded into the convolution (kgdet_amd/backbone.py _ConvBNActFold) ---------------------------------------
// forward:  z = [relu](conv(x, w * s) + t [+ r])  in the convolution's store (csrc/conv1x1.hip), y = conv(x, w) is never formed.
// backward: g' = g * [z > 0]  (this kernel; it also IS the residual branch's gradient), grad_x = conv_grad_input(g', w * s),
//           G = conv_grad_weight(x, g'), grad_w = s (.) G, grad_beta = sum g',
//           grad_gamma = invstd * (sum g' * y - mean * sum g')  with  sum_p g'[o, p] * y[o, p] = <w[o], G[o]>  (bn_fold_finish) --
//           the identity needs neither y nor gamma != 0.
template <bool RELU>
__global__ __launch_bounds__(256) void relu_sum_bwd_kernel(const float *__restrict__ gz, const float *__restrict__ z,
                                                           float *__restrict__ g, float *__restrict__ partial, int C, int HW,
                                                           int per, int P) {
  __shared__ float red[4];
  const int plane = blockIdx.y, c = plane % C, n = plane / C;
  const long long base = (long long)plane * HW;
  const int lo = blockIdx.x * per, hi = min(HW, lo + per);
  float s1 = 0.f;
  {
    // 16-byte accesses need 4-byte alignment only on gfx950 (tools/microbench/unaligned_x4.hip), so planes whose pixel count is
    // not a multiple of 4 (25 x 42: the whole of layer 4) take the vector loop too; `per` is a multiple of 4, only the plane's
    // last chunk can end on a ragged piece, which the first lanes finish one pixel at a time (round 3: a scalar loop for
    // every such plane)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int hi4 = lo + ((hi - lo) & ~3);
    for (int i = lo + threadIdx.x * 4; i < hi4; i += 1024) {
      f32x4u v = *reinterpret_cast<const f32x4u *>(gz + base + i);
      if (RELU) {
        const f32x4u zv = *reinterpret_cast<const f32x4u *>(z + base + i);
        v[0] = zv[0] > 0.f ? v[0] : 0.f; v[1] = zv[1] > 0.f ? v[1] : 0.f; v[2] = zv[2] > 0.f ? v[2] : 0.f; v[3] = zv[3] > 0.f ? v[3] : 0.f;
        *reinterpret_cast<f32x4u *>(g + base + i) = v;
      }
      s1 += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (int i = hi4 + threadIdx.x; i < hi; i += 256) {
      float v = gz[base + i];
      if (RELU) { v = z[base + i] > 0.f ? v : 0.f; g[base + i] = v; }
      s1 += v;
    }
  }
  s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s1;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long long)c * P + n * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per output channel o: grad_beta[o] = sum of the P partials (slot order), dot = <w[o], G[o]> over the CK
// elements of the row (fixed order: lane-strided partial sums, wave butterflies, the four waves in order),
// grad_gamma[o] = invstd[o] * (dot - mean[o] * grad_beta[o]), and the row of G is scaled by s[o] in place (= grad_w).
__global__ __launch_bounds__(256) void bn_fold_finish_kernel(const float *__restrict__ partial, int P,
                                                             const float *__restrict__ w, float *__restrict__ G,
                                                             const float *__restrict__ s, const float *__restrict__ mean,
                                                             const float *__restrict__ var, float eps,
                                                             float *__restrict__ gbeta, float *__restrict__ ggamma, int CK) {
  __shared__ float red[2][4];
  const int o = blockIdx.x;
  float sb = 0.f;
  for (int k = threadIdx.x; k < P; k += 256) sb += partial[(long long)o * P + k];
  float dot = 0.f;
  const float so = s[o];
  const float *wr = w + (long long)o * CK;
  float *gr = G ? G + (long long)o * CK : nullptr;
  if (gr) {
    for (int k = threadIdx.x; k < CK; k += 256) {
      const float gv = gr[k];
      dot += wr[k] * gv;
      gr[k] = gv * so;
    }
  }
  sb = wave_sum(sb);
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sb; red[1][threadIdx.x >> 6] = dot; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float b = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float d = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    if (gbeta) gbeta[o] = b;
    if (ggamma) ggamma[o] = (d - mean[o] * b) / sqrtf(var[o] + eps);
  }
}

namespace {

// chunks per plane: enough workgroups to fill 256 CUs several times over, at least 1024 elements each
int chunks_for(int64_t planes, int64_t HW) {
  int64_t want = (4096 + planes - 1) / planes;
  const int64_t most = (HW + 1023) / 1024;
  if (want > most) want = most;
  return want < 1 ? 1 : (int)want;
}

}  // namespace

}  // namespace kgdet

using namespace kgdet;

// The frozen stem (resnet.py:487-491, 528: maxpool(relu(norm1(conv1(x)))) with conv1 / norm1 never trained): BatchNorm, ReLU
// and the 3x3 stride-2 padding-1 max pooling in one pass -- the full-resolution normalised map (137 MB at 2 x 64 x 400 x 672)
// is neither written nor read back.  relu(.) >= 0 and every window holds a valid element, so the padding never wins.
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ var, float eps,
                                                              float *__restrict__ y, int C, int H, int W, int Ho, int Wo) {
  const int plane = blockIdx.y, c = plane % C;
  const ChannelAffine a = channel_affine(gamma, beta, mean, var, eps, c);
  const float *xp = x + (long long)plane * H * W;
  float *yp = y + (long long)plane * Ho * Wo;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < Ho * Wo; o += gridDim.x * 256) {
    const int i = o / Wo, j = o - i * Wo;
    float m = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int r = 2 * i + dy;
      if (r < 0 || r >= H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int q = 2 * j + dx;
        if (q < 0 || q >= W) continue;
        m = fmaxf(m, xp[(long long)r * W + q] * a.s + a.t);
      }
    }
    yp[o] = m;
  }
}

extern "C" int kgdet_bn_relu_maxpool(const float *x, const float *gamma, const float *beta, const float *mean,
                                     const float *var, float eps, float *y, int64_t N, int32_t C, int32_t H, int32_t W,
                                     void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && H > 0 && W > 0, "bad sizes");
  if (N == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x && y && mean && var, "null pointer");
  KGDET_CHECK_SHAPE(N * C <= 65535, "N*C = %lld exceeds the grid limit", (long long)(N * C));
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;     // floor((H + 2 - 3) / 2) + 1
  int chunks = (Ho * Wo + 1023) / 1024;                      // four outputs per thread
  hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(chunks, (unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                     beta, mean, var, eps, y, C, H, W, Ho, Wo);
  KGDET_CHECK_LAUNCH("bn_relu_maxpool");
  return KGDET_OK;
}

extern "C" int32_t kgdet_bn_act_partials(int64_t N, int32_t C, int64_t HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return (int32_t)(N * chunks_for(N * C, HW));
}

extern "C" int kgdet_bn_act_forward(const float *x, const float *gamma, const float *beta, const float *mean,
                                    const float *var, float eps, const float *residual, float *y, int64_t N, int32_t C,
                                    int64_t HW, int32_t relu, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && HW >= 0 && HW < (1LL << 31), "bad sizes");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x && y && mean && var, "null pointer");
  KGDET_CHECK_SHAPE(N * C <= 65535, "N*C = %lld exceeds the grid limit", (long long)(N * C));
  const int chunks = chunks_for(N * C, HW);
  const int per = (int)(((HW + chunks - 1) / chunks + 3) / 4 * 4);
  dim3 grid(chunks, (unsigned)(N * C));
#define LAUNCH(RES, RELU)                                                                                      \
  hipLaunchKernelGGL((bn_act_fwd_kernel<RES, RELU>), grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, \
                     mean, var, eps, residual, y, C, (int)HW, per)
  if (residual) { if (relu) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (relu) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  KGDET_CHECK_LAUNCH("bn_act_forward");
  return KGDET_OK;
}

extern "C" int kgdet_bn_act_backward(const float *grad_y, const float *x, const float *y, const float *gamma,
                                     const float *beta, const float *mean, const float *var, float eps,
                                     int32_t has_residual, int32_t relu, float *grad_x, float *grad_residual,
                                     float *partial, float *sums, int64_t N, int32_t C, int64_t HW, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && HW >= 0 && HW < (1LL << 31), "bad sizes");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(grad_y && x && mean && var && partial, "null pointer");
  KGDET_CHECK_SHAPE(!(has_residual && relu) || (y && grad_residual),
                    "residual + relu needs the forward output and a grad_residual buffer");
  KGDET_CHECK_SHAPE(N * C <= 65535, "N*C = %lld exceeds the grid limit", (long long)(N * C));
  const int chunks = chunks_for(N * C, HW);
  const int per = (int)(((HW + chunks - 1) / chunks + 3) / 4 * 4);
  const int P = (int)(N * chunks);
  dim3 grid(chunks, (unsigned)(N * C));
#define LAUNCH(RES, RELU)                                                                                        \
  hipLaunchKernelGGL((bn_act_bwd_kernel<RES, RELU>), grid, dim3(256), 0, (hipStream_t)stream, grad_y, x, y, gamma, \
                     beta, mean, var, eps, grad_x, grad_residual, partial, C, (int)HW, per, P)
  if (has_residual) { if (relu) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (relu) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  KGDET_CHECK_LAUNCH("bn_act_backward");
  if (sums) {   // [2][C]: grad_beta, grad_gamma
    hipLaunchKernelGGL(bn_partial_sum, dim3((2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, sums, 2 * C,
                       P);
    KGDET_CHECK_LAUNCH("bn_partial_sum");
  }
  return KGDET_OK;
}

extern "C" int kgdet_bn_fold_backward(const float *grad_z, const float *z, int32_t relu, float *g, float *partial, int64_t N,
                                      int32_t C, int64_t HW, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && HW >= 0 && HW < (1LL << 31), "bad sizes");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(grad_z && partial && (!relu || (z && g)), "null pointer");
  KGDET_CHECK_SHAPE(N * C <= 65535, "N*C = %lld exceeds the grid limit", (long long)(N * C));
  const int chunks = chunks_for(N * C, HW);
  const int per = (int)(((HW + chunks - 1) / chunks + 3) / 4 * 4);
  const int P = (int)(N * chunks);
  dim3 grid(chunks, (unsigned)(N * C));
  if (relu)
    hipLaunchKernelGGL(relu_sum_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, grad_z, z, g, partial, C, (int)HW,
                       per, P);
  else
    hipLaunchKernelGGL(relu_sum_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, grad_z, z, g, partial, C,
                       (int)HW, per, P);
  KGDET_CHECK_LAUNCH("bn_fold_backward");
  return KGDET_OK;
}

extern "C" int kgdet_bn_fold_finish(const float *partial, int32_t P, const float *w, float *G, const float *s,
                                    const float *mean, const float *var, float eps, float *grad_beta, float *grad_gamma,
                                    int32_t O, int32_t CK, void *stream) {
  KGDET_CHECK_SHAPE(O > 0 && CK > 0 && P > 0, "bad sizes");
  KGDET_CHECK_SHAPE(partial && w && s && mean && var, "null pointer");
  KGDET_CHECK_SHAPE(G || !grad_gamma, "grad_gamma needs the raw weight gradient");
  hipLaunchKernelGGL(bn_fold_finish_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, partial, P, w, G, s, mean, var, eps,
                     grad_beta, grad_gamma, CK);
  KGDET_CHECK_LAUNCH("bn_fold_finish");
  return KGDET_OK;
}
