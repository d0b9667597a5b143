// Small element-wise glue of the KGDet step as single passes (gfx950): launch-latency items of the torch chain.
//
//  reppts_offsets_forward / _backward: the offsets of a Kp3RepBlock's three deformable kernel sizes from the previous
//    stage's reppoints (reppoints_head_kp3rep_cas_1_assign_once.py:131-143): for the 9 / 25 / 49-point slices of
//    reppts [B, 166, H, W]:  offset_k = (gm * part + (1 - gm) * part.detach()) - base_k   -- value = the reference's float
//    expression, gradient = gm * grad (the "gradient_mul" trick) -- three contiguous [B, 2 k^2, H, W] outputs from one
//    launch (12 torch launches forward, ~15 backward per stage before).
//  subsample2_forward / _backward: x[:, :, ::2, ::2] as a contiguous tensor (the stride-2 1x1 downsample branch of
//    resnet.py:180-186 runs as a 1x1 convolution of the subsampled input) and its backward
//    grad_x = zero-stuffed grad_xs [+ other] in ONE pass: `other` is the gradient the trunk already holds for x, so the
//    zero fill, the strided scatter and the accumulation of the torch chain are one write of grad_x.
//  pts_from_offsets_forward / _backward: a level's predicted offsets [B, 2n, H, W] (channel = (point, y | x)) as image
//    coordinates [B, H*W, 2n] (x, y interleaved) = offset * stride + centre (reppoints_head_kp_serial.py offset_to_pts:
//    permute + reshape copy, flip, multiply, add -- four passes over a [2, 588, 100, 168] tensor, and their four backward
//    passes): a 32-pixel x 64-channel transpose through LDS, the pair swap in the channel index.
#include "common.h"

// (the offsets must equal torch's separate multiply / multiply / add / subtract bit for bit)
#pragma clang fp contract(off)

namespace kgdet {

struct RepOffsets {
  float *out[3];
  int first[3], count[3], k[3];   // channel range of reppts [first, first + count) for kernel size k
};

__global__ __launch_bounds__(256) void reppts_offsets_forward(const float *__restrict__ reppts, RepOffsets d, int B, int C,
                                                              int HW, float gm) {
  const long long total = (long long)B * C * HW;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((long long)HW * C));
    const int g = c < d.first[1] ? 0 : (c < d.first[2] ? 1 : 2);
    const int cc = c - d.first[g];
    if (cc >= d.count[g]) continue;
    const float v = reppts[i];
    const float part = gm * v + (1 - gm) * v;
    // regular grid (y, x) pairs, row-major (KP3:37-46): tap t = cc / 2, component cc & 1
    const int t = cc >> 1, k = d.k[g], pad = (k - 1) / 2;
    const float base = (cc & 1) ? (float)(t % k - pad) : (float)(t / k - pad);
    d.out[g][((long long)b * d.count[g] + cc) * HW + p] = part - base;
  }
}

struct RepGrads {
  const float *g[3];
  int first[3], count[3];
};

__global__ __launch_bounds__(256) void reppts_offsets_backward(RepGrads d, float *__restrict__ grad_reppts, int B, int C, int HW,
                                                               float gm) {
  const long long total = (long long)B * C * HW;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((long long)HW * C));
    const int g = c < d.first[1] ? 0 : (c < d.first[2] ? 1 : 2);
    const int cc = c - d.first[g];
    float v = 0.f;
    if (cc < d.count[g] && d.g[g] != nullptr) v = gm * d.g[g][((long long)b * d.count[g] + cc) * HW + p];
    grad_reppts[i] = v;
  }
}

__global__ __launch_bounds__(256) void subsample2_forward(const float *__restrict__ x, float *__restrict__ y, long long planes, int H,
                                                          int W, int Ho, int Wo) {
  const long long total = planes * Ho * Wo;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
    const long long pl = i / ((long long)Wo * Ho);
    y[i] = x[(pl * H + 2 * yo) * W + 2 * xo];
  }
}

// grad_x[pl, y, x] = (y, x both even ? grad_y[pl, y / 2, x / 2] : 0) + (other ? other[...] : 0); four x per thread
__global__ __launch_bounds__(256) void subsample2_backward(const float *__restrict__ gy, const float *__restrict__ other,
                                                           float *__restrict__ gx, long long planes, int H, int W, int Ho, int Wo) {
  const int W4 = W >> 2;                       // (W % 4 == 0 checked by the caller)
  const long long total = planes * H * W4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x4 = (int)(i % W4), y = (int)((i / W4) % H);
    const long long pl = i / ((long long)W4 * H);
    float4 v = other ? *reinterpret_cast<const float4 *>(other + (pl * H + y) * W + 4 * x4) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(y & 1)) {
      const float2 g = *reinterpret_cast<const float2 *>(gy + (pl * Ho + (y >> 1)) * Wo + 2 * x4);
      v.x += g.x;
      v.z += g.y;
    }
    *reinterpret_cast<float4 *>(gx + (pl * H + y) * W + 4 * x4) = v;
  }
}

// pred [B][C][HW] -> out [B][HW][C]: out[b][p][c ^ swap] = pred[b][c][p] * stride + centres[b][p][(c ^ swap) & 1]
// BACKWARD == true: the other way round, grad_pred[b][c][p] = g[b][p][c ^ swap] * stride.
// grid = (ceil(HW / 32), ceil(C / 64), B), 256 threads.
template <bool BACKWARD>
__global__ __launch_bounds__(256) void pts_from_offsets(const float *__restrict__ src, const float *__restrict__ centres,
                                                        float *__restrict__ dst, int C, int HW, float stride, int swap) {
  __shared__ float tile[64][33];
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 64, b = blockIdx.z;
  const float *pred = (BACKWARD ? dst : src) + (long long)b * C * HW;       // channel-major side
  const float *pts = (BACKWARD ? src : dst) + (long long)b * HW * C;        // pixel-major side
  const int tid = threadIdx.x;
  if constexpr (!BACKWARD) {
    const int px = tid & 31;
    for (int r = tid >> 5; r < 64; r += 8) {
      const int c = c0 + r, p = p0 + px;
      tile[r][px] = (c < C && p < HW) ? pred[(long long)c * HW + p] : 0.0f;
    }
    __syncthreads();
    const int cc = tid & 63;
    for (int r = tid >> 6; r < 32; r += 4) {
      const int p = p0 + r, co = c0 + cc;            // output channel; its source channel is co ^ swap (same 64-block)
      if (p < HW && co < C) {
        const float v = tile[cc ^ swap][r] * stride;
        const_cast<float *>(pts)[(long long)p * C + co] = v + centres[((long long)b * HW + p) * 2 + (co & 1)];
      }
    }
  } else {
    const int cc = tid & 63;
    for (int r = tid >> 6; r < 32; r += 4) {
      const int p = p0 + r, co = c0 + cc;
      tile[cc][r] = (p < HW && co < C) ? pts[(long long)p * C + co] : 0.0f;
    }
    __syncthreads();
    const int px = tid & 31;
    for (int r = tid >> 5; r < 64; r += 8) {
      const int c = c0 + r, p = p0 + px;
      if (c < C && p < HW) const_cast<float *>(pred)[(long long)c * HW + p] = tile[r ^ swap][px] * stride;
    }
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" {

int kgdet_reppts_offsets_forward(const float *reppts, int32_t B, int32_t C, int32_t HW, const int32_t *kernel_sizes, float gm,
                                 float *out0, float *out1, float *out2, void *stream) {
  KGDET_CHECK_SHAPE(reppts && kernel_sizes && out0 && out1 && out2 && B > 0 && C > 0 && HW > 0, "bad arguments");
  RepOffsets d;
  float *outs[3] = {out0, out1, out2};
  int first = 0;
  for (int g = 0; g < 3; ++g) {
    d.out[g] = outs[g]; d.k[g] = kernel_sizes[g]; d.first[g] = first; d.count[g] = 2 * kernel_sizes[g] * kernel_sizes[g];
    first += d.count[g];
  }
  KGDET_CHECK_SHAPE(first <= C, "reppoints tensor has %d channels, the three kernel sizes need %d", C, first);
  const long long total = (long long)B * C * HW;
  hipLaunchKernelGGL(reppts_offsets_forward, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, reppts, d, B, C, HW, gm);
  KGDET_CHECK_LAUNCH("reppts_offsets_forward");
  return KGDET_OK;
}

int kgdet_reppts_offsets_backward(const float *g0, const float *g1, const float *g2, int32_t B, int32_t C, int32_t HW,
                                  const int32_t *kernel_sizes, float gm, float *grad_reppts, void *stream) {
  KGDET_CHECK_SHAPE(kernel_sizes && grad_reppts && B > 0 && C > 0 && HW > 0, "bad arguments");
  RepGrads d;
  const float *gs[3] = {g0, g1, g2};
  int first = 0;
  for (int g = 0; g < 3; ++g) {
    d.g[g] = gs[g]; d.first[g] = first; d.count[g] = 2 * kernel_sizes[g] * kernel_sizes[g];
    first += d.count[g];
  }
  KGDET_CHECK_SHAPE(first <= C, "reppoints tensor has %d channels, the three kernel sizes need %d", C, first);
  const long long total = (long long)B * C * HW;
  hipLaunchKernelGGL(reppts_offsets_backward, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d, grad_reppts, B, C, HW, gm);
  KGDET_CHECK_LAUNCH("reppts_offsets_backward");
  return KGDET_OK;
}

int kgdet_subsample2_forward(const float *x, float *y, int64_t planes, int32_t H, int32_t W, void *stream) {
  KGDET_CHECK_SHAPE(x && y && planes > 0 && H > 0 && W > 0, "bad arguments");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long total = planes * Ho * Wo;
  hipLaunchKernelGGL(subsample2_forward, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, (long long)planes, H, W, Ho, Wo);
  KGDET_CHECK_LAUNCH("subsample2_forward");
  return KGDET_OK;
}

int kgdet_subsample2_backward(const float *grad_y, const float *other, float *grad_x, int64_t planes, int32_t H, int32_t W,
                              void *stream) {
  KGDET_CHECK_SHAPE(grad_y && grad_x && planes > 0 && H > 0 && W > 0 && W % 4 == 0, "bad arguments (W must be a multiple of 4)");
  const int Ho = (H + 1) / 2, Wo = W / 2;
  const long long total = planes * H * (W / 4);
  hipLaunchKernelGGL(subsample2_backward, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, grad_y, other, grad_x, (long long)planes, H, W, Ho, Wo);
  KGDET_CHECK_LAUNCH("subsample2_backward");
  return KGDET_OK;
}

int kgdet_pts_from_offsets_forward(const float *pred, const float *centres, float *pts, int64_t B, int32_t C, int64_t HW,
                                   float stride, int32_t y_first, void *stream) {
  KGDET_CHECK_SHAPE(pred && centres && pts && B > 0 && C > 0 && C % 2 == 0 && HW > 0 && HW < (1LL << 31) && B < 65536,
                    "bad arguments");
  hipLaunchKernelGGL(pts_from_offsets<false>, dim3((unsigned)((HW + 31) / 32), (unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, pred, centres, pts, C, (int)HW, stride, y_first ? 1 : 0);
  KGDET_CHECK_LAUNCH("pts_from_offsets_forward");
  return KGDET_OK;
}

int kgdet_pts_from_offsets_backward(const float *grad_pts, float *grad_pred, int64_t B, int32_t C, int64_t HW, float stride,
                                    int32_t y_first, void *stream) {
  KGDET_CHECK_SHAPE(grad_pts && grad_pred && B > 0 && C > 0 && C % 2 == 0 && HW > 0 && HW < (1LL << 31) && B < 65536,
                    "bad arguments");
  hipLaunchKernelGGL(pts_from_offsets<true>, dim3((unsigned)((HW + 31) / 32), (unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, grad_pts, (const float *)nullptr, grad_pred, C, (int)HW, stride, y_first ? 1 : 0);
  KGDET_CHECK_LAUNCH("pts_from_offsets_backward");
  return KGDET_OK;
}

}  // extern "C"
