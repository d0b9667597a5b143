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

// T = float, or __bf16 (autocast inference: torch runs group_norm in fp32 between a bf16 -> fp32 cast of the convolution's output
// and the fp32 -> bf16 cast in front of the next convolution; here the bf16 tensor is read and written directly, fp32 arithmetic,
// one rounding at the same place)
// NHWC: x and y are channels-last ([N][HW][C], the layout the inference backbone / neck hand over and MIOpen's bf16 kernels
// work in: a tower that stays channels-last needs none of MIOpen's layout-conversion launches around its convolutions).
template <typename T, bool NHWC = false>
__global__ __launch_bounds__(kGnThreads) void gn_act_forward(const T *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps, int relu,
                                                             T *__restrict__ y, float *__restrict__ mean_out,
                                                             float *__restrict__ rstd_out, int C, int G, int HW) {
  __shared__ float red[16];
  const int n = blockIdx.x / G, g = blockIdx.x % G, D = C / G;
  const long long base = ((long long)n * C + (long long)g * D) * HW;      // of the group in [N][C][HW]
  const long long xbase = NHWC ? (long long)n * HW * C + (long long)g * D : base;
  const int total = D * HW;
  // element i of the group: NHWC walks the D channels of a pixel first (contiguous in x)
  auto x_at = [&](int i) { return NHWC ? xbase + (long long)(i / D) * C + (i % D) : xbase + i; };
  if constexpr (NHWC && sizeof(T) == 2) {
    if (D == 8 && C % 8 == 0) {      // the group's 8 bf16 channels of a pixel: one 16-byte access (thread = pixel)
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      auto px = [&](int p) { return *reinterpret_cast<const u32x4 *>(x + xbase + (long long)p * C); };
      float s8 = 0.f;
      for (int p = threadIdx.x; p < HW; p += kGnThreads) {
        const u32x4 u = px(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) s8 += __uint_as_float(u[k] << 16) + __uint_as_float(u[k] & 0xffff0000u);
      }
      const float mean8 = gn_block_sum(s8, red) / (float)total;
      float q8 = 0.f;
      for (int p = threadIdx.x; p < HW; p += kGnThreads) {
        const u32x4 u = px(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = __uint_as_float(u[k] << 16) - mean8, b = __uint_as_float(u[k] & 0xffff0000u) - mean8;
          q8 += a * a + b * b;
        }
      }
      const float rstd8 = 1.0f / sqrtf(gn_block_sum(q8, red) / (float)total + eps);
      if (threadIdx.x == 0 && mean_out) {
        mean_out[blockIdx.x] = mean8;
        rstd_out[blockIdx.x] = rstd8;
      }
      float ga[8], be[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        ga[d] = gamma ? gamma[g * 8 + d] : 1.0f;
        be[d] = beta ? beta[g * 8 + d] : 0.0f;
      }
      for (int p = threadIdx.x; p < HW; p += kGnThreads) {
        const u32x4 u = px(p);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v0 = (__uint_as_float(u[k] << 16) - mean8) * rstd8 * ga[2 * k] + be[2 * k];
          float v1 = (__uint_as_float(u[k] & 0xffff0000u) - mean8) * rstd8 * ga[2 * k + 1] + be[2 * k + 1];
          if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
          o[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
        }
        *reinterpret_cast<u32x4 *>(y + xbase + (long long)p * C) = o;
      }
      return;
    }
  }
  float s = 0.f;
  for (int i = threadIdx.x; i < total; i += kGnThreads) s += (float)x[x_at(i)];
  const float mean = gn_block_sum(s, red) / (float)total;
  float q = 0.f;
  for (int i = threadIdx.x; i < total; i += kGnThreads) {
    const float d = (float)x[x_at(i)] - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(gn_block_sum(q, red) / (float)total + eps);
  if (threadIdx.x == 0 && mean_out) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = rstd;
  }
  for (int i = threadIdx.x; i < total; i += kGnThreads) {
    const int c = g * D + (NHWC ? i % D : i / HW);
    const long long at = x_at(i);
    float v = ((float)x[at] - mean) * rstd * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f);
    if (relu) v = fmaxf(v, 0.0f);
    y[at] = (T)v;
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

// ---- large groups (config 5's 100 x 168 level: 8 channels x 16800 pixels per group, 64 (image, group) pairs): the pixels of
// an (image, group) are cut into S slices, one workgroup each.  Forward: slice moments (mean, centred squares; two passes over
// the slice, the second one from L2) combined with the parallel-variance formula by every workgroup of the normalising pass.
// Backward: per-channel (ds, db) of every slice, added in slice order by every workgroup of the pass that writes grad_x.
// grid = (N * G, S).
__device__ __forceinline__ void gn_slice(int HW, int S, int s, int &p0, int &p1) {
  const int per = (HW + S - 1) / S;
  p0 = min(HW, s * per);
  p1 = min(HW, p0 + per);
}

// (T = __bf16, NHWC: the inference towers' largest level -- 8 channels x 16800 pixels per group -- which the one-workgroup kernel
//  does not take; element (d, p) of the group at x[n][p][g D + d])
template <typename T = float, bool NHWC = false>
__global__ __launch_bounds__(kGnThreads) void gn_split_moments(const T *__restrict__ x, float *__restrict__ parts, int C,
                                                               int G, int HW, int S) {
  __shared__ float red[16];
  const int n = blockIdx.x / G, g = blockIdx.x % G, D = C / G;
  const long long base = NHWC ? (long long)n * HW * C + (long long)g * D : ((long long)n * C + (long long)g * D) * HW;
  auto at = [&](int d, int p) { return NHWC ? base + (long long)p * C + d : base + (long long)d * HW + p; };
  int p0, p1;
  gn_slice(HW, S, blockIdx.y, p0, p1);
  const float cnt = (float)D * (float)(p1 - p0);
  float sum = 0.f, q = 0.f;
  float mean;
  if (NHWC && D == 8 && sizeof(T) == 2 && C % 8 == 0) {
    // the usual case (256 channels in 32 groups, bf16): the group's 8 channels of a pixel are ONE 16-byte load
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto px = [&](int p) { return *reinterpret_cast<const u32x4 *>(x + base + (long long)p * C); };
    auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
    auto hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
    for (int p = p0 + threadIdx.x; p < p1; p += kGnThreads) {
      const u32x4 u = px(p);
#pragma unroll
      for (int k = 0; k < 4; ++k) sum += lo(u[k]) + hi(u[k]);
    }
    mean = cnt > 0.f ? gn_block_sum(sum, red) / cnt : 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += kGnThreads) {
      const u32x4 u = px(p);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = lo(u[k]) - mean, b = hi(u[k]) - mean;
        q += a * a + b * b;
      }
    }
  } else if (NHWC) {      // the D channels of a pixel are contiguous: thread = (pixel, channel)
    const int ni = (p1 - p0) * D;
    for (int i = threadIdx.x; i < ni; i += kGnThreads) sum += (float)x[at(i % D, p0 + i / D)];
    mean = cnt > 0.f ? gn_block_sum(sum, red) / cnt : 0.f;
    for (int i = threadIdx.x; i < ni; i += kGnThreads) {
      const float v = (float)x[at(i % D, p0 + i / D)] - mean;
      q += v * v;
    }
  } else {
    for (int d = 0; d < D; ++d)
      for (int p = p0 + threadIdx.x; p < p1; p += kGnThreads) sum += (float)x[at(d, p)];
    mean = cnt > 0.f ? gn_block_sum(sum, red) / cnt : 0.f;
    for (int d = 0; d < D; ++d)
      for (int p = p0 + threadIdx.x; p < p1; p += kGnThreads) {
        const float v = (float)x[at(d, p)] - mean;
        q += v * v;
      }
  }
  q = gn_block_sum(q, red);
  if (threadIdx.x == 0) {
    float *dst = parts + ((long long)blockIdx.x * S + blockIdx.y) * 2;
    dst[0] = mean;
    dst[1] = q;
  }
}

template <typename T = float, bool NHWC = false>
__global__ __launch_bounds__(kGnThreads) void gn_split_forward(const T *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, float eps, int relu,
                                                               T *__restrict__ y, float *__restrict__ mean_out,
                                                               float *__restrict__ rstd_out, const float *__restrict__ parts,
                                                               int C, int G, int HW, int S) {
  const int n = blockIdx.x / G, g = blockIdx.x % G, D = C / G;
  const long long base = NHWC ? (long long)n * HW * C + (long long)g * D : ((long long)n * C + (long long)g * D) * HW;
  const float *pp = parts + (long long)blockIdx.x * S * 2;
  const float total = (float)D * (float)HW;
  float mean = 0.f;
  for (int s = 0; s < S; ++s) {
    int a, b;
    gn_slice(HW, S, s, a, b);
    mean += pp[2 * s] * ((float)D * (float)(b - a));
  }
  mean /= total;
  float m2 = 0.f;
  for (int s = 0; s < S; ++s) {
    int a, b;
    gn_slice(HW, S, s, a, b);
    const float dm = pp[2 * s] - mean;
    m2 += pp[2 * s + 1] + (float)D * (float)(b - a) * dm * dm;
  }
  const float rstd = 1.0f / sqrtf(m2 / total + eps);
  if (threadIdx.x == 0 && blockIdx.y == 0 && mean_out) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = rstd;
  }
  int p0, p1;
  gn_slice(HW, S, blockIdx.y, p0, p1);
  if (NHWC && D == 8 && sizeof(T) == 2 && C % 8 == 0) {      // 16 bytes = the group's 8 bf16 channels of a pixel
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float sc[8], sh[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int c = g * 8 + d;
      sc[d] = rstd * (gamma ? gamma[c] : 1.0f);
      sh[d] = (beta ? beta[c] : 0.0f) - mean * sc[d];
    }
    for (int p = p0 + threadIdx.x; p < p1; p += kGnThreads) {
      const long long a = base + (long long)p * C;
      const u32x4 u = *reinterpret_cast<const u32x4 *>(x + a);
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v0 = __uint_as_float(u[k] << 16) * sc[2 * k] + sh[2 * k];
        float v1 = __uint_as_float(u[k] & 0xffff0000u) * sc[2 * k + 1] + sh[2 * k + 1];
        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
        o[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
      }
      *reinterpret_cast<u32x4 *>(y + a) = o;
    }
    return;
  }
  if (NHWC) {      // the D channels of a pixel are contiguous: thread = (pixel, channel)
    const int cnt = (p1 - p0) * D;
    for (int i = threadIdx.x; i < cnt; i += kGnThreads) {
      const int p = p0 + i / D, d = i % D, c = g * D + d;
      const float sc = rstd * (gamma ? gamma[c] : 1.0f), sh = (beta ? beta[c] : 0.0f) - mean * sc;
      const long long a = base + (long long)p * C + d;
      float v = (float)x[a] * sc + sh;
      if (relu) v = fmaxf(v, 0.0f);
      y[a] = (T)v;
    }
    return;
  }
  for (int d = 0; d < D; ++d) {
    const int c = g * D + d;
    const float sc = rstd * (gamma ? gamma[c] : 1.0f), sh = (beta ? beta[c] : 0.0f) - mean * sc;
    const long long cb = base + (long long)d * HW;
    for (int p = p0 + threadIdx.x; p < p1; p += kGnThreads) {
      float v = (float)x[cb + p] * sc + sh;
      if (relu) v = fmaxf(v, 0.0f);
      y[cb + p] = (T)v;
    }
  }
}

// parts: [N * G][S][2][D]  (ds, then db, of the slice)
__global__ __launch_bounds__(kGnThreads) void gn_split_bwd_sums(const float *__restrict__ gy, const float *__restrict__ x,
                                                                const float *__restrict__ y, int relu,
                                                                float *__restrict__ parts, int C, int G, int HW, int S) {
  __shared__ float part_ds[64][16], part_db[64][16];   // [channel][wave slice]
  const int n = blockIdx.x / G, g = blockIdx.x % G, D = C / G;
  const long long base = ((long long)n * C + (long long)g * D) * HW;
  int p0, p1;
  gn_slice(HW, S, blockIdx.y, p0, p1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpc = D <= 16 ? 16 / D : 1;
  for (int d0 = 0; d0 < D; d0 += 16 / wpc) {
    const int d = d0 + wave / wpc, sub = wave % wpc;
    if (d < D && wave < (16 / wpc) * wpc) {
      const long long cb = base + (long long)d * HW;
      float ds = 0.f, db = 0.f;
      for (int i = p0 + sub * 64 + lane; i < p1; i += wpc * 64) {
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
    float *dst = parts + ((long long)blockIdx.x * S + blockIdx.y) * 2 * D;
    dst[threadIdx.x] = ds;
    dst[D + threadIdx.x] = db;
  }
}

__global__ __launch_bounds__(kGnThreads) void gn_split_backward(const float *__restrict__ gy, const float *__restrict__ x,
                                                                const float *__restrict__ y, const float *__restrict__ gamma,
                                                                const float *__restrict__ mean_in,
                                                                const float *__restrict__ rstd_in, int relu,
                                                                float *__restrict__ gx, float *__restrict__ dgb,
                                                                const float *__restrict__ parts, int N, int C, int G, int HW,
                                                                int S) {
  __shared__ float ch_ds[64], ch_db[64];
  const int n = blockIdx.x / G, g = blockIdx.x % G, D = C / G;
  const long long base = ((long long)n * C + (long long)g * D) * HW;
  const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
  if ((int)threadIdx.x < D) {
    const float *pp = parts + (long long)blockIdx.x * S * 2 * D;
    float ds = 0.f, db = 0.f;
    for (int s = 0; s < S; ++s) { ds += pp[s * 2 * D + threadIdx.x]; db += pp[s * 2 * D + D + threadIdx.x]; }
    ch_ds[threadIdx.x] = ds;
    ch_db[threadIdx.x] = db;
    if (blockIdx.y == 0) {
      const int c = g * D + threadIdx.x;
      dgb[(long long)n * C + c] = (ds - mean * db) * rstd;
      dgb[((long long)N + n) * C + c] = db;
    }
  }
  __syncthreads();
  if (!gx) return;
  float S1 = 0.f, S2 = 0.f;
  for (int d = 0; d < D; ++d) {
    const float gm = gamma ? gamma[g * D + d] : 1.0f;
    S1 += gm * ch_db[d];
    S2 += gm * ch_ds[d];
  }
  const float inv = 1.0f / ((float)D * (float)HW);
  const float c2 = (S1 * mean - S2) * rstd * rstd * rstd * inv;
  const float c3 = -c2 * mean - S1 * rstd * inv;
  int p0, p1;
  gn_slice(HW, S, blockIdx.y, p0, p1);
  for (int d = 0; d < D; ++d) {
    const float gr = (gamma ? gamma[g * D + d] : 1.0f) * rstd;
    const long long cb = base + (long long)d * HW;
    for (int p = p0 + threadIdx.x; p < p1; p += kGnThreads) {
      float gv = gy[cb + p];
      if (relu && !(y[cb + p] > 0.0f)) gv = 0.0f;
      gx[cb + p] = gv * gr + c2 * x[cb + p] + c3;
    }
  }
}

}  // namespace kgdet

using namespace kgdet;

// slices per (image, group) of the split kernels (1: the one-workgroup kernels); scratch floats they need
namespace {
constexpr long long kGnSplitElems = 65536;   // beyond: one workgroup per (image, group) leaves most of the chip idle
constexpr int kGnMaxSlices = 16;
int gn_slices(int64_t N, int32_t C, int32_t groups, int64_t HW) {
  const long long elems = (long long)(C / groups) * HW;
  if (elems <= kGnSplitElems) return 1;
  long long S = (elems + 16383) / 16384;
  // (enough workgroups to fill the chip twice is plenty; never back to ONE slice: the one-workgroup kernels reject groups
  // beyond kGnSplitElems, and large batches -- N * groups >= 1024 -- would be sent there)
  while (S > 2 && N * groups * S > 1024) --S;
  return (int)(S > kGnMaxSlices ? kGnMaxSlices : S);
}
}  // namespace

extern "C" int32_t kgdet_gn_act_slices(int64_t N, int32_t C, int32_t groups, int64_t HW) {
  if (C <= 0 || groups <= 0 || C % groups) return 1;
  return gn_slices(N, C, groups, HW);
}
extern "C" size_t kgdet_gn_act_scratch_floats(int64_t N, int32_t C, int32_t groups, int64_t HW) {
  if (C <= 0 || groups <= 0 || C % groups) return 0;
  const int S = gn_slices(N, C, groups, HW);
  return S > 1 ? (size_t)N * groups * S * 2 * (size_t)(C / groups) : 0;
}

extern "C" int kgdet_gn_act_forward_split(const float *x, const float *gamma, const float *beta, int32_t groups, float eps,
                                          int32_t relu, float *y, float *mean, float *rstd, float *scratch, int64_t N,
                                          int32_t C, int64_t HW, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && groups > 0 && C % groups == 0 && HW >= 0 && HW < (1LL << 31) &&
                    (long long)(C / groups) * HW < (1LL << 31), "bad sizes");
  KGDET_CHECK_SHAPE(C / groups <= 64 && N * groups < (1LL << 31), "at most 64 channels per group");
  if (N * HW == 0) return KGDET_OK;
  const int S = gn_slices(N, C, groups, HW);
  KGDET_CHECK_SHAPE(x && y && mean && rstd && (S == 1 || scratch), "null pointer");
  if (S == 1) {
    hipLaunchKernelGGL(gn_act_forward<float>, dim3((unsigned)(N * groups)), dim3(kGnThreads), 0, (hipStream_t)stream, x, gamma, beta,
                       eps, relu, y, mean, rstd, C, groups, (int)HW);
  } else {
    hipLaunchKernelGGL((gn_split_moments<float, false>), dim3((unsigned)(N * groups), S), dim3(kGnThreads), 0, (hipStream_t)stream, x, scratch,
                       C, groups, (int)HW, S);
    hipLaunchKernelGGL((gn_split_forward<float, false>), dim3((unsigned)(N * groups), S), dim3(kGnThreads), 0, (hipStream_t)stream, x, gamma,
                       beta, eps, relu, y, mean, rstd, (const float *)scratch, C, groups, (int)HW, S);
  }
  KGDET_CHECK_LAUNCH("gn_act_forward_split");
  return KGDET_OK;
}

extern "C" int kgdet_gn_act_backward_split(const float *grad_y, const float *x, const float *y, const float *gamma,
                                           const float *mean, const float *rstd, int32_t groups, int32_t relu, float *grad_x,
                                           float *dgamma_dbeta, float *scratch, int64_t N, int32_t C, int64_t HW,
                                           void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && groups > 0 && C % groups == 0 && HW >= 0 && HW < (1LL << 31) &&
                    (long long)(C / groups) * HW < (1LL << 31), "bad sizes");
  KGDET_CHECK_SHAPE(C / groups <= 64 && N * groups < (1LL << 31), "at most 64 channels per group");
  if (N * HW == 0) return KGDET_OK;
  const int S = gn_slices(N, C, groups, HW);
  KGDET_CHECK_SHAPE(grad_y && x && mean && rstd && dgamma_dbeta && (!relu || y) && (S == 1 || scratch), "null pointer");
  if (S == 1) {
    hipLaunchKernelGGL(gn_act_backward, dim3((unsigned)(N * groups)), dim3(kGnThreads), 0, (hipStream_t)stream, grad_y, x, y,
                       gamma, mean, rstd, relu, grad_x, dgamma_dbeta, (int)N, C, groups, (int)HW);
  } else {
    hipLaunchKernelGGL(gn_split_bwd_sums, dim3((unsigned)(N * groups), S), dim3(kGnThreads), 0, (hipStream_t)stream, grad_y, x,
                       y, relu, scratch, C, groups, (int)HW, S);
    hipLaunchKernelGGL(gn_split_backward, dim3((unsigned)(N * groups), S), dim3(kGnThreads), 0, (hipStream_t)stream, grad_y, x,
                       y, gamma, mean, rstd, relu, grad_x, dgamma_dbeta, (const float *)scratch, (int)N, C, groups, (int)HW, S);
  }
  KGDET_CHECK_LAUNCH("gn_act_backward_split");
  return KGDET_OK;
}

// bf16 in, bf16 out, fp32 arithmetic, groups of ANY size: beyond 65536 elements the pixels of a group are cut into
// kgdet_gn_act_slices(...) slices (moments per slice, combined in slice order; `scratch` of kgdet_gn_act_scratch_floats(...) floats)
extern "C" int kgdet_gn_act_forward_bf16_split(const void *x, int32_t x_channels_last, const float *gamma, const float *beta,
                                               int32_t groups, float eps, int32_t relu, void *y, float *scratch, int64_t N,
                                               int32_t C, int64_t HW, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && groups > 0 && C % groups == 0 && HW >= 0 && HW < (1LL << 31) &&
                    (long long)(C / groups) * HW < (1LL << 31), "bad sizes");
  KGDET_CHECK_SHAPE(C / groups <= 64 && N * groups < (1LL << 31), "at most 64 channels per group");
  if (N * HW == 0) return KGDET_OK;
  const int S = gn_slices(N, C, groups, HW);
  KGDET_CHECK_SHAPE(x && y && (S == 1 || scratch), "null pointer");
  if (S == 1)
    return kgdet_gn_act_forward_bf16(x, x_channels_last, gamma, beta, groups, eps, relu, y, N, C, HW, stream);
  const dim3 grid((unsigned)(N * groups), S);
  if (x_channels_last) {
    hipLaunchKernelGGL((gn_split_moments<__bf16, true>), grid, dim3(kGnThreads), 0, (hipStream_t)stream, (const __bf16 *)x, scratch,
                       C, groups, (int)HW, S);
    hipLaunchKernelGGL((gn_split_forward<__bf16, true>), grid, dim3(kGnThreads), 0, (hipStream_t)stream, (const __bf16 *)x, gamma,
                       beta, eps, relu, (__bf16 *)y, (float *)nullptr, (float *)nullptr, (const float *)scratch, C, groups, (int)HW, S);
  } else {
    hipLaunchKernelGGL((gn_split_moments<__bf16, false>), grid, dim3(kGnThreads), 0, (hipStream_t)stream, (const __bf16 *)x, scratch,
                       C, groups, (int)HW, S);
    hipLaunchKernelGGL((gn_split_forward<__bf16, false>), grid, dim3(kGnThreads), 0, (hipStream_t)stream, (const __bf16 *)x, gamma,
                       beta, eps, relu, (__bf16 *)y, (float *)nullptr, (float *)nullptr, (const float *)scratch, C, groups, (int)HW, S);
  }
  KGDET_CHECK_LAUNCH("gn_act_forward_bf16_split");
  return KGDET_OK;
}

// bf16 in, bf16 out, fp32 arithmetic (inference under autocast); groups of at most 65536 elements
extern "C" int kgdet_gn_act_forward_bf16(const void *x, int32_t x_channels_last, const float *gamma, const float *beta,
                                         int32_t groups, float eps, int32_t relu, void *y, int64_t N, int32_t C, int64_t HW,
                                         void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && groups > 0 && C % groups == 0 && HW >= 0 && (long long)(C / groups) * HW <= kGnSplitElems,
                    "bad sizes (at most %lld elements per group)", kGnSplitElems);
  KGDET_CHECK_SHAPE(C / groups <= 64 && N * groups < (1LL << 31), "at most 64 channels per group");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x && y, "null pointer");
  if (x_channels_last)
    hipLaunchKernelGGL((gn_act_forward<__bf16, true>), dim3((unsigned)(N * groups)), dim3(kGnThreads), 0, (hipStream_t)stream,
                       (const __bf16 *)x, gamma, beta, eps, relu, (__bf16 *)y, (float *)nullptr, (float *)nullptr, C, groups, (int)HW);
  else
    hipLaunchKernelGGL((gn_act_forward<__bf16, false>), dim3((unsigned)(N * groups)), dim3(kGnThreads), 0, (hipStream_t)stream,
                       (const __bf16 *)x, gamma, beta, eps, relu, (__bf16 *)y, (float *)nullptr, (float *)nullptr, C, groups, (int)HW);
  KGDET_CHECK_LAUNCH("gn_act_forward_bf16");
  return KGDET_OK;
}

extern "C" int kgdet_gn_act_forward(const float *x, const float *gamma, const float *beta, int32_t groups, float eps,
                                    int32_t relu, float *y, float *mean, float *rstd, int64_t N, int32_t C, int64_t HW,
                                    void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && groups > 0 && C % groups == 0 && HW >= 0 && (long long)(C / groups) * HW < (1LL << 31),
                    "bad sizes");
  KGDET_CHECK_SHAPE(C / groups <= 64 && N * groups < (1LL << 31), "at most 64 channels per group");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x && y && mean && rstd, "null pointer");
  hipLaunchKernelGGL(gn_act_forward<float>, dim3((unsigned)(N * groups)), dim3(kGnThreads), 0, (hipStream_t)stream, x, gamma, beta, eps,
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
