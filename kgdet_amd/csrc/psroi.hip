// Deformable position-sensitive RoI pooling forward / backward for gfx950.
// Replaces DeformablePSROIPoolForwardKernel / BackwardAccKernel
// (mmdet/ops/dcn/src/deform_pool_cuda_kernel.cu:53-140, 143-263).  One thread per output bin
// (n, ctop, ph, pw); <= sample_per_part^2 bilinear samples per bin.  Backward accumulates with
// float atomics exactly where the reference does (feature grad: 4 per sample, trans grad: 2).
// The mixed float/double arithmetic of the reference (0.5, 0.1, 1. literals) is kept so that
// bin boundaries round the same way.
#include "common.h"

namespace kgdet {

namespace {

struct Roi {
  int batch;
  float w0, h0, rw, rh, bin_w, bin_h, sub_w, sub_h;
};

__device__ __forceinline__ Roi roi_setup(const float *roi, float scale, int P, int S) {
  Roi r;
  r.batch = (int)roi[0];
  r.w0 = (float)((float)(round(roi[1])) * scale - 0.5);
  r.h0 = (float)((float)(round(roi[2])) * scale - 0.5);
  const float w1 = (float)((float)(round(roi[3]) + 1.) * scale - 0.5);
  const float h1 = (float)((float)(round(roi[4]) + 1.) * scale - 0.5);
  r.rw = (float)fmax((double)(w1 - r.w0), 0.1);
  r.rh = (float)fmax((double)(h1 - r.h0), 0.1);
  r.bin_h = r.rh / (float)P;
  r.bin_w = r.rw / (float)P;
  r.sub_h = r.bin_h / (float)S;
  r.sub_w = r.bin_w / (float)S;
  return r;
}

struct Bin {
  int n, ctop, ph, pw, part_h, part_w, class_id, c;
  long long tix, tiy;
  float wstart, hstart;
  Roi r;
};

__device__ __forceinline__ Bin bin_setup(const kgdet_psroi_shape &s, long long idx, const float *rois,
                                         const float *trans) {
  Bin b;
  const int P = s.pooled_size;
  b.pw = (int)(idx % P);
  b.ph = (int)((idx / P) % P);
  b.ctop = (int)((idx / P / P) % s.out_dim);
  b.n = (int)(idx / P / P / s.out_dim);
  b.r = roi_setup(rois + 5 * b.n, s.spatial_scale, P, s.sample_per_part);
  b.part_h = (int)floorf((float)(b.ph) / P * s.part_size);
  b.part_w = (int)floorf((float)(b.pw) / P * s.part_size);
  const int ch_each_class = s.no_trans ? s.out_dim : s.out_dim / s.num_classes;
  b.class_id = b.ctop / ch_each_class;
  b.tix = (((long long)(b.n * s.num_classes + b.class_id) * 2) * s.part_size + b.part_h) * s.part_size + b.part_w;
  b.tiy = (((long long)(b.n * s.num_classes + b.class_id) * 2 + 1) * s.part_size + b.part_h) * s.part_size + b.part_w;
  const float tx = s.no_trans ? 0.0f : trans[b.tix] * s.trans_std;
  const float ty = s.no_trans ? 0.0f : trans[b.tiy] * s.trans_std;
  b.wstart = (float)(b.pw) * b.r.bin_w + b.r.w0;
  b.wstart += tx * b.r.rw;
  b.hstart = (float)(b.ph) * b.r.bin_h + b.r.h0;
  b.hstart += ty * b.r.rh;
  int gw = (int)floorf((float)(b.pw) * s.group_size / P);
  int gh = (int)floorf((float)(b.ph) * s.group_size / P);
  gw = min(max(gw, 0), s.group_size - 1);
  gh = min(max(gh, 0), s.group_size - 1);
  b.c = (b.ctop * s.group_size + gh) * s.group_size + gw;
  return b;
}

}  // namespace

__global__ __launch_bounds__(256) void psroi_forward(const kgdet_psroi_shape s, const float *__restrict__ data,
                                                     const float *__restrict__ rois, const float *__restrict__ trans,
                                                     float *__restrict__ out, float *__restrict__ count) {
  const long long total = (long long)s.R * s.out_dim * s.pooled_size * s.pooled_size;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const Bin b = bin_setup(s, idx, rois, trans);
    const float *plane = data + ((long long)b.r.batch * s.C + b.c) * s.H * s.W;
    float sum = 0.f;
    int cnt = 0;
    for (int ih = 0; ih < s.sample_per_part; ++ih)
      for (int iw = 0; iw < s.sample_per_part; ++iw) {
        float w = b.wstart + iw * b.r.sub_w;
        float h = b.hstart + ih * b.r.sub_h;
        if (w < -0.5 || w > s.W - 0.5 || h < -0.5 || h > s.H - 0.5) continue;
        w = (float)fmin(fmax((double)w, 0.), s.W - 1.);
        h = (float)fmin(fmax((double)h, 0.), s.H - 1.);
        const int xa = (int)floorf(w), xb = (int)ceilf(w), ya = (int)floorf(h), yb = (int)ceilf(h);
        const float fx = w - xa, fy = h - ya;
        const float v = (1 - fx) * (1 - fy) * plane[ya * s.W + xa] + (1 - fx) * fy * plane[yb * s.W + xa] +
                        fx * (1 - fy) * plane[ya * s.W + xb] + fx * fy * plane[yb * s.W + xb];
        sum += v;
        cnt++;
      }
    out[idx] = cnt == 0 ? 0.0f : sum / cnt;
    count[idx] = (float)cnt;
  }
}

__global__ __launch_bounds__(256) void psroi_backward(const kgdet_psroi_shape s, const float *__restrict__ grad_out,
                                                      const float *__restrict__ count, const float *__restrict__ data,
                                                      const float *__restrict__ rois, const float *__restrict__ trans,
                                                      float *__restrict__ grad_data, float *__restrict__ grad_trans) {
  const long long total = (long long)s.R * s.out_dim * s.pooled_size * s.pooled_size;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    if (count[idx] <= 0) continue;
    const Bin b = bin_setup(s, idx, rois, trans);
    const float diff = grad_out[idx] / count[idx];
    const float *plane = data + ((long long)b.r.batch * s.C + b.c) * s.H * s.W;
    float *gplane = grad_data + ((long long)b.r.batch * s.C + b.c) * s.H * s.W;
    for (int ih = 0; ih < s.sample_per_part; ++ih)
      for (int iw = 0; iw < s.sample_per_part; ++iw) {
        float w = b.wstart + iw * b.r.sub_w;
        float h = b.hstart + ih * b.r.sub_h;
        if (w < -0.5 || w > s.W - 0.5 || h < -0.5 || h > s.H - 0.5) continue;
        w = (float)fmin(fmax((double)w, 0.), s.W - 1.);
        h = (float)fmin(fmax((double)h, 0.), s.H - 1.);
        const int xa = (int)floorf(w), xb = (int)ceilf(w), ya = (int)floorf(h), yb = (int)ceilf(h);
        const float fx = w - xa, fy = h - ya;
        atomicAdd(gplane + ya * s.W + xa, (1 - fx) * (1 - fy) * diff);
        atomicAdd(gplane + yb * s.W + xa, (1 - fx) * fy * diff);
        atomicAdd(gplane + ya * s.W + xb, fx * (1 - fy) * diff);
        atomicAdd(gplane + yb * s.W + xb, fx * fy * diff);
        if (s.no_trans) continue;
        const float U00 = plane[ya * s.W + xa], U01 = plane[yb * s.W + xa];
        const float U10 = plane[ya * s.W + xb], U11 = plane[yb * s.W + xb];
        float dx = (U11 * fy + U10 * (1 - fy) - U01 * fy - U00 * (1 - fy)) * s.trans_std * diff;
        dx *= b.r.rw;
        float dy = (U11 * fx + U01 * (1 - fx) - U10 * fx - U00 * (1 - fx)) * s.trans_std * diff;
        dy *= b.r.rh;
        atomicAdd(grad_trans + b.tix, dx);
        atomicAdd(grad_trans + b.tiy, dy);
      }
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" {

static int psroi_check(const kgdet_psroi_shape *s) {
  KGDET_CHECK_SHAPE(s != nullptr, "null shape");
  KGDET_CHECK_SHAPE(s->pooled_size > 0 && s->part_size > 0 && s->sample_per_part > 0 && s->group_size > 0 &&
                        s->out_dim > 0, "bad pooling geometry");
  KGDET_CHECK_SHAPE(s->trans_std >= 0.0f && s->trans_std <= 1.0f, "trans_std must be in [0, 1]");
  KGDET_CHECK_SHAPE(s->no_trans || (s->num_classes > 0 && s->out_dim % s->num_classes == 0),
                    "out_dim must divide num_classes");
  KGDET_CHECK_SHAPE(s->out_dim * s->group_size * s->group_size <= s->C,
                    "input has %d channels, position-sensitive pooling needs %d", s->C,
                    s->out_dim * s->group_size * s->group_size);
  return KGDET_OK;
}

int kgdet_deform_psroi_forward(const kgdet_psroi_shape *s, const float *data, const float *rois, const float *trans,
                               float *out, float *count, void *stream) {
  if (int rc = psroi_check(s)) return rc;
  const long long total = (long long)s->R * s->out_dim * s->pooled_size * s->pooled_size;
  if (total == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(data && rois && out && count && (s->no_trans || trans), "null pointer");
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(psroi_forward, dim3(grid), dim3(256), 0, (hipStream_t)stream, *s, data, rois, trans, out, count);
  KGDET_CHECK_LAUNCH("psroi_forward");
  return KGDET_OK;
}

int kgdet_deform_psroi_backward(const kgdet_psroi_shape *s, const float *grad_out, const float *count,
                                const float *data, const float *rois, const float *trans, float *grad_data,
                                float *grad_trans, void *stream) {
  if (int rc = psroi_check(s)) return rc;
  const long long total = (long long)s->R * s->out_dim * s->pooled_size * s->pooled_size;
  if (total == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(grad_out && count && data && rois && grad_data && (s->no_trans || (trans && grad_trans)),
                    "null pointer");
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(psroi_backward, dim3(grid), dim3(256), 0, (hipStream_t)stream, *s, grad_out, count, data, rois,
                     trans, grad_data, grad_trans);
  KGDET_CHECK_LAUNCH("psroi_backward");
  return KGDET_OK;
}

}  // extern "C"
