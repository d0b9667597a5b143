// Deformable position-sensitive RoI pooling for gfx950 -- forward, grad_trans, grad_data.
// Computes what DeformablePSROIPoolForwardKernel / BackwardAccKernel compute
// (mmdet/ops/dcn/src/deform_pool_cuda_kernel.cu:53-140, 143-263) with a different decomposition:
//
//  * The SAMPLING GEOMETRY of an RoI (P*P bins x S*S samples: two corner columns, two corner rows, two bilinear
//    fractions) depends on (RoI, offset class) only, not on the channel.  A workgroup evaluates it ONCE, cooperatively,
//    into 16-byte records in LDS; every channel of the class then reads the records (LDS broadcast, wave-uniform) instead
//    of redoing ~40 VALU instructions per sample and channel.
//  * COALESCED GATHERS: the feature map is re-laid-out once per call (psroi_to_cell_major, ~6 us for 8.6 MB) as
//    dataT[image][group cell][pixel][output channel] -- channels-last for group_size 1 -- so that lanes = 64 consecutive
//    output channels read ONE contiguous 256-byte piece per bilinear corner.  The reference's mapping (a thread per bin,
//    NCHW) makes every corner a scattered 4-byte read; measured on [2,256,50,84] x 512 RoIs x 7x7 bins the scattered form
//    runs at the texture addresser's ~2 lanes per clock and CU (115 us at 2x2, 359 us at 4x4 samples per bin), and
//    staging per-RoI feature windows in LDS (built and measured: 167 / 364 us) only moves the scattered reads to the
//    staging loop.
//  * psroi_gather<0> (forward): workgroup = (RoI, offset class, 64 output channels); its four waves split the bins; results
//    leave through an LDS staging buffer as one contiguous block.  psroi_gather<1> (grad_trans): workgroup = (RoI, class)
//    walks all channels of the class, per-lane partial sums, ONE fixed-order wave reduction per bin: no atomics.
//  * psroi_grad_data: a GATHER by output tile.  One single-wave workgroup per (image, 4x6-pixel tile, 64 input channels) keeps the
//    tile's gradient in LDS, walks the RoIs of its image in index order (culled by a per-RoI bounding box; the sample
//    records come prepared from psroi_prepare, one visit ahead), and thread = channel adds the in-tile corners of every
//    sample in the reference's own serial order (RoI, bin row, bin column, sample row, sample column, corner).  Each
//    (channel, pixel) sum has exactly one writer: NO atomics, bit-repeatable, no pre-zeroed output, and the float result
//    equals a serial CPU evaluation of the reference loop bit for bit (this file is compiled without fp contraction for
//    that reason).  psroi_prepare also writes grad_out / count transposed to [RoI][bin][channel] so the per-bin read is
//    coalesced.
//
// The mixed float / double arithmetic of the reference (0.5, 0.1, 1. literals) is kept so that bin boundaries round the
// same way.  Envelope of the LDS record table: pooled_size^2 * sample_per_part^2 <= 2048 records, group_size <= 16.
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

#pragma clang fp contract(off)

namespace kgdet {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxRecords = 2048;
constexpr int kMaxGroup = 16;
// grad_data tile (round 4, same shape as profiles/r03_psroi.md, S = 2): 2 x 6 570 us, 4 x 3 580, 4 x 6 616 (rounds 2-3), 8 x 4 711,
// 2 x 3 725 -- the kernel's time is nearly independent of the tile (total visits x per-visit work ~ constant), i.e. it is
// not occupancy-bound as round 3 assumed
#ifndef KGDET_PSROI_TILE_H
#define KGDET_PSROI_TILE_H 2
#define KGDET_PSROI_TILE_W 6
#endif
constexpr int kTileH = KGDET_PSROI_TILE_H, kTileW = KGDET_PSROI_TILE_W;
constexpr int kBinBatch = 16;          // bins whose quotients psroi_grad_data fetches together
constexpr int kListCap = 4096;         // RoIs per overlap-list segment of psroi_grad_data
constexpr int kTilePix = kTileH * kTileW;
constexpr int kGdThreads = 64;         // psroi_grad_data: one wave per workgroup
constexpr int kGdStride = kGdThreads + 1;

struct Roi {
  int batch;
  float w0, h0, rw, rh, bin_w, bin_h, sub_w, sub_h;
};

// deform_pool_cuda_kernel.cu:76-97
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

// one bilinear sample: corner columns / rows and fractions; xa < 0: outside the map (the reference's `continue`)
struct __attribute__((aligned(16))) Rec {
  short xa, xb, ya, yb;
  float fx, fy;
};

__device__ __forceinline__ int group_cell(int p, int G, int P) {   // :101-106
  const int g = (int)floorf((float)(p)*G / P);
  return min(max(g, 0), G - 1);
}

__device__ __forceinline__ long long trans_index(const kgdet_psroi_shape &s, int n, int cls, int dir, int ph, int pw) {
  const int part_h = (int)floorf((float)(ph) / s.pooled_size * s.part_size);
  const int part_w = (int)floorf((float)(pw) / s.pooled_size * s.part_size);
  return (((long long)(n * s.num_classes + cls) * 2 + dir) * s.part_size + part_h) * s.part_size + part_w;
}

// sample (ih, iw) of bin (ph, pw) of RoI r under the offsets of class `cls` (:99-134)
__device__ __forceinline__ Rec make_record(const kgdet_psroi_shape &s, const Roi &r, int n, int cls, int ph, int pw,
                                           int ih, int iw, const float *__restrict__ trans) {
  const float tx = s.no_trans ? 0.0f : trans[trans_index(s, n, cls, 0, ph, pw)] * s.trans_std;
  const float ty = s.no_trans ? 0.0f : trans[trans_index(s, n, cls, 1, ph, pw)] * s.trans_std;
  float wstart = (float)(pw)*r.bin_w + r.w0;
  wstart += tx * r.rw;
  float hstart = (float)(ph)*r.bin_h + r.h0;
  hstart += ty * r.rh;
  float w = wstart + iw * r.sub_w;
  float h = hstart + ih * r.sub_h;
  Rec q;
  if (w < -0.5 || w > s.W - 0.5 || h < -0.5 || h > s.H - 0.5) {
    q.xa = q.xb = q.ya = q.yb = -1;
    q.fx = q.fy = 0.f;
    return q;
  }
  w = (float)fmin(fmax((double)w, 0.), s.W - 1.);
  h = (float)fmin(fmax((double)h, 0.), s.H - 1.);
  const int xa = (int)floorf(w), xb = (int)ceilf(w), ya = (int)floorf(h), yb = (int)ceilf(h);
  q.xa = (short)xa; q.xb = (short)xb; q.ya = (short)ya; q.yb = (short)yb;
  q.fx = w - xa;
  q.fy = h - ya;
  return q;
}

__device__ __forceinline__ Rec lds_record(const Rec *recs, int i) {
  const int4 v = *reinterpret_cast<const int4 *>(recs + i);      // one ds_read_b128
  Rec q;
  q.xa = (short)(v.x & 0xffff); q.xb = (short)(v.x >> 16);
  q.ya = (short)(v.y & 0xffff); q.yb = (short)(v.y >> 16);
  q.fx = __int_as_float(v.z); q.fy = __int_as_float(v.w);
  return q;
}

// first list of (image, offset class, group cell of bin (ph, pw)); lists are indexed [image][class][cell][pixel]
__device__ __forceinline__ long long list_base(const kgdet_psroi_shape &s, int batch, int classes, int cls, int ph, int pw) {
  const int G = s.group_size;
  const int k = group_cell(ph, G, s.pooled_size) * G + group_cell(pw, G, s.pooled_size);
  return (((long long)batch * classes + cls) * G * G + k) * s.H * s.W;
}

// This lane's slot in list L (live lanes only): lanes of a wave that name the same list share ONE atomic and take consecutive
// slots (up to four distinct lists per call are served that way, the remaining lanes one atomic each).  A tiny RoI puts
// hundreds of samples on one pixel: without this its corners serialise on a single counter.  Called by ALL lanes of the wave.
__device__ __forceinline__ int list_take(int *__restrict__ cnt, int L, bool live) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  int slot = 0;
  bool pending = live;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned long long act = __ballot(pending);
    if (act == 0ull) break;                                   // (uniform)
    const int leader = __ffsll((long long)act) - 1;
    const int Lf = __shfl(L, leader);
    const bool mine = pending && L == Lf;
    const unsigned long long m = __ballot(mine);
    int base = 0;
    if (lane == leader) base = atomicAdd(cnt + Lf, __popcll(m));
    base = __shfl(base, leader);
    if (mine) {
      slot = base + __popcll(m & below);
      pending = false;
    }
  }
  if (pending) slot = atomicAdd(cnt + L, 1);
  return slot;
}

}  // namespace

// data [B, C, H, W] -> dataT[b][k][pixel][ctop] with input channel c = ctop * G*G + k (k = group cell): lanes = output
// channels become contiguous.  Block = (64 pixels, 64 output channels, b * G*G + k); 64 x 65 LDS tile.
__global__ __launch_bounds__(kThreads) void psroi_to_cell_major(const kgdet_psroi_shape s, const float *__restrict__ data,
                                                                float *__restrict__ dataT) {
  __shared__ float tile[64 * 65];
  const int GG = s.group_size * s.group_size, HW = s.H * s.W;
  const int b = blockIdx.z / GG, k = blockIdx.z - b * GG;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int e = tid; e < 64 * 64; e += kThreads) {
    const int c = e >> 6, p = e & 63;
    float v = 0.f;
    if (c0 + c < s.out_dim && p0 + p < HW) v = data[((long long)b * s.C + (long long)(c0 + c) * GG + k) * HW + p0 + p];
    tile[c * 65 + p] = v;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += kThreads) {
    const int p = e >> 6, c = e & 63;
    if (c0 + c < s.out_dim && p0 + p < HW)
      dataT[(((long long)b * GG + k) * HW + p0 + p) * s.out_dim + c0 + c] = tile[c * 65 + p];
  }
}

// MODE 0: out / count; block = (RoI, offset class, chunk of 64 output channels).
// MODE 1: grad_trans; block = (RoI, offset class), all channels of the class, diffT = grad_out / count as [RoI][bin][ctop].
// Record of the gather kernels: element offsets of the four corners inside one (image, cell) plane of dataT (o00 < 0: the
// sample lies outside the map) and, MODE 0, the four bilinear weights / MODE 1, the two fractions.  Wave-uniform: read by
// every lane as an LDS broadcast, moved to SGPRs, so that a corner load is `global_load v, v_lane_offset, s[base + o]`
// with no vector address arithmetic.
struct __attribute__((aligned(16))) GRec {
  int o00, o01, o10, o11;       // (ya, xa), (yb, xa), (ya, xb), (yb, xb)
  float a, b, c, d;             // MODE 0: w00, w01, w10, w11;  MODE 1: fx, fy, -, -
};

template <int MODE>
__global__ __launch_bounds__(kThreads) void psroi_gather(const kgdet_psroi_shape s, const float *__restrict__ dataT,
                                                         const float *__restrict__ rois, const float *__restrict__ trans,
                                                         float *__restrict__ out, float *__restrict__ count,
                                                         const float *__restrict__ diffT, float *__restrict__ grad_trans,
                                                         int classes, int ch_per_class, int chunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int P = s.pooled_size, S = s.sample_per_part, G = s.group_size;
  const int PP = P * P, SS = S * S, GG = G * G, nrec = PP * SS;
  GRec *recs = reinterpret_cast<GRec *>(smem);
  int *bin_cnt = reinterpret_cast<int *>(recs + nrec);
  float *stage = reinterpret_cast<float *>(bin_cnt + PP);           // MODE 0: [64][PP] results; MODE 1: [2][PP] bin sums

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int blk = blockIdx.x;
  const int chunk = blk % chunks; blk /= chunks;
  const int cls = blk % classes;
  const int n = blk / classes;
  const Roi r = roi_setup(rois + 5 * n, s.spatial_scale, P, S);
  const bool roi_ok = r.batch >= 0 && r.batch < s.B;
  const int HW = s.H * s.W;

  for (int b = tid; b < PP; b += kThreads) bin_cnt[b] = 0;
  __syncthreads();
  for (int i = tid; i < nrec; i += kThreads) {
    const int bin = i / SS, si = i - bin * SS;
    const int ph = bin / P, pw = bin - ph * P;
    const int ih = si / S, iw = si - ih * S;
    const Rec q = make_record(s, r, n, cls, ph, pw, ih, iw, trans);
    GRec g;
    if (q.xa < 0 || !roi_ok) {
      g.o00 = -1; g.o01 = g.o10 = g.o11 = 0;
      g.a = g.b = g.c = g.d = 0.f;
    } else {
      g.o00 = (q.ya * s.W + q.xa) * s.out_dim; g.o01 = (q.yb * s.W + q.xa) * s.out_dim;
      g.o10 = (q.ya * s.W + q.xb) * s.out_dim; g.o11 = (q.yb * s.W + q.xb) * s.out_dim;
      if (MODE == 0) {
        g.a = (1 - q.fx) * (1 - q.fy); g.b = (1 - q.fx) * q.fy; g.c = q.fx * (1 - q.fy); g.d = q.fx * q.fy;
      } else {
        g.a = q.fx; g.b = q.fy; g.c = g.d = 0.f;
      }
      atomicAdd(bin_cnt + bin, 1);
    }
    recs[i] = g;
  }
  __syncthreads();
  const float *img = dataT + (long long)(roi_ok ? r.batch : 0) * GG * HW * s.out_dim;
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto unif = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };

  if (MODE == 0) {
    const int c0 = cls * ch_per_class + chunk * 64;
    const int nch = min(64, (cls + 1) * ch_per_class - c0);
    const int ctop = c0 + min(lane, nch - 1);                 // (idle lanes repeat the last channel: no divergent loads)
    for (int bin = wave; bin < PP; bin += kThreads / 64) {
      const int ph = bin / P, pw = bin - ph * P;
      const int k = group_cell(ph, G, P) * G + group_cell(pw, G, P);
      const int voff = k * HW * s.out_dim + ctop;             // the lane's part of every corner address of this bin
      float sum = 0.f;
      // four samples per round: their sixteen corner loads are issued together (branch-free: a sample outside the map
      // reads element 0 and is dropped by a select), then summed in sample order
      for (int si0 = 0; si0 < SS; si0 += 4) {
        float u00[4], u01[4], u10[4], u11[4], wa[4], wb[4], wc[4], wd[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int4 qo = *reinterpret_cast<const int4 *>(recs + bin * SS + min(si0 + u, SS - 1));
          const float4 qw = *(reinterpret_cast<const float4 *>(recs + bin * SS + min(si0 + u, SS - 1)) + 1);
          const int o00 = uni(qo.x);
          ok[u] = si0 + u < SS && o00 >= 0;
          u00[u] = (img + max(o00, 0))[voff]; u01[u] = (img + uni(qo.y))[voff];
          u10[u] = (img + uni(qo.z))[voff];   u11[u] = (img + uni(qo.w))[voff];
          wa[u] = unif(qw.x); wb[u] = unif(qw.y); wc[u] = unif(qw.z); wd[u] = unif(qw.w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float v = wa[u] * u00[u] + wb[u] * u01[u] + wc[u] * u10[u] + wd[u] * u11[u];   // :37-49
          sum = ok[u] ? sum + v : sum;
        }
      }
      const int cnt = bin_cnt[bin];
      if (lane < nch) stage[lane * PP + bin] = cnt == 0 ? 0.0f : sum / cnt;
    }
    __syncthreads();
    const long long base = ((long long)n * s.out_dim + c0) * PP;
    for (int e = tid; e < nch * PP; e += kThreads) {
      out[base + e] = stage[e];
      count[base + e] = (float)bin_cnt[e % PP];
    }
  } else {
    const int c_begin = cls * ch_per_class;
    for (int bin = wave; bin < PP; bin += kThreads / 64) {
      const int ph = bin / P, pw = bin - ph * P;
      const int k = group_cell(ph, G, P) * G + group_cell(pw, G, P);
      float dx = 0.f, dy = 0.f;
      if (bin_cnt[bin] > 0) {
        for (int cc = 0; cc < ch_per_class; cc += 64) {
          const bool on = cc + lane < ch_per_class;
          const int ctop = c_begin + min(cc + lane, ch_per_class - 1);
          const int voff = k * HW * s.out_dim + ctop;
          const float diff = on ? diffT[((long long)n * PP + bin) * s.out_dim + ctop] : 0.f;
          for (int si0 = 0; si0 < SS; si0 += 4) {
            float u00[4], u01[4], u10[4], u11[4], fxs[4], fys[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int4 qo = *reinterpret_cast<const int4 *>(recs + bin * SS + min(si0 + u, SS - 1));
              const float4 qw = *(reinterpret_cast<const float4 *>(recs + bin * SS + min(si0 + u, SS - 1)) + 1);
              const int o00 = uni(qo.x);
              ok[u] = si0 + u < SS && o00 >= 0;
              u00[u] = (img + max(o00, 0))[voff]; u01[u] = (img + uni(qo.y))[voff];
              u10[u] = (img + uni(qo.z))[voff];   u11[u] = (img + uni(qo.w))[voff];
              fxs[u] = unif(qw.x); fys[u] = unif(qw.y);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float fx = fxs[u], fy = fys[u];
              float ddx = (u11[u] * fy + u10[u] * (1 - fy) - u01[u] * fy - u00[u] * (1 - fy)) * s.trans_std * diff;   // :253-258
              ddx *= r.rw;
              float ddy = (u11[u] * fx + u01[u] * (1 - fx) - u10[u] * fx - u00[u] * (1 - fx)) * s.trans_std * diff;
              ddy *= r.rh;
              dx = ok[u] ? dx + ddx : dx;
              dy = ok[u] ? dy + ddy : dy;
            }
          }
        }
      }
      // fixed-order butterfly over the 64 lanes
      for (int m = 32; m >= 1; m >>= 1) {
        dx += __shfl_xor(dx, m);
        dy += __shfl_xor(dy, m);
      }
      if (lane == 0) { stage[bin] = dx; stage[PP + bin] = dy; }
    }
    __syncthreads();
    // bins of one offset cell, row-major, into grad_trans[n, cls, dir, part_h, part_w]: every entry written once
    const int part = s.part_size;
    for (int e = tid; e < 2 * part * part; e += kThreads) {
      const int dir = e / (part * part), cell = e - dir * part * part;
      const int qh = cell / part, qw = cell - qh * part;
      float a = 0.f;
      for (int ph = 0; ph < P; ++ph) {
        if ((int)floorf((float)(ph) / P * part) != qh) continue;
        for (int pw = 0; pw < P; ++pw)
          if ((int)floorf((float)(pw) / P * part) == qw) a += stage[dir * PP + ph * P + pw];
      }
      grad_trans[(((long long)(n * s.num_classes + cls) * 2 + dir) * part + qh) * part + qw] = a;
    }
  }
}

// Backward preparation.  Block = RoI: the sample records of every offset class, recs[n][class][bin][sample], the
// bounding box of every valid sample corner over all classes (bbox[n] = {xmin, xmax, ymin, ymax}; xmax < 0: the
// RoI touches nothing) and, for grad_data's lists, the number of corners per (image, class, group cell, pixel).
__global__ __launch_bounds__(kThreads) void psroi_prepare(const kgdet_psroi_shape s, const float *__restrict__ rois,
                                                          const float *__restrict__ trans, int4 *__restrict__ bbox,
                                                          Rec *__restrict__ recs_out, int classes,
                                                          int *__restrict__ list_cnt /*nullable*/) {
  __shared__ int bb[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int P = s.pooled_size, S = s.sample_per_part, PP = P * P, SS = S * S;
  const Roi r = roi_setup(rois + 5 * n, s.spatial_scale, P, S);
  if (tid == 0) { bb[0] = INT_MAX; bb[1] = -1; bb[2] = INT_MAX; bb[3] = -1; }
  __syncthreads();
  if (r.batch >= 0 && r.batch < s.B) {      // (other RoIs are never visited: their records stay unwritten)
    int xmin = INT_MAX, xmax = -1, ymin = INT_MAX, ymax = -1;
    const int total = classes * PP * SS;
    for (int i0 = 0; i0 < total; i0 += kThreads) {       // (uniform trip count: list_take is a wave-wide operation)
      const int i = min(i0 + tid, total - 1);
      const bool in_range = i0 + tid < total;
      const int cls = i / (PP * SS), j = i - cls * PP * SS;
      const int bin = j / SS, si = j - bin * SS;
      const int ph = bin / P, pw = bin - ph * P, ih = si / S, iw = si - ih * S;
      const Rec q = make_record(s, r, n, cls, ph, pw, ih, iw, trans);
      if (in_range) recs_out[(long long)n * classes * PP * SS + i] = q;
      const bool live = in_range && q.xa >= 0;
      if (list_cnt) {     // entries of the per-pixel contribution lists (psroi_list_*): one per corner
        const int lb = (int)list_base(s, r.batch, classes, cls, ph, pw);
        list_take(list_cnt, lb + q.ya * s.W + q.xa, live); list_take(list_cnt, lb + q.yb * s.W + q.xa, live);
        list_take(list_cnt, lb + q.ya * s.W + q.xb, live); list_take(list_cnt, lb + q.yb * s.W + q.xb, live);
      }
      if (!live) continue;
      xmin = min(xmin, (int)q.xa); xmax = max(xmax, (int)q.xb);
      ymin = min(ymin, (int)q.ya); ymax = max(ymax, (int)q.yb);
    }
    if (xmax >= 0) {
      atomicMin(bb + 0, xmin); atomicMax(bb + 1, xmax); atomicMin(bb + 2, ymin); atomicMax(bb + 3, ymax);
    }
  }
  __syncthreads();
  if (tid == 0) bbox[n] = make_int4(bb[0], bb[1], bb[2], bb[3]);
}

// diffT[n][bin][ctop] = grad_out / count (0 where count <= 0), transposed through LDS so that both sides are coalesced.
// Block = (RoI, 64 output channels): [64][PP] -> [PP][64] in 64 x 64 tiles.  (Round 3 did this inside psroi_prepare, one block per
// RoI walking the channel tiles in sequence: 58 us for 77 MB; 4 x as many blocks in flight: see profiles/r04_psroi.md.)
__global__ __launch_bounds__(kThreads) void psroi_quotient_t(const kgdet_psroi_shape s, const float *__restrict__ grad_out,
                                                             const float *__restrict__ count, float *__restrict__ diffT) {
  __shared__ float tile[64 * 65];
  const int n = blockIdx.x, c0 = blockIdx.y * 64, tid = threadIdx.x;
  const int PP = s.pooled_size * s.pooled_size;
  const long long base = (long long)n * s.out_dim * PP;
  for (int b0 = 0; b0 < PP; b0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += kThreads) {
      const int c = e >> 6, b = e & 63;
      float v = 0.f;
      if (c0 + c < s.out_dim && b0 + b < PP) {
        const long long idx = base + (long long)(c0 + c) * PP + b0 + b;
        const float cn = count[idx];
        v = cn > 0.f ? grad_out[idx] / cn : 0.f;
      }
      tile[c * 65 + b] = v;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += kThreads) {
      const int b = e >> 6, c = e & 63;
      if (c0 + c < s.out_dim && b0 + b < PP) diffT[base + (long long)(b0 + b) * s.out_dim + c0 + c] = tile[c * 65 + b];
    }
  }
}

// grad_data as a gather by output tile; block = ONE WAVE = (tile, 64-channel chunk, image); thread = input channel.
// (A workgroup of four waves = 256 channels met at two barriers per visit; one-wave workgroups need none -- a wave's LDS
// operations complete in order -- and take the same time, 625 against 616 us: what bounds the kernel is each wave's chain of
// dependent LDS read-add-write operations, ~1500 waves resident either way.)
// Phase 0: the ordered list of the RoIs of this image whose bounding box meets the tile (ballot compaction, ascending).
// Per visit (RoI of the list x offset class of this channel chunk): the prepared records come from the workspace into
// registers ONE VISIT AHEAD, are filtered against the tile into LDS, and every thread walks the bins of its group cell
// that have a sample in the tile -- the bins' grad_out / count quotients are fetched eight bins at a time (coalesced
// over channels) before they are used.
template <int kRecRegs>      // records of one visit held per thread (kRecRegs * 64 >= P*P*S*S): 4, 13 or 32 registers of 16 bytes
__global__ __launch_bounds__(kGdThreads) void psroi_grad_data(const kgdet_psroi_shape s, const float *__restrict__ rois,
                                                            const int4 *__restrict__ bbox,
                                                            const Rec *__restrict__ recs_in,
                                                            const float *__restrict__ diffT,
                                                            float *__restrict__ grad_data, int classes,
                                                            int ch_per_class, int tiles_x, int list_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int P = s.pooled_size, S = s.sample_per_part, G = s.group_size;
  const int PP = P * P, SS = S * S, GG = G * G, nrec = PP * SS;
  Rec *recs = reinterpret_cast<Rec *>(smem);
  float *acc = reinterpret_cast<float *>(recs + nrec);              // [kTilePix][kGdStride]
  int *list = reinterpret_cast<int *>(acc + kTilePix * kGdStride);  // [list_cap]

  const int tid = threadIdx.x, lane = tid;
  const int tile = blockIdx.x, b = blockIdx.z;
  const int ty0 = (tile / tiles_x) * kTileH, tx0 = (tile % tiles_x) * kTileW;
  const int ty1 = min(ty0 + kTileH, s.H) - 1, tx1 = min(tx0 + kTileW, s.W) - 1;
  const int c_first = blockIdx.y * kGdThreads;
  const int c = c_first + tid;
  const bool active = c < s.out_dim * GG && c < s.C;
  const int ctop = active ? c / GG : 0;
  const int k = active ? c - ctop * GG : 0;
  const int gh = k / G, gw = k - gh * G;
  const int my_cls = ctop / ch_per_class;
  // this channel's bins: the contiguous row / column ranges that map to its group cell
  int ph_lo = P, ph_hi = 0, pw_lo = P, pw_hi = 0;
  for (int p = 0; p < P; ++p) {
    if (group_cell(p, G, P) == gh) { ph_lo = min(ph_lo, p); ph_hi = max(ph_hi, p + 1); }
    if (group_cell(p, G, P) == gw) { pw_lo = min(pw_lo, p); pw_hi = max(pw_hi, p + 1); }
  }
  const int nw = max(pw_hi - pw_lo, 0), my_bins = active ? max(ph_hi - ph_lo, 0) * nw : 0;
  const int last_ch = min(min(c_first + kGdThreads, s.out_dim * GG), s.C) - 1;
  const int cls_lo = last_ch >= c_first ? (c_first / GG) / ch_per_class : 1;
  const int cls_hi = last_ch >= c_first ? (last_ch / GG) / ch_per_class : 0;
  const int ncls = max(cls_hi - cls_lo + 1, 0);

  for (int e = tid; e < kTilePix * kGdStride; e += kGdThreads) acc[e] = 0.f;
  float *col = acc + tid;

  for (int seg = 0; seg < s.R && ncls > 0; seg += list_cap) {
    // ---- phase 0: RoIs [seg, seg + list_cap) of image b that meet the tile, ascending
    int len = 0;      // (wave-uniform)
    for (int base = seg; base < min(seg + list_cap, s.R); base += kGdThreads) {
      const int n = base + tid;
      bool hit = false;
      if (n < s.R && n < seg + list_cap && (int)rois[5 * n] == b) {
        const int4 bx = bbox[n];
        hit = !(bx.y < tx0 || bx.x > tx1 || bx.w < ty0 || bx.z > ty1);
      }
      const unsigned long long m = __ballot(hit);
      if (hit) list[len + __popcll(m & ((1ull << lane) - 1ull))] = n;
      len += __popcll(m);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const int list_len = len;
    const int visits = list_len * ncls;

    int4 pre[kRecRegs];
    auto fetch = [&](int v) {      // records of visit v -> registers (clamped addresses: always the same number of loads)
      const int n = list[min(v, visits - 1) / ncls], cls = cls_lo + min(v, visits - 1) % ncls;
      const int4 *src = reinterpret_cast<const int4 *>(recs_in + ((long long)n * classes + cls) * nrec);
#pragma unroll
      for (int j = 0; j < kRecRegs; ++j)
        if (j * kGdThreads < nrec) pre[j] = src[min(tid + j * kGdThreads, nrec - 1)];
    };
    if (visits > 0) fetch(0);
    for (int v = 0; v < visits; ++v) {
      const int n = list[v / ncls], cls = cls_lo + v % ncls;
      // (one wave: LDS operations complete in order, a wave barrier keeps the compiler from reordering across the phases)
      __builtin_amdgcn_wave_barrier();
      unsigned long long hit_bits[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
      for (int j = 0; j < kRecRegs; ++j) {
        const int i = tid + j * kGdThreads;
        if (j * kGdThreads < nrec) {
          int4 q = pre[j];
          const int xa = (short)(q.x & 0xffff), xb = (short)(q.x >> 16), ya = (short)(q.y & 0xffff), yb = (short)(q.y >> 16);
          const bool in = i < nrec && xa >= 0 && !(xb < tx0 || xa > tx1 || yb < ty0 || ya > ty1);     // a corner in the tile
          if (!in) q.x = -1;
          if (i < nrec) *reinterpret_cast<int4 *>(recs + i) = q;
          // bins of this lane's record with a sample in the tile -> OR over the wave (uniform 256-bit set)
          const int bin = min(i, nrec - 1) / SS;
          if (in) hit_bits[bin >> 6] |= 1ull << (bin & 63);
        }
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        unsigned long long x = hit_bits[w];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const unsigned lo = __shfl_xor((unsigned)x, d), hi = __shfl_xor((unsigned)(x >> 32), d);
          x |= ((unsigned long long)hi << 32) | lo;
        }
        hit_bits[w] = x;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the records are in LDS
      __builtin_amdgcn_wave_barrier();
      fetch(v + 1);                                           // in flight while this visit is accumulated
      if (my_cls != cls) continue;
      for (int j0 = 0; j0 < my_bins; j0 += kBinBatch) {
        // up to kBinBatch bins of this channel's cell per round: first which of them have a sample in the tile, then ALL their
        // quotients (one coalesced load per bin and wave, all in flight together), then the adds in bin order
        float dq[kBinBatch];
        unsigned long long mine = 0ull;
#pragma unroll
        for (int u = 0; u < kBinBatch; ++u) {
          const int j = min(j0 + u, my_bins - 1);
          const int bin = (ph_lo + j / nw) * P + pw_lo + j % nw;
          if (j0 + u < my_bins && ((hit_bits[bin >> 6] >> (bin & 63)) & 1ull)) mine |= 1ull << u;
        }
        if (mine == 0ull) continue;
#pragma unroll
        for (int u = 0; u < kBinBatch; ++u) {
          const int j = min(j0 + u, my_bins - 1);
          const int bin = (ph_lo + j / nw) * P + pw_lo + j % nw;
          dq[u] = 0.f;
          if ((mine >> u) & 1ull) dq[u] = diffT[((long long)n * PP + bin) * s.out_dim + ctop];
        }
#pragma unroll
        for (int u = 0; u < kBinBatch; ++u) {
          if (!((mine >> u) & 1ull)) continue;
          const int j = j0 + u;
          const int bin = (ph_lo + j / nw) * P + pw_lo + j % nw;
          const float diff = dq[u];
          for (int si = 0; si < SS; ++si) {
            const Rec q = lds_record(recs, bin * SS + si);
            if (q.xa < 0) continue;
            const float fx = q.fx, fy = q.fy;
            const bool xa_in = q.xa >= tx0 && q.xa <= tx1, xb_in = q.xb >= tx0 && q.xb <= tx1;
            const bool ya_in = q.ya >= ty0 && q.ya <= ty1, yb_in = q.yb >= ty0 && q.yb <= ty1;
            const int oa = (q.ya - ty0) * kTileW, ob = (q.yb - ty0) * kTileW;
            const int pa = q.xa - tx0, pb = q.xb - tx0;
            // the reference's four atomicAdds (:236-245), same order, same expressions
            // (plain read-add-write: ds_add_f32 was measured 1.7x slower here -- LDS float atomics run at a fraction of the
            //  read / write rate)
            if (ya_in && xa_in) col[(oa + pa) * kGdStride] += (1 - fx) * (1 - fy) * diff;
            if (yb_in && xa_in) col[(ob + pa) * kGdStride] += (1 - fx) * fy * diff;
            if (ya_in && xb_in) col[(oa + pb) * kGdStride] += fx * (1 - fy) * diff;
            if (yb_in && xb_in) col[(ob + pb) * kGdStride] += fx * fy * diff;
          }
        }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // write the tile: every element of grad_data[b, c_first .. c_first + 63] inside the tile, used channel or not
  const int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
  for (int e = tid; e < kGdThreads * kTilePix; e += kGdThreads) {
    const int ch = e / kTilePix, p = e - ch * kTilePix;
    const int y = p / kTileW, x = p - y * kTileW;
    if (c_first + ch >= s.C || y >= th || x >= tw) continue;
    grad_data[(((long long)b * s.C + c_first + ch) * s.H + ty0 + y) * s.W + tx0 + x] = acc[p * kGdStride + ch];
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// grad_data from per-pixel contribution lists (round 4; the tile gather above stays as the fallback / A-B path).
// What the tile gather pays per (tile, 64 channels, overlapping RoI) -- fetch and filter all P*P*S*S records to find the
// few with a corner in the tile -- does not depend on the channel, and grows with the number of tiles an RoI covers.
// Here the channel-independent part is done ONCE per call:
//   list L = (image, offset class, group cell, pixel) holds every (RoI, bin, sample, corner) that lands on the pixel as
//   (row of diffT, bilinear weight), in the reference loop's serial order (RoI, bin row, bin column, sample row, sample
//   column, corner);
//   psroi_prepare counts, psroi_list_alloc hands out storage (a bump allocation: where a list lies is arbitrary, what it
//   holds is not), psroi_list_fill drops the entries in (slot order arbitrary), psroi_list_sort ranks every list by the
//   entries' sequence numbers (unique keys: rank = number of smaller keys) and writes it out in order;
// and the channel-wide part is a pure stream: one wave per (list, run of channels) walks its list -- entries through scalar
// loads, eight diffT rows in flight, lanes = 4 consecutive output channels (one 1 KB row piece per load and wave) -- and
// adds w * diff in list order.  Exactly one writer per element: no atomics on floats, bit-repeatable, and the same bits as
// the serial float32 evaluation (same products, same order).
constexpr int kSortChunk = 1024;       // keys of a list a wave holds in LDS at a time
constexpr int kWaveSortMax = 512;      // longest list the per-wave rank sort takes
constexpr int kLongSortLds = 8192;     // entries psroi_list_sort_long sorts in LDS (64 KB); longer lists in place in global memory

// storage for every list (wave-aggregated bump allocation), counters reset to serve as fill cursors
__global__ __launch_bounds__(kThreads) void psroi_list_alloc(int *__restrict__ cnt, int *__restrict__ start,
                                                             int *__restrict__ total, long long NL) {
  const long long L = (long long)blockIdx.x * kThreads + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int c = L < NL ? cnt[L] : 0;
  int incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const int wave_sum = __shfl(incl, 63);
  int base = 0;
  if (lane == 63 && wave_sum > 0) base = atomicAdd(total, wave_sum);
  base = __shfl(base, 63);
  if (L < NL) {
    start[L] = base + incl - c;
    cnt[L] = 0;
  }
}

// thread = record: its four corner entries (sequence number, weight) into their lists
__global__ __launch_bounds__(kThreads) void psroi_list_fill(const kgdet_psroi_shape s, const float *__restrict__ rois,
                                                            const Rec *__restrict__ recs, const int *__restrict__ start,
                                                            int *__restrict__ cursor, int2 *__restrict__ ent, int classes) {
  const int P = s.pooled_size, SS = s.sample_per_part * s.sample_per_part, nrec = P * P * SS;
  const long long total = (long long)s.R * classes * nrec;
  const long long i_raw = (long long)blockIdx.x * kThreads + threadIdx.x;
  const long long i = i_raw < total ? i_raw : total - 1;            // (no early exits: list_take is a wave-wide operation)
  const int nc = (int)(i / nrec), rec = (int)(i - (long long)nc * nrec);
  const int n = nc / classes, cls = nc - n * classes;
  const int batch = (int)rois[5 * n];
  const bool roi_ok = batch >= 0 && batch < s.B;
  const int4 v = roi_ok ? reinterpret_cast<const int4 *>(recs)[i] : make_int4(-1, -1, 0, 0);
  const int xa = (short)(v.x & 0xffff), xb = (short)(v.x >> 16), ya = (short)(v.y & 0xffff), yb = (short)(v.y >> 16);
  const bool live = i_raw < total && roi_ok && xa >= 0;
  const float fx = __int_as_float(v.z), fy = __int_as_float(v.w);
  const int bin = rec / SS, ph = bin / P, pw = bin - ph * P;
  const int lb = live ? (int)list_base(s, batch, classes, cls, ph, pw) : 0;
  const int seq = (int)i * 4;
  // the reference's four atomicAdds (:236-245), their order and their weight expressions
  const int L0 = lb + ya * s.W + xa, L1 = lb + yb * s.W + xa, L2 = lb + ya * s.W + xb, L3 = lb + yb * s.W + xb;
  const float w0 = (1 - fx) * (1 - fy), w1 = (1 - fx) * fy, w2 = fx * (1 - fy), w3 = fx * fy;
  const int s0 = list_take(cursor, L0, live), s1 = list_take(cursor, L1, live);
  const int s2 = list_take(cursor, L2, live), s3 = list_take(cursor, L3, live);
  if (!live) return;
  ent[start[L0] + s0] = make_int2(seq, __float_as_int(w0));
  ent[start[L1] + s1] = make_int2(seq + 1, __float_as_int(w1));
  ent[start[L2] + s2] = make_int2(seq + 2, __float_as_int(w2));
  ent[start[L3] + s3] = make_int2(seq + 3, __float_as_int(w3));
}

// wave = list: entries (sequence number, weight) in slot order -> (diffT row, weight) in sequence order.  Rank sort: the keys
// are unique, an entry's place is the number of smaller keys; up to 64 entries by lane reads, beyond through an LDS copy of the
// keys read as broadcasts (four keys per ds_read_b128 against four entries per lane).
__global__ __launch_bounds__(kThreads) void psroi_list_sort(const int *__restrict__ start, const int *__restrict__ len,
                                                            const int2 *__restrict__ ent_in, int2 *__restrict__ ent_out,
                                                            long long NL, int nrec, int classes, int SS, int PP,
                                                            int *__restrict__ long_q, int *__restrict__ long_n) {
  __shared__ __attribute__((aligned(16))) int keys[kThreads / 64][kSortChunk];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long long L = (long long)blockIdx.x * (kThreads / 64) + wave;
  if (L >= NL) return;
  const int n = __builtin_amdgcn_readfirstlane(len[L]);
  if (n == 0) return;
  if (n > kWaveSortMax) {      // the rank sort is quadratic: long lists go to psroi_list_sort_long (order of the queue: irrelevant)
    if (lane == 0) long_q[atomicAdd(long_n, 1)] = (int)L;
    return;
  }
  const int base = __builtin_amdgcn_readfirstlane(start[L]);
  int *kw = keys[wave];
  auto emit = [&](int2 e, int rank) {
    const int q = e.x >> 2, nc = q / nrec, rec = q - nc * nrec;
    ent_out[base + rank] = make_int2((nc / classes) * PP + rec / SS, e.y);
  };
  if (n <= 64) {
    const int2 e = lane < n ? ent_in[base + lane] : make_int2(INT_MAX, 0);
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += __shfl(e.x, j) < e.x ? 1 : 0;
    if (lane < n) emit(e, rank);
    return;
  }
  for (int i0 = 0; i0 < n; i0 += 256) {
    int2 e[4];
    int rank[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = i0 + u * 64 + lane;
      e[u] = idx < n ? ent_in[base + idx] : make_int2(INT_MAX, 0);
    }
    for (int c0 = 0; c0 < n; c0 += kSortChunk) {
      if (n > kSortChunk || i0 == 0) {
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < kSortChunk; j += 64) kw[j] = c0 + j < n ? ent_in[base + c0 + j].x : INT_MAX;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      }
      const int m4 = (min(kSortChunk, n - c0) + 3) & ~3;
      for (int j = 0; j < m4; j += 4) {
        const int4 k4 = *reinterpret_cast<const int4 *>(kw + j);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          rank[u] += (k4.x < e[u].x ? 1 : 0) + (k4.y < e[u].x ? 1 : 0) + (k4.z < e[u].x ? 1 : 0) + (k4.w < e[u].x ? 1 : 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * 64 + lane < n) emit(e[u], rank[u]);
  }
}

// workgroup = one LONG list (more than kWaveSortMax entries: pixels under many small RoIs): a bitonic network whose every
// compare-exchange puts the smaller key at the lower index (the first step of a merge pairs i with its mirror image inside
// the block, the others i with i + j), so the power-of-two padding can stay virtual -- a pair whose upper index is beyond
// the list is a pair with +infinity and does nothing.  O(n log^2 n); up to kLongSortLds entries in LDS, longer lists in place
// in the slot-order array (a workgroup barrier orders its global accesses).
__global__ __launch_bounds__(kThreads) void psroi_list_sort_long(const int *__restrict__ start, const int *__restrict__ len,
                                                                 int2 *__restrict__ ent_in, int2 *__restrict__ ent_out,
                                                                 int nrec, int classes, int SS, int PP,
                                                                 const int *__restrict__ long_q,
                                                                 const int *__restrict__ long_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int2 *buf = reinterpret_cast<int2 *>(smem);
  const int tid = threadIdx.x, count = *long_n;
  for (int qi = blockIdx.x; qi < count; qi += gridDim.x) {
    const int L = long_q[qi], n = len[L], base = start[L];
    const bool in_lds = n <= kLongSortLds;
    int2 *a = in_lds ? buf : ent_in + base;
    if (in_lds)
      for (int t = tid; t < n; t += kThreads) buf[t] = ent_in[base + t];
    __syncthreads();
    int n_pad = 1;
    while (n_pad < n) n_pad <<= 1;
    for (int k = 2; k <= n_pad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        const bool flip = j == (k >> 1);
        for (int t = tid; t < (n_pad >> 1); t += kThreads) {
          const int i = 2 * t - (t & (j - 1));              // the t-th index with bit j clear
          const int p2 = flip ? (i ^ (k - 1)) : (i + j);
          if (p2 < n) {
            const int2 x = a[i], y = a[p2];
            if (x.x > y.x) { a[i] = y; a[p2] = x; }
          }
        }
        __syncthreads();
      }
    for (int t = tid; t < n; t += kThreads) {
      const int2 e = a[t];
      const int q = e.x >> 2, nc = q / nrec, rec = q - nc * nrec;
      ent_out[base + t] = make_int2((nc / classes) * PP + rec / SS, e.y);
    }
    __syncthreads();
  }
}

// wave = (list, run of 64 * VEC output channels of the list's offset class); lane = VEC consecutive output channels.
// Workgroups are renumbered so that one XCD (workgroup id mod 8) walks a contiguous range of lists = neighbouring pixels,
// whose entries share diffT rows: the rows stay in that XCD's L2.
template <int VEC>
__global__ __launch_bounds__(kThreads) void psroi_grad_data_lists(const kgdet_psroi_shape s, const int *__restrict__ start,
                                                                  const int *__restrict__ len,
                                                                  const int2 *__restrict__ ent,
                                                                  const float *__restrict__ diffT,
                                                                  float *__restrict__ grad_data, int classes,
                                                                  int ch_per_class, int chunks, long long items) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int per_xcd = gridDim.x >> 3;
  const long long blk = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long long item = blk * (kThreads / 64) + wave;
  if (item >= items) return;
  const int GG = s.group_size * s.group_size, HW = s.H * s.W;
  const long long L = item / chunks;
  const int chunk = (int)(item - L * chunks);
  long long t = L / HW;
  const int pix = (int)(L - t * HW);
  const int k = (int)(t % GG);
  t /= GG;
  const int cls = (int)(t % classes), b = (int)(t / classes);
  const int n = __builtin_amdgcn_readfirstlane(len[L]);
  const int2 *ep = ent + __builtin_amdgcn_readfirstlane(start[L]);
  const int ct = chunk * 64 * VEC + lane * VEC;                 // first channel of this lane inside the class
  const bool live = ct < ch_per_class;                          // (VEC = 4: ch_per_class % 4 == 0, all four or none)
  const float *dcol = diffT + cls * ch_per_class + (live ? ct : 0);
  const long long od = s.out_dim;
  vec_t acc = 0.f;
  int e = 0;
  for (; e + 8 <= n; e += 8) {
    int2 q[8];
    vec_t d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = ep[e + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = *reinterpret_cast<const vec_t *>(dcol + q[u].x * od);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = acc + __int_as_float(q[u].y) * d[u];
  }
  for (; e < n; ++e) {
    const int2 q = ep[e];
    acc = acc + __int_as_float(q.y) * *reinterpret_cast<const vec_t *>(dcol + q.x * od);
  }
  if (!live) return;
  float *g = grad_data + (((long long)b * s.C + (long long)(cls * ch_per_class + ct) * GG + k) * HW + pix);
#pragma unroll
  for (int v = 0; v < VEC; ++v)
    if (VEC == 1) g[0] = acc[0]; else g[(long long)v * GG * HW] = acc[v];
}

// channels beyond out_dim * group_size^2 take no part in the pooling: their gradient is zero
__global__ __launch_bounds__(kThreads) void psroi_zero_tail(float *__restrict__ grad_data, long long image_stride,
                                                            long long first, long long count) {
  float *g = grad_data + blockIdx.y * image_stride + first;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < count; i += (long long)gridDim.x * kThreads) g[i] = 0.f;
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
  KGDET_CHECK_SHAPE(s->pooled_size * s->pooled_size * s->sample_per_part * s->sample_per_part <= kMaxRecords &&
                        s->group_size <= kMaxGroup && s->pooled_size * s->pooled_size <= 256,
                    "pooled_size^2 * sample_per_part^2 must be <= %d, pooled_size <= 16 and group_size <= %d (LDS record "
                    "table of the gfx950 kernels)", kMaxRecords, kMaxGroup);
  KGDET_CHECK_SHAPE(s->H < 32768 && s->W < 32768, "feature map beyond 32767 pixels per side");
  KGDET_CHECK_SHAPE((long long)s->group_size * s->group_size * s->H * s->W * s->out_dim < (1ll << 30),
                    "one image of the cell-major map beyond 2^30 elements (32-bit corner offsets)");
  return KGDET_OK;
}

namespace {
size_t cell_major_bytes(const kgdet_psroi_shape *s) {
  return align_up((size_t)s->B * s->group_size * s->group_size * s->H * s->W * s->out_dim * sizeof(float), 256);
}
size_t gather_lds(const kgdet_psroi_shape *s, int mode) {
  const size_t PP = (size_t)s->pooled_size * s->pooled_size, SS = (size_t)s->sample_per_part * s->sample_per_part;
  return PP * SS * sizeof(GRec) + PP * 4 + (mode == 0 ? 64 : 2) * PP * 4;
}
int launch_cell_major(const kgdet_psroi_shape *s, const float *data, float *dataT, hipStream_t stream) {
  const int GG = s->group_size * s->group_size;
  KGDET_CHECK_SHAPE((long long)s->B * GG <= 65535, "batch x group cells beyond the launch grid");
  hipLaunchKernelGGL(psroi_to_cell_major, dim3(ceil_div(s->H * s->W, 64), ceil_div(s->out_dim, 64), s->B * GG),
                     dim3(kThreads), 0, stream, *s, data, dataT);
  KGDET_CHECK_LAUNCH("psroi_to_cell_major");
  return KGDET_OK;
}
}  // namespace

size_t kgdet_deform_psroi_forward_workspace_bytes(const kgdet_psroi_shape *s) {
  return s == nullptr ? 0 : cell_major_bytes(s) + 256;
}

int kgdet_deform_psroi_forward(const kgdet_psroi_shape *s, const float *data, const float *rois, const float *trans,
                               float *out, float *count, void *workspace, size_t workspace_bytes, void *stream) {
  if (int rc = psroi_check(s)) return rc;
  const long long total = (long long)s->R * s->out_dim * s->pooled_size * s->pooled_size;
  if (total == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(data && rois && out && count && (s->no_trans || trans), "null pointer");
  const size_t need = kgdet_deform_psroi_forward_workspace_bytes(s);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("deform_psroi_forward: needs %zu bytes of workspace (kgdet_deform_psroi_forward_workspace_bytes), got %zu",
              need, workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  float *dataT = reinterpret_cast<float *>(workspace);
  if (int rc = launch_cell_major(s, data, dataT, (hipStream_t)stream)) return rc;
  const int classes = s->no_trans ? 1 : s->num_classes;
  const int ch_per_class = s->out_dim / classes, chunks = ceil_div(ch_per_class, 64);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)psroi_gather<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      150 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(psroi_gather<0>, dim3((unsigned)((long long)s->R * classes * chunks)), dim3(kThreads),
                     gather_lds(s, 0), (hipStream_t)stream, *s, dataT, rois, trans, out, count, (const float *)nullptr,
                     (float *)nullptr, classes, ch_per_class, chunks);
  KGDET_CHECK_LAUNCH("psroi_gather<forward>");
  return KGDET_OK;
}

namespace {
inline long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }
// per-pixel contribution lists of grad_data: counters / cursors [NL] + the allocation counter, starts [NL], two entry arrays
long long lists_count(const kgdet_psroi_shape *s) {
  const long long classes = s->no_trans ? 1 : (s->num_classes > 0 ? s->num_classes : 1);
  return (long long)(s->B > 0 ? s->B : 0) * classes * s->group_size * s->group_size * s->H * s->W;
}
long long lists_entries(const kgdet_psroi_shape *s) {
  const long long classes = s->no_trans ? 1 : (s->num_classes > 0 ? s->num_classes : 1);
  return (long long)(s->R > 0 ? s->R : 0) * classes * s->pooled_size * s->pooled_size * s->sample_per_part *
         s->sample_per_part * 4;
}
bool lists_ok(const kgdet_psroi_shape *s) {
  static const bool off = [] { const char *e = getenv("KGDET_PSROI_GRAD_DATA"); return e && strcmp(e, "tiles") == 0; }();   // A/B switch
  return !off && lists_count(s) <= (1ll << 26) && lists_entries(s) < (1ll << 31) - 4;
}
size_t lists_bytes(const kgdet_psroi_shape *s) {
  if (!lists_ok(s)) return 0;
  const size_t NL = (size_t)lists_count(s), T = (size_t)lists_entries(s);
  return align_up((NL + 64) * sizeof(int), 256) + align_up(NL * sizeof(int), 256) + 2 * align_up(T * sizeof(int2), 256) +
         align_up((T / kWaveSortMax + 64) * sizeof(int), 256);
}
}  // namespace

size_t kgdet_deform_psroi_backward_workspace_bytes(const kgdet_psroi_shape *s) {
  if (s == nullptr) return 0;
  const size_t R = (size_t)(s->R > 0 ? s->R : 0);
  const size_t total = R * s->out_dim * s->pooled_size * s->pooled_size;
  const size_t classes = s->no_trans ? 1 : (s->num_classes > 0 ? s->num_classes : 1);
  const size_t nrec = (size_t)s->pooled_size * s->pooled_size * s->sample_per_part * s->sample_per_part;
  return align_up(R * sizeof(int4), 256) + align_up(total * sizeof(float), 256) +
         align_up(R * classes * nrec * sizeof(Rec), 256) + (s->no_trans ? 0 : cell_major_bytes(s)) + lists_bytes(s) + 256;
}

int kgdet_deform_psroi_backward(const kgdet_psroi_shape *s, const float *grad_out, const float *count,
                                const float *data, const float *rois, const float *trans, float *grad_data,
                                float *grad_trans, void *workspace, size_t workspace_bytes, void *stream) {
  if (int rc = psroi_check(s)) return rc;
  KGDET_CHECK_SHAPE(grad_data != nullptr, "null pointer");
  const long long total = (long long)s->R * s->out_dim * s->pooled_size * s->pooled_size;
  const size_t data_bytes = (size_t)s->B * s->C * s->H * s->W * sizeof(float);
  if (data_bytes == 0) return KGDET_OK;
  if (total == 0) {   // no RoI: the gradient is zero (the reference's zero-filled grad_input stays as it is)
    KGDET_HIP_TRY(hipMemsetAsync(grad_data, 0, data_bytes, (hipStream_t)stream));
    return KGDET_OK;
  }
  KGDET_CHECK_SHAPE(grad_out && count && data && rois && (s->no_trans || (trans && grad_trans)), "null pointer");
  const size_t need = kgdet_deform_psroi_backward_workspace_bytes(s);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("deform_psroi_backward: needs %zu bytes of workspace (kgdet_deform_psroi_backward_workspace_bytes), got %zu",
              need, workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  const int classes = s->no_trans ? 1 : s->num_classes;
  const int ch_per_class = s->out_dim / classes;
  const size_t nrec = (size_t)s->pooled_size * s->pooled_size * s->sample_per_part * s->sample_per_part;
  unsigned char *wsb = reinterpret_cast<unsigned char *>(workspace);
  int4 *bbox = reinterpret_cast<int4 *>(wsb);
  wsb += align_up((size_t)s->R * sizeof(int4), 256);
  float *diffT = reinterpret_cast<float *>(wsb);
  wsb += align_up((size_t)total * sizeof(float), 256);
  Rec *recs_ws = reinterpret_cast<Rec *>(wsb);
  wsb += align_up((size_t)s->R * classes * nrec * sizeof(Rec), 256);
  float *dataT = reinterpret_cast<float *>(wsb);
  wsb += s->no_trans ? 0 : cell_major_bytes(s);
  const bool use_lists = lists_ok(s);
  const long long NL = lists_count(s), T = lists_entries(s);
  int *list_cnt = reinterpret_cast<int *>(wsb);
  int *list_total = list_cnt + NL;
  wsb += align_up((size_t)(NL + 64) * sizeof(int), 256);
  int *list_start = reinterpret_cast<int *>(wsb);
  wsb += align_up((size_t)NL * sizeof(int), 256);
  int2 *ent_slot = reinterpret_cast<int2 *>(wsb);
  wsb += align_up((size_t)T * sizeof(int2), 256);
  int2 *ent_sorted = reinterpret_cast<int2 *>(wsb);
  wsb += align_up((size_t)T * sizeof(int2), 256);
  int *long_q = reinterpret_cast<int *>(wsb);                 // lists beyond the per-wave sort (at most T / kWaveSortMax)
  int *long_n = list_cnt + NL + 1;                            // (zeroed with the counters)
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)psroi_gather<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      150 * 1024));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)psroi_grad_data<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)psroi_grad_data<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)psroi_grad_data<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)psroi_list_sort_long, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(kLongSortLds * sizeof(int2))));
    attr_set = true;
  }
  if (use_lists) KGDET_HIP_TRY(hipMemsetAsync(list_cnt, 0, (size_t)(NL + 64) * sizeof(int), (hipStream_t)stream));
  hipLaunchKernelGGL(psroi_prepare, dim3(s->R), dim3(kThreads), 0, (hipStream_t)stream, *s, rois, trans, bbox, recs_ws, classes,
                     use_lists ? list_cnt : (int *)nullptr);
  KGDET_CHECK_LAUNCH("psroi_prepare");
  KGDET_CHECK_SHAPE(ceil_div(s->out_dim, 64) <= 65535, "out_dim beyond the launch grid");
  hipLaunchKernelGGL(psroi_quotient_t, dim3(s->R, ceil_div(s->out_dim, 64)), dim3(kThreads), 0, (hipStream_t)stream, *s, grad_out,
                     count, diffT);
  KGDET_CHECK_LAUNCH("psroi_quotient_t");
  if (use_lists) {
    hipStream_t st = (hipStream_t)stream;
    const int list_waves = kThreads / 64;
    hipLaunchKernelGGL(psroi_list_alloc, dim3((unsigned)ceil_div(NL, (long long)kThreads)), dim3(kThreads), 0, st, list_cnt,
                       list_start, list_total, NL);
    KGDET_CHECK_LAUNCH("psroi_list_alloc");
    hipLaunchKernelGGL(psroi_list_fill, dim3((unsigned)ceil_div(T / 4, (long long)kThreads)), dim3(kThreads), 0, st, *s, rois,
                       recs_ws, list_start, list_cnt, ent_slot, classes);
    KGDET_CHECK_LAUNCH("psroi_list_fill");
    hipLaunchKernelGGL(psroi_list_sort, dim3((unsigned)ceil_div(NL, (long long)list_waves)), dim3(kThreads), 0, st, list_start,
                       list_cnt, ent_slot, ent_sorted, NL, (int)nrec, classes, s->sample_per_part * s->sample_per_part,
                       s->pooled_size * s->pooled_size, long_q, long_n);
    KGDET_CHECK_LAUNCH("psroi_list_sort");
    hipLaunchKernelGGL(psroi_list_sort_long, dim3(512), dim3(kThreads), kLongSortLds * sizeof(int2), st, list_start, list_cnt,
                       ent_slot, ent_sorted, (int)nrec, classes, s->sample_per_part * s->sample_per_part,
                       s->pooled_size * s->pooled_size, (const int *)long_q, (const int *)long_n);
    KGDET_CHECK_LAUNCH("psroi_list_sort_long");
    const bool vec4 = ch_per_class % 4 == 0;
    const int chunks = ceil_div(ch_per_class, vec4 ? 256 : 64);
    const long long items = NL * chunks;
    const long long blocks = ceil_div(ceil_div(items, (long long)list_waves), 8ll) * 8;
    KGDET_CHECK_SHAPE(blocks < (1ll << 31), "grad_data: too many (pixel, channel run) items for one launch");
    if (vec4)
      hipLaunchKernelGGL(psroi_grad_data_lists<4>, dim3((unsigned)blocks), dim3(kThreads), 0, st, *s, list_start, list_cnt,
                         ent_sorted, diffT, grad_data, classes, ch_per_class, chunks, items);
    else
      hipLaunchKernelGGL(psroi_grad_data_lists<1>, dim3((unsigned)blocks), dim3(kThreads), 0, st, *s, list_start, list_cnt,
                         ent_sorted, diffT, grad_data, classes, ch_per_class, chunks, items);
    KGDET_CHECK_LAUNCH("psroi_grad_data_lists");
    const long long used = (long long)s->out_dim * s->group_size * s->group_size, HW = (long long)s->H * s->W;
    if (used < s->C) {
      const long long cnt = (s->C - used) * HW, zb = ceil_div(cnt, (long long)kThreads);
      hipLaunchKernelGGL(psroi_zero_tail, dim3((unsigned)(zb > 1024 ? 1024 : zb), s->B), dim3(kThreads), 0, st, grad_data,
                         (long long)s->C * HW, used * HW, cnt);
      KGDET_CHECK_LAUNCH("psroi_zero_tail");
    }
  } else {
    const int tiles_x = ceil_div(s->W, kTileW), tiles_y = ceil_div(s->H, kTileH);
    const int list_cap = s->R < kListCap ? s->R : kListCap;
    const size_t lds = nrec * sizeof(Rec) + (size_t)kTilePix * kGdStride * sizeof(float) + (size_t)list_cap * sizeof(int);
    KGDET_CHECK_SHAPE(s->B <= 65535 && ceil_div(s->C, kGdThreads) <= 65535, "batch / channel count beyond the launch grid");
    const dim3 gd_grid(tiles_x * tiles_y, ceil_div(s->C, kGdThreads), s->B);
#define KGDET_GD_ARGS (hipStream_t)stream, *s, rois, bbox, recs_ws, diffT, grad_data, classes, ch_per_class, tiles_x, list_cap
    if (nrec <= 4 * kGdThreads) hipLaunchKernelGGL(psroi_grad_data<4>, gd_grid, dim3(kGdThreads), lds, KGDET_GD_ARGS);
    else if (nrec <= 13 * kGdThreads) hipLaunchKernelGGL(psroi_grad_data<13>, gd_grid, dim3(kGdThreads), lds, KGDET_GD_ARGS);
    else hipLaunchKernelGGL(psroi_grad_data<32>, gd_grid, dim3(kGdThreads), lds, KGDET_GD_ARGS);
#undef KGDET_GD_ARGS
    KGDET_CHECK_LAUNCH("psroi_grad_data");
  }
  if (!s->no_trans) {
    if (int rc = launch_cell_major(s, data, dataT, (hipStream_t)stream)) return rc;
    hipLaunchKernelGGL(psroi_gather<1>, dim3((unsigned)((long long)s->R * classes)), dim3(kThreads), gather_lds(s, 1),
                       (hipStream_t)stream, *s, dataT, rois, trans, (float *)nullptr, (float *)nullptr, diffT, grad_trans,
                       classes, ch_per_class, 1);
    KGDET_CHECK_LAUNCH("psroi_gather<grad_trans>");
  }
  return KGDET_OK;
}

}  // extern "C"
