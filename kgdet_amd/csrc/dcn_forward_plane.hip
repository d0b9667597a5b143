// Deformable convolution forward, "plane" variant for gfx950: feature planes resident in LDS, bilinear
// gathers served by LDS, bf16 MFMA with an fp32-accurate hi/lo operand split.
//
// Same reference path as dcn_forward.hip (deformable_im2col + addmm_,
// mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:190-276, 570-632; deform_conv_cuda.cpp:221-245, 534-563)
// and the same GEMM view / stream-K split, but the two things that bound the f32 kernel are removed:
//   * gathers: the f32 kernel fetches every bilinear corner pair from L2 through the texture addresser
//     (~4 lanes/clk/CU).  Here a 16-channel slice of ONE image ([16][H*W] fp32, 67 KB at 25x42) is copied
//     into LDS once and reused by all K taps of that channel chunk, so a corner pair is one ds_read2_b32;
//     the reduction therefore runs chunk-major / tap-minor and pixel tiles never straddle images.
//   * MFMA rate: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  Every fp32 operand v is split
//     into hi = bf16(v), lo = bf16(v - hi) and the product is taken as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
//     with v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 3 MFMAs of 32 cycles replace 8 of 64, and the
//     dropped a_lo*b_lo term plus the two representation residuals are <= 2^-16 relative per product
//     (measured against the f64 oracle: ~1e-6 of the output scale, tests/test_gpu_dcn.py).
//     PARTS = 1 keeps only the hi parts: plain bf16 operands for autocast inference.
//
// Operand images (identical in global memory and LDS, so the weight stage is a lane-linear copy):
//   A stage (tap t, channel chunk c16, 256 output channels): [part][khalf][o 256][8 bf16]   8 KB / part
//   B stage (same reduction slice, 128 pixels):              [part][khalf][px 128][8 bf16]  4 KB / part
//   lane l of a wave reads row/col (l & 31) of k-half (l >> 5) as one 16-byte ds_read_b128; reduction
//   element k = khalf*8 + j is channel c16*16 + k for both operands.
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kAPart = 2 * kTileM * 8 * 2;  // bytes of one part of an A stage
constexpr int kBPart = 2 * kTileN * 8 * 2;
constexpr int kPlaneLoads = 12;             // 16-byte loads per thread that cover a [16][1536] plane

template <int PARTS>
struct PlaneStageRegs {
  f32x4 a[PARTS];    // this thread's 16 B of each part of the weight stage
  float ry, rx, rm;  // RAW learned offset (dy, dx) and modulation of the stage's tap for this thread's pixel;
                     // nothing is computed from them until produce(), so the loads stay in flight
};

}  // namespace

#ifndef KGDET_ABL
#define KGDET_ABL 0
#endif
template <int PARTS>
__global__ __launch_bounds__(kThreads, 1) void dcn_fwd_plane(const DcnFwdGroup grp, float *__restrict__ slabs) {
  constexpr int ABL = KGDET_ABL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *As = smem;                         // [2][PARTS][kAPart]
  unsigned char *Bs = smem + 2 * PARTS * kAPart;    // [2][PARTS][kBPart]
  float *plane = reinterpret_cast<float *>(smem + 2 * PARTS * (kAPart + kBPart));  // [16][H*W]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int n_local = tid & (kTileN - 1);  // pixel column this thread samples for
  const int cq = tid >> 7;                 // which 4 of the chunk's 16 channels
  const long long G = gridDim.x, g = blockIdx.x;
  const long long total = grp.unit_begin[grp.n];
  const long long my_begin = unit_begin(g, total, G);
  const long long my_end = unit_begin(g + 1, total, G);

  long long cur = my_begin;
  while (cur < my_end) {
    int pi = 0;
    while (pi + 1 < grp.n && cur >= grp.unit_begin[pi + 1]) ++pi;
    const DcnProblem &p = grp.p[pi];
    const long long p_begin = grp.unit_begin[pi];
    const int HW = p.H * p.W;
    const int K = p.K;
    const int n_c16 = p.chunks_per_tap;
    const int cpt = p.chunks_per_tile;
    const int tile = (int)((cur - p_begin) / cpt);
    const long long tile_begin = p_begin + (long long)tile * cpt;
    const int s_begin = (int)(cur - tile_begin);
    const int s_end = (int)((my_end - tile_begin) < cpt ? (my_end - tile_begin) : cpt);
    const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;

    int pb, hw;
    const bool live = tile_pixel(p, nt, n_local, pb, hw);
    const int tile_b = nt / p.tiles_per_image;  // wave-uniform image of the tile
    const int oy = hw / p.Wo, ox = hw - oy * p.Wo;

    f32x16 acc[2][2];
    zero_acc(acc);

    typedef PlaneStageRegs<PARTS> Regs;
    Regs R0, R1;

    // unconditional loads from clamped addresses (dead pixels read pixel 0 of the image and get zero
    // weights in produce()); without a mask the modulation slot re-reads the offset and is ignored
    const int hw_c = live ? hw : 0;
    const float *mod_src = p.mask ? p.mask : p.offset;
    auto fetch_raw = [&](int s, Regs &R) {
      const int c16 = s / K, t = s - c16 * K;
      const int dgi = (p.c_base + min(c16 * kChunk + cq * 4, p.Cg - 1)) / p.cpdg;
      const unsigned bd = (unsigned)(tile_b * p.DG + dgi);
      const unsigned obase = (bd * 2u * (unsigned)K + 2u * (unsigned)t) * (unsigned)p.HoWo + (unsigned)hw_c;
      R.ry = p.offset[obase];
      R.rx = p.offset[obase + (unsigned)p.HoWo];
      R.rm = mod_src[(bd * (unsigned)K + (unsigned)t) * (unsigned)p.HoWo + (unsigned)hw_c];
    };
    auto issue_weights = [&](int s, Regs &R) {
      const int c16 = s / K, t = s - c16 * K;
      const size_t stage = (size_t)((mt * n_c16 + c16) * K + t) * (2 * kAPart);
      const unsigned char *src = reinterpret_cast<const unsigned char *>(p.wq) + stage + tid * 16;
#pragma unroll
      for (int part = 0; part < PARTS; ++part) R.a[part] = *reinterpret_cast<const f32x4 *>(src + part * kAPart);
    };
    auto commit_weights = [&](int buf, const Regs &R) {
#pragma unroll
      for (int part = 0; part < PARTS; ++part)
        *reinterpret_cast<f32x4 *>(As + (buf * PARTS + part) * kAPart + tid * 16) = R.a[part];
    };
    // Copy x[tile_b, c_base + 16*c16 .. +15, :, :] into LDS.  All loads are issued before the first store and
    // are unconditional from clamped addresses (a guarded load makes hipcc branch and drain the queue).
    auto load_plane = [&](int c16) {
      const int c0 = c16 * kChunk;
      const long long base = ((long long)tile_b * p.C_total + p.c_base + c0) * HW;
      if (c0 + kChunk <= p.Cg && (base & 3) == 0) {
        const int n4 = 4 * HW;  // float4 units in 16 planes
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.x + base);
        f32x4 v[kPlaneLoads];
#pragma unroll
        for (int r = 0; r < kPlaneLoads; ++r) v[r] = src[min(tid + r * kThreads, n4 - 1)];
#pragma unroll
        for (int r = 0; r < kPlaneLoads; ++r)
          if (tid + r * kThreads < n4) reinterpret_cast<f32x4 *>(plane)[tid + r * kThreads] = v[r];
      } else {  // ragged last chunk or unaligned planes: dword copy, padded channels re-read the last real one
        const int n = kChunk * HW;
        for (int i0 = 0; i0 < n; i0 += 8 * kThreads) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int i = min(i0 + tid + r * kThreads, n - 1);
            const int k = i / HW, e = i - k * HW;
            const int c = min(c0 + k, p.Cg - 1);
            v[r] = p.x[((long long)tile_b * p.C_total + p.c_base + c) * HW + e];
          }
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (i0 + tid + r * kThreads < n) plane[i0 + tid + r * kThreads] = v[r];
        }
      }
    };
    // B stage: sample 4 channels of this thread's pixel at the stage's tap, split, store.
    auto produce = [&](int s, int buf, const Regs &R) {
      const int t = s % K;
      const int ti = t / p.kw, tj = t - ti * p.kw;
      const float y = (float)(oy * p.sh - p.ph + ti * p.dh) + R.ry;
      const float x = (float)(ox * p.sw - p.pw + tj * p.dw) + R.rx;
      TapPair tap;
      make_tap_pair(y, x, p.H, p.W, live, p.mask ? R.rm : 1.0f, tap);
      bf16x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float *pl = plane + (cq * 4 + j) * HW;
        f32x2u v0, v1;
        if constexpr (ABL & 1) { v0[0] = R.ry; v0[1] = R.rx; v1 = v0; }
        else {
        v0 = *reinterpret_cast<const f32x2u *>(pl + tap.o[0]);
        v1 = *reinterpret_cast<const f32x2u *>(pl + tap.o[1]);
        }
        const float sv = tap.w[0] * v0[0] + tap.w[1] * v0[1] + tap.w[2] * v1[0] + tap.w[3] * v1[1];
        hi[j] = (__bf16)sv;
        if constexpr (PARTS == 2) lo[j] = (__bf16)(sv - (float)hi[j]);
      }
      unsigned char *dst = Bs + buf * PARTS * kBPart + (cq >> 1) * (kTileN * 16) + n_local * 16 + (cq & 1) * 8;
      *reinterpret_cast<bf16x4 *>(dst) = hi;
      if constexpr (PARTS == 2) *reinterpret_cast<bf16x4 *>(dst + kBPart) = lo;
    };
    auto multiply = [&](int buf) {
      const unsigned char *A = As + buf * PARTS * kAPart + (lane >> 5) * (kTileM * 16) + (wm * 64 + (lane & 31)) * 16;
      const unsigned char *B = Bs + buf * PARTS * kBPart + (lane >> 5) * (kTileN * 16) + (wn * 64 + (lane & 31)) * 16;
      bf16x8 a[PARTS][2], b[PARTS][2];
#pragma unroll
      for (int part = 0; part < PARTS; ++part)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[part][i] = *reinterpret_cast<const bf16x8 *>(A + part * kAPart + i * 32 * 16);
          b[part][i] = *reinterpret_cast<const bf16x8 *>(B + part * kBPart + i * 32 * 16);
        }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if constexpr (PARTS == 2) {  // small terms first
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[1][ni], acc[mi][ni], 0, 0, 0);
          }
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
        }
    };

    // prologue: plane of the first chunk, stage s_begin in buffer 0, loads of s_begin+1 in flight in R1
    fetch_raw(s_begin, R0);
    issue_weights(s_begin, R0);
    fetch_raw(min(s_begin + 1, s_end - 1), R1);
    issue_weights(min(s_begin + 1, s_end - 1), R1);
    __syncthreads();  // previous tile's readers of plane / A / B are done
    if (!(ABL & 16)) load_plane(s_begin / K);
    commit_weights(0, R0);
    __syncthreads();
    produce(s_begin, 0, R0);
    __syncthreads();

    // stage s: multiply buffer s&1; meanwhile build stage s+1 (registers RC) into the other buffer and
    // put the loads of stage s+2 in flight (registers RI)
    auto run_stage = [&](int s, Regs &RI, Regs &RC) {
      const int buf = (s - s_begin) & 1;
      const bool has_next = (s + 1) < s_end;
      const bool same_plane = has_next && ((s + 1) / K == s / K);
      // unconditional (the tail re-loads the last stage): a guarded issue makes the number of loads in
      // flight path-dependent and hipcc then drains the queue (vmcnt(0)) at every consumer
      const int s2 = min(s + 2, s_end - 1);
      if (!(ABL & 32)) fetch_raw(s2, RI);
      if (!(ABL & 4)) issue_weights(s2, RI);
      if (same_plane && !(ABL & 8)) produce(s + 1, buf ^ 1, RC);
      if (!(ABL & 2)) multiply(buf);
      if (has_next && !(ABL & 4)) commit_weights(buf ^ 1, RC);
      if (has_next && !same_plane) {  // chunk boundary: everybody is done sampling the old plane
        __syncthreads();
        if (!(ABL & 16)) load_plane((s + 1) / K);
        __syncthreads();
        produce(s + 1, buf ^ 1, RC);
      }
      __syncthreads();
    };
    for (int s = s_begin; s < s_end; s += 2) {
      run_stage(s, R0, R1);
      if (s + 1 < s_end) run_stage(s + 1, R1, R0);
    }

    if (s_begin == 0 && s_end == cpt) {
      store_output(p, mt, nt, tid, acc);
    } else {
      float *slab = slabs + ((long long)g * 2 + slab_slot(cur, my_begin)) * kTileElems;
      store_slab(slab, tid, acc);
    }
    cur = tile_begin + s_end;
  }
}

template __global__ void dcn_fwd_plane<1>(const DcnFwdGroup grp, float *__restrict__ slabs);
template __global__ void dcn_fwd_plane<2>(const DcnFwdGroup grp, float *__restrict__ slabs);

size_t dcn_fwd_plane_lds_bytes(int parts, int HW) {
  return (size_t)2 * parts * (kAPart + kBPart) + (size_t)kChunk * HW * sizeof(float);
}

// ----------------------------------------------------------------------------------------------
// Weight packing, all three images in one pass over the weights.  [O, Cg, K] (one group) ->
//   wpk[t][c pad16][o pad256]                       fp32  (exact forward kernel, dcn_forward.hip)
//   wpt[t][o pad16][c pad256]                       fp32  (backward-input, dcn_backward*.hip)
//   wq [mt][c16][t][part (hi, lo)][khalf][o 256][8] bf16  (plane forward kernel; hi = bf16(w), lo = bf16(w - hi))
// all zero padded.  One workgroup per (8 channels = one k-half, 32 output channels): each wave reads
// whole runs of K contiguous floats into an LDS tile [8][33][K+1]; the three images are written from it
// in runs of 128 B (wpk), 32 B (wpt) and 512 B (wq).
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_pack_weight_all(const float *__restrict__ w, float *__restrict__ wpk,
                                                            float *__restrict__ wpt, void *__restrict__ wq,
                                                            int Og, int Cg, int K, int Cg_pad, int Og_pad,
                                                            int Og_pad16, int Cg_pad256) {
  extern __shared__ float tile[];  // [(cc * 33 + oo)][K+1]
  const int c8 = blockIdx.x, o0 = blockIdx.y * 32, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ld = K + 1;
  for (int t0 = 0; t0 < K; t0 += 64) {
    const int t = t0 + lane;
    for (int r0 = wave; r0 < 256; r0 += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // 8 independent (clamped, unconditional) loads in flight per lane
        const int r = r0 + 4 * u, cc = r >> 5, oo = r & 31;
        const int c = min(c8 * 8 + cc, Cg - 1), o = min(o0 + oo, Og - 1);
        v[u] = w[((long long)o * Cg + c) * K + min(t, K - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 4 * u, cc = r >> 5, oo = r & 31;
        const bool real = (c8 * 8 + cc < Cg) && (o0 + oo < Og);
        if (t < K) tile[(cc * 33 + oo) * ld + t] = real ? v[u] : 0.0f;
      }
    }
  }
  __syncthreads();
  const int cbase = c8 * 8;
  if (cbase < Cg_pad && o0 < Og_pad) {
    const int oo = tid & 31, cc = tid >> 5;
    for (int t = 0; t < K; ++t)
      wpk[((long long)t * Cg_pad + cbase + cc) * Og_pad + o0 + oo] = tile[(cc * 33 + oo) * ld + t];
    if (wq) {
      const int n_c16 = Cg_pad / kChunk;
      const int c16 = c8 >> 1, khalf = c8 & 1;
      const int mt = o0 / kTileM, o_in = o0 % kTileM;
      for (int e = tid; e < K * 64; e += 256) {
        const int o2 = e & 31, part = (e >> 5) & 1, t = e >> 6;
        bf16x8 v;
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          const float f = tile[(c2 * 33 + o2) * ld + t];
          const __bf16 hi = (__bf16)f;
          v[c2] = part == 0 ? hi : (__bf16)(f - (float)hi);
        }
        const size_t stage = (size_t)((mt * n_c16 + c16) * K + t) * (2 * kAPart);
        unsigned char *dst = reinterpret_cast<unsigned char *>(wq) + stage + part * kAPart + khalf * (kTileM * 16) +
                             (o_in + o2) * 16;
        *reinterpret_cast<bf16x8 *>(dst) = v;
      }
    }
  }
  if (o0 < Og_pad16) {
    const int cc = tid & 7, oo = tid >> 3;
    if (o0 + oo < Og_pad16)
      for (int t = 0; t < K; ++t)
        wpt[((long long)t * Og_pad16 + o0 + oo) * Cg_pad256 + cbase + cc] = tile[(cc * 33 + oo) * ld + t];
  }
}

}  // namespace kgdet
