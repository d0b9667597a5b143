// Deformable convolution backward for gfx950 (same gather -> LDS -> f32 MFMA geometry as forward).
//
// Reference path replaced (mmdet/ops/dcn/src):
//   deform_conv_backward_input_cuda      deform_conv_cuda.cpp:260-371  = addmm_ (W^T g) into a
//       [C*K, N*Ho*Wo] column matrix in HBM, then deformable_col2im_coord (:373-435 of the .cu)
//       and deformable_col2im with float atomics (:279-334);
//   deform_conv_backward_parameters_cuda deform_conv_cuda.cpp:373-484 = a second im2col + addmm_.
// Neither column matrix exists here.
//
// (backward w.r.t. input / offset / mask: dcn_backward_plane.hip, dcn_backward_offset.hip, dcn_backward_gather.hip and, for maps
//  beyond the LDS plane and any grouping, dcn_backward_large.hip.  Rounds 1-4 kept dcn_bwd_input_mfma here -- the column gradient
//  scattered with float atomics, as the reference does -- for weight groups / deformable groups on large maps: removed in round 5,
//  the library has no float atomic left.)
// dcn_bwd_weight_mfma: per (tap t, 256 out-channels o, 128 channels c) tile
//       gW[t][c][o] = sum_p g[o,p] * sample_c(p,t)             (MFMA, reduction over pixels,
//   stream-K over pixel stages with deterministic slab fix-up), written in the packed layout.
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

// ------------------------------------------------------------------------------------------------
// backward w.r.t. weight
// tile = (tap t, o-tile of 256, c-tile of 128); reduction over pixels in stages of 16.
// p.out = packed gradient image gWpk[t][c (pad)][o (pad)] of this group.
// ------------------------------------------------------------------------------------------------

namespace {

constexpr int kLdbW = kTileN + 1;  // padded: the staging threads write [pixel][channel] with pixels across lanes
constexpr int kLdaW = kTileM + 1;

// tile register image -> gWpk, transposed through LDS so that stores run along o (contiguous)
__device__ __forceinline__ void store_wgrad(const DcnProblem &p, const DcnBwdWeightArgs &a, int tile, int tid,
                                            float *scratch, const f32x16 (&acc)[2][2]) {
  const int lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2;
  const int ct = tile % a.n_ctiles;
  const int ot = (tile / a.n_ctiles) % a.n_otiles;
  const int t = tile / (a.n_ctiles * a.n_otiles);
  float *T = scratch + wave * (32 * 33);  // private 32x32 (+1 pad) transpose buffer per wave
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
      for (int r = 0; r < 16; ++r) T[(lane & 31) * 33 + mfma_row(r, lane)] = acc[mi][ni][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
      const int o = ot * kTileM + wm * 64 + mi * 32 + (lane & 31);
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const int cl = cc * 2 + (lane >> 5);
        const int c = ct * kTileN + wn * 64 + ni * 32 + cl;
        const float v = T[cl * 33 + (lane & 31)];
        if (c < p.Cg_pad && o < p.Og_pad) p.out[((long long)t * p.Cg_pad + c) * p.Og_pad + o] = v;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

}  // namespace

__global__ __launch_bounds__(kThreads, 2) void dcn_bwd_weight_mfma(const DcnProblem p, const DcnBwdWeightArgs a,
                                                                   float *__restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kChunk * kLdaW + 2 * kChunk * kLdbW];
  float *As = lds;                        // [2][16][257]   A[k = pixel][i = o]
  float *Bs = lds + 2 * kChunk * kLdaW;   // [2][16][129]   B[k = pixel][j = c]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const long long G = gridDim.x, g = blockIdx.x;
  const long long my_begin = unit_begin(g, a.total_units, G);
  const long long my_end = unit_begin(g + 1, a.total_units, G);
  const int HW = p.H * p.W;
  // staging roles: every thread handles pixel k_local of the stage
  const int k_local = tid & 15;
  // 0..31; the four values inside a wave are 8 apart so that [k][sub] LDS writes are <= 2-way
  const int sub = ((lane >> 4) << 3) + wave;

  long long cur = my_begin;
  while (cur < my_end) {
    const int spt = a.stages_per_tile;
    const int tile = (int)(cur / spt);
    const long long tile_begin = (long long)tile * spt;
    const int s_begin = (int)(cur - tile_begin);
    const int s_end = (int)((my_end - tile_begin) < spt ? (my_end - tile_begin) : spt);
    const int ct = tile % a.n_ctiles;
    const int ot = (tile / a.n_ctiles) % a.n_otiles;
    const int t = tile / (a.n_ctiles * a.n_otiles);

    f32x16 acc[2][2];
    zero_acc(acc);
    float gv[8];     // grad_out values: 8 output channels for this thread's pixel
    float v[4][4];   // gathered corners: 4 channels x 4 corners
    Tap tap;

    // channels this thread samples: c = ct*128 + sub + 32*j  (j = 0..3)
    auto issue = [&](int s) {
      const int pix = s * kChunk + k_local;
      const bool live = pix < p.P;
      const int b = live ? pix / p.HoWo : 0;
      const int hw = live ? pix - b * p.HoWo : 0;
      const int oy = hw / p.Wo, ox = hw - oy * p.Wo;
      // A: g[o, pix] for o = ot*256 + sub + 32*j
      const float *gsrc = a.grad_out + ((long long)b * p.O_total + p.o_base) * p.HoWo + hw;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // unconditional load from a clamped row: padded o rows are never unpacked, dead pixels carry
        // zero samples on the B side
        const int o = min(ot * kTileM + sub + 32 * j, p.Og - 1);
        gv[j] = gsrc[(long long)o * p.HoWo];
      }
      // B: samples; all four channels must share a deformable group -> tap per thread and stage
      const int c_first = ct * kTileN + sub;
      const int dgi = (p.c_base + min(c_first, p.Cg - 1)) / p.cpdg;
      float y = 0.f, x = 0.f, m = 0.f;
      if (live) tap_position(p, b, dgi, t, hw, oy, ox, y, x, m);
      TapGeom geo;
      make_tap(y, x, p.H, p.W, live, m, tap, geo);
      const float *xb = p.x + ((long long)b * p.C_total + p.c_base) * HW;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = min(c_first + 32 * j, p.Cg - 1);
        const float *plane = xb + (long long)c * HW;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = plane[tap.o[q]];
      }
    };
    auto commit = [&](float *Adst, float *Bdst) {
#pragma unroll
      for (int j = 0; j < 8; ++j) Adst[k_local * kLdaW + sub + 32 * j] = gv[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = ct * kTileN + sub + 32 * j;
        float sv = tap.w[0] * v[j][0] + tap.w[1] * v[j][1] + tap.w[2] * v[j][2] + tap.w[3] * v[j][3];
        if (c >= p.Cg) sv = 0.0f;  // padded channels
        Bdst[k_local * kLdbW + sub + 32 * j] = sv;
      }
    };

    issue(s_begin);
    commit(As, Bs);
    __syncthreads();
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
      const bool more = (s + 1) < s_end;
      if (more) issue(s + 1);
      mfma_stage(As + buf * kChunk * kLdaW, kLdaW, Bs + buf * kChunk * kLdbW, kLdbW, wm * 64, wn * 64, lane, acc);
      if (more) commit(As + (buf ^ 1) * kChunk * kLdaW, Bs + (buf ^ 1) * kChunk * kLdbW);
      __syncthreads();
      buf ^= 1;
    }

    if (s_begin == 0 && s_end == spt) {
      store_wgrad(p, a, tile, tid, lds, acc);
      __syncthreads();
    } else {
      float *slab = slabs + ((long long)g * 2 + slab_slot(cur, my_begin)) * kTileElems;
      store_slab(slab, tid, acc);
    }
    cur = tile_begin + s_end;
  }
}

// grid = (tiles, 4): block (tile, j) owns the 32x32 accumulator block (mi, ni) = (j >> 1, j & 1) of
// every wave, adds the slabs in workgroup order and stores it transposed (runs along o).
__global__ __launch_bounds__(kThreads) void dcn_bwd_weight_fixup(const DcnProblem p, const DcnBwdWeightArgs a,
                                                                 const float *__restrict__ slabs, int G) {
  __shared__ float scratch[8 * 32 * 33];
  const int tile = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
  const int spt = a.stages_per_tile;
  const long long tb = (long long)tile * spt, te = tb + spt;
  long long g = tb * G / a.total_units;
  while (unit_begin(g + 1, a.total_units, G) <= tb) ++g;
  while (unit_begin(g, a.total_units, G) > tb) --g;
  const long long gb = unit_begin(g, a.total_units, G), ge = unit_begin(g + 1, a.total_units, G);
  if (gb <= tb && ge >= te) return;
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (; g < G; ++g) {
    const long long b0 = unit_begin(g, a.total_units, G);
    if (b0 >= te) break;
    if (unit_begin(g + 1, a.total_units, G) == b0) continue;
    const long long seg_begin = b0 > tb ? b0 : tb;
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(slabs + ((long long)g * 2 + slab_slot(seg_begin, b0)) * kTileElems);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = s4[(j * 4 + q) * kThreads + tid];
      acc[4 * q] += v[0]; acc[4 * q + 1] += v[1]; acc[4 * q + 2] += v[2]; acc[4 * q + 3] += v[3];
    }
  }
  const int mi = j >> 1, ni = j & 1;
  const int lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2;
  const int ct = tile % a.n_ctiles;
  const int ot = (tile / a.n_ctiles) % a.n_otiles;
  const int t = tile / (a.n_ctiles * a.n_otiles);
  float *T = scratch + wave * (32 * 33);
#pragma unroll
  for (int r = 0; r < 16; ++r) T[(lane & 31) * 33 + mfma_row(r, lane)] = acc[r];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const int o = ot * kTileM + wm * 64 + mi * 32 + (lane & 31);
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) {
    const int cl = cc * 2 + (lane >> 5);
    const int c = ct * kTileN + wn * 64 + ni * 32 + cl;
    if (c < p.Cg_pad && o < p.Og_pad) p.out[((long long)t * p.Cg_pad + c) * p.Og_pad + o] = T[cl * 33 + (lane & 31)];
  }
}

// grad_bias[o] = sum_{b,hw} grad_out[b,o,hw]   (deform_conv_cuda.cpp:659-665)
__global__ __launch_bounds__(256) void dcn_bias_grad(const float *__restrict__ grad_out, float *__restrict__ grad_bias,
                                                     int N, int O, int HoWo, int accumulate) {
  __shared__ float part[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < N; ++b) {
    const float *src = grad_out + ((long long)b * O + o) * HoWo;
    for (int i = tid; i < HoWo; i += 256) s += src[i];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const float tot = part[0] + part[1] + part[2] + part[3];
    grad_bias[o] = accumulate ? grad_bias[o] + tot : tot;
  }
}

}  // namespace kgdet
