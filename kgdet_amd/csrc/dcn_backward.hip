// Deformable convolution backward for gfx950 (same gather -> LDS -> f32 MFMA geometry as forward).
//
// Reference path replaced (mmdet/ops/dcn/src):
//   deform_conv_backward_input_cuda      deform_conv_cuda.cpp:260-371  = addmm_ (W^T g) into a
//       [C*K, N*Ho*Wo] column matrix in HBM, then deformable_col2im_coord (:373-435 of the .cu)
//       and deformable_col2im with float atomics (:279-334);
//   deform_conv_backward_parameters_cuda deform_conv_cuda.cpp:373-484 = a second im2col + addmm_.
// Neither column matrix exists here.
//
// dcn_bwd_input_mfma : per (tap t, 256 channels c, 128 pixels p) tile
//       colgrad[c,p] = sum_o Wt[t][o][c] * g[o,p]            (MFMA, reduction over o)
//   and, while the tile is still in registers,
//       grad_offset[p,t,{y,x}] = sum_c colgrad * d(sample_c)/d{y,x}   (wave shuffle + LDS reduce)
//       grad_mask[p,t]         = sum_c colgrad * sample_c             (v2)
//       grad_input[b,c,corner] += colgrad * bilinear weight           (float atomics, as the reference)
// dcn_bwd_weight_mfma: per (tap t, 256 out-channels o, 128 channels c) tile
//       gW[t][c][o] = sum_p g[o,p] * sample_c(p,t)             (MFMA, reduction over pixels,
//   stream-K over pixel stages with deterministic slab fix-up), written in the packed layout.
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

namespace {
constexpr int kLdsA = kChunk * kTileM;
constexpr int kLdsB = kChunk * kTileN;
}  // namespace

// ------------------------------------------------------------------------------------------------
// backward w.r.t. input / offset / mask
// p.wpk here is the TRANSPOSED packed image Wt[t][o (pad 16)][c (pad 256)]; p.out is unused.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads, 2) void dcn_bwd_input_mfma(const DcnProblem p, const DcnBwdInputArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kLdsA + 2 * kLdsB];
  float *As = lds;
  float *Bs = lds + 2 * kLdsA;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int n_local = tid & (kTileN - 1), kq = tid >> 7;
  const int HW = p.H * p.W;
  const int n_stages = a.Og_pad16 / kChunk;

  for (int unit = blockIdx.x; unit < a.n_units; unit += gridDim.x) {
    // unit -> (tap, channel tile, pixel tile); pixel tile fastest so neighbours share Wt[t]
    const int nt = unit % a.n_ntiles;
    const int ct = (unit / a.n_ntiles) % a.n_ctiles;
    const int t = unit / (a.n_ntiles * a.n_ctiles);
    const int c_tile0 = ct * kTileM;

    // B operand source: grad_out[b, o, hw] for this thread's pixel
    const int pix = nt * kTileN + n_local;
    const bool live = pix < p.P;
    const int pb = live ? pix / p.HoWo : 0;
    const int hw = live ? pix - pb * p.HoWo : 0;
    const float *gsrc = a.grad_out + ((long long)pb * p.O_total + p.o_base) * p.HoWo + hw;

    f32x16 acc[2][2];
    zero_acc(acc);
    float gv[4];

    auto stage_w = [&](int s, float *Adst) {  // Wt[t][16 o][256 c] -> LDS, lane-linear
      const int wave_base = wave << 6;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int q = tid + kThreads * r;
        const int k = q >> 6, col4 = q & 63;
        const float *src = p.wpk + ((long long)(t * a.Og_pad16 + s * kChunk + k) * a.Cg_pad256 + c_tile0 + col4 * 4);
        float *dst = Adst + (wave_base + kThreads * r) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
    };
    auto g_issue = [&](int s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // unconditional load from a clamped row (a select would make hipcc branch + drain vmcnt);
        // rows >= Og meet zero weights, dead pixels are never stored
        const int o = min(s * kChunk + kq * 4 + j, p.Og - 1);
        gv[j] = gsrc[(long long)o * p.HoWo];
      }
    };
    auto g_commit = [&](float *Bdst) {
#pragma unroll
      for (int j = 0; j < 4; ++j) Bdst[(kq * 4 + j) * kTileN + n_local] = gv[j];
    };

    stage_w(0, As);
    g_issue(0);
    g_commit(Bs);
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < n_stages; ++s) {
      const bool more = (s + 1) < n_stages;
      if (more) {
        stage_w(s + 1, As + (buf ^ 1) * kLdsA);
        g_issue(s + 1);
      }
      mfma_stage(As + buf * kLdsA, kTileM, Bs + buf * kLdsB, kTileN, wm * 64, wn * 64, lane, acc);
      if (more) g_commit(Bs + (buf ^ 1) * kLdsB);
      __syncthreads();
      buf ^= 1;
    }

    // ---- epilogue: the tile holds colgrad[c, p] for tap t -------------------------------------
    // lane owns pixel columns (ni) and 32 channel rows per column.
    float *red = lds;  // [4 wm][128 px][3] partial sums, reuses the stage buffers (all reads done)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int px = nt * kTileN + wn * 64 + ni * 32 + (lane & 31);
      const bool plive = px < p.P;
      const int b = plive ? px / p.HoWo : 0;
      const int phw = plive ? px - b * p.HoWo : 0;
      const int oy = phw / p.Wo, ox = phw - oy * p.Wo;
      const int dgi = (p.c_base + min(c_tile0, p.Cg - 1)) / p.cpdg;  // whole tile in one deformable group
      float y = 0.f, x = 0.f, m = 0.f;
      if (plive) tap_position(p, b, dgi, t, phw, oy, ox, y, x, m);
      Tap tap;
      TapGeom geo;
      make_tap(y, x, p.H, p.W, plive, m, tap, geo);
      const float hy = 1.0f - geo.ly, hx = 1.0f - geo.lx;
      // unmasked bilinear weights for grad_mask; slopes use the raw corner values
      const float ua = geo.va ? hy * hx : 0.f, ub = geo.vb ? hy * geo.lx : 0.f;
      const float uc = geo.vc ? geo.ly * hx : 0.f, ud = geo.vd ? geo.ly * geo.lx : 0.f;
      float sum_y = 0.f, sum_x = 0.f, sum_m = 0.f;
      const float *xin = p.x + ((long long)b * p.C_total + p.c_base) * HW;
      float *gin = a.grad_input + ((long long)b * p.C_total + p.c_base) * HW;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = c_tile0 + wm * 64 + mi * 32 + mfma_row(r, lane);
          if (c >= p.Cg || !geo.in_range) continue;
          const float cg = acc[mi][ni][r];
          const float *plane = xin + (long long)c * HW;
          float *gplane = gin + (long long)c * HW;
          const float va = geo.va ? plane[tap.o[0]] : 0.f;
          const float vb = geo.vb ? plane[tap.o[1]] : 0.f;
          const float vc = geo.vc ? plane[tap.o[2]] : 0.f;
          const float vd = geo.vd ? plane[tap.o[3]] : 0.f;
          // d/dy and d/dx of the bilinear sample (deform_conv_cuda_kernel.cu:144-187)
          sum_y += cg * (hx * (vc - va) + geo.lx * (vd - vb));
          sum_x += cg * (hy * (vb - va) + geo.ly * (vd - vc));
          sum_m += cg * (ua * va + ub * vb + uc * vc + ud * vd);
          // scatter to the four corners (deform_conv_cuda_kernel.cu:279-334)
          if (geo.va) atomicAdd(gplane + tap.o[0], tap.w[0] * cg);
          if (geo.vb) atomicAdd(gplane + tap.o[1], tap.w[1] * cg);
          if (geo.vc) atomicAdd(gplane + tap.o[2], tap.w[2] * cg);
          if (geo.vd) atomicAdd(gplane + tap.o[3], tap.w[3] * cg);
        }
      // the other half-wave holds the remaining rows of the same pixel
      sum_y += __shfl_xor(sum_y, 32);
      sum_x += __shfl_xor(sum_x, 32);
      sum_m += __shfl_xor(sum_m, 32);
      if (lane < 32) {
        float *dst = red + ((wm * kTileN) + wn * 64 + ni * 32 + lane) * 3;
        dst[0] = sum_y * m;  // v2: grad_offset carries the mask factor (:757); m == 1 for v1
        dst[1] = sum_x * m;
        dst[2] = sum_m;
      }
    }
    __syncthreads();
    if (tid < kTileN) {
      const int px = nt * kTileN + tid;
      if (px < p.P) {
        float gy = 0.f, gx = 0.f, gm = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float *src = red + (w * kTileN + tid) * 3;
          gy += src[0]; gx += src[1]; gm += src[2];
        }
        const int b = px / p.HoWo, phw = px - b * p.HoWo;
        const int dgi = (p.c_base + min(c_tile0, p.Cg - 1)) / p.cpdg;
        // several channel tiles / weight groups of one deformable group add up
        float *go = a.grad_offset + ((long long)(b * p.DG + dgi) * 2 * p.K + 2 * t) * p.HoWo + phw;
        if (a.direct) {
          go[0] = gy; go[p.HoWo] = gx;
          if (a.grad_mask) a.grad_mask[((long long)(b * p.DG + dgi) * p.K + t) * p.HoWo + phw] = gm;
        } else {
          atomicAdd(go, gy); atomicAdd(go + p.HoWo, gx);
          if (a.grad_mask) atomicAdd(a.grad_mask + ((long long)(b * p.DG + dgi) * p.K + t) * p.HoWo + phw, gm);
        }
      }
    }
    __syncthreads();
  }
}

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
