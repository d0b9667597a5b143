// Backward w.r.t. input / offset / mask WITHOUT atomics (gfx950).
//
// The reference scatters colgrad = W^T g into grad_input with one float atomicAdd per
// (channel, tap, pixel, corner): 4*C*K*P = 105 M atomics for one 7x7 call
// (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:279-334).  Measured on MI355X that is 2.8 ms with
// global atomics and 0.43 ms with ds_add_f32 into an LDS-resident plane set (LDS float atomics retire
// ~0.4 lanes/clk/CU), against 86 us of MFMA work.  This file turns the scatter into a gather:
//
//  1. dcn_bwd_build_index: per (image, deformable group, tap) one workgroup inverts the sampling map
//     into CSR form: for every INPUT cell q the list of (output pixel p, bilinear weight w) that
//     touch it.  Lists are sorted by p, so everything downstream is order-deterministic.
//  2. dcn_bwd_input_gather: a workgroup owns (image b, 32-channel slice) and a range of taps.
//       phase 1  colgrad[32 c, all pixels of b] for the tap: 32x32 MFMA blocks (operands streamed from
//                L2, register double-buffered), epilogue forms this slice's partial
//                grad_offset / grad_mask (wave shuffle reduce) and parks colgrad in LDS [32][HWp]
//                (139 KB at 25x42 -- the 160 KB LDS of CDNA4 is what makes the whole tile resident);
//       phase 2  every thread owns ~17 input cells x 4 channels IN REGISTERS and adds
//                sum_e w_e * colgrad_lds[c][p_e] over its cells' lists: plain ds_read_b32 + FMA.
//     After its taps the workgroup writes its register image once.
//  3. dcn_bwd_input_fixup / dcn_bwd_offset_fixup (dcn_backward_fixup.hip) add the partial planes and
//     the per-slice offset gradients in fixed order.
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

// ------------------------------------------------------------------------------------------------
// 1. inverse sampling index
//    row_ptr [groups_bd][K][HW + 1]   (groups_bd = N * DG)
//    entries [groups_bd][K][4 * HoWo] as (pixel, weight-bits) pairs
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_bwd_build_index(const DcnProblem p, int *__restrict__ row_ptr,
                                                           int2 *__restrict__ entries) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  const int HW = p.H * p.W;
  int *cnt = sm;                                        // [HW + 1]
  int *cursor = sm + (HW + 1);                          // [HW]
  int2 *ent = reinterpret_cast<int2 *>(sm + 2 * HW + 2);  // [4 * HoWo]
  __shared__ int wave_tot[4];

  const int t = blockIdx.x % p.K, bd = blockIdx.x / p.K;
  const int b = bd / p.DG, dgi = bd - b * p.DG;
  const int tid = threadIdx.x;

  for (int i = tid; i <= HW; i += 256) cnt[i] = 0;
  __syncthreads();
  // pass 1: count the corners landing in each cell
  for (int px = tid; px < p.HoWo; px += 256) {
    const int oy = px / p.Wo, ox = px - oy * p.Wo;
    float y, x, m;
    tap_position(p, b, dgi, t, px, oy, ox, y, x, m);
    Tap tap;
    TapGeom geo;
    make_tap(y, x, p.H, p.W, true, m, tap, geo);
    if (geo.va) atomicAdd(&cnt[tap.o[0]], 1);
    if (geo.vb) atomicAdd(&cnt[tap.o[1]], 1);
    if (geo.vc) atomicAdd(&cnt[tap.o[2]], 1);
    if (geo.vd) atomicAdd(&cnt[tap.o[3]], 1);
  }
  __syncthreads();
  // exclusive scan of cnt[0..HW) -> cursor (each thread owns a contiguous run)
  const int per = (HW + 255) / 256;
  const int lo = min(HW, tid * per), hi = min(HW, lo + per);
  int local = 0;
  for (int i = lo; i < hi; ++i) local += cnt[i];
  int incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int n = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += n;
  }
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
  int run = base + incl - local;
  for (int i = lo; i < hi; ++i) {
    const int c = cnt[i];
    cursor[i] = run;
    run += c;
  }
  __syncthreads();
  int *rp = row_ptr + (long long)(bd * p.K + t) * (HW + 1);
  for (int i = tid; i < HW; i += 256) rp[i] = cursor[i];
  if (tid == 0) rp[HW] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  __syncthreads();
  // pass 2: fill (slot order inside a cell is arbitrary here ...)
  for (int px = tid; px < p.HoWo; px += 256) {
    const int oy = px / p.Wo, ox = px - oy * p.Wo;
    float y, x, m;
    tap_position(p, b, dgi, t, px, oy, ox, y, x, m);
    Tap tap;
    TapGeom geo;
    make_tap(y, x, p.H, p.W, true, m, tap, geo);
    const int valid[4] = {geo.va, geo.vb, geo.vc, geo.vd};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (valid[q]) {
        const int slot = atomicAdd(&cursor[tap.o[q]], 1);
        ent[slot] = make_int2(px, __float_as_int(tap.w[q]));
      }
  }
  __syncthreads();
  // ... so sort every cell's (short) list by pixel; two corners of one pixel never share a cell
  for (int cell = tid; cell < HW; cell += 256) {
    const int e1 = cursor[cell];      // end (cursor advanced by the fill)
    const int e0 = e1 - cnt[cell];
    for (int i = e0 + 1; i < e1; ++i) {
      const int2 key = ent[i];
      int j = i - 1;
      while (j >= e0 && ent[j].x > key.x) { ent[j + 1] = ent[j]; --j; }
      ent[j + 1] = key;
    }
  }
  __syncthreads();
  int2 *dst = entries + (long long)(bd * p.K + t) * 4 * p.HoWo;
  const int total = rp[HW];
  for (int i = tid; i < total; i += 256) dst[i] = ent[i];
}

// ------------------------------------------------------------------------------------------------
// 2. colgrad per tap -> LDS, grad_offset partials, grad_input by gather
// ------------------------------------------------------------------------------------------------
constexpr int kWin = 256;          // CSR entries per wave-private LDS window
constexpr int kMaxCellIters = 17;  // cells per lane: H*W <= 1088 (KGDet's stride-32 map has 1050)

__global__ __launch_bounds__(kThreads, 2) void dcn_bwd_input_gather(const DcnProblem p, const DcnBwdInputLdsArgs a,
                                                                   const int *__restrict__ row_ptr,
                                                                   const int2 *__restrict__ entries) {
  extern __shared__ __attribute__((aligned(16))) float colgrad[];  // [32][HWp]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kk = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = p.H * p.W;
  const int HWp = a.HWp;  // colgrad row stride: HoWo rounded up to 64 (pixels), >= HoWo
  const int n_cell_iters = (HW + 63) >> 6;

  const int pair = blockIdx.x / a.S, split = blockIdx.x - pair * a.S;
  const int b = pair / a.n_cslices, cs = pair - b * a.n_cslices;
  const int c0 = cs * 32;
  const int dgi = (p.c_base + min(c0, p.Cg - 1)) / p.cpdg;
  const int t_begin = (int)((long long)split * p.K / a.S), t_end = (int)((long long)(split + 1) * p.K / a.S);

  const float *gimg = a.grad_out + ((long long)b * p.O_total + p.o_base) * p.HoWo;
  const float *ximg = p.x + ((long long)b * p.C_total + p.c_base) * HW;
  const int slice_id = a.slice_base + cs;
  const int c_last = p.Cg - 1;
  const int n_pblocks = (p.HoWo + 31) >> 5;  // 32-pixel MFMA blocks of this image

  float gin[kMaxCellIters][4];  // grad_input of cells (lane + 64 i), channels c0 + 4*wave + j
#pragma unroll
  for (int i = 0; i < kMaxCellIters; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) gin[i][j] = 0.f;

  for (int t = t_begin; t < t_end; ++t) {
    // ---------------- phase 1: colgrad blocks -----------------------------------------------------
    for (int pb = wave; pb < n_pblocks; pb += 8) {
      const int px = pb * 32 + l31;
      const bool plive = px < p.HoWo;
      const float *wt = p.wpk + ((long long)t * a.Og_pad16 + kk) * a.Cg_pad256 + c0 + l31;
      const float *gp = gimg + (long long)kk * p.HoWo + (plive ? px : 0);
      f32x16 acc = {0};
      constexpr int U = 8;
      const int n_groups = a.Og_pad16 / (2 * U);
      const int o_last = p.Og - 1;
      float aA[U], pA[U], aB[U], pB[U];
      auto load_set = [&](int grp, float (&av)[U], float (&pv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k2 = grp * U + u;
          const int o = 2 * k2 + kk;
          const int oc = min(o, o_last) - kk;
          av[u] = wt[(long long)(2 * k2) * a.Cg_pad256];  // rows >= Og are zero in the packed image
          // UNCONDITIONAL load from a clamped address: a select here makes hipcc sink the load into a
          // branch and wait vmcnt(0) after it, which serialises the whole pipeline (measured 14x).
          // No masking is needed: padded o rows multiply zero weights, dead pixels are never stored.
          pv[u] = gp[(long long)oc * p.HoWo];
        }
      };
      auto mma_set = [&](const float (&av)[U], const float (&pv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], pv[u], acc, 0, 0, 0);
      };
      load_set(0, aA, pA);
      for (int grp = 0; grp < n_groups; grp += 2) {
        if (grp + 1 < n_groups) load_set(grp + 1, aB, pB);
        mma_set(aA, pA);
        if (grp + 2 < n_groups) load_set(grp + 2, aA, pA);
        if (grp + 1 < n_groups) mma_set(aB, pB);
      }

      // epilogue: rows = channels c0 + mfma_row(r), column = this lane's pixel
      const int oy = plive ? px / p.Wo : 0, ox = plive ? px - oy * p.Wo : 0;
      float y = 0.f, x = 0.f, m = 0.f;
      if (plive) tap_position(p, b, dgi, t, px, oy, ox, y, x, m);
      Tap tap;
      TapGeom geo;
      make_tap(y, x, p.H, p.W, plive, m, tap, geo);
      const float hy = 1.0f - geo.ly, hx = 1.0f - geo.lx;
      const float ka = geo.va ? 1.f : 0.f, kb = geo.vb ? 1.f : 0.f, kc = geo.vc ? 1.f : 0.f, kd = geo.vd ? 1.f : 0.f;
      const float ua = ka * hy * hx, ub = kb * hy * geo.lx, uc = kc * geo.ly * hx, ud = kd * geo.ly * geo.lx;
      float sum_y = 0.f, sum_x = 0.f, sum_m = 0.f;
#pragma unroll
      for (int half = 0; half < 4; ++half) {
        float cv[4][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {  // 16 gathers in flight together
          const int cl = mfma_row(half * 4 + rr, lane);
          const float *plane = ximg + (long long)min(c0 + cl, c_last) * HW;
#pragma unroll
          for (int q = 0; q < 4; ++q) cv[rr][q] = plane[tap.o[q]];
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = half * 4 + rr;
          const int cl = mfma_row(r, lane);
          const float cg = ((c0 + cl) <= c_last) ? acc[r] : 0.0f;
          const float va = cv[rr][0] * ka, vb = cv[rr][1] * kb, vc = cv[rr][2] * kc, vd = cv[rr][3] * kd;
          sum_y += cg * (hx * (vc - va) + geo.lx * (vd - vb));   // deform_conv_cuda_kernel.cu:144-187
          sum_x += cg * (hy * (vb - va) + geo.ly * (vd - vc));
          sum_m += cg * (ua * va + ub * vb + uc * vc + ud * vd);
          if (plive) colgrad[cl * HWp + px] = cg;
        }
      }
      sum_y += __shfl_xor(sum_y, 32);
      sum_x += __shfl_xor(sum_x, 32);
      sum_m += __shfl_xor(sum_m, 32);
      if (kk == 0 && plive) {
        float *dst = a.off_part + (((long long)slice_id * p.N + b) * 2 * p.K + 2 * t) * p.HoWo + px;
        dst[0] = sum_y * m;      // v2: grad_offset carries the mask factor (:757); m == 1 for v1
        dst[p.HoWo] = sum_x * m;
        if (a.mask_part) a.mask_part[(((long long)slice_id * p.N + b) * p.K + t) * p.HoWo + px] = sum_m;
      }
    }
    __syncthreads();
    // ---------------- phase 2: every thread pulls its cells' contributions -------------------------
    // Iteration i serves cells [64 i, 64 i + 64): their CSR entries are one contiguous range, which the
    // wave copies into its private 2 KiB LDS window (coalesced 8-byte loads, prefetched one iteration
    // ahead) so that the per-lane list walk costs LDS latency instead of an L2 round trip per entry.
    {
      const int *rp = row_ptr + (long long)((b * p.DG + dgi) * p.K + t) * (HW + 1);
      const int2 *ent = entries + (long long)((b * p.DG + dgi) * p.K + t) * 4 * p.HoWo;
      const float *cgrow = colgrad + (4 * wave) * HWp;
      int2 *win = reinterpret_cast<int2 *>(colgrad + 32 * HWp) + wave * kWin;  // this wave's window
      const int ent_total = rp[HW];
      int2 nxt[kWin / 64];
      int cb = __builtin_amdgcn_readfirstlane(rp[0]);  // first entry of the upcoming iteration's range
#pragma unroll
      for (int k = 0; k < kWin / 64; ++k) nxt[k] = ent[min(cb + lane + 64 * k, ent_total - 1 < 0 ? 0 : ent_total - 1)];
#pragma unroll
      for (int i = 0; i < kMaxCellIters; ++i) {
        if (i < n_cell_iters) {
          const int cell = lane + 64 * i;
          int e = 0, e_end = 0;
          if (cell < HW) { e = rp[cell]; e_end = rp[cell + 1]; }
          const int range_end = __builtin_amdgcn_readfirstlane(rp[min(64 * (i + 1), HW)]);
          const int cur = cb;
#pragma unroll
          for (int k = 0; k < kWin / 64; ++k) win[lane + 64 * k] = nxt[k];
          cb = range_end;  // next iteration's range starts where this one ends
          if (i + 1 < n_cell_iters) {
#pragma unroll
            for (int k = 0; k < kWin / 64; ++k)
              nxt[k] = ent[min(cb + lane + 64 * k, ent_total - 1 < 0 ? 0 : ent_total - 1)];
          }
          const int in_win_end = min(e_end, cur + kWin);
          for (; e < in_win_end; ++e) {  // entries inside the window: LDS
            const int2 en = win[e - cur];
            const float w = __int_as_float(en.y);
            const float *src = cgrow + en.x;
            gin[i][0] += w * src[0];
            gin[i][1] += w * src[HWp];
            gin[i][2] += w * src[2 * HWp];
            gin[i][3] += w * src[3 * HWp];
          }
          for (; e < e_end; ++e) {       // rare: more than kWin entries for 64 cells -> straight from L2
            const int2 en = ent[e];
            const float w = __int_as_float(en.y);
            const float *src = cgrow + en.x;
            gin[i][0] += w * src[0];
            gin[i][1] += w * src[HWp];
            gin[i][2] += w * src[2 * HWp];
            gin[i][3] += w * src[3 * HWp];
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep iterations apart: hoisting 17 iterations of loads spills
      }
    }
    __syncthreads();
  }
  // park the partial planes: slab [32][HW] per workgroup (same layout dcn_bwd_input_fixup reads)
  float *dst = a.slabs + (long long)blockIdx.x * 32 * HW + (long long)(4 * wave) * HW;
#pragma unroll
  for (int i = 0; i < kMaxCellIters; ++i) {
    const int cell = lane + 64 * i;
    if (i < n_cell_iters && cell < HW) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[(long long)j * HW + cell] = gin[i][j];
    }
  }
}

}  // namespace kgdet
