// Backward w.r.t. input / offset / mask with LDS-privatised accumulation (gfx950).
//
// Why: scattering colgrad to grad_input with global float atomics (what the reference does,
// deform_conv_cuda_kernel.cu:329) is bound by L2 atomic throughput: 4*C*K*P atomics = 105 M for
// one 7x7 call, measured ~40 G atomics/s => 2.8 ms against 86 us of MFMA work.  Here each
// workgroup owns one (image b, 32-channel slice) and keeps that slice's whole grad_input plane
// set [32][H*W] in LDS (134 KB at 25x42 -- this is what the 160 KB LDS of CDNA4 buys), adds into
// it with ds_add_f32, and writes it out once.  Partial planes of the workgroups that share a
// (b, slice) pair and the per-slice partial offset gradients are summed in a fixed order by
// dcn_bwd_input_fixup, so no global atomics and no pre-zeroed outputs are needed.
//
// Per wave-unit = (tap t, 64 output pixels of image b):
//     colgrad[32 c, 64 p] = sum_o Wt[t][o][c] * g[o, p]     128 x v_mfma_f32_32x32x2_f32 x 2
//   operands go global -> VGPR directly (both are 128 B-coalesced rows; all 8 waves share A via L1),
//   then the epilogue of dcn_backward.hip runs on the 2x16 accumulators of each lane.
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

__global__ __launch_bounds__(kThreads, 2) void dcn_bwd_input_lds(const DcnProblem p, const DcnBwdInputLdsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float slab[];  // [32][HW]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kk = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = p.H * p.W;

  const int pair = blockIdx.x / a.S, split = blockIdx.x - pair * a.S;
  const int b = pair / a.n_cslices, cs = pair - b * a.n_cslices;
  const int c0 = cs * 32;  // first channel (inside the weight group) of this slice
  const int dgi = (p.c_base + min(c0, p.Cg - 1)) / p.cpdg;

  for (int i = tid; i < 32 * HW; i += kThreads) slab[i] = 0.0f;
  __syncthreads();

  const int n_units = p.K * a.n_pblocks;
  const int u_begin = (int)((long long)split * n_units / a.S);
  const int u_end = (int)((long long)(split + 1) * n_units / a.S);
  const float *gimg = a.grad_out + ((long long)b * p.O_total + p.o_base) * p.HoWo;
  const float *ximg = p.x + ((long long)b * p.C_total + p.c_base) * HW;
  const int slice_id = a.slice_base + cs;  // position among all slices of all weight groups

  for (int u = u_begin + wave; u < u_end; u += 8) {
    const int t = u / a.n_pblocks, pb = u - t * a.n_pblocks;
    const int px0 = pb * 64 + l31, px1 = px0 + 32;  // pixel (inside image b) of each accumulator
    const bool live0 = px0 < p.HoWo, live1 = px1 < p.HoWo;
    const float *wt = p.wpk + ((long long)t * a.Og_pad16 + kk) * a.Cg_pad256 + c0 + l31;
    const float *g0 = gimg + (long long)kk * p.HoWo + (live0 ? px0 : 0);
    const float *g1 = gimg + (long long)kk * p.HoWo + (live1 ? px1 : 0);

    f32x16 acc0 = {0}, acc1 = {0};
    // Reduction over o, two k-steps' worth of operands per lane pair.  Operands come straight from
    // L2 (no LDS left for staging), so the loop is software-pipelined by hand: two register sets of
    // U k-steps each; the loads of set B are in flight while the MFMAs consume set A.
    constexpr int U = 8;
    const int n_groups = a.Og_pad16 / (2 * U);
    const int o_last = p.Og - 1;
    float aA[U], pA[U], qA[U], aB[U], pB[U], qB[U];
    auto load_set = [&](int grp, float (&av)[U], float (&pv)[U], float (&qv)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k2 = grp * U + u;
        const int o = 2 * k2 + kk;
        const int oc = min(o, o_last) - kk;  // clamped row (relative to the kk row baked into g0/g1)
        av[u] = wt[(long long)(2 * k2) * a.Cg_pad256];  // rows >= Og are zero in the packed image
        const float v0 = g0[(long long)oc * p.HoWo];
        const float v1 = g1[(long long)oc * p.HoWo];
        pv[u] = (o <= o_last && live0) ? v0 : 0.0f;
        qv[u] = (o <= o_last && live1) ? v1 : 0.0f;
      }
    };
    auto mma_set = [&](const float (&av)[U], const float (&pv)[U], const float (&qv)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], pv[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], qv[u], acc1, 0, 0, 0);
      }
    };
    load_set(0, aA, pA, qA);
    for (int grp = 0; grp < n_groups; grp += 2) {
      if (grp + 1 < n_groups) load_set(grp + 1, aB, pB, qB);
      mma_set(aA, pA, qA);
      if (grp + 2 < n_groups) load_set(grp + 2, aA, pA, qA);
      if (grp + 1 < n_groups) mma_set(aB, pB, qB);
    }

    // epilogue on the two 32x32 blocks: rows = channels c0 + mfma_row(r), column = this lane's pixel
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int px = ni ? px1 : px0;
      const bool plive = ni ? live1 : live0;
      const int oy = plive ? px / p.Wo : 0, ox = plive ? px - oy * p.Wo : 0;
      float y = 0.f, x = 0.f, m = 0.f;
      if (plive) tap_position(p, b, dgi, t, px, oy, ox, y, x, m);
      Tap tap;
      TapGeom geo;
      make_tap(y, x, p.H, p.W, plive, m, tap, geo);
      const float hy = 1.0f - geo.ly, hx = 1.0f - geo.lx;
      const float ua = geo.va ? hy * hx : 0.f, ub = geo.vb ? hy * geo.lx : 0.f;
      const float uc = geo.vc ? geo.ly * hx : 0.f, ud = geo.vd ? geo.ly * geo.lx : 0.f;
      float sum_y = 0.f, sum_x = 0.f, sum_m = 0.f;
      // zero the weights of dead corners once; the gathers themselves are unconditional (offsets are
      // clamped into the plane) so that all of them are in flight together instead of one L2 round
      // trip per branch
      const float ka = geo.va ? 1.f : 0.f, kb = geo.vb ? 1.f : 0.f, kc = geo.vc ? 1.f : 0.f, kd = geo.vd ? 1.f : 0.f;
      const int c_last = p.Cg - 1;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float cv[8][4];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int cl = mfma_row(half * 8 + rr, lane);
          const float *plane = ximg + (long long)min(c0 + cl, c_last) * HW;
#pragma unroll
          for (int q = 0; q < 4; ++q) cv[rr][q] = plane[tap.o[q]];
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int r = half * 8 + rr;
          const int cl = mfma_row(r, lane);
          const bool c_ok = (c0 + cl) <= c_last;
          const float cg = c_ok ? (ni ? acc1[r] : acc0[r]) : 0.0f;
          const float va = cv[rr][0] * ka, vb = cv[rr][1] * kb, vc = cv[rr][2] * kc, vd = cv[rr][3] * kd;
          sum_y += cg * (hx * (vc - va) + geo.lx * (vd - vb));   // deform_conv_cuda_kernel.cu:144-187
          sum_x += cg * (hy * (vb - va) + geo.ly * (vd - vc));
          sum_m += cg * (ua * va + ub * vb + uc * vc + ud * vd);
          float *gplane = slab + cl * HW;
          if (c_ok) {
            if (geo.va) atomicAdd(gplane + tap.o[0], tap.w[0] * cg);  // LDS ds_add_f32
            if (geo.vb) atomicAdd(gplane + tap.o[1], tap.w[1] * cg);
            if (geo.vc) atomicAdd(gplane + tap.o[2], tap.w[2] * cg);
            if (geo.vd) atomicAdd(gplane + tap.o[3], tap.w[3] * cg);
          }
        }
      }
      sum_y += __shfl_xor(sum_y, 32);
      sum_x += __shfl_xor(sum_x, 32);
      sum_m += __shfl_xor(sum_m, 32);
      if (kk == 0 && plive) {  // one plain store per (slice, b, t, pixel): partial over this slice's channels
        float *dst = a.off_part + (((long long)slice_id * p.N + b) * 2 * p.K + 2 * t) * p.HoWo + px;
        dst[0] = sum_y * m;
        dst[p.HoWo] = sum_x * m;
        if (a.mask_part) a.mask_part[(((long long)slice_id * p.N + b) * p.K + t) * p.HoWo + px] = sum_m;
      }
    }
  }
  __syncthreads();
  // park this workgroup's partial planes; dcn_bwd_input_fixup adds the S of them
  float *dst = a.slabs + (long long)blockIdx.x * 32 * HW;
  for (int i = tid; i < 32 * HW; i += kThreads) dst[i] = slab[i];
}

// grad_input[b, c] = sum over the S partial planes;  grad_offset / grad_mask = sum over the slices
// of each deformable group.  One workgroup per (b, slice) for the planes, then a grid-stride pass
// over the offset gradient.
__global__ __launch_bounds__(256) void dcn_bwd_input_fixup(const DcnProblem p, const DcnBwdInputLdsArgs a,
                                                           float *__restrict__ grad_input) {
  const int HW = p.H * p.W;
  const int pair = blockIdx.y;
  const int b = pair / a.n_cslices, cs = pair - b * a.n_cslices;
  const int c0 = cs * 32;
  const int n_c = min(32, p.Cg - c0);
  float *dst = grad_input + ((long long)b * p.C_total + p.c_base + c0) * HW;
  const float *src = a.slabs + (long long)pair * a.S * 32 * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_c * HW; i += gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < a.S; ++k) s += src[(long long)k * 32 * HW + i];
    dst[i] = s;
  }
}

__global__ __launch_bounds__(256) void dcn_bwd_offset_fixup(const float *__restrict__ off_part,
                                                            const float *__restrict__ mask_part,
                                                            float *__restrict__ grad_offset,
                                                            float *__restrict__ grad_mask, int n_slices, int N,
                                                            int DG, int K, int HoWo, int n_cslices, int Cg, int cpdg) {
  // element e over [N][DG][2K][HoWo]
  const long long total = (long long)N * DG * 2 * K * HoWo;
  for (long long e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int hw = (int)(e % HoWo);
    const int ch = (int)((e / HoWo) % (2 * K));
    const int dgi = (int)((e / HoWo / (2 * K)) % DG);
    const int b = (int)(e / HoWo / (2 * K) / DG);
    // slice sl = (weight group, 32-channel slice); its deformable group follows from its first channel
    auto slice_dg = [&](int sl) {
      const int wg = sl / n_cslices, cs = sl - wg * n_cslices;
      return (wg * Cg + min(cs * 32, Cg - 1)) / cpdg;
    };
    float s = 0.f;
    for (int sl = 0; sl < n_slices; ++sl)
      if (slice_dg(sl) == dgi) s += off_part[(((long long)sl * N + b) * 2 * K + ch) * HoWo + hw];
    grad_offset[e] = s;
    if (grad_mask && (ch & 1) == 0) {
      float sm = 0.f;
      for (int sl = 0; sl < n_slices; ++sl)
        if (slice_dg(sl) == dgi) sm += mask_part[(((long long)sl * N + b) * K + (ch >> 1)) * HoWo + hw];
      grad_mask[(((long long)b * DG + dgi) * K + (ch >> 1)) * HoWo + hw] = sm;
    }
  }
}

}  // namespace kgdet
