// Fix-up kernels of the backward-input pass (gfx950): deterministic, fixed-order sums of
//   - the partial grad_input planes written by the workgroups that share an (image, 32-channel slice)
//     pair (dcn_bwd_input_gather, dcn_backward_gather.hip), and
//   - the per-slice partial offset / mask gradients of each deformable group.
// No atomics, so outputs need no pre-zeroing.  (An earlier version of this pass accumulated into an
// LDS-resident plane set with ds_add_f32; LDS float atomics retire ~0.4 lanes/clk/CU on MI355X, which made
// it atomics-bound at 13 % of MFMA peak -- see DESIGN.md section 3.)
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

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
