// Kernel declarations shared between the .hip translation units and the C-ABI launchers.
#pragma once
#include "dcn_common.h"

namespace kgdet {

__global__ void dcn_fwd_mfma(const DcnProblem p, float *__restrict__ slabs);
__global__ void dcn_fwd_fixup(const DcnFwdGroup grp, const float *__restrict__ slabs, int G);
__global__ void dcn_fwd_fixup_static(const DcnFwdGroup grp, const float *__restrict__ slabs, int G);
// plane variant (dcn_forward_plane.hip): PARTS = 2 hi/lo split (fp32-accurate), 1 = plain bf16 operands
template <int PARTS>
__global__ void dcn_fwd_plane(const DcnFwdGroup grp, float *__restrict__ slabs);
size_t dcn_fwd_plane_lds_bytes(int parts, int HW);
int dcn_plane_wave_layout();
template <int PARTS>
__global__ void dcn_fwd_gather(const DcnFwdGroup grp, float *__restrict__ slabs);
__global__ void dcn_to_pixel_major(const float *__restrict__ src, float *__restrict__ dst, int C, long long P,
                                   long long src_image_stride);
size_t dcn_fwd_plane_fixed_lds_bytes(int parts);
int dcn_fwd_plane_threads();
// large-map v1 backward without atomics (dcn_backward_large.hip)
bool dcn_bwd_large_ok(const DcnProblem &p, bool has_mask, int groups);
size_t dcn_bwd_large_workspace_bytes(const DcnProblem &p);
int dcn_bwd_large(const DcnProblem &p, const float *grad_output, int out_channels_total, int out_channel_offset,
                  float *grad_input, float *grad_offset, float *grad_mask, void *workspace, size_t workspace_bytes, void *stream,
                  int x_channels, int c_base, int accumulate_offset);
__global__ void dcn_build_taps(const DcnFwdGroup grp);
// grad_input on the plane kernel (dcn_backward_plane.hip)
template <int PARTS>
__global__ void dcn_bwd_input_plane(const DcnFwdGroup grp, float *__restrict__ slabs);
size_t dcn_bwd_input_plane_lds_bytes(int parts, int plane_pixels);
size_t dcn_bwd_input_plane_fixed_lds_bytes(int parts);
__global__ void dcn_build_inverse_taps(const DcnProblem p, uint4 *__restrict__ inv, int *__restrict__ hdr,
                                       DcnInvOvfCell *__restrict__ cells, int2 *__restrict__ spill);
size_t dcn_build_inverse_taps_lds_bytes(int HW, int HoWo);
struct DcnPackOne {
  const float *w;
  float *wpk, *wpt;
  void *wq, *wqt;
  int Og, Cg, K, Cg_pad, Og_pad, Og_pad16, Cg_pad256;
};
struct DcnPackGroup {
  DcnPackOne e[kMaxFwdGroup];
};
__global__ void dcn_pack_weight_all_multi(const DcnPackGroup grp);
struct DcnInvBuild {
  DcnProblem p;
  uint4 *inv;
  int *hdr;
  DcnInvOvfCell *cells;
  int2 *spill;
  int4 *hot_cols;    // nullable: the column list of dcn_hot_gemm, filled by the builder
  int *hot_count;
  int hot_max;       // columns the list may take (<= kHotMaxCols)
};
struct DcnInvBuildGroup {
  int n;
  DcnInvBuild e[kMaxFwdGroup];
};
__global__ void dcn_build_inverse_taps_multi(const DcnInvBuildGroup grp);
// sums of the cells with more than 8 contributions, all output channels (dcn_backward_plane.hip)
constexpr int kInvSumSplit = 4;
struct DcnInvSum {
  const int *hdr;
  const DcnInvOvfCell *cells;
  const int2 *spill;
  const float *gout_t;   // [N][HoWo][O]: pixel-major copy of the convolution's grad_output channels
  float *gov;            // [N * K][max_slots][O_ld], O_ld = groups * Og_pad16
  int NK, K, HoWo, O, Og, Og_pad16, O_ld, max_slots;
  int W;                 // width of the INPUT map (cells q = y * W + x): neighbours of a cell for the cluster rule
  const int *hot_count;  // nullable: [N] hot cells of the image's offset tensor (dcn_hot_gemm takes them if <= hot_max)
  int hot_max;
};
// XCD-local schedule of the grouped launch: workgroups are dealt to the 8 XCDs round-robin by linear id, so workgroup L runs on XCD
// L % 8; every (problem, image) -- whose 1 MB pixel-major grad_output all its (tap, split) units read row by row -- is given to ONE
// XCD, where those rows stay in that XCD's L2 (with units dealt in launch order every XCD streamed all of the group's 12 MB:
// 350 MB of fabric reads per head stage on converged offsets, ~95 us per launch in the training step's profile of round 5).
constexpr int kInvSumSegs = 6;
struct DcnInvSumSeg { int z, b, first, n, u0; };      // problem, image, first unit index r on its XCD, units, the segment's first unit
                                                      // within the (problem, image) -- a pair's K * kInvSumSplit units come as two segments
struct DcnInvSumSched {
  int on;                                             // 0: grid = (N * K, kInvSumSplit, problems), unit = blockIdx
  int n_seg[8];
  DcnInvSumSeg seg[8][kInvSumSegs];
};
struct DcnInvSumGroup {
  int n;
  DcnInvSum e[kMaxFwdGroup];
  DcnInvSumSched sched;
  int hot_gemm;            // cells whose .pad is set are left to dcn_hot_gemm
};
__global__ void dcn_inv_overflow_sums(const DcnInvSumGroup grp);   // cells above 64 contributions (clusters)
__global__ void dcn_inv_medium_sums(const DcnInvSumGroup grp);     // cells with 9 .. 64 (same grid)
// HOT cells (more than 64 contributions: the key-point cells of a converged head) as the columns of ONE split-operand GEMM per
// (problem, image) over the pixels (round 6, dcn_backward_plane.hip): per distinct offset tensor the hot cells of ALL taps are
// compacted into a column list and their by-pixel weights laid out densely (dcn_hot_build); per problem
// Gov[(t, slot)][o] = sum_px Wd[col][px] * grad_out[o][px] (dcn_hot_gemm).  dcn_inv_overflow_sums skips the cells marked handled.
constexpr int kHotMaxCols = 4096;     // columns per (offset tensor, image) the list holds (see dcn_hot_gemm: never exceeded up to 7x7 kernels)
constexpr int kHotRange = 1152;       // pixels of a map one pass of the GEMM covers (its A tile lives in LDS: 32 rows of
constexpr int kHotRowLd = kHotRange + 4;   // kHotRowLd floats -- 4 mod 64, so the 16-byte row reads of 16 lanes meet 64 distinct banks)
struct DcnHotGemm {        // per problem
  const int4 *cols;        // [N][kHotMaxCols] (tap, slot, start, n), listed by the inverse-record builder
  const int *count;        // [N]
  const int2 *spill;       // the builder's contribution lists (pixel, weight bits)
  const float *gout_t;     // [N][HoWo][O]: the pixel-major grad_output copy of the sums kernel
  float *gov;              // [N * K][max_slots][O_ld]
  int N, K, HoWo, O, O_ld, max_slots, max_cols;
};
struct DcnHotGemmGroup { int n; DcnHotGemm e[kMaxFwdGroup]; };
constexpr int kHotBlocksPerXcd = 32;  // one workgroup per CU (its A tile takes most of the LDS)
__global__ void dcn_hot_gemm(const DcnHotGemmGroup grp, const DcnInvSumSched sched);
size_t dcn_hot_gemm_lds_bytes();
struct DcnPixelMajorItem {
  const float *src;
  float *dst;
  int N, C, P;
  long long src_image_stride;
};
struct DcnPixelMajorGroup {
  int n;
  DcnPixelMajorItem e[kMaxFwdGroup];
};
__global__ void dcn_gout_pixel_major_multi(const DcnPixelMajorGroup grp);
__global__ void dcn_bwd_input_prepare(const DcnInvBuildGroup grp, const DcnPixelMajorGroup pm, int build_blocks, int pm_bx, int pm_by);
// grad_offset on an LDS-resident plane (dcn_backward_offset.hip)
template <int PARTS>
__global__ void dcn_bwd_offset_plane(const DcnFwdGroup grp, float *__restrict__ slabs, int max_K);
__global__ void dcn_bwd_offset_plane_fixup(const DcnFwdGroup grp, const float *__restrict__ slabs, int G, int max_K);
__global__ void dcn_build_grad_taps(const DcnFwdGroup grp);
size_t dcn_bwd_offset_plane_lds_bytes(int parts, int K, int HW, int masked = 0);
__global__ void dcn_bwd_offset_plane_masked(const DcnFwdGroup grp, float *__restrict__ slabs, int max_K);
__global__ void dcn_bwd_offset_plane_fixup_masked(const DcnFwdGroup grp, const float *__restrict__ slabs, int G, int max_K);
int dcn_bwd_offset_plane_threads();
// round 6: tap pairs on 32x32 MFMA blocks, two alternating wave halves, W^T by LDS-DMA (dcn_backward_offset_pair.hip)
__global__ void dcn_bwd_offset_pair(const DcnFwdGroup grp, float *__restrict__ slabs, int max_K);
size_t dcn_bwd_offset_pair_lds_bytes();
int dcn_bwd_offset_pair_threads();
// grad_weight on an LDS-resident plane (dcn_backward_weight_plane.hip)
template <int PARTS>
__global__ void dcn_bwd_weight_plane(const DcnFwdGroup grp, float *__restrict__ slabs);
template <int PARTS>
__global__ void dcn_bwd_weight_gather(const DcnFwdGroup grp, float *__restrict__ slabs);
__global__ void dcn_bwd_weight_plane_fixup(const DcnFwdGroup grp, const float *__restrict__ slabs, int G);
// output-stationary grad_weight (dcn_backward_weight_os.hip): one 256 x (16 channels x 13 taps) tile per workgroup
template <int PARTS>
__global__ void dcn_bwd_weight_os(const DcnFwdGroup grp);
size_t dcn_bwd_weight_os_lds_bytes(int parts, int HW);
int dcn_bwd_weight_os_threads();
int dcn_bwd_weight_os_taps();
struct DcnPackGradOutItem {
  const float *gout;
  void *gq;
  int N, O_total, o_base, Og, HoWo, n_px16, n_mtiles;
};
struct DcnPackGradOut {
  DcnPackGradOutItem item[kMaxFwdGroup];
};
__global__ void dcn_pack_grad_out(const DcnPackGradOut g, int parts);
size_t dcn_bwd_weight_plane_lds_bytes(int parts, int HW);
size_t dcn_bwd_weight_gather_lds_bytes(int parts);
int dcn_bwd_weight_plane_threads();
__global__ void dcn_pack_weight_all(const float *__restrict__ w, float *__restrict__ wpk, float *__restrict__ wpt,
                                    void *__restrict__ wq /*nullable*/, void *__restrict__ wqt /*nullable*/, int Og,
                                    int Cg, int K, int Cg_pad, int Og_pad, int Og_pad16, int Cg_pad256);
__global__ void dcn_pack_weight(const float *__restrict__ w, float *__restrict__ wpk, int Og, int Cg, int K,
                                int Cg_pad, int Og_pad);
__global__ void dcn_unpack_weight(const float *__restrict__ wpk, float *__restrict__ w, int Og, int Cg, int K,
                                  int Cg_pad, int Og_pad, int accumulate);

__global__ void dcn_pack_weight_t(const float *__restrict__ w, float *__restrict__ wpt, int Og, int Cg, int K,
                                  int Og_pad16, int Cg_pad256);

struct DcnBwdInputArgs {
  const float *grad_out;
  float *grad_input;
  float *grad_offset;
  float *grad_mask;
  int Og_pad16, Cg_pad256, n_ctiles, n_ntiles, n_units;
  int direct;
};
struct DcnBwdWeightArgs {
  const float *grad_out;
  int n_ctiles, n_otiles, stages_per_tile, Cg_pad128;
  long long total_units;
};
struct DcnBwdInputLdsArgs {
  const float *grad_out;
  float *slabs;      // [N * n_cslices * S][32][H*W] partial grad_input planes
  float *off_part;   // [n_slices_total][N][2K][HoWo] per-slice partial offset gradients
  float *mask_part;  // [n_slices_total][N][K][HoWo] or nullptr
  int Og_pad16, Cg_pad256, n_cslices, S, n_pblocks, slice_base;
  int HWp;           // row stride of the LDS colgrad tile (output pixels of one image, padded)
};
__global__ void dcn_bwd_build_index(const DcnProblem p, int *__restrict__ row_ptr, int2 *__restrict__ entries);
__global__ void dcn_bwd_input_gather(const DcnProblem p, const DcnBwdInputLdsArgs a, const int *__restrict__ row_ptr,
                                     const int2 *__restrict__ entries);
__global__ void dcn_bwd_input_fixup(const DcnProblem p, const DcnBwdInputLdsArgs a, float *__restrict__ grad_input);
__global__ void dcn_bwd_offset_fixup(const float *__restrict__ off_part, const float *__restrict__ mask_part,
                                     float *__restrict__ grad_offset, float *__restrict__ grad_mask, int n_slices,
                                     int N, int DG, int K, int HoWo, int n_cslices, int Cg, int cpdg);
__global__ void dcn_bwd_weight_mfma(const DcnProblem p, const DcnBwdWeightArgs a, float *__restrict__ slabs);
__global__ void dcn_bwd_weight_fixup(const DcnProblem p, const DcnBwdWeightArgs a, const float *__restrict__ slabs,
                                     int G);
__global__ void dcn_bias_grad(const float *__restrict__ grad_out, float *__restrict__ grad_bias, int N, int O,
                              int HoWo, int accumulate);

}  // namespace kgdet
