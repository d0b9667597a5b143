// C-ABI launchers for deformable convolution (see include/kgdet_hip.h for the contract and the
// reference entry points each one replaces).
#include "common.h"
#include <algorithm>

#include "dcn_kernels.h"

namespace kgdet {

namespace {

struct Derived {
  int Ho, Wo, K, Cg, Og, Cg_pad, Og_pad;
  int Og_pad16, Cg_pad256;  // transposed image for backward-input
  size_t fwd_image_floats() const { return (size_t)K * Cg_pad * Og_pad; }
  size_t bwd_image_floats() const { return (size_t)K * Og_pad16 * Cg_pad256; }
  // bf16 hi/lo image of the plane forward kernel: same element count, 2 x 2 bytes each
  size_t plane_image_floats() const { return (size_t)K * Cg_pad * Og_pad; }
  // its transpose (rows = input channels) for the plane grad_input kernel
  size_t plane_t_image_floats() const { return (size_t)K * Og_pad16 * Cg_pad256; }
};

int derive(const kgdet_dcn_shape *s, Derived &d) {
  // the reference's shape_check (mmdet/ops/dcn/src/deform_conv_cuda.cpp:61-149)
  KGDET_CHECK_SHAPE(s != nullptr, "null shape");
  KGDET_CHECK_SHAPE(s->kh > 0 && s->kw > 0, "kernel size should be greater than zero, but got kH: %d kW: %d",
                    s->kh, s->kw);
  KGDET_CHECK_SHAPE(s->stride_h > 0 && s->stride_w > 0, "stride should be greater than zero, but got dH: %d dW: %d",
                    s->stride_h, s->stride_w);
  KGDET_CHECK_SHAPE(s->dil_h > 0 && s->dil_w > 0,
                    "dilation should be greater than 0, but got dilationH: %d dilationW: %d", s->dil_h, s->dil_w);
  KGDET_CHECK_SHAPE(s->N > 0 && s->C > 0 && s->O > 0 && s->H > 0 && s->W > 0, "empty tensor dimension");
  KGDET_CHECK_SHAPE(s->groups > 0 && s->C % s->groups == 0 && s->O % s->groups == 0,
                    "channels must divide groups (C=%d O=%d groups=%d)", s->C, s->O, s->groups);
  KGDET_CHECK_SHAPE(s->deformable_groups > 0 && s->C % s->deformable_groups == 0,
                    "input channels must divide deformable group size");
  KGDET_CHECK_SHAPE(s->out_channel_offset >= 0 &&
                        (s->out_channels_total == 0 ? s->out_channel_offset == 0
                                                    : s->out_channel_offset + s->O <= s->out_channels_total),
                    "output channel window [%d, %d) does not fit %d channels", s->out_channel_offset,
                    s->out_channel_offset + s->O, s->out_channels_total);
  d.Ho = (s->H + 2 * s->pad_h - (s->dil_h * (s->kh - 1) + 1)) / s->stride_h + 1;
  d.Wo = (s->W + 2 * s->pad_w - (s->dil_w * (s->kw - 1) + 1)) / s->stride_w + 1;
  KGDET_CHECK_SHAPE(d.Ho >= 1 && d.Wo >= 1,
                    "Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). Output size is too small",
                    s->C, s->H, s->W, s->O, d.Ho, d.Wo);
  d.K = s->kh * s->kw;
  d.Cg = s->C / s->groups;
  d.Og = s->O / s->groups;
  d.Cg_pad = (int)align_up(d.Cg, kChunk);
  d.Og_pad = (int)align_up(d.Og, kTileM);
  d.Og_pad16 = (int)align_up(d.Og, kChunk);
  d.Cg_pad256 = (int)align_up(d.Cg, kTileM);
  const long long in_bytes = 4LL * s->N * s->C * s->H * s->W;
  const long long out_bytes = 4LL * s->N * (s->out_channels_total > 0 ? s->out_channels_total : s->O) * d.Ho * d.Wo;
  const long long off_bytes = 4LL * s->N * s->deformable_groups * 2 * d.K * d.Ho * d.Wo;
  KGDET_CHECK_SHAPE(in_bytes < (1LL << 31) && out_bytes < (1LL << 31) && off_bytes < (1LL << 31),
                    "tensor larger than 2 GiB is not supported");
  return KGDET_OK;
}

int grid_size() {
  const int cus = cu_count();
  return cus > 0 ? cus : 256;
}

constexpr int kSlabSlots = 8;  // partial tiles a workgroup may write per launch (plane kernel: one per range it meets)
size_t slab_bytes() { return (size_t)grid_size() * kSlabSlots * kTileElems * sizeof(float); }

// A stream-K workgroup writes one slab per range its slice meets.  `tiles` tiles of `cpt` stages each, dealt to the
// grid alone in a launch: does the slice of one workgroup stay within kSlabSlots ranges?  (Large N / many M tiles /
// a device with few CUs: beyond the bound the slab index would run into the next workgroup's slabs.)
bool slab_slots_ok(long long tiles, int cpt) {
  const long long per_wg = (tiles * cpt + grid_size() - 1) / grid_size();
  return (per_wg + cpt - 1) / cpt + 2 <= kSlabSlots;
}

constexpr size_t kMaxLds = 160 * 1024;

// plane kernels (forward / grad_input): LDS of the launch = the largest need over the group's problems.
// (A second plane buffer, filled by the producer waves while the current chunk runs, was measured and dropped: the
// launch got slower, 256 -> 271 us at B=2 and 598 -> 686 us at B=8 -- the producers are the critical path of a stage
// and the copy adds to exactly their work, while the synchronous copy by all twelve waves is ~6 % of a workgroup.)
size_t plan_plane_lds(DcnFwdGroup &grp, size_t single, size_t fixed) {
  grp.plane_bytes = (int)(single - fixed);
  grp.dbl_plane = 0;
  return single;
}

// Static split-K for the plane kernels (forward / grad_input): cut every problem's reduction into parts of about one
// workgroup's share (whole channel chunks) so that the (problem, part, tile) ranges can be dealt ONE per workgroup.
// Ranges of one (problem, part) are consecutive and consecutive slices sit on one XCD (sk_slice_of_block), so the
// workgroups of an XCD stream the same weight stages at the same time: L2 hits instead of one fabric read per pixel
// tile.  Measured on the head-stage forward (6 problems, B=2): fabric reads 883 -> 174 MB per launch, 265 -> 207 us.
// Falls back to stream-K (static_ranges = 0, kparts = 1) when the ranges outnumber the workgroups or would be too
// uneven.  Problems must have kparts == 1 on entry.
// max_rounds > 1 (forward / grad_input plane kernels): when a workgroup's share is long (B = 8 inference: 747 stages) the
// ranges are cut to ~190 stages all the same and dealt in ROUNDS -- workgroup r computes ranges r, r + G, ... -- instead
// of falling back to stream-K: every range still starts on a channel chunk, the tiles of one (problem, part) still run
// together on one XCD, and the fix-up stays the cheap static one.
void plan_static_ranges(DcnFwdGroup &grp, int G, bool force = false, int max_rounds = 1) {
  grp.static_ranges = 0;
  grp.rounds = 1;
  const long long total = grp.unit_begin[grp.n];
  static const int env_rounds = getenv("KGDET_DCN_ROUNDS") ? atoi(getenv("KGDET_DCN_ROUNDS")) : 0;   // A/B switch
  if (env_rounds > 0 && max_rounds > 1) max_rounds = env_rounds;
  int R = (int)((double)total / G / 220.0 + 0.999);
  R = R < 1 ? 1 : (R > max_rounds ? max_rounds : R);
  const double share = (double)total / G / R;   // target length of a range
  int ranges = 0;
  double longest = 0;
  int kp[kMaxFwdGroup];
  for (int i = 0; i < grp.n; ++i) {
    const DcnProblem &q = grp.p[i];
    int k = (int)((double)q.chunks_per_tile / share + 0.5);
    k = k < 1 ? 1 : (k > q.chunks_per_tap ? q.chunks_per_tap : k);
    kp[i] = k;
    ranges += q.n_ntiles * q.n_mtiles * k;
    const double len = (double)((q.chunks_per_tap + k - 1) / k) * q.seg_stages;
    longest = len > longest ? len : longest;
  }
  static const bool off = getenv("KGDET_DCN_STREAMK") != nullptr;   // A/B switch
  if (force) {   // a kernel without a stream-K fix-up (v2 grad_offset): fewer, longer parts until the ranges fit
    R = 1;
    while (ranges > G) {
      int worst = 0;
      for (int i = 1; i < grp.n; ++i)
        if (kp[i] > kp[worst]) worst = i;
      if (kp[worst] == 1) return;   // more tiles than workgroups: no static schedule
      ranges -= grp.p[worst].n_ntiles * grp.p[worst].n_mtiles * (kp[worst] - (kp[worst] + 1) / 2);
      kp[worst] = (kp[worst] + 1) / 2;
    }
  } else if (off || ranges > R * G || longest > 1.25 * share) {
    return;
  }
  grp.static_ranges = 1;
  grp.rounds = (ranges + G - 1) / G;
  for (int i = 0; i < grp.n; ++i) {
    grp.p[i].kparts = kp[i];
    grp.range_begin[i + 1] = grp.range_begin[i] + grp.p[i].n_ntiles * grp.p[i].n_mtiles * kp[i];
  }
}

// fix-up launch of a plane-kernel group: the static schedule has its own, cheaper kernel
void launch_plane_fixup(const DcnFwdGroup &grp, const void *workspace, int G, void *stream) {
  if (grp.static_ranges)
    hipLaunchKernelGGL(dcn_fwd_fixup_static, dim3(grp.tile_begin[grp.n], 2), dim3(kThreads), 0, (hipStream_t)stream, grp,
                       (const float *)workspace, G);
  else
    hipLaunchKernelGGL(dcn_fwd_fixup, dim3(grp.tile_begin[grp.n], 16), dim3(kThreads), 0, (hipStream_t)stream, grp,
                       (const float *)workspace, G);
}

// Grid of a plane-kernel launch: small groups run on fewer workgroups, so that a workgroup's share is at least ~16 stages.
// (With all 256 workgroups on the 7 x 11 level of a five-level head -- 2 tiles x 144 stages -- every workgroup computed one stage
// and wrote a 128 KB slab; the fix-up then added 128 slabs per tile with two workgroups: 128 us per launch, four launches per step.)
int small_launch_grid(const DcnFwdGroup &grp, int G) {
  static const bool off = getenv("KGDET_DCN_FULL_GRID") != nullptr;   // A/B switch
  int min_len = 1 << 30;                                 // (a slice must not meet more ranges than it has slab slots)
  for (int i = 0; i < grp.n; ++i) min_len = std::min(min_len, grp.p[i].chunks_per_tile);
  const long long per = std::max<long long>(1, std::min<long long>(16, (long long)(kSlabSlots - 2) * min_len));
  const long long g = grp.unit_begin[grp.n] / per;
  return off ? G : (int)std::max<long long>(1, std::min<long long>(G, g));
}

// LDS-privatised backward-input: one (image, 32-channel slice) plane set must fit in LDS
struct BwdLdsPlan {
  bool ok;
  int n_cslices, S, n_blocks, n_pblocks, n_slices_total, HWp;
  size_t lds_bytes, slab_floats, off_floats, mask_floats, rowptr_ints, entry_pairs, index_lds_bytes;
};

BwdLdsPlan plan_bwd_lds(const kgdet_dcn_shape *s, const Derived &d) {
  BwdLdsPlan pl{};
  pl.HWp = (int)align_up((size_t)d.Ho * d.Wo, 64);
  pl.lds_bytes = (size_t)32 * pl.HWp * sizeof(float) + 8 * 256 * 8;  // colgrad tile [32][HWp] + 8 entry windows
  const int cpdg = s->C / s->deformable_groups;
  // a 32-channel slice must not straddle deformable groups
  const bool dg_ok = s->deformable_groups == 1 || cpdg % d.Cg == 0 || (d.Cg % cpdg == 0 && cpdg % 32 == 0);
  pl.index_lds_bytes = ((size_t)2 * s->H * s->W + 2) * sizeof(int) + (size_t)4 * d.Ho * d.Wo * 8;
  pl.ok = pl.lds_bytes <= kMaxLds && pl.index_lds_bytes <= kMaxLds - 64 && s->H * s->W <= 17 * 64 && dg_ok;
  pl.n_cslices = ceil_div(d.Cg, 32);
  const int pairs = s->N * pl.n_cslices;
  pl.S = grid_size() / pairs;
  if (pl.S < 1) pl.S = 1;
  if (pl.S > 64) pl.S = 64;
  pl.n_blocks = pairs * pl.S;
  pl.n_pblocks = ceil_div(d.Ho * d.Wo, 64);
  pl.n_slices_total = s->groups * pl.n_cslices;
  pl.slab_floats = (size_t)pl.n_blocks * 32 * s->H * s->W;
  pl.off_floats = (size_t)pl.n_slices_total * s->N * 2 * d.K * d.Ho * d.Wo;
  pl.mask_floats = pl.off_floats / 2;
  pl.rowptr_ints = (size_t)s->N * s->deformable_groups * d.K * ((size_t)s->H * s->W + 1);
  pl.entry_pairs = (size_t)s->N * s->deformable_groups * d.K * 4 * d.Ho * d.Wo;
  return pl;
}

// the MFMA kernels gather 4 consecutive channels per thread with one Tap, so a deformable group
// boundary must not fall inside such a quad
bool mfma_ok(const kgdet_dcn_shape *s) {
  return s->W >= 2 && ((s->C / s->deformable_groups) % 4 == 0 || s->deformable_groups == 1);
}
// plane forward kernel: a 16-channel slice of one input image must fit in LDS next to the operand stages
bool plane_ok(const kgdet_dcn_shape *s, const Derived &d) {
  const int cpdg = s->C / s->deformable_groups;  // a producer thread samples a 16-channel chunk with one tap record
  return mfma_ok(s) && (s->deformable_groups == 1 || cpdg % 16 == 0) && s->H * s->W <= kPlaneMaxHW &&
         (size_t)8 * 33 * (d.K + 1) * sizeof(float) <= kMaxLds - 64 &&
         slab_slots_ok((long long)s->N * ceil_div(d.Ho * d.Wo, kTileN) * (d.Og_pad / kTileM), d.K * (d.Cg_pad / kChunk));
}
// forward on maps beyond the LDS plane: producers gather from a pixel-major copy of x (plane_role MODE 2); offsets into
// that copy and into the weight image must stay below 4 GB
bool gather_ok(const kgdet_dcn_shape *s, const Derived &d) {
  const int cpdg = s->C / s->deformable_groups;
  return mfma_ok(s) && (s->deformable_groups == 1 || cpdg % 16 == 0) && !plane_ok(s, d) &&
         s->C % 4 == 0 &&                                     // (a pixel's channel quads are 16-byte loads)
         (size_t)s->N * s->H * s->W * s->C * 4 < ((size_t)1 << 32) - 4096 &&
         (size_t)s->N * s->deformable_groups * d.K * d.Ho * d.Wo * sizeof(DcnTapRec) < ((size_t)1 << 32) &&
         (size_t)8 * 33 * (d.K + 1) * sizeof(float) <= kMaxLds - 64;
}
size_t gather_table_bytes(const kgdet_dcn_shape *s, const Derived &d) {
  return align_up((size_t)s->N * s->deformable_groups * d.K * d.Ho * d.Wo * sizeof(DcnTapRec), 256);
}
size_t gather_image_bytes(const kgdet_dcn_shape *s) { return align_up((size_t)s->N * s->H * s->W * s->C * 4 + 256, 256); }
size_t tap_table_bytes(const kgdet_dcn_shape *s, const Derived &d) {
  return plane_ok(s, d) ? (size_t)s->N * s->deformable_groups * d.K * d.Ho * d.Wo * sizeof(DcnTapRec) : 0;
}
// Channel runs of the backward plane kernels: maximal runs of input channels inside one weight group AND one deformable
// group (one run = one sub-problem: its own tap / inverse records, its weight group's operand image and grad_out window).
struct ChannelRun { int c0, c1, g, dgi; };
int channel_runs(const kgdet_dcn_shape *s, const Derived &d, ChannelRun *runs, int cap) {
  const int cpdg = s->C / s->deformable_groups;
  int n = 0;
  for (int c0 = 0; c0 < s->C;) {
    const int g = c0 / d.Cg, dgi = c0 / cpdg;
    const int c1 = std::min((g + 1) * d.Cg, (dgi + 1) * cpdg);
    if (n < cap) runs[n] = ChannelRun{c0, c1, g, dgi};
    ++n;
    c0 = c1;
  }
  return n;
}
// grad_input on the plane kernel: a 16-channel slice of one grad_output image in LDS; a channel run must start on a
// 256-row tile of the transposed weight image or end inside the tile it starts in
bool plane_bwd_input_ok(const kgdet_dcn_shape *s, const Derived &d) {
  ChannelRun runs[kMaxFwdGroup];
  const int n_runs = channel_runs(s, d, runs, kMaxFwdGroup);
  if (n_runs > kMaxFwdGroup) return false;
  for (int i = 0; i < n_runs; ++i) {
    const int row0 = (runs[i].c0 - runs[i].g * d.Cg) % kTileM;
    if (row0 != 0 && row0 + (runs[i].c1 - runs[i].c0) > kTileM) return false;
  }
  return d.Ho * d.Wo <= kPlaneMaxHW && s->H * s->W <= kPlaneMaxHW && s->W >= 1 &&
         (size_t)8 * 33 * (d.K + 1) * sizeof(float) <= kMaxLds - 64 &&
         dcn_build_inverse_taps_lds_bytes(s->H * s->W, d.Ho * d.Wo) <= kMaxLds - 64 &&
         slab_slots_ok((long long)n_runs * s->N * ceil_div(s->H * s->W, kTileN) * (d.Cg_pad256 / kTileM),
                       d.K * (d.Og_pad16 / kChunk));
}
// grad_offset on the plane kernel: <= 256 output channels per group; every deformable group inside ONE weight group (its
// sum then comes from one sub-problem; a deformable group spread over several weight groups would need a sum over them),
// in whole 16-channel chunks
bool plane_bwd_offset_ok(const kgdet_dcn_shape *s, const Derived &d, bool masked = false) {
  const int cpdg = s->C / s->deformable_groups;
  const bool runs_ok = (s->groups == 1 && s->deformable_groups == 1) ||
                       (d.Cg % cpdg == 0 && cpdg % kChunk == 0 && s->deformable_groups <= kMaxFwdGroup);
  return runs_ok && d.Og <= 256 && d.K <= 64 && s->H * s->W <= kPlaneMaxHW &&
         dcn_bwd_offset_plane_lds_bytes(2, d.K, s->H * s->W, masked) <= kMaxLds &&
         (size_t)8 * 33 * (d.K + 1) * sizeof(float) <= kMaxLds - 64 &&
         slab_slots_ok((long long)s->N * ceil_div(d.Ho * d.Wo, kTileN), d.K * (ceil_div(cpdg, kChunk)));
}
size_t grad_tap_bytes(const kgdet_dcn_shape *s, const Derived &d) { return (size_t)s->N * s->deformable_groups * d.K * d.Ho * d.Wo * 64; }   // (v2 records: 64 B; one table per deformable group)
// Sum groups of a backward launch (DcnProblem::sum_count): problems whose OUTPUT POINTERS coincide are summed into that one tensor
// by the fix-up.  Members must tile alike (same images, pixels, rows); at most four per group.  Returns false otherwise.
bool assign_sum_groups(DcnFwdGroup &grp, const void *const *outs, bool same_taps) {
  for (int i = 0; i < grp.n; ++i) { grp.p[i].sum_count = 0; }
  for (int i = 0; i < grp.n; ++i) {
    int lead = i;
    for (int j = 0; j < i; ++j)
      if (outs[j] == outs[i]) { lead = j; break; }
    if (lead == i) continue;
    DcnProblem &L = grp.p[lead];
    const DcnProblem &q = grp.p[i];
    if (L.sum_count == 0) { L.sum_count = 1; L.sum_members[0] = lead; }
    if (L.sum_count >= 4 || q.N != L.N || q.HoWo != L.HoWo || q.Og != L.Og || q.n_ntiles != L.n_ntiles || q.n_mtiles != L.n_mtiles ||
        q.tiles_per_image != L.tiles_per_image || (same_taps && q.K != L.K))
      return false;
    L.sum_members[L.sum_count++] = i;
  }
  for (int i = 0; i < grp.n; ++i) {        // every member carries the group (the kernels test sum_count, the fix-up the leader)
    const DcnProblem &L = grp.p[i];
    if (L.sum_count > 1 && L.sum_members[0] == i)
      for (int m = 1; m < L.sum_count; ++m) {
        DcnProblem &q = grp.p[L.sum_members[m]];
        q.sum_count = L.sum_count;
        for (int e = 0; e < 4; ++e) q.sum_members[e] = L.sum_members[e];
      }
  }
  return true;
}
bool has_sum_groups(const DcnFwdGroup &grp) {
  for (int i = 0; i < grp.n; ++i)
    if (grp.p[i].sum_count > 1) return true;
  return false;
}
// grad_offset on tap pairs (dcn_backward_offset_pair.hip): v1, split operands, one static range per workgroup, K >= 3 (a tap's
// running sum is re-read one segment later, one slot ahead of its use: with a single pair per segment that would be the slot it is
// written in)
bool offset_pair_ok(const DcnFwdGroup &grp) {
  static const bool off = getenv("KGDET_DCN_OFFSET_PAIR") && atoi(getenv("KGDET_DCN_OFFSET_PAIR")) == 0;   // A/B switch
  if (off || !grp.static_ranges || grp.rounds != 1) return false;
  for (int i = 0; i < grp.n; ++i) {
    const DcnProblem &q = grp.p[i];
    if (q.mask || q.K < 3 || q.Og > 256 || q.Og % 32 != 0 || q.H * q.W > kPlaneMaxHW) return false;   // (Og % 32: a DMA piece = two 16-o chunks)
  }
  return true;
}
// Blocked copies of the distinct inputs of a grad_offset group behind its record tables (written by dcn_build_grad_taps), where
// the workspace has room for them; without one a problem's plane switches copy through registers.
void place_offset_xblk(DcnFwdGroup &grp, unsigned char *table_base, size_t used, size_t table_cap) {
  static const bool off = getenv("KGDET_DCN_OFFSET_XBLK") && atoi(getenv("KGDET_DCN_OFFSET_XBLK")) == 0;   // A/B switch
  used = align_up(used, 256);
  for (int i = 0; i < grp.n; ++i) {
    DcnProblem &q = grp.p[i];
    q.xblk = nullptr; q.build_xblk = 0;
    if (off) continue;
    for (int j = 0; j < i && !q.xblk; ++j) {
      const DcnProblem &o = grp.p[j];
      if (o.xblk && o.x == q.x && o.c_base == q.c_base && o.Cg == q.Cg && o.N == q.N && o.H == q.H && o.W == q.W && o.C_total == q.C_total)
        q.xblk = o.xblk;
    }
    if (q.xblk) continue;
    const size_t xb = align_up(dcn_xblk_bytes(q.N, q.Cg_pad, q.H * q.W), 256);
    if (used + xb > table_cap) continue;
    q.xblk = reinterpret_cast<const float *>(table_base + used);
    q.build_xblk = 1;
    used += xb;
  }
}
int launch_offset_pair(const DcnFwdGroup &grp, int G, void *workspace, int max_K, void *stream) {
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_offset_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
    attr_set = true;
  }
  hipLaunchKernelGGL(dcn_bwd_offset_pair, dim3(G), dim3(dcn_bwd_offset_pair_threads()), dcn_bwd_offset_pair_lds_bytes(),
                     (hipStream_t)stream, grp, (float *)workspace, max_K);
  return KGDET_OK;
}
struct InvTables {
  size_t rec_bytes, hdr_bytes, cell_bytes, spill_bytes;
  size_t hot_cols_bytes, hot_count_bytes;   // the hot cells' column list (dcn_hot_gemm)
  size_t total() const { return rec_bytes + hdr_bytes + cell_bytes + spill_bytes + hot_cols_bytes + hot_count_bytes; }
};
InvTables inv_tables(const kgdet_dcn_shape *s, const Derived &d) {   // of ONE deformable group (functions of the offsets alone)
  InvTables t;
  const int slots = dcn_inv_max_slots(s->H * s->W, d.Ho * d.Wo);
  t.rec_bytes = (size_t)s->N * d.K * s->H * s->W * 64;
  t.hdr_bytes = align_up((size_t)s->N * d.K * sizeof(int), 64);
  t.cell_bytes = align_up((size_t)s->N * d.K * slots * sizeof(DcnInvOvfCell), 64);
  t.spill_bytes = (size_t)s->N * d.K * 4 * d.Ho * d.Wo * 8;
  t.hot_cols_bytes = align_up((size_t)s->N * kHotMaxCols * sizeof(int4), 256);
  t.hot_count_bytes = align_up((size_t)s->N * sizeof(int), 256);
  return t;
}
size_t inv_tables_all(const kgdet_dcn_shape *s, const Derived &d) { return s->deformable_groups * align_up(inv_tables(s, d).total(), 256); }
// what depends on grad_output as well: the pre-aggregated sums of the long cells (one table per deformable group) and
// the pixel-major copy of the convolution's grad_output channels they are formed from
int inv_gov_ld(const kgdet_dcn_shape *s, const Derived &d) { return s->groups * d.Og_pad16; }
size_t inv_gov_bytes(const kgdet_dcn_shape *s, const Derived &d) {   // of ONE deformable group
  return align_up((size_t)s->N * d.K * dcn_inv_max_slots(s->H * s->W, d.Ho * d.Wo) * inv_gov_ld(s, d) * sizeof(float), 256);
}
size_t inv_gout_t_bytes(const kgdet_dcn_shape *s, const Derived &d) { return align_up((size_t)s->N * d.Ho * d.Wo * s->O * sizeof(float), 256); }
size_t inv_sums_all(const kgdet_dcn_shape *s, const Derived &d) { return s->deformable_groups * inv_gov_bytes(s, d) + inv_gout_t_bytes(s, d); }
// backward tiles (256 / 128 channels wide) must lie inside one deformable group
bool mfma_bwd_ok(const kgdet_dcn_shape *s) {
  const int cpdg = s->C / s->deformable_groups, Cg = s->C / s->groups;
  return s->deformable_groups == 1 || cpdg % Cg == 0 || (Cg % cpdg == 0 && cpdg % kTileM == 0);
}

void fill_problem(const kgdet_dcn_shape *s, const Derived &d, int group, DcnProblem &p) {
  p = DcnProblem{};
  p.N = s->N; p.C_total = s->C; p.c_base = group * d.Cg; p.Cg = d.Cg; p.Cg_pad = d.Cg_pad;
  p.O_total = s->out_channels_total > 0 ? s->out_channels_total : s->O;
  p.o_base = s->out_channel_offset + group * d.Og; p.Og = d.Og; p.Og_pad = d.Og_pad;
  p.bias_base = group * d.Og;
  p.H = s->H; p.W = s->W; p.Ho = d.Ho; p.Wo = d.Wo; p.HoWo = d.Ho * d.Wo; p.P = s->N * p.HoWo;
  p.kh = s->kh; p.kw = s->kw; p.K = d.K; p.seg_stages = d.K;
  p.sh = s->stride_h; p.sw = s->stride_w; p.ph = s->pad_h; p.pw = s->pad_w; p.dh = s->dil_h; p.dw = s->dil_w;
  p.DG = s->deformable_groups; p.cpdg = s->C / s->deformable_groups;
}

}  // namespace

}  // namespace kgdet

using namespace kgdet;

extern "C" {

int kgdet_dcn_output_size(const kgdet_dcn_shape *s, int32_t *Ho, int32_t *Wo) {
  Derived d;
  if (int rc = derive(s, d)) return rc;
  if (Ho) *Ho = d.Ho;
  if (Wo) *Wo = d.Wo;
  return KGDET_OK;
}

size_t kgdet_dcn_packed_weight_bytes(const kgdet_dcn_shape *s) {
  Derived d;
  if (derive(s, d)) return 0;
  // [forward image per group ...][transposed (backward-input) image per group ...]
  // ... [bf16 hi/lo image of the plane forward kernel per group ...]
  // ... [its transpose per group ...]
  return (size_t)s->groups * (d.fwd_image_floats() + d.bwd_image_floats() + d.plane_image_floats() +
                              d.plane_t_image_floats()) * sizeof(float);
}

size_t kgdet_dcn_workspace_bytes(const kgdet_dcn_shape *s) {
  Derived d;
  if (derive(s, d)) return 0;
  // slabs for stream-K partial tiles + (forward) the tap records / (backward-weight) a packed gradient image
  const size_t after_slabs = (size_t)s->groups * d.fwd_image_floats() * sizeof(float);
  // (forward: the tap records and the blocked copy of the input the chained plane hand-over reads, behind them)
  const size_t fwd_tables = tap_table_bytes(s, d) +
                            (plane_ok(s, d) ? 512 + s->groups * align_up(dcn_xblk_bytes(s->N, d.Cg_pad, s->H * s->W), 256) : 0);
  const size_t fwd_and_wgrad = slab_bytes() + (after_slabs > fwd_tables ? after_slabs : fwd_tables);
  const BwdLdsPlan pl = plan_bwd_lds(s, d);
  const size_t bwd_in = pl.ok ? (pl.slab_floats + pl.off_floats + pl.mask_floats) * sizeof(float) +
                                    pl.rowptr_ints * sizeof(int) + pl.entry_pairs * 8 + 64
                              : 0;
  size_t bwd_in_plane = plane_bwd_input_ok(s, d) ? slab_bytes() + inv_tables_all(s, d) + inv_sums_all(s, d) : 0;
  const size_t off_tabs = align_up(grad_tap_bytes(s, d), 256) + s->deformable_groups * align_up(dcn_xblk_bytes(s->N, d.Cg_pad, s->H * s->W), 256);
  if (plane_bwd_offset_ok(s, d) && slab_bytes() + off_tabs > bwd_in_plane) bwd_in_plane = slab_bytes() + off_tabs;
  size_t need = fwd_and_wgrad > bwd_in ? fwd_and_wgrad : bwd_in;
  need = need > bwd_in_plane ? need : bwd_in_plane;
  if (plane_ok(s, d)) {   // grad_weight on the plane kernel: records + one grad_out image per weight group
    const size_t wplane = slab_bytes() + align_up(tap_table_bytes(s, d), 256) +
                          (size_t)s->groups * (d.Og_pad / kTileM) * s->N * ceil_div(d.Ho * d.Wo, kChunk) * 16384;
    need = need > wplane ? need : wplane;
  }
  if (gather_ok(s, d)) {   // forward / grad_weight on large maps: records (+ grad_out images) + the pixel-major copy of x
    const size_t gw = slab_bytes() + gather_table_bytes(s, d) + gather_image_bytes(s) +
                      (size_t)s->groups * (d.Og_pad / kTileM) * s->N * ceil_div(d.Ho * d.Wo, kChunk) * 16384;
    need = need > gw ? need : gw;
  }
  if (!(pl.ok && mfma_bwd_ok(s)) || !plane_bwd_input_ok(s, d)) {   // the materialised column gradient of dcn_backward_large.hip, per channel run
    DcnProblem p;
    fill_problem(s, d, 0, p);
    const int cpdg_ = s->C / s->deformable_groups;
    p.C_total = d.Cg < cpdg_ ? d.Cg : cpdg_;          // (the longest channel run)
    if (dcn_bwd_large_ok(p, false, 1)) {
      const size_t big = dcn_bwd_large_workspace_bytes(p);
      need = need > big ? need : big;
    }
  }
  return need;
}

size_t kgdet_dcn_group_workspace_bytes(int32_t n, const kgdet_dcn_shape *const *shapes) {
  size_t tables = 0, bwd_tables = 0, wgrad_tables = 0, off_tables = 0, single = 0;
  for (int i = 0; i < n; ++i) {
    Derived d;
    if (!shapes || derive(shapes[i], d)) return 0;
    tables += tap_table_bytes(shapes[i], d);
    if (plane_ok(shapes[i], d))   // (a blocked copy of the input behind the tap records: the forward's chained plane hand-over)
      tables += 512 + shapes[i]->groups * align_up(dcn_xblk_bytes(shapes[i]->N, d.Cg_pad, shapes[i]->H * shapes[i]->W), 256);
    if (plane_bwd_input_ok(shapes[i], d)) bwd_tables += inv_tables_all(shapes[i], d) + inv_sums_all(shapes[i], d);
    if (plane_bwd_offset_ok(shapes[i], d))   // (grad_offset phase: records + the blocked copy of the input of the tap-pair kernel)
      off_tables += align_up(grad_tap_bytes(shapes[i], d), 256) + align_up(dcn_xblk_bytes(shapes[i]->N, d.Cg_pad, shapes[i]->H * shapes[i]->W), 256) + 256;
    wgrad_tables += align_up(tap_table_bytes(shapes[i], d), 256) +
                    (size_t)shapes[i]->groups * (d.Og_pad / kTileM) * shapes[i]->N * ceil_div(d.Ho * d.Wo, kChunk) * 16384;
    const size_t w = kgdet_dcn_workspace_bytes(shapes[i]);
    single = w > single ? w : single;
  }
  if (wgrad_tables > bwd_tables) bwd_tables = wgrad_tables;
  if (off_tables > bwd_tables) bwd_tables = off_tables;
  const size_t grouped = slab_bytes() + (tables > bwd_tables ? tables : bwd_tables);
  return grouped > single ? grouped : single;
}

int kgdet_dcn_pack_weight(const kgdet_dcn_shape *s, const float *weight, float *packed, void *stream) {
  Derived d;
  if (int rc = derive(s, d)) return rc;
  KGDET_CHECK_SHAPE(weight && packed, "null pointer");
  const size_t lds_all = (size_t)8 * 33 * (d.K + 1) * sizeof(float);
  const bool fused = lds_all <= kMaxLds - 64;
  const size_t lds = (size_t)64 * (d.K + 1) * sizeof(float);
  KGDET_CHECK_SHAPE(fused || lds <= 64 * 1024, "kernel %dx%d too large to pack", s->kh, s->kw);
  if (fused) {
    static thread_local bool attr_set = false;
    if (!attr_set) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_pack_weight_all,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds - 64));
      attr_set = true;
    }
  }
  for (int g = 0; g < s->groups; ++g) {
    const float *w = weight + (size_t)g * d.Og * d.Cg * d.K;
    float *dst = packed + (size_t)g * d.fwd_image_floats();
    float *dst_t = packed + (size_t)s->groups * d.fwd_image_floats() + (size_t)g * d.bwd_image_floats();
    float *dst_q = packed + (size_t)s->groups * (d.fwd_image_floats() + d.bwd_image_floats()) +
                   (size_t)g * d.plane_image_floats();
    float *dst_qt = packed + (size_t)s->groups * (d.fwd_image_floats() + d.bwd_image_floats() +
                                                   d.plane_image_floats()) + (size_t)g * d.plane_t_image_floats();
    if (fused) {
      dim3 grid(d.Cg_pad256 / 8, d.Og_pad / 32);
      // the bf16 images depend on the WEIGHT alone (never on the map size or batch: a packed weight is reused across
      // pyramid levels and batch sizes -- kgdet_amd/dcn.py caches it for inference)
      const int cpdg_ = s->C / s->deformable_groups;
      const bool plane = ((s->C / s->deformable_groups) % 4 == 0 || s->deformable_groups == 1) &&
                         (s->deformable_groups == 1 || cpdg_ % 16 == 0);
      hipLaunchKernelGGL(dcn_pack_weight_all, grid, dim3(256), lds_all, (hipStream_t)stream, w, dst, dst_t,
                         plane ? (void *)dst_q : nullptr, plane ? (void *)dst_qt : nullptr, d.Og, d.Cg, d.K,
                         d.Cg_pad, d.Og_pad, d.Og_pad16, d.Cg_pad256);
    } else {
      dim3 grid(d.Cg_pad, d.Og_pad / 64);
      hipLaunchKernelGGL(dcn_pack_weight, grid, dim3(256), lds, (hipStream_t)stream, w, dst, d.Og, d.Cg, d.K,
                         d.Cg_pad, d.Og_pad);
      dim3 grid_t(d.Og_pad16, d.Cg_pad256 / 64);
      hipLaunchKernelGGL(dcn_pack_weight_t, grid_t, dim3(256), lds, (hipStream_t)stream, w, dst_t, d.Og, d.Cg, d.K,
                         d.Og_pad16, d.Cg_pad256);
    }
  }
  KGDET_CHECK_LAUNCH("dcn_pack_weight");
  return KGDET_OK;
}

// n weights in ONE launch (every weight: one weight group, fused pack kernel); anything else: one call each
int kgdet_dcn_pack_weight_multi(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *weights,
                                float *const *packeds, void *stream) {
  return kgdet_dcn_pack_weight_images(n, shapes, weights, packeds, 3, stream);
}

// 1 when EVERY product of a convolution of this shape -- forward, grad_input, grad_offset, grad_weight -- has a split-operand
// kernel: such a convolution never reads the two fp32 images of its packed weight
int32_t kgdet_dcn_split_path_complete(const kgdet_dcn_shape *s) {
  Derived d;
  if (!s || derive(s, d)) return 0;
  return s->groups == 1 && s->deformable_groups == 1 && plane_ok(s, d) && plane_bwd_input_ok(s, d) && plane_bwd_offset_ok(s, d)
             ? 1 : 0;
}

int kgdet_dcn_pack_weight_images(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *weights,
                                 float *const *packeds, uint32_t images, void *stream) {
  KGDET_CHECK_SHAPE(n >= 1 && shapes && weights && packeds, "null pointer");
  KGDET_CHECK_SHAPE((images & 3u) != 0 && (images & ~3u) == 0, "images: bit 0 = the fp32 images, bit 1 = the split (bf16 hi/lo) images");
  int i = 0;
  while (i < n) {
    DcnPackGroup grp;
    int m = 0, gx = 0, gy = 0;
    size_t lds = 0;
    for (; i < n && m < kMaxFwdGroup; ++i) {
      const kgdet_dcn_shape *s = shapes[i];
      Derived d;
      if (int rc = derive(s, d)) return rc;
      KGDET_CHECK_SHAPE(weights[i] && packeds[i], "null pointer (weight %d)", i);
      const size_t lds_all = (size_t)8 * 33 * (d.K + 1) * sizeof(float);
      if (s->groups != 1 || lds_all > kMaxLds - 64) {   // not the fused single-group case: on its own (all images)
        if (m > 0) break;
        if (int rc = kgdet_dcn_pack_weight(s, weights[i], packeds[i], stream)) return rc;
        continue;
      }
      const int cpdg_ = s->C / s->deformable_groups;
      const bool plane = ((s->C / s->deformable_groups) % 4 == 0 || s->deformable_groups == 1) &&
                         (s->deformable_groups == 1 || cpdg_ % 16 == 0);
      DcnPackOne &e = grp.e[m++];
      float *packed = packeds[i];
      e.w = weights[i];
      // (a weight without split images keeps its fp32 ones whatever the mask says: nothing else could serve it)
      const bool fp32_images = (images & 1u) || !plane || !(images & 2u);
      e.wpk = fp32_images ? packed : nullptr;
      e.wpt = fp32_images ? packed + d.fwd_image_floats() : nullptr;
      e.wq = plane && (images & 2u) ? (void *)(packed + d.fwd_image_floats() + d.bwd_image_floats()) : nullptr;
      e.wqt = plane && (images & 2u) ? (void *)(packed + d.fwd_image_floats() + d.bwd_image_floats() + d.plane_image_floats()) : nullptr;
      e.Og = d.Og; e.Cg = d.Cg; e.K = d.K; e.Cg_pad = d.Cg_pad; e.Og_pad = d.Og_pad; e.Og_pad16 = d.Og_pad16;
      e.Cg_pad256 = d.Cg_pad256;
      gx = d.Cg_pad256 / 8 > gx ? d.Cg_pad256 / 8 : gx;
      gy = d.Og_pad / 32 > gy ? d.Og_pad / 32 : gy;
      lds = lds_all > lds ? lds_all : lds;
    }
    if (m > 0) {
      static thread_local bool attr_set = false;
      if (!attr_set) {
        KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_pack_weight_all_multi,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds - 64));
        attr_set = true;
      }
      hipLaunchKernelGGL(dcn_pack_weight_all_multi, dim3(gx, gy, m), dim3(256), lds, (hipStream_t)stream, grp);
      KGDET_CHECK_LAUNCH("dcn_pack_weight_all_multi");
    }
  }
  return KGDET_OK;
}


int kgdet_dcn_unpack_weight_grad(const kgdet_dcn_shape *s, const float *packed, float *grad_weight,
                                 int accumulate, void *stream) {
  Derived d;
  if (int rc = derive(s, d)) return rc;
  KGDET_CHECK_SHAPE(grad_weight && packed, "null pointer");
  const size_t lds = (size_t)64 * (d.K + 1) * sizeof(float);
  KGDET_CHECK_SHAPE(lds <= 64 * 1024, "kernel %dx%d too large to unpack", s->kh, s->kw);
  for (int g = 0; g < s->groups; ++g) {
    float *w = grad_weight + (size_t)g * d.Og * d.Cg * d.K;
    const float *src = packed + (size_t)g * d.fwd_image_floats();
    dim3 grid(d.Cg, d.Og_pad / 64);
    hipLaunchKernelGGL(dcn_unpack_weight, grid, dim3(256), lds, (hipStream_t)stream, src, w, d.Og, d.Cg, d.K,
                       d.Cg_pad, d.Og_pad, accumulate);
  }
  KGDET_CHECK_LAUNCH("dcn_unpack_weight");
  return KGDET_OK;
}

int kgdet_deform_conv_forward_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                      const float *const *offsets, const float *const *masks,
                                      const float *const *packed_weights, const float *const *biases,
                                      float *const *outputs, uint32_t flags, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  KGDET_CHECK_SHAPE(n >= 1 && shapes && inputs && offsets && packed_weights && outputs, "null pointer / empty group");
  if (workspace_bytes < slab_bytes() || workspace == nullptr) {
    set_error("workspace too small: need %zu bytes, got %zu", slab_bytes(), workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  const int G = grid_size();
  DcnFwdGroup grp;  // problems collected for one launch of the plane kernel
  grp.n = 0;
  grp.xcd_slices = 1;
  grp.slots = kSlabSlots;
  grp.dbl_plane = 0;
  grp.plane_bytes = 0;
  grp.pair_mode = 0; grp.gather_mode = 0;
  grp.rounds = 1;
  grp.wave_layout = 1;
  size_t lds = 0;
  int min_len = 1 << 30;  // shortest range (stages) in the pending group
  unsigned char *const table_base = (unsigned char *)workspace + slab_bytes();
  size_t table_used = 0;  // bytes of tap records placed behind the slabs for the pending group
  const int parts = (flags & KGDET_DCN_BF16) ? 1 : 2;
  const bool grp_pair = false;   // (the tap-pair kernel of rounds 2-4 left the library in round 5: tools/experiments/dcn_plane_pairs.h)
  int max_hw = 0;
  auto flush = [&]() -> int {
    if (grp.n == 0) return KGDET_OK;
    const int Gf = G;                                  // (the full grid: record builders)
    const int G = small_launch_grid(grp, Gf);
    plan_static_ranges(grp, G, false, kSlabSlots - 2);
    static thread_local bool attr_set = false;
    if (!attr_set) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_fwd_plane<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kMaxLds));
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_fwd_plane<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kMaxLds));
      attr_set = true;
    }
    grp.pair_mode = grp_pair ? 1 : 0;
    static const int chain_dma = getenv("KGDET_DCN_CHAIN_DMA") ? atoi(getenv("KGDET_DCN_CHAIN_DMA")) : 1;   // A/B switch
    const bool use_xblk = chain_dma && parts == 2 && !grp_pair;   // (plane kernel: chained segment hand-over by LDS-DMA)
    for (int i = 0; i < grp.n; ++i) { grp.p[i].xblk = nullptr; grp.p[i].build_xblk = 0; }
    if (use_xblk) {   // blocked copies of the distinct inputs behind the tap records (written by dcn_build_taps' blocks)
      size_t used = align_up(table_used, 256);
      for (int i = 0; i < grp.n; ++i) {
        DcnProblem &q = grp.p[i];
        for (int j = 0; j < i && !q.xblk; ++j) {
          const DcnProblem &o = grp.p[j];
          if (o.x == q.x && o.c_base == q.c_base && o.Cg == q.Cg && o.N == q.N && o.H == q.H && o.W == q.W && o.C_total == q.C_total)
            q.xblk = o.xblk;
        }
        if (!q.xblk) {
          const size_t xb = align_up(dcn_xblk_bytes(q.N, q.Cg_pad, q.H * q.W), 256);
          if (slab_bytes() + used + xb > workspace_bytes) {
            set_error("workspace too small for the blocked inputs: need %zu bytes, got %zu (kgdet_dcn_group_workspace_bytes)",
                      slab_bytes() + used + xb, workspace_bytes);
            return KGDET_E_WORKSPACE;
          }
          q.xblk = reinterpret_cast<const float *>(table_base + used);
          q.build_xblk = 1;
          used += xb;
        }
      }
    }
    hipLaunchKernelGGL(dcn_build_taps, dim3(2 * Gf, grp.n), dim3(256), 0, (hipStream_t)stream, grp);
    const int threads = dcn_fwd_plane_threads();
    {
      const size_t lds2 = plan_plane_lds(grp, lds, dcn_fwd_plane_fixed_lds_bytes(parts));
      grp.wave_layout = dcn_plane_wave_layout();
      if (parts == 1)
        hipLaunchKernelGGL(dcn_fwd_plane<1>, dim3(G), dim3(threads), lds2, (hipStream_t)stream, grp, (float *)workspace);
      else
        hipLaunchKernelGGL(dcn_fwd_plane<2>, dim3(G), dim3(threads), lds2, (hipStream_t)stream, grp, (float *)workspace);
    }
    launch_plane_fixup(grp, workspace, G, stream);
    grp.n = 0;
    lds = 0;
    max_hw = 0;
    min_len = 1 << 30;
    table_used = 0;
    return KGDET_OK;
  };
  for (int i = 0; i < n; ++i) {
    const kgdet_dcn_shape *s = shapes[i];
    Derived d;
    if (int rc = derive(s, d)) return rc;
    KGDET_CHECK_SHAPE(inputs[i] && offsets[i] && packed_weights[i] && outputs[i], "null pointer (problem %d)", i);
    if (!mfma_ok(s)) {
      set_error("deformable_groups=%d with %d channels per group is not supported by the MFMA path",
                s->deformable_groups, s->C / s->deformable_groups);
      return KGDET_E_UNSUPPORTED;
    }
    const bool use_plane = plane_ok(s, d) && !(flags & KGDET_DCN_EXACT_FP32);
    for (int g = 0; g < s->groups; ++g) {
      DcnProblem p;
      fill_problem(s, d, g, p);
      p.x = inputs[i]; p.offset = offsets[i]; p.mask = masks ? masks[i] : nullptr;
      p.bias = biases ? biases[i] : nullptr; p.out = outputs[i];
      p.wpk = packed_weights[i] + (size_t)g * d.fwd_image_floats();
      p.flags = flags;
      p.n_ntiles = ceil_div(p.P, kTileN);
      p.n_mtiles = d.Og_pad / kTileM;
      p.chunks_per_tap = d.Cg_pad / kChunk;
      p.chunks_per_tile = d.K * p.chunks_per_tap;
      if (use_plane) {
        max_hw = s->H * s->W > max_hw ? s->H * s->W : max_hw;
        p.wq = packed_weights[i] + (size_t)s->groups * (d.fwd_image_floats() + d.bwd_image_floats()) +
               (size_t)g * d.plane_image_floats();
        p.tiles_per_image = ceil_div(p.HoWo, kTileN);
        p.n_ntiles = p.N * p.tiles_per_image;
      }
      p.total_units = (long long)p.n_ntiles * p.n_mtiles * p.chunks_per_tile;
      p.kparts = 1;
      if (use_plane) {
        // kparts > 1 would cut the reduction so that one part's weights fit a per-XCD L2 (units ordered problem, part,
        // tile, stage -- supported by dcn_unit_pos and the fix-up).  Measured on MI355X: no gain -- the weight stream
        // is not what bounds the kernel (an always-hot weight stage bought 6 %) -- while the extra partial tiles cost
        // 25 %; it stays 1.
        // a workgroup writes one slab per range its slice meets: keep that within kSlabSlots
        const int len = p.chunks_per_tile / p.kparts;
        const int new_min = len < min_len ? len : min_len;
        const long long units_after = (grp.n ? grp.unit_begin[grp.n] : 0) + p.total_units;
        if (grp.n > 0 && ceil_div((int)ceil_div(units_after, (long long)G), new_min) + 2 > kSlabSlots)
          if (int rc = flush()) return rc;
        min_len = len < min_len ? len : min_len;
        if (grp.n == 0) { grp.tile_begin[0] = 0; grp.range_begin[0] = 0; grp.unit_begin[0] = 0; }
        // tap records: shared with an earlier problem of the group that samples at the same positions
        p.taps = nullptr;
        p.build_taps = 0;
        for (int q = 0; q < grp.n && !p.taps; ++q) {
          const DcnProblem &o = grp.p[q];
          if (o.offset == p.offset && o.mask == p.mask && o.N == p.N && o.H == p.H && o.W == p.W && o.kh == p.kh &&
              o.kw == p.kw && o.sh == p.sh && o.sw == p.sw && o.ph == p.ph && o.pw == p.pw && o.dh == p.dh &&
              o.dw == p.dw && o.DG == p.DG)
            p.taps = o.taps;
        }
        if (!p.taps) {
          const size_t tb = tap_table_bytes(s, d);
          if (slab_bytes() + table_used + tb > workspace_bytes) {
            set_error("workspace too small for the tap records: need %zu bytes, got %zu (kgdet_dcn_group_workspace_bytes)",
                      slab_bytes() + table_used + tb, workspace_bytes);
            return KGDET_E_WORKSPACE;
          }
          p.taps = reinterpret_cast<const DcnTapRec *>(table_base + table_used);
          p.build_taps = 1;
          table_used += tb;
        }
        grp.p[grp.n] = p;
        grp.tile_begin[grp.n + 1] = grp.tile_begin[grp.n] + p.n_ntiles * p.n_mtiles;
        grp.range_begin[grp.n + 1] = grp.range_begin[grp.n] + p.n_ntiles * p.n_mtiles * p.kparts;
        grp.unit_begin[grp.n + 1] = grp.unit_begin[grp.n] + p.total_units;
        ++grp.n;
        const size_t need = dcn_fwd_plane_lds_bytes(parts, s->H * s->W);
        lds = need > lds ? need : lds;
        if (grp.n == kMaxFwdGroup)
          if (int rc = flush()) return rc;
      } else if (!(flags & KGDET_DCN_EXACT_FP32) && gather_ok(s, d) &&
                 slab_slots_ok((long long)s->N * ceil_div(d.Ho * d.Wo, kTileN) * (d.Og_pad / kTileM), d.K * (d.Cg_pad / kChunk))) {
        // map beyond the LDS plane: split operands, producers gather from a pixel-major copy of x (one launch per problem)
        if (int rc = flush()) return rc;
        const size_t tb = gather_table_bytes(s, d), ib = gather_image_bytes(s);
        if (slab_bytes() + tb + ib > workspace_bytes) {
          set_error("workspace too small for the large-map forward: need %zu bytes, got %zu (kgdet_dcn_workspace_bytes)",
                    slab_bytes() + tb + ib, workspace_bytes);
          return KGDET_E_WORKSPACE;
        }
        float *xT = reinterpret_cast<float *>(table_base + tb);
        const long long HW = (long long)s->H * s->W;
        if (g == 0)
          hipLaunchKernelGGL(dcn_to_pixel_major, dim3((unsigned)((HW + 31) / 32), (s->C + 31) / 32, s->N), dim3(256), 0,
                             (hipStream_t)stream, inputs[i], xT, s->C, HW, (long long)s->C * HW);
        p.x = xT;
        p.wq = packed_weights[i] + (size_t)s->groups * (d.fwd_image_floats() + d.bwd_image_floats()) +
               (size_t)g * d.plane_image_floats();
        p.tiles_per_image = ceil_div(p.HoWo, kTileN);
        p.n_ntiles = p.N * p.tiles_per_image;
        p.total_units = (long long)p.n_ntiles * p.n_mtiles * p.chunks_per_tile;
        p.taps = reinterpret_cast<const DcnTapRec *>(table_base);
        p.build_taps = g == 0;
        DcnFwdGroup gg;
        gg.n = 1; gg.xcd_slices = 1; gg.slots = kSlabSlots; gg.dbl_plane = 0; gg.plane_bytes = 0; gg.static_ranges = 0;
        gg.pair_mode = 0; gg.gather_mode = 1; gg.rounds = 1; gg.wave_layout = dcn_plane_wave_layout();
        gg.tile_begin[0] = 0; gg.range_begin[0] = 0; gg.unit_begin[0] = 0;
        gg.p[0] = p;
        gg.tile_begin[1] = p.n_ntiles * p.n_mtiles;
        gg.range_begin[1] = p.n_ntiles * p.n_mtiles;
        gg.unit_begin[1] = p.total_units;
        if (ceil_div((int)ceil_div(p.total_units, (long long)G), p.chunks_per_tile) + 2 > kSlabSlots) {
          set_error("large-map forward: more tiles per workgroup than slab slots");
          return KGDET_E_UNSUPPORTED;
        }
        plan_static_ranges(gg, G, false, kSlabSlots - 2);
        static thread_local bool gattr_set = false;
        if (!gattr_set) {
          KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_fwd_gather<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)kMaxLds));
          KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_fwd_gather<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)kMaxLds));
          gattr_set = true;
        }
        if (p.build_taps) hipLaunchKernelGGL(dcn_build_taps, dim3(2 * G, 1), dim3(256), 0, (hipStream_t)stream, gg);
        const size_t glds = dcn_fwd_plane_fixed_lds_bytes(parts);
        if (parts == 1)
          hipLaunchKernelGGL(dcn_fwd_gather<1>, dim3(G), dim3(dcn_fwd_plane_threads()), glds, (hipStream_t)stream, gg,
                             (float *)workspace);
        else
          hipLaunchKernelGGL(dcn_fwd_gather<2>, dim3(G), dim3(dcn_fwd_plane_threads()), glds, (hipStream_t)stream, gg,
                             (float *)workspace);
        launch_plane_fixup(gg, workspace, G, stream);
      } else {  // exact-fp32 kernel: one launch per problem (slabs are shared, so flush the pending group first)
        if (int rc = flush()) return rc;
        DcnFwdGroup one;
        one.n = 1; one.xcd_slices = 0; one.slots = 2; one.dbl_plane = 0; one.plane_bytes = 0; one.static_ranges = 0; one.pair_mode = 0; one.gather_mode = 0; one.rounds = 1; one.wave_layout = 0; one.range_begin[0] = 0;
        one.range_begin[1] = p.n_ntiles * p.n_mtiles; one.tile_begin[0] = 0; one.tile_begin[1] = p.n_ntiles * p.n_mtiles;
        one.unit_begin[0] = 0; one.unit_begin[1] = p.total_units;
        one.p[0] = p;
        hipLaunchKernelGGL(dcn_fwd_mfma, dim3(G), dim3(kThreads), 0, (hipStream_t)stream, p, (float *)workspace);
        hipLaunchKernelGGL(dcn_fwd_fixup, dim3(one.tile_begin[1], 16), dim3(kThreads), 0, (hipStream_t)stream, one,
                           (const float *)workspace, G);
      }
    }
  }
  if (int rc = flush()) return rc;
  KGDET_CHECK_LAUNCH("dcn_fwd");
  return KGDET_OK;
}

int kgdet_deform_conv_forward(const kgdet_dcn_shape *s, const float *input, const float *offset,
                              const float *mask, const float *packed_weight, const float *bias,
                              float *output, uint32_t flags, void *workspace, size_t workspace_bytes,
                              void *stream) {
  KGDET_CHECK_SHAPE(s != nullptr, "null shape");
  return kgdet_deform_conv_forward_grouped(1, &s, &input, &offset, &mask, &packed_weight, &bias, &output, flags,
                                           workspace, workspace_bytes, stream);
}

int kgdet_deform_conv_grad_input(const kgdet_dcn_shape *s, const float *offset, const float *mask,
                                 const float *packed_weight, const float *grad_output, float *grad_input,
                                 uint32_t flags, void *workspace, size_t workspace_bytes, void *stream) {
  Derived d;
  if (int rc = derive(s, d)) return rc;
  KGDET_CHECK_SHAPE(offset && packed_weight && grad_output && grad_input, "null pointer");
  if (!plane_bwd_input_ok(s, d)) {
    set_error("grad_input plane kernel: maps of at most %d pixels, at most %d (weight group, deformable group) channel runs, "
              "each starting on a 256-channel tile or ending inside one", kPlaneMaxHW, kMaxFwdGroup);
    return KGDET_E_UNSUPPORTED;
  }
  const InvTables it = inv_tables(s, d);
  const size_t it_stride = align_up(it.total(), 256);
  if (workspace == nullptr || workspace_bytes < slab_bytes() + inv_tables_all(s, d) + inv_sums_all(s, d)) {
    set_error("workspace too small: need %zu bytes, got %zu", slab_bytes() + inv_tables_all(s, d) + inv_sums_all(s, d), workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  unsigned char *base = (unsigned char *)workspace + slab_bytes();
  unsigned char *sums_base = base + inv_tables_all(s, d);              // [DG] Gov tables, then the pixel-major grad_output
  float *gout_t = (float *)(sums_base + (size_t)s->deformable_groups * inv_gov_bytes(s, d));
  const int gov_slots = dcn_inv_max_slots(s->H * s->W, d.Ho * d.Wo), gov_ld = inv_gov_ld(s, d);
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_build_inverse_taps, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds - 64));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_input_plane<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_input_plane<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    attr_set = true;
  }
  for (int dgi = 0; dgi < s->deformable_groups; ++dgi) {  // inverse sampling records from the forward geometry, per deformable group
    DcnProblem f;
    fill_problem(s, d, 0, f);
    f.offset = offset; f.mask = mask; f.dgi = dgi;
    unsigned char *tb = base + (size_t)dgi * it_stride;
    hipLaunchKernelGGL(dcn_build_inverse_taps, dim3(s->N * d.K), dim3(256),
                       dcn_build_inverse_taps_lds_bytes(s->H * s->W, d.Ho * d.Wo), (hipStream_t)stream, f, (uint4 *)tb,
                       (int *)(tb + it.rec_bytes), (DcnInvOvfCell *)(tb + it.rec_bytes + it.hdr_bytes),
                       (int2 *)(tb + it.rec_bytes + it.hdr_bytes + it.cell_bytes));
  }
  {   // sums of the cells with more than 8 contributions, for all output channels (dcn_backward_plane.hip)
    const int O_total_ = s->out_channels_total > 0 ? s->out_channels_total : s->O;
    DcnPixelMajorGroup pm;
    pm.n = 1;
    pm.e[0] = DcnPixelMajorItem{grad_output + (size_t)s->out_channel_offset * d.Ho * d.Wo, gout_t, s->N, s->O, d.Ho * d.Wo,
                                (long long)O_total_ * d.Ho * d.Wo};
    hipLaunchKernelGGL(dcn_gout_pixel_major_multi, dim3(ceil_div(d.Ho * d.Wo, 32), ceil_div(s->O, 32), s->N), dim3(256), 0,
                       (hipStream_t)stream, pm);
    DcnInvSumGroup sg;
    sg.n = 0;
    for (int dgi = 0; dgi < s->deformable_groups; ++dgi) {
      if (sg.n == kMaxFwdGroup) {
        sg.sched.on = 0; sg.hot_gemm = 0;
        hipLaunchKernelGGL(dcn_inv_medium_sums, dim3(s->N * d.K, kInvSumSplit, sg.n), dim3(256), 0, (hipStream_t)stream, sg);
        hipLaunchKernelGGL(dcn_inv_overflow_sums, dim3(s->N * d.K, kInvSumSplit, sg.n), dim3(256), 0, (hipStream_t)stream, sg);
        sg.n = 0;
      }
      unsigned char *tb = base + (size_t)dgi * it_stride;
      sg.e[sg.n++] = DcnInvSum{(const int *)(tb + it.rec_bytes), (const DcnInvOvfCell *)(tb + it.rec_bytes + it.hdr_bytes),
                               (const int2 *)(tb + it.rec_bytes + it.hdr_bytes + it.cell_bytes), gout_t,
                               (float *)(sums_base + (size_t)dgi * inv_gov_bytes(s, d)), s->N * d.K, d.K, d.Ho * d.Wo, s->O,
                               d.Og, d.Og_pad16, gov_ld, gov_slots, s->W};
    }
    sg.sched.on = 0; sg.hot_gemm = 0;
    hipLaunchKernelGGL(dcn_inv_medium_sums, dim3(s->N * d.K, kInvSumSplit, sg.n), dim3(256), 0, (hipStream_t)stream, sg);
    hipLaunchKernelGGL(dcn_inv_overflow_sums, dim3(s->N * d.K, kInvSumSplit, sg.n), dim3(256), 0, (hipStream_t)stream, sg);
  }
  const int G = grid_size();
  const int parts = (flags & KGDET_DCN_BF16) ? 1 : 2;
  DcnFwdGroup grp;
  grp.n = 0; grp.xcd_slices = 1; grp.slots = kSlabSlots; grp.dbl_plane = 0; grp.plane_bytes = 0; grp.static_ranges = 0; grp.pair_mode = 0; grp.gather_mode = 0; grp.rounds = 1; grp.wave_layout = 0;
  grp.tile_begin[0] = 0; grp.range_begin[0] = 0; grp.unit_begin[0] = 0;
  const int O_total = s->out_channels_total > 0 ? s->out_channels_total : s->O;
  ChannelRun runs[kMaxFwdGroup];
  const int n_runs = channel_runs(s, d, runs, kMaxFwdGroup);
  for (int r = 0; r < n_runs; ++r) {
    const int g = runs[r].g, len = runs[r].c1 - runs[r].c0, row = runs[r].c0 - g * d.Cg;
    // the transposed problem: "input" = grad_output window of the run's weight group, "output" = the run's grad_input channels
    DcnProblem p{};
    p.x = grad_output; p.out = grad_input; p.bias = nullptr; p.offset = nullptr; p.mask = nullptr;
    p.N = s->N;
    p.C_total = O_total; p.c_base = s->out_channel_offset + g * d.Og; p.Cg = d.Og; p.Cg_pad = d.Og_pad16;
    p.O_total = s->C; p.o_base = runs[r].c0; p.Og = len; p.Og_pad = ceil_div(len, kTileM) * kTileM; p.bias_base = 0;
    p.mt_base = row / kTileM; p.row0 = row % kTileM;
    p.H = d.Ho; p.W = d.Wo;                       // plane geometry = grad_output
    p.Ho = s->H; p.Wo = s->W; p.HoWo = s->H * s->W; p.P = s->N * p.HoWo;  // "pixels" = input cells
    p.kh = s->kh; p.kw = s->kw; p.K = d.K; p.seg_stages = d.K;
    p.DG = 1; p.cpdg = p.C_total;
    p.tiles_per_image = ceil_div(p.HoWo, kTileN);
    p.n_ntiles = p.N * p.tiles_per_image;
    p.n_mtiles = p.Og_pad / kTileM;
    p.chunks_per_tap = d.Og_pad16 / kChunk;
    p.chunks_per_tile = d.K * p.chunks_per_tap;
    p.total_units = (long long)p.n_ntiles * p.n_mtiles * p.chunks_per_tile;
    p.kparts = 1;
    p.flags = 0;
    p.wq = packed_weight + (size_t)s->groups * (d.fwd_image_floats() + d.bwd_image_floats() + d.plane_image_floats()) +
           (size_t)g * d.plane_t_image_floats();
    unsigned char *tb = base + (size_t)runs[r].dgi * it_stride;
    p.taps = reinterpret_cast<const DcnTapRec *>(tb);
    p.inv_gov = (const float *)(sums_base + (size_t)runs[r].dgi * inv_gov_bytes(s, d));
    p.gov_slots = gov_slots; p.gov_ld = gov_ld; p.gov_c0 = g * d.Og_pad16;
    p.build_taps = 0;
    grp.p[grp.n] = p;
    grp.tile_begin[grp.n + 1] = grp.tile_begin[grp.n] + p.n_ntiles * p.n_mtiles;
    grp.range_begin[grp.n + 1] = grp.range_begin[grp.n] + p.n_ntiles * p.n_mtiles;
    grp.unit_begin[grp.n + 1] = grp.unit_begin[grp.n] + p.total_units;
    ++grp.n;
  }
  const size_t lds = plan_plane_lds(grp, dcn_bwd_input_plane_lds_bytes(parts, d.Ho * d.Wo),
                                    dcn_bwd_input_plane_fixed_lds_bytes(parts));
  const int threads = dcn_fwd_plane_threads();
  grp.wave_layout = dcn_plane_wave_layout();   // plane_role's accumulator layout (slabs are decoded by dcn_fwd_fixup)
  const int Gs = small_launch_grid(grp, G);
  plan_static_ranges(grp, Gs, false, kSlabSlots - 2);
  if (parts == 1)
    hipLaunchKernelGGL(dcn_bwd_input_plane<1>, dim3(Gs), dim3(threads), lds, (hipStream_t)stream, grp, (float *)workspace);
  else
    hipLaunchKernelGGL(dcn_bwd_input_plane<2>, dim3(Gs), dim3(threads), lds, (hipStream_t)stream, grp, (float *)workspace);
  launch_plane_fixup(grp, workspace, Gs, stream);
  KGDET_CHECK_LAUNCH("dcn_bwd_input_plane");
  return KGDET_OK;
}

static int grad_offset_plane(const kgdet_dcn_shape *s, const float *input, const float *offset, const float *mask,
                             const float *packed_weight, const float *grad_output, float *grad_offset, float *grad_mask,
                             uint32_t flags, void *workspace, size_t workspace_bytes, void *stream);

int kgdet_deform_conv_grad_offset(const kgdet_dcn_shape *s, const float *input, const float *offset,
                                  const float *packed_weight, const float *grad_output, float *grad_offset,
                                  uint32_t flags, void *workspace, size_t workspace_bytes, void *stream) {
  return grad_offset_plane(s, input, offset, nullptr, packed_weight, grad_output, grad_offset, nullptr, flags, workspace,
                           workspace_bytes, stream);
}

// v1 (mask == nullptr) or v2 (mask, grad_mask; split operands and static ranges only: KGDET_E_UNSUPPORTED otherwise, before
// anything is launched)
static int grad_offset_plane(const kgdet_dcn_shape *s, const float *input, const float *offset, const float *mask,
                             const float *packed_weight, const float *grad_output, float *grad_offset, float *grad_mask,
                             uint32_t flags, void *workspace, size_t workspace_bytes, void *stream) {
  Derived d;
  if (int rc = derive(s, d)) return rc;
  KGDET_CHECK_SHAPE(input && offset && packed_weight && grad_output && grad_offset, "null pointer");
  KGDET_CHECK_SHAPE((mask == nullptr) == (grad_mask == nullptr), "mask and grad_mask must come together");
  if (!plane_bwd_offset_ok(s, d, mask != nullptr) || (mask && (flags & KGDET_DCN_BF16))) {
    set_error("grad_offset plane kernel: needs every deformable group inside one weight group (whole 16-channel chunks), "
              "<= 256 output channels per group, H*W <= %d", kPlaneMaxHW);
    return KGDET_E_UNSUPPORTED;
  }
  const size_t need = slab_bytes() + grad_tap_bytes(s, d);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_offset_plane<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_offset_plane<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_offset_plane_masked, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    attr_set = true;
  }
  const int G = grid_size();
  const int parts = (flags & KGDET_DCN_BF16) ? 1 : 2;
  DcnFwdGroup grp;
  grp.n = 0; grp.xcd_slices = 1; grp.slots = kSlabSlots; grp.dbl_plane = 0; grp.plane_bytes = 0; grp.static_ranges = 0; grp.pair_mode = 0; grp.gather_mode = 0; grp.rounds = 1; grp.wave_layout = 0;
  grp.tile_begin[0] = 0; grp.range_begin[0] = 0; grp.unit_begin[0] = 0;
  // one sub-problem per deformable group (= channel run: plane_bwd_offset_ok), each with its own record table
  const unsigned char *recs = (const unsigned char *)workspace + slab_bytes();
  const size_t rec_stride = (size_t)s->N * d.K * d.Ho * d.Wo * 64;
  ChannelRun runs[kMaxFwdGroup];
  const int n_runs = channel_runs(s, d, runs, kMaxFwdGroup);
  for (int r = 0; r < n_runs; ++r) {
    const int g = runs[r].g;
    DcnProblem p;
    fill_problem(s, d, g, p);
    p.x = input; p.offset = offset; p.mask = mask; p.gout = grad_output; p.goff = grad_offset; p.gmask = grad_mask;
    p.c_base = runs[r].c0; p.Cg = runs[r].c1 - runs[r].c0; p.Cg_pad = ceil_div(p.Cg, kChunk) * kChunk;
    p.dgi = runs[r].dgi; p.c16_base = (runs[r].c0 - g * d.Cg) / kChunk;
    p.wq = packed_weight + (size_t)s->groups * (d.fwd_image_floats() + d.bwd_image_floats() + d.plane_image_floats()) +
           (size_t)g * d.plane_t_image_floats();
    p.taps = reinterpret_cast<const DcnTapRec *>(recs + (size_t)runs[r].dgi * rec_stride); p.build_taps = 1;
    p.tiles_per_image = ceil_div(p.HoWo, kTileN);
    p.n_ntiles = p.N * p.tiles_per_image;
    p.n_mtiles = 1;
    p.chunks_per_tap = p.Cg_pad / kChunk;
    p.chunks_per_tile = d.K * p.chunks_per_tap;
    p.total_units = (long long)p.n_ntiles * p.chunks_per_tile;
    p.kparts = 1;
    p.flags = flags;
    grp.p[grp.n] = p;
    grp.tile_begin[grp.n + 1] = grp.tile_begin[grp.n] + p.n_ntiles;
    grp.range_begin[grp.n + 1] = grp.range_begin[grp.n] + p.n_ntiles;
    grp.unit_begin[grp.n + 1] = grp.unit_begin[grp.n] + p.total_units;
    ++grp.n;
  }
  const int Gs = mask ? G : small_launch_grid(grp, G);      // (v2 needs one workgroup per range: the full grid)
  plan_static_ranges(grp, Gs, mask != nullptr);
  if (mask && !grp.static_ranges) {
    set_error("grad_offset plane kernel (v2): more (part, tile) ranges than workgroups");
    return KGDET_E_UNSUPPORTED;
  }
  const bool use_pair = parts == 2 && !mask && offset_pair_ok(grp);
  if (use_pair)
    place_offset_xblk(grp, (unsigned char *)workspace + slab_bytes(), grad_tap_bytes(s, d), workspace_bytes - slab_bytes());
  hipLaunchKernelGGL(dcn_build_grad_taps, dim3(2 * G, grp.n), dim3(256), 0, (hipStream_t)stream, grp);
  const size_t lds = dcn_bwd_offset_plane_lds_bytes(parts, d.K, s->H * s->W, mask != nullptr);
  const int threads = dcn_bwd_offset_plane_threads();
  if (mask) {
    hipLaunchKernelGGL(dcn_bwd_offset_plane_masked, dim3(G), dim3(threads), lds, (hipStream_t)stream, grp, (float *)workspace,
                       d.K);
    hipLaunchKernelGGL(dcn_bwd_offset_plane_fixup_masked, dim3(grp.tile_begin[grp.n], 8), dim3(256), 0, (hipStream_t)stream,
                       grp, (const float *)workspace, G, d.K);
    KGDET_CHECK_LAUNCH("dcn_bwd_offset_plane_masked");
    return KGDET_OK;
  }
  if (parts == 1)
    hipLaunchKernelGGL(dcn_bwd_offset_plane<1>, dim3(Gs), dim3(threads), lds, (hipStream_t)stream, grp,
                       (float *)workspace, d.K);
  else if (use_pair) {
    if (int rc = launch_offset_pair(grp, Gs, workspace, d.K, stream)) return rc;
  } else
    hipLaunchKernelGGL(dcn_bwd_offset_plane<2>, dim3(Gs), dim3(threads), lds, (hipStream_t)stream, grp,
                       (float *)workspace, d.K);
  hipLaunchKernelGGL(dcn_bwd_offset_plane_fixup, dim3(grp.tile_begin[grp.n], 8), dim3(256), 0, (hipStream_t)stream, grp,
                     (const float *)workspace, Gs, d.K);
  KGDET_CHECK_LAUNCH("dcn_bwd_offset_plane");
  return KGDET_OK;
}

// test hook: with KGDET_DCN_HOT_DEBUG=1 in the environment, the grouped backward copies (synchronously, before its second phase
// reuses the tables) how many hot cells its builder handed to dcn_hot_gemm, per (distinct offset tensor, image) in launch order; this
// returns the last call's counts (raw: values above kHotMaxCols mean the rest went the cluster path).
namespace { struct HotDbg { const int *ptr[kMaxFwdGroup]; int N[kMaxFwdGroup]; int n; int vals[64]; int n_vals; }; HotDbg g_hot_dbg = {}; }
int kgdet_debug_dcn_hot_columns(int *out, int max_out) {
  int w = 0;
  for (; w < g_hot_dbg.n_vals && w < max_out; ++w) out[w] = g_hot_dbg.vals[w];
  return w;
}

// n v1 problems: grad_input of all of them in one dcn_bwd_input_plane launch, grad_offset of all of them in one
// dcn_bwd_offset_plane launch (inverse / gradient records are built once per distinct offset tensor).
int kgdet_deform_conv_backward_input_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                             const float *const *offsets, const float *const *packed_weights,
                                             const float *const *grad_outputs, float *const *grad_inputs,
                                             float *const *grad_offsets, void *workspace, size_t workspace_bytes,
                                             void *stream) {
  KGDET_CHECK_SHAPE(n >= 1 && n <= kMaxFwdGroup && shapes && inputs && offsets && packed_weights && grad_outputs &&
                        grad_inputs && grad_offsets, "null pointer / group size not in [1, %d]", kMaxFwdGroup);
  const int G = grid_size();
  Derived dd[kMaxFwdGroup];
  int same_as[kMaxFwdGroup];  // earlier problem with the same offsets and geometry (shares its records)
  size_t inv_off[kMaxFwdGroup], rec_off[kMaxFwdGroup], sum_off[kMaxFwdGroup], inv_total = 0, rec_total = 0;
  int max_K = 0;
  for (int i = 0; i < n; ++i) {
    const kgdet_dcn_shape *s = shapes[i];
    if (int rc = derive(s, dd[i])) return rc;
    KGDET_CHECK_SHAPE(inputs[i] && offsets[i] && packed_weights[i] && grad_outputs[i] && grad_inputs[i] &&
                          grad_offsets[i], "null pointer (problem %d)", i);
    if (s->groups != 1 || s->deformable_groups != 1 || !plane_bwd_input_ok(s, dd[i]) || !plane_bwd_offset_ok(s, dd[i])) {
      set_error("problem %d is not eligible for the plane backward kernels", i);
      return KGDET_E_UNSUPPORTED;
    }
    same_as[i] = -1;
    for (int q = 0; q < i && same_as[i] < 0; ++q) {
      const kgdet_dcn_shape *o = shapes[q];
      if (offsets[q] == offsets[i] && o->N == s->N && o->H == s->H && o->W == s->W && o->kh == s->kh && o->kw == s->kw &&
          o->stride_h == s->stride_h && o->stride_w == s->stride_w && o->pad_h == s->pad_h && o->pad_w == s->pad_w &&
          o->dil_h == s->dil_h && o->dil_w == s->dil_w)
        same_as[i] = same_as[q] >= 0 ? same_as[q] : q;
    }
    if (same_as[i] < 0) {
      inv_off[i] = inv_total; inv_total += align_up(inv_tables(s, dd[i]).total(), 256);
      rec_off[i] = rec_total; rec_total += align_up(grad_tap_bytes(s, dd[i]), 256);
    } else {
      inv_off[i] = inv_off[same_as[i]]; rec_off[i] = rec_off[same_as[i]];
    }
    max_K = dd[i].K > max_K ? dd[i].K : max_K;
  }
  for (int i = 0; i < n; ++i) {   // per problem (functions of its grad_output): the long cells' sums + the pixel-major grad_output
    sum_off[i] = inv_total; inv_total += inv_sums_all(shapes[i], dd[i]);
  }
  const size_t tables = inv_total > rec_total ? inv_total : rec_total;  // the two phases run one after the other
  if (workspace == nullptr || workspace_bytes < slab_bytes() + tables) {
    set_error("workspace too small: need %zu bytes, got %zu (kgdet_dcn_group_workspace_bytes)", slab_bytes() + tables,
              workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_build_inverse_taps, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds - 64));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_input_plane<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_offset_plane<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    attr_set = true;
  }
  unsigned char *tab = (unsigned char *)workspace + slab_bytes();
  auto check_slots = [&](const DcnFwdGroup &grp) -> bool {  // a slice must not meet more ranges than it has slab slots
    int min_len = 1 << 30;
    for (int i = 0; i < grp.n; ++i) min_len = grp.p[i].chunks_per_tile < min_len ? grp.p[i].chunks_per_tile : min_len;
    return ceil_div((int)ceil_div((int)grp.unit_begin[grp.n], G), min_len) + 2 <= kSlabSlots;
  };

  // ---- phase 1: grad_input (transposed sampling) ----
  DcnFwdGroup grp;
  grp.n = 0; grp.xcd_slices = 1; grp.slots = kSlabSlots; grp.dbl_plane = 0; grp.plane_bytes = 0; grp.static_ranges = 0; grp.pair_mode = 0; grp.gather_mode = 0; grp.rounds = 1; grp.wave_layout = 0;
  grp.tile_begin[0] = 0; grp.range_begin[0] = 0; grp.unit_begin[0] = 0;
  size_t lds = 0;
  DcnInvBuildGroup builds;
  builds.n = 0;
  int build_blocks = 0;
  size_t build_lds = 0;
  DcnPixelMajorGroup pmg;      // grad_output windows -> pixel-major copies, one launch
  pmg.n = 0;
  int pm_px = 0, pm_c = 0, pm_images = 0;
  DcnInvSumGroup sums;         // the long cells' sums of every problem, one launch
  sums.n = 0;
  // (problem, image) groups onto XCDs, largest first onto the least loaded one (DcnInvSumSched): the sums kernel and the hot-cell GEMM
  // both read a group's 1 MB pixel-major grad_output from one XCD's L2
  int sched_longest = 0;
  {
    static const bool xcd_off = getenv("KGDET_DCN_SUMS_XCD") && atoi(getenv("KGDET_DCN_SUMS_XCD")) == 0;   // A/B switch
    DcnInvSumSched &sc = sums.sched;
    sc.on = 0;
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = 0; x < 8; ++x) sc.n_seg[x] = 0;
    // (KGDET_DCN_SUMS_HALVES=1, measured and left off: every pair as TWO segments -- its first and second half of the taps; six
    // problems x two images are 12 pairs for 8 XCDs.  dcn_inv_medium_sums 43.6 against 44.1 us, dcn_hot_gemm 36 against 21 us)
    struct G_ { int z, b, units, u0; } gs[kMaxFwdGroup * 128];
    int ng = 0;
    bool fits = !xcd_off;
    for (int z = 0; z < n && fits; ++z)
      for (int b = 0; b < shapes[z]->N && fits; ++b) {
        if (ng + 2 > kMaxFwdGroup * 128) { fits = false; break; }
        static const bool no_halves = !(getenv("KGDET_DCN_SUMS_HALVES") && atoi(getenv("KGDET_DCN_SUMS_HALVES")) == 1);   // (experiment, off)
        const int units = dd[z].K * kInvSumSplit, h0 = no_halves ? units : (dd[z].K + 1) / 2 * kInvSumSplit;
        gs[ng++] = G_{z, b, h0, 0};
        if (units > h0) gs[ng++] = G_{z, b, units - h0, h0};
      }
    std::stable_sort(gs, gs + ng, [](const G_ &a, const G_ &b) { return a.units > b.units; });
    for (int i = 0; i < ng && fits; ++i) {
      int x = 0;
      for (int y = 1; y < 8; ++y) if (load[y] < load[x]) x = y;
      if (sc.n_seg[x] >= kInvSumSegs) { fits = false; break; }
      sc.seg[x][sc.n_seg[x]++] = DcnInvSumSeg{gs[i].z, gs[i].b, load[x], gs[i].units, gs[i].u0};
      load[x] += gs[i].units;
    }
    if (fits) {
      sc.on = 1;
      for (int x = 0; x < 8; ++x) sched_longest = load[x] > sched_longest ? load[x] : sched_longest;
    }
  }
  static const bool hot_off = getenv("KGDET_DCN_HOT_GEMM") && atoi(getenv("KGDET_DCN_HOT_GEMM")) == 0;   // A/B switch
  bool hot_ok = !hot_off;       // hot cells (> 64 contributions) as one GEMM per (problem, image): dcn_hot_gemm
  g_hot_dbg.n = 0; g_hot_dbg.n_vals = 0;
  // (test switch: a shorter column list, to reach the overflow rule with small kernels)
  const int hot_max = getenv("KGDET_DCN_HOT_MAX_COLS") ? std::max(32, std::min(kHotMaxCols, atoi(getenv("KGDET_DCN_HOT_MAX_COLS")))) : kHotMaxCols;
  hot_ok = hot_ok && sums.sched.on;   // (its workgroups follow the same XCD schedule)
  for (int i = 0; i < n; ++i) hot_ok = hot_ok && inv_gov_ld(shapes[i], dd[i]) == shapes[i]->O;
  DcnHotGemmGroup hotg;
  hotg.n = 0;
  int sums_blocks = 0;
  for (int i = 0; i < n; ++i) {
    const kgdet_dcn_shape *s = shapes[i];
    const Derived &d = dd[i];
    const InvTables it = inv_tables(s, d);
    uint4 *inv = (uint4 *)(tab + inv_off[i]);
    int *hdr = (int *)(tab + inv_off[i] + it.rec_bytes);
    DcnInvOvfCell *cells = (DcnInvOvfCell *)(tab + inv_off[i] + it.rec_bytes + it.hdr_bytes);
    int2 *spill = (int2 *)(tab + inv_off[i] + it.rec_bytes + it.hdr_bytes + it.cell_bytes);
    float *gov = (float *)(tab + sum_off[i]);
    float *gout_t = (float *)(tab + sum_off[i] + inv_gov_bytes(s, d));
    {
      const int O_total_ = s->out_channels_total > 0 ? s->out_channels_total : s->O;
      pmg.e[pmg.n++] = DcnPixelMajorItem{grad_outputs[i] + (size_t)s->out_channel_offset * d.Ho * d.Wo, gout_t, s->N, s->O,
                                         d.Ho * d.Wo, (long long)O_total_ * d.Ho * d.Wo};
      pm_px = d.Ho * d.Wo > pm_px ? d.Ho * d.Wo : pm_px; pm_c = s->O > pm_c ? s->O : pm_c; pm_images += s->N;
      sums.e[sums.n++] = DcnInvSum{hdr, cells, spill, gout_t, gov, s->N * d.K, d.K, d.Ho * d.Wo, s->O, d.Og, d.Og_pad16,
                                   inv_gov_ld(s, d), dcn_inv_max_slots(s->H * s->W, d.Ho * d.Wo), s->W, nullptr, 0};
      sums_blocks = s->N * d.K > sums_blocks ? s->N * d.K : sums_blocks;
      const int src = same_as[i] < 0 ? i : same_as[i];    // (the column list lives with the offset tensor's records)
      const InvTables its = inv_tables(shapes[src], dd[src]);
      unsigned char *hot = tab + inv_off[src] + its.rec_bytes + its.hdr_bytes + its.cell_bytes + its.spill_bytes;
      hotg.e[hotg.n++] = DcnHotGemm{(const int4 *)hot, (const int *)(hot + its.hot_cols_bytes), spill, gout_t, gov, s->N, d.K,
                                    d.Ho * d.Wo, s->O, inv_gov_ld(s, d), dcn_inv_max_slots(s->H * s->W, d.Ho * d.Wo), hot_max};
      if (hot_ok) { sums.e[sums.n - 1].hot_count = (const int *)(hot + its.hot_cols_bytes); sums.e[sums.n - 1].hot_max = hot_max; }
    }
    if (same_as[i] < 0) {   // (all distinct offset tensors of the group: one builder launch below)
      DcnInvBuild &e = builds.e[builds.n++];
      fill_problem(s, d, 0, e.p);
      e.p.offset = offsets[i]; e.p.mask = nullptr;
      e.inv = inv; e.hdr = hdr; e.cells = cells; e.spill = spill;
      e.hot_cols = hot_ok ? (int4 *)(tab + inv_off[i] + it.rec_bytes + it.hdr_bytes + it.cell_bytes + it.spill_bytes) : nullptr;
      e.hot_count = hot_ok ? (int *)(tab + inv_off[i] + it.rec_bytes + it.hdr_bytes + it.cell_bytes + it.spill_bytes + it.hot_cols_bytes) : nullptr;
      e.hot_max = hot_max;
      if (hot_ok) {
        KGDET_HIP_TRY(hipMemsetAsync(e.hot_count, 0, (size_t)s->N * sizeof(int), (hipStream_t)stream));
        if (g_hot_dbg.n < kMaxFwdGroup) { g_hot_dbg.ptr[g_hot_dbg.n] = e.hot_count; g_hot_dbg.N[g_hot_dbg.n++] = s->N; }
      }
      const int blocks = s->N * d.K;
      build_blocks = blocks > build_blocks ? blocks : build_blocks;
      const size_t need = dcn_build_inverse_taps_lds_bytes(s->H * s->W, d.Ho * d.Wo);
      build_lds = need > build_lds ? need : build_lds;
    }
    DcnProblem p{};
    p.x = grad_outputs[i]; p.out = grad_inputs[i];
    p.N = s->N;
    p.C_total = s->out_channels_total > 0 ? s->out_channels_total : s->O;
    p.c_base = s->out_channel_offset; p.Cg = d.Og; p.Cg_pad = d.Og_pad16;
    p.O_total = s->C; p.o_base = 0; p.Og = d.Cg; p.Og_pad = d.Cg_pad256;
    p.H = d.Ho; p.W = d.Wo;
    p.Ho = s->H; p.Wo = s->W; p.HoWo = s->H * s->W; p.P = s->N * p.HoWo;
    p.kh = s->kh; p.kw = s->kw; p.K = d.K; p.seg_stages = d.K;
    p.DG = 1; p.cpdg = p.C_total;
    p.tiles_per_image = ceil_div(p.HoWo, kTileN);
    p.n_ntiles = p.N * p.tiles_per_image;
    p.n_mtiles = d.Cg_pad256 / kTileM;
    p.chunks_per_tap = d.Og_pad16 / kChunk;
    p.chunks_per_tile = d.K * p.chunks_per_tap;
    p.total_units = (long long)p.n_ntiles * p.n_mtiles * p.chunks_per_tile;
    p.kparts = 1;
    p.wq = packed_weights[i] + (d.fwd_image_floats() + d.bwd_image_floats() + d.plane_image_floats());
    p.taps = reinterpret_cast<const DcnTapRec *>(inv);
    p.inv_gov = gov; p.gov_slots = dcn_inv_max_slots(s->H * s->W, d.Ho * d.Wo); p.gov_ld = inv_gov_ld(s, d); p.gov_c0 = 0;
    grp.p[grp.n] = p;
    grp.tile_begin[grp.n + 1] = grp.tile_begin[grp.n] + p.n_ntiles * p.n_mtiles;
    grp.range_begin[grp.n + 1] = grp.range_begin[grp.n] + p.n_ntiles * p.n_mtiles;
    grp.unit_begin[grp.n + 1] = grp.unit_begin[grp.n] + p.total_units;
    ++grp.n;
    const size_t need = dcn_bwd_input_plane_lds_bytes(2, d.Ho * d.Wo);
    lds = need > lds ? need : lds;
  }
  if (!check_slots(grp)) { set_error("group too uneven for the slab slots"); return KGDET_E_UNSUPPORTED; }
  // problems that share a grad_input pointer are summed into it by the fix-up (sum groups); planned before anything is launched
  if (!assign_sum_groups(grp, (const void *const *)grad_inputs, false)) {
    set_error("aliased grad_input pointers: the problems do not tile alike (or more than four share one)");
    return KGDET_E_UNSUPPORTED;
  }
  lds = plan_plane_lds(grp, lds, dcn_bwd_input_plane_fixed_lds_bytes(2));
  grp.wave_layout = dcn_plane_wave_layout();   // plane_role's accumulator layout (slabs are decoded by dcn_fwd_fixup)
  const int Gs = small_launch_grid(grp, G);
  plan_static_ranges(grp, Gs, false, kSlabSlots - 2);
  if (has_sum_groups(grp) && !grp.static_ranges) {
    set_error("aliased grad_input pointers need the static schedule (one range per workgroup)");
    return KGDET_E_UNSUPPORTED;
  }
  const int only_phase = g_options[KGDET_OPT_BWD_PHASE];   // (measurement switch: 1 = grad_input only, 2 = grad_offset only)
  if (only_phase != 2) {
  {
    static const bool fused_off = getenv("KGDET_DCN_BWD_PREPARE") && atoi(getenv("KGDET_DCN_BWD_PREPARE")) == 0;   // A/B switch
    static thread_local bool multi_attr_set = false;
    if (!multi_attr_set) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_build_inverse_taps_multi, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kMaxLds - 64));
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_input_prepare, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kMaxLds - 64 - 2 * 32 * 33 * 4 - 64));
      multi_attr_set = true;
    }
    const int pm_bx = ceil_div(pm_px, 32), pm_by = ceil_div(pm_c, 32);
    if (!fused_off && build_lds + 2 * 32 * 33 * 4 + 128 <= kMaxLds - 64) {
      const int pm_blocks = ceil_div(pm_bx * pm_by * pm_images, 2);
      hipLaunchKernelGGL(dcn_bwd_input_prepare, dim3(build_blocks * builds.n + pm_blocks), dim3(512), build_lds, (hipStream_t)stream,
                         builds, pmg, build_blocks, pm_bx, pm_by);
    } else {
      if (builds.n > 0)
        hipLaunchKernelGGL(dcn_build_inverse_taps_multi, dim3(build_blocks, builds.n), dim3(256), build_lds, (hipStream_t)stream,
                           builds);
      hipLaunchKernelGGL(dcn_gout_pixel_major_multi, dim3(pm_bx, pm_by, pm_images), dim3(256), 0, (hipStream_t)stream, pmg);
    }
  }
  sums.hot_gemm = hot_ok ? 1 : 0;
  if (hot_ok) {
    static thread_local bool hot_attr = false;
    if (!hot_attr) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_hot_gemm, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)dcn_hot_gemm_lds_bytes()));
      hot_attr = true;
    }
    hipLaunchKernelGGL(dcn_hot_gemm, dim3(8 * kHotBlocksPerXcd), dim3(512), dcn_hot_gemm_lds_bytes(), (hipStream_t)stream, hotg,
                       sums.sched);
  }
  // the cluster kernel serves the cells above 64 contributions that dcn_hot_gemm does not take: none, if the GEMM is on and no image
  // can overflow its column list (a tap has at most 4 HoWo / 65 such cells)
  bool clusters_needed = !hot_ok;
  for (int i = 0; i < n && !clusters_needed; ++i)
    clusters_needed = (long long)dd[i].K * (4 * dd[i].Ho * dd[i].Wo / 65) > hot_max;
  if (sums.sched.on) {
    hipLaunchKernelGGL(dcn_inv_medium_sums, dim3(8 * sched_longest), dim3(256), 0, (hipStream_t)stream, sums);
    if (clusters_needed)
      hipLaunchKernelGGL(dcn_inv_overflow_sums, dim3(8 * sched_longest), dim3(256), 0, (hipStream_t)stream, sums);
  } else {
    hipLaunchKernelGGL(dcn_inv_medium_sums, dim3(sums_blocks, kInvSumSplit, sums.n), dim3(256), 0, (hipStream_t)stream, sums);
    hipLaunchKernelGGL(dcn_inv_overflow_sums, dim3(sums_blocks, kInvSumSplit, sums.n), dim3(256), 0, (hipStream_t)stream, sums);
  }
  if (g_hot_dbg.n > 0 && getenv("KGDET_DCN_HOT_DEBUG") && atoi(getenv("KGDET_DCN_HOT_DEBUG")) == 1) {   // (test hook, see above)
    KGDET_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < g_hot_dbg.n; ++i)
      for (int b = 0; b < g_hot_dbg.N[i] && g_hot_dbg.n_vals < 64; ++b)
        KGDET_HIP_TRY(hipMemcpy(&g_hot_dbg.vals[g_hot_dbg.n_vals++], g_hot_dbg.ptr[i] + b, sizeof(int), hipMemcpyDeviceToHost));
  }
  hipLaunchKernelGGL(dcn_bwd_input_plane<2>, dim3(Gs), dim3(dcn_fwd_plane_threads()), lds, (hipStream_t)stream, grp,
                     (float *)workspace);
  launch_plane_fixup(grp, workspace, Gs, stream);
  }
  if (only_phase == 1) {   // (measurement switch: grad_offset was NOT written -- never reported as success)
    KGDET_CHECK_LAUNCH("dcn_bwd_plane_grouped");
    set_error("KGDET_OPT_BWD_PHASE = 1: grad_offset not computed");
    return KGDET_E_PARTIAL;
  }

  // ---- phase 2: grad_offset (column gradient in registers) ----
  grp.n = 0;
  grp.wave_layout = 0;
  grp.static_ranges = 0;
  lds = 0;
  for (int i = 0; i < n; ++i) {
    const kgdet_dcn_shape *s = shapes[i];
    const Derived &d = dd[i];
    DcnProblem p;
    fill_problem(s, d, 0, p);
    p.x = inputs[i]; p.offset = offsets[i]; p.mask = nullptr; p.gout = grad_outputs[i]; p.goff = grad_offsets[i];
    p.wq = packed_weights[i] + (d.fwd_image_floats() + d.bwd_image_floats() + d.plane_image_floats());
    p.taps = reinterpret_cast<const DcnTapRec *>(tab + rec_off[i]);
    p.build_taps = same_as[i] < 0;
    p.tiles_per_image = ceil_div(p.HoWo, kTileN);
    p.n_ntiles = p.N * p.tiles_per_image;
    p.n_mtiles = 1;
    p.chunks_per_tap = d.Cg_pad / kChunk;
    p.chunks_per_tile = d.K * p.chunks_per_tap;
    p.total_units = (long long)p.n_ntiles * p.chunks_per_tile;
    p.kparts = 1;
    p.flags = 0;
    grp.p[grp.n] = p;
    grp.tile_begin[grp.n + 1] = grp.tile_begin[grp.n] + p.n_ntiles;
    grp.range_begin[grp.n + 1] = grp.range_begin[grp.n] + p.n_ntiles;
    grp.unit_begin[grp.n + 1] = grp.unit_begin[grp.n] + p.total_units;
    ++grp.n;
    const size_t need = dcn_bwd_offset_plane_lds_bytes(2, max_K, s->H * s->W);
    lds = need > lds ? need : lds;
  }
  if (lds > kMaxLds || !check_slots(grp)) { set_error("group does not fit the grad_offset kernel"); return KGDET_E_UNSUPPORTED; }
  if (!assign_sum_groups(grp, (const void *const *)grad_offsets, true)) {
    set_error("aliased grad_offset pointers: the problems do not tile alike (or more than four share one)");
    return KGDET_E_UNSUPPORTED;
  }
  const int Go = small_launch_grid(grp, G);
  plan_static_ranges(grp, Go);
  if (has_sum_groups(grp) && !grp.static_ranges) {
    set_error("aliased grad_offset pointers need the static schedule (one range per workgroup)");
    return KGDET_E_UNSUPPORTED;
  }
  const bool use_pair = offset_pair_ok(grp);
  if (use_pair) place_offset_xblk(grp, tab, rec_total, workspace_bytes - slab_bytes());
  hipLaunchKernelGGL(dcn_build_grad_taps, dim3(2 * G, grp.n), dim3(256), 0, (hipStream_t)stream, grp);
  if (use_pair) {
    if (int rc = launch_offset_pair(grp, Go, workspace, max_K, stream)) return rc;
  } else
    hipLaunchKernelGGL(dcn_bwd_offset_plane<2>, dim3(Go), dim3(dcn_bwd_offset_plane_threads()), lds, (hipStream_t)stream, grp,
                       (float *)workspace, max_K);
  hipLaunchKernelGGL(dcn_bwd_offset_plane_fixup, dim3(grp.tile_begin[grp.n], 8), dim3(256), 0, (hipStream_t)stream, grp,
                     (const float *)workspace, Go, max_K);
  KGDET_CHECK_LAUNCH("dcn_bwd_plane_grouped");
  if (only_phase == 2) {   // (measurement switch: grad_input was NOT written)
    set_error("KGDET_OPT_BWD_PHASE = 2: grad_input not computed");
    return KGDET_E_PARTIAL;
  }
  return KGDET_OK;
}

// grad_weight of n v1 problems in one launch of the plane kernel (+ record / grad_out image builders, fix-up).
static int grad_weight_plane_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                     const float *const *offsets, const float *const *masks, const float *const *grad_outputs,
                                     float *const *grad_weights, void *workspace, size_t workspace_bytes, void *stream,
                                     bool gather = false);

int kgdet_deform_conv_grad_weight_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                          const float *const *offsets, const float *const *grad_outputs,
                                          float *const *grad_weights, void *workspace, size_t workspace_bytes,
                                          void *stream) {
  return grad_weight_plane_grouped(n, shapes, inputs, offsets, nullptr, grad_outputs, grad_weights, workspace, workspace_bytes,
                                   stream);
}

// masks: nullptr (v1) or one mask pointer per problem (v2: the tap records carry mask x bilinear weight)
// Weight groups and deformable groups: a problem is cut into runs of input channels that share both (sub-problems: the
// input window, the weight group's grad_out image and output rows, the deformable group's tap records); at most
// kMaxFwdGroup sub-problems per launch.
// gather: one problem on a map beyond the LDS plane -- pixel-major copy of x behind the tables, dcn_bwd_weight_gather.
static int grad_weight_plane_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                     const float *const *offsets, const float *const *masks, const float *const *grad_outputs,
                                     float *const *grad_weights, void *workspace, size_t workspace_bytes, void *stream,
                                     bool gather) {
  KGDET_CHECK_SHAPE(n >= 1 && n <= kMaxFwdGroup && shapes && inputs && offsets && grad_outputs && grad_weights,
                    "null pointer / group size not in [1, %d]", kMaxFwdGroup);
  KGDET_CHECK_SHAPE(!gather || n == 1, "the large-map grad_weight kernel takes one problem per launch");
  const int G = grid_size();
  Derived dd[kMaxFwdGroup];
  int same_taps[kMaxFwdGroup], same_gq[kMaxFwdGroup];
  size_t taps_off[kMaxFwdGroup], gq_off[kMaxFwdGroup], total = 0;
  int n_sub = 0;
  for (int i = 0; i < n; ++i) {
    const kgdet_dcn_shape *s = shapes[i];
    if (int rc = derive(s, dd[i])) return rc;
    KGDET_CHECK_SHAPE(inputs[i] && offsets[i] && grad_outputs[i] && grad_weights[i], "null pointer (problem %d)", i);
    if (!(gather ? gather_ok(s, dd[i]) : plane_ok(s, dd[i])) || dd[i].K > 128) {   // (the fix-up: at most 16 tap groups per channel chunk)
      set_error("problem %d is not eligible for the plane grad_weight kernel", i);
      return KGDET_E_UNSUPPORTED;
    }
    const int cpdg = s->C / s->deformable_groups;
    for (int c0 = 0; c0 < s->C;) {   // runs of channels inside one weight group and one deformable group
      const int c1 = std::min((c0 / dd[i].Cg + 1) * dd[i].Cg, (c0 / cpdg + 1) * cpdg);
      ++n_sub;
      c0 = c1;
    }
    same_taps[i] = same_gq[i] = -1;
    for (int q = 0; q < i; ++q) {
      const kgdet_dcn_shape *o = shapes[q];
      const bool geo = o->N == s->N && o->H == s->H && o->W == s->W && o->kh == s->kh && o->kw == s->kw &&
                       o->stride_h == s->stride_h && o->stride_w == s->stride_w && o->pad_h == s->pad_h &&
                       o->pad_w == s->pad_w && o->dil_h == s->dil_h && o->dil_w == s->dil_w &&
                       o->deformable_groups == s->deformable_groups;
      if (same_taps[i] < 0 && geo && offsets[q] == offsets[i] && (!masks || masks[q] == masks[i]))
        same_taps[i] = same_taps[q] >= 0 ? same_taps[q] : q;
      if (same_gq[i] < 0 && grad_outputs[q] == grad_outputs[i] && o->N == s->N && o->O == s->O && o->groups == s->groups &&
          o->out_channel_offset == s->out_channel_offset && o->out_channels_total == s->out_channels_total &&
          dd[q].Ho == dd[i].Ho && dd[q].Wo == dd[i].Wo)
        same_gq[i] = same_gq[q] >= 0 ? same_gq[q] : q;
    }
    if (same_taps[i] < 0) { taps_off[i] = total; total += gather ? gather_table_bytes(s, dd[i]) : align_up(tap_table_bytes(s, dd[i]), 256); }
    else taps_off[i] = taps_off[same_taps[i]];
    const int n_px16 = ceil_div(dd[i].Ho * dd[i].Wo, kChunk);
    if (same_gq[i] < 0) { gq_off[i] = total; total += (size_t)s->groups * (dd[i].Og_pad / kTileM) * s->N * n_px16 * 16384; }
    else gq_off[i] = gq_off[same_gq[i]];
  }
  if (n_sub > kMaxFwdGroup) {
    set_error("more than %d (weight group, deformable group) channel runs in one launch", kMaxFwdGroup);
    return KGDET_E_UNSUPPORTED;
  }
  const size_t image_off = total;   // (gather) the pixel-major copy of x
  if (gather) total += gather_image_bytes(shapes[0]);
  // Output-stationary kernel (dcn_backward_weight_os.hip, round 4): a workgroup owns a 256 x (16 channels x 13 taps) tile for
  // the whole reduction over pixels -- no partial tiles, no fix-up -- when all tiles of the call fit ONE round over the CUs
  // (a KGDet head stage: 224).  Otherwise (and KGDET_DCN_WGRAD_OS=0, A/B) the stream-K kernel + fix-up below.
  bool use_os = false;
  if (!gather) {
    static const int os_on = [] { const char *e = getenv("KGDET_DCN_WGRAD_OS"); return e ? atoi(e) : 1; }();
    long long os_tiles = 0;
    for (int i = 0; i < n; ++i) {
      const kgdet_dcn_shape *s = shapes[i];
      const int cpdg = s->C / s->deformable_groups;
      for (int c0 = 0; c0 < s->C;) {
        const int c1 = std::min((c0 / dd[i].Cg + 1) * dd[i].Cg, (c0 / cpdg + 1) * cpdg);
        os_tiles += (long long)(dd[i].Og_pad / kTileM) * ceil_div(c1 - c0, kChunk) * ceil_div(dd[i].K, dcn_bwd_weight_os_taps());
        c0 = c1;
      }
    }
    use_os = os_on && g_options[KGDET_OPT_WGRAD_STREAMK] == 0 && os_tiles <= G;
  }
  if (workspace == nullptr || workspace_bytes < slab_bytes() + total) {
    set_error("workspace too small: need %zu bytes, got %zu (kgdet_dcn_group_workspace_bytes)", slab_bytes() + total,
              workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_weight_plane<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_weight_gather<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kMaxLds));
    attr_set = true;
  }
  unsigned char *tab = (unsigned char *)workspace + slab_bytes();
  if (gather) {
    const kgdet_dcn_shape *s = shapes[0];
    const long long HW = (long long)s->H * s->W;
    hipLaunchKernelGGL(dcn_to_pixel_major, dim3((unsigned)((HW + 31) / 32), (s->C + 31) / 32, s->N), dim3(256), 0,
                       (hipStream_t)stream, inputs[0], reinterpret_cast<float *>(tab + image_off), s->C, HW, (long long)s->C * HW);
  }
  DcnFwdGroup grp;
  grp.n = 0; grp.xcd_slices = 1; grp.slots = kSlabSlots; grp.dbl_plane = 0; grp.plane_bytes = 0; grp.static_ranges = 0; grp.pair_mode = 0; grp.gather_mode = gather ? 1 : 0; grp.rounds = 1; grp.wave_layout = 0;
  grp.tile_begin[0] = 0; grp.range_begin[0] = 0; grp.unit_begin[0] = 0;
  size_t lds = 0;
  int min_len = 1 << 30;
  DcnPackGradOut pack;          // the grad_out images of the call: one launch (at most one per channel run: <= kMaxFwdGroup)
  int n_pack = 0, pack_x = 0, pack_y = 0;
  for (int i = 0; i < n; ++i) {
    const kgdet_dcn_shape *s = shapes[i];
    const Derived &d = dd[i];
    const int cpdg = s->C / s->deformable_groups;
    const int n_px16 = ceil_div(d.Ho * d.Wo, kChunk);
    const size_t gq_group_bytes = (size_t)(d.Og_pad / kTileM) * s->N * n_px16 * 16384;
    int packed_group = -1;   // grad_out images of the weight groups are packed as their first channel run is met
    bool first_sub = true;
    for (int c0 = 0; c0 < s->C;) {
      const int g = c0 / d.Cg, dgi = c0 / cpdg;
      const int c1 = std::min((g + 1) * d.Cg, (dgi + 1) * cpdg);
      DcnProblem p;
      fill_problem(s, d, g, p);
      p.x = gather ? reinterpret_cast<const float *>(tab + image_off) : inputs[i];
      p.offset = offsets[i]; p.mask = masks ? masks[i] : nullptr;
      p.c_base = c0; p.Cg = c1 - c0; p.Cg_pad = ceil_div(p.Cg, kChunk) * kChunk;
      p.dgi = dgi;
      p.w_ld = d.Cg;                                                      // a weight row holds the whole group's channels
      p.out = grad_weights[i] + ((size_t)g * d.Og * d.Cg + (size_t)(c0 - g * d.Cg)) * d.K;
      p.taps = reinterpret_cast<const DcnTapRec *>(tab + taps_off[i]);
      p.build_taps = same_taps[i] < 0 && first_sub;
      p.wq = tab + gq_off[i] + (size_t)g * gq_group_bytes;
      if (same_gq[i] < 0 && packed_group < g) {
        DcnPackGradOutItem &it = pack.item[n_pack++];
        it.gout = grad_outputs[i]; it.gq = (void *)(tab + gq_off[i] + (size_t)g * gq_group_bytes);
        it.N = s->N; it.O_total = p.O_total; it.o_base = p.o_base; it.Og = d.Og; it.HoWo = p.HoWo; it.n_px16 = n_px16;
        it.n_mtiles = d.Og_pad / kTileM;
        pack_x = std::max(pack_x, s->N * n_px16);
        pack_y = std::max(pack_y, d.Og_pad / kTileM);
        packed_group = g;
      }
      p.n_mtiles = d.Og_pad / kTileM;
      p.tiles_per_image = ceil_div(d.K, use_os ? dcn_bwd_weight_os_taps() : 8);   // tap groups per channel chunk
      p.n_ntiles = (p.Cg_pad / kChunk) * p.tiles_per_image;    // (chunk, tap group) column tiles
      p.chunks_per_tap = n_px16;                               // stages per image
      p.chunks_per_tile = s->N * n_px16;                       // the reduction runs over the pixels of all images
      p.total_units = (long long)p.n_ntiles * p.n_mtiles * p.chunks_per_tile;
      p.kparts = 1;
      p.flags = 0;
      grp.p[grp.n] = p;
      grp.tile_begin[grp.n + 1] = grp.tile_begin[grp.n] + p.n_ntiles * p.n_mtiles;
      grp.range_begin[grp.n + 1] = grp.range_begin[grp.n] + p.n_ntiles * p.n_mtiles;
      grp.unit_begin[grp.n + 1] = grp.unit_begin[grp.n] + p.total_units;
      ++grp.n;
      min_len = p.chunks_per_tile < min_len ? p.chunks_per_tile : min_len;
      first_sub = false;
      c0 = c1;
    }
    const size_t need = gather ? dcn_bwd_weight_gather_lds_bytes(2)
                               : use_os ? dcn_bwd_weight_os_lds_bytes(2, s->H * s->W) : dcn_bwd_weight_plane_lds_bytes(2, s->H * s->W);
    lds = need > lds ? need : lds;
  }
  if (use_os) {
    static thread_local bool os_attr_set = false;
    if (!os_attr_set) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_weight_os<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kMaxLds));
      os_attr_set = true;
    }
    if (n_pack > 0)
      hipLaunchKernelGGL(dcn_pack_grad_out, dim3(pack_x, pack_y, n_pack), dim3(256), 0, (hipStream_t)stream, pack, 2);
    hipLaunchKernelGGL(dcn_build_taps, dim3(2 * G, grp.n), dim3(256), 0, (hipStream_t)stream, grp);
    const int Gos = ceil_div(grp.tile_begin[grp.n], 8) * 8;      // (a multiple of 8: the XCD-contiguous tile order is a bijection)
    hipLaunchKernelGGL(dcn_bwd_weight_os<2>, dim3(Gos), dim3(dcn_bwd_weight_os_threads()), lds, (hipStream_t)stream, grp);
    KGDET_CHECK_LAUNCH("dcn_bwd_weight_os");
    return KGDET_OK;
  }
  // The fix-up lists at most 32 slabs per tile: small problems run on fewer workgroups, so that a workgroup's share of the
  // units is at least 1/30 of the longest tile.
  int Gw = G;
  {
    int max_len = 0;
    for (int i = 0; i < grp.n; ++i) max_len = grp.p[i].chunks_per_tile > max_len ? grp.p[i].chunks_per_tile : max_len;
    const long long need = ceil_div(max_len, 30);
    const long long fit = grp.unit_begin[grp.n] / need;
    if (fit < Gw) Gw = (int)std::max<long long>(1, fit);
  }
  if (ceil_div((int)ceil_div((int)grp.unit_begin[grp.n], Gw), min_len) + 2 > kSlabSlots) {
    set_error("group too uneven for the slab slots");
    return KGDET_E_UNSUPPORTED;
  }
  if (n_pack > 0)
    hipLaunchKernelGGL(dcn_pack_grad_out, dim3(pack_x, pack_y, n_pack), dim3(256), 0, (hipStream_t)stream, pack, 2);
  hipLaunchKernelGGL(dcn_build_taps, dim3(2 * G, grp.n), dim3(256), 0, (hipStream_t)stream, grp);
  if (gather)
    hipLaunchKernelGGL(dcn_bwd_weight_gather<2>, dim3(Gw), dim3(dcn_bwd_weight_plane_threads()), lds, (hipStream_t)stream, grp,
                       (float *)workspace);
  else
    hipLaunchKernelGGL(dcn_bwd_weight_plane<2>, dim3(Gw), dim3(dcn_bwd_weight_plane_threads()), lds, (hipStream_t)stream, grp,
                       (float *)workspace);
  {
    int fix_blocks = 0, max_K = 0;
    for (int i = 0; i < grp.n; ++i) {
      fix_blocks += (grp.p[i].n_ntiles / grp.p[i].tiles_per_image) * grp.p[i].n_mtiles * 32;
      max_K = grp.p[i].K > max_K ? grp.p[i].K : max_K;
    }
    const size_t fix_lds = (size_t)8 * 16 * max_K * sizeof(float);
    static thread_local bool fix_attr_set = false;
    if (!fix_attr_set) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_weight_plane_fixup, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        80 * 1024));   // (beside its static slab lists)
      fix_attr_set = true;
    }
    hipLaunchKernelGGL(dcn_bwd_weight_plane_fixup, dim3(fix_blocks), dim3(256), fix_lds, (hipStream_t)stream, grp,
                       (const float *)workspace, Gw);
  }
  KGDET_CHECK_LAUNCH("dcn_bwd_weight_plane");
  return KGDET_OK;
}

int kgdet_deform_conv_backward_input(const kgdet_dcn_shape *s, const float *input, const float *offset,
                                     const float *mask, const float *packed_weight, const float *grad_output,
                                     float *grad_input, float *grad_offset, float *grad_mask, void *workspace,
                                     size_t workspace_bytes, void *stream) {
  Derived d;
  if (int rc = derive(s, d)) return rc;
  KGDET_CHECK_SHAPE(input && offset && packed_weight && grad_output && grad_input && grad_offset, "null pointer");
  KGDET_CHECK_SHAPE((mask == nullptr) == (grad_mask == nullptr), "mask and grad_mask must come together");
  // v1 on small maps: the two plane kernels (grad_input by transposed sampling, grad_offset with the column
  // gradient in registers), bf16 hi/lo split MFMA -- 2x faster than the f32 gather kernel below
  const bool plane_off = g_options[KGDET_OPT_EXACT_BACKWARD] != 0;
  if (!plane_off && plane_bwd_input_ok(s, d) && plane_bwd_offset_ok(s, d, mask != nullptr) &&
      workspace_bytes >= kgdet_dcn_workspace_bytes(s)) {
    // (grad_offset first: the v2 variant declines -- before launching anything -- when its ranges outnumber the
    // workgroups, and the gather path below then does everything)
    const int rc = grad_offset_plane(s, input, offset, mask, packed_weight, grad_output, grad_offset, grad_mask, 0, workspace,
                                     workspace_bytes, stream);
    if (rc == KGDET_OK)
      return kgdet_deform_conv_grad_input(s, offset, mask, packed_weight, grad_output, grad_input, 0, workspace,
                                          workspace_bytes, stream);
    if (rc != KGDET_E_UNSUPPORTED) return rc;
  }
  const int cpdg = s->C / s->deformable_groups;
  const BwdLdsPlan pl = plan_bwd_lds(s, d);
  if (pl.ok && mfma_bwd_ok(s)) {   // (channel tiles that straddle deformable groups: the channel runs of the path below)
    // gather path: no atomics anywhere, outputs need no pre-zeroing
    const size_t need = (pl.slab_floats + pl.off_floats + pl.mask_floats) * sizeof(float) +
                        pl.rowptr_ints * sizeof(int) + pl.entry_pairs * 8 + 64;
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
      return KGDET_E_WORKSPACE;
    }
    float *slabs = (float *)workspace;
    float *off_part = slabs + pl.slab_floats;
    float *mask_part = mask ? off_part + pl.off_floats : nullptr;
    int2 *entries = (int2 *)align_up((size_t)(off_part + pl.off_floats + pl.mask_floats), 16);
    int *row_ptr = (int *)(entries + pl.entry_pairs);
    static thread_local bool attr_set = false;
    if (!attr_set) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_input_gather, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kMaxLds));
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)dcn_bwd_build_index, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kMaxLds - 64));
      attr_set = true;
    }
    {
      DcnProblem p;
      fill_problem(s, d, 0, p);
      p.x = input; p.offset = offset; p.mask = mask;
      hipLaunchKernelGGL(dcn_bwd_build_index, dim3(s->N * s->deformable_groups * d.K), dim3(256),
                         pl.index_lds_bytes, (hipStream_t)stream, p, row_ptr, entries);
    }
    for (int g = 0; g < s->groups; ++g) {
      DcnProblem p;
      fill_problem(s, d, g, p);
      p.x = input; p.offset = offset; p.mask = mask;
      p.wpk = packed_weight + (size_t)s->groups * d.fwd_image_floats() + (size_t)g * d.bwd_image_floats();
      DcnBwdInputLdsArgs a{};
      a.grad_out = grad_output; a.slabs = slabs; a.off_part = off_part; a.mask_part = mask_part;
      a.Og_pad16 = d.Og_pad16; a.Cg_pad256 = d.Cg_pad256;
      a.n_cslices = pl.n_cslices; a.S = pl.S; a.n_pblocks = pl.n_pblocks; a.slice_base = g * pl.n_cslices;
      a.HWp = pl.HWp;
      hipLaunchKernelGGL(dcn_bwd_input_gather, dim3(pl.n_blocks), dim3(kThreads), pl.lds_bytes, (hipStream_t)stream,
                         p, a, (const int *)row_ptr, (const int2 *)entries);
      hipLaunchKernelGGL(dcn_bwd_input_fixup, dim3(ceil_div(32 * s->H * s->W, 256 * 4), s->N * pl.n_cslices), dim3(256),
                         0, (hipStream_t)stream, p, a, grad_input);
    }
    const long long off_elems = (long long)s->N * s->deformable_groups * 2 * d.K * d.Ho * d.Wo;
    int fix_grid = (int)((off_elems + 255) / 256);
    if (fix_grid > 4096) fix_grid = 4096;
    hipLaunchKernelGGL(dcn_bwd_offset_fixup, dim3(fix_grid), dim3(256), 0, (hipStream_t)stream,
                       (const float *)off_part, (const float *)mask_part, grad_offset, grad_mask, pl.n_slices_total,
                       s->N, s->deformable_groups, d.K, d.Ho * d.Wo, pl.n_cslices, d.Cg, cpdg);
    KGDET_CHECK_LAUNCH("dcn_bwd_input_gather");
    return KGDET_OK;
  }
  // large feature maps, v1 and v2, any weight groups / deformable groups: materialised transposed column gradient + inverse index
  // per CHANNEL RUN (channels that share weight group and deformable group), no atomics (dcn_backward_large.hip).  Rounds 1-4 kept
  // a float-atomic scatter kernel (dcn_bwd_input_mfma) for groups on such maps -- the reference's one non-deterministic piece
  // (deform_conv_cuda_kernel.cu:329) -- and -munsafe-fp-atomics in the Makefile for it: both gone.
  {
    const int n_runs_max = s->C;   // (upper bound; channel_runs counts)
    (void)n_runs_max;
    const int cpdg_ = s->C / s->deformable_groups;
    bool seen_dg[64] = {false};
    if (s->deformable_groups > 64) { set_error("more than 64 deformable groups"); return KGDET_E_UNSUPPORTED; }
    // eligibility first (nothing launched on failure)
    for (int c0 = 0; c0 < s->C;) {
      const int g = c0 / d.Cg, dgi = c0 / cpdg_;
      const int c1 = std::min((g + 1) * d.Cg, (dgi + 1) * cpdg_);
      DcnProblem q;
      fill_problem(s, d, g, q);
      q.C_total = c1 - c0; q.mask = mask;
      if (!dcn_bwd_large_ok(q, mask != nullptr, 1)) {
        set_error("backward_input on a map of %d pixels: channel run [%d, %d) is not eligible for the column-gradient path "
                  "(output channels per group %% 16, (taps x channels) %% 2, at most 49 taps)", s->H * s->W, c0, c1);
        return KGDET_E_UNSUPPORTED;
      }
      c0 = c1;
    }
    for (int c0 = 0; c0 < s->C;) {
      const int g = c0 / d.Cg, dgi = c0 / cpdg_;
      const int c1 = std::min((g + 1) * d.Cg, (dgi + 1) * cpdg_);
      DcnProblem q;
      fill_problem(s, d, g, q);
      q.x = input + (size_t)c0 * s->H * s->W; q.offset = offset; q.mask = mask;
      q.C_total = c1 - c0; q.dgi = dgi;
      q.wpk = packed_weight + (size_t)g * d.fwd_image_floats() + (size_t)(c0 - g * d.Cg) * d.Og_pad;   // [K][Cg_pad][Og_pad]: from the run's first channel
      if (int rc = dcn_bwd_large(q, grad_output, s->out_channels_total > 0 ? s->out_channels_total : s->O,
                                 s->out_channel_offset + g * d.Og, grad_input, grad_offset,
                                 grad_mask, workspace, workspace_bytes, stream, s->C, c0, seen_dg[dgi] ? 1 : 0))
        return rc;
      seen_dg[dgi] = true;
      c0 = c1;
    }
  }
  return KGDET_OK;
}

int kgdet_deform_conv_backward_weight(const kgdet_dcn_shape *s, const float *input, const float *offset,
                                      const float *mask, const float *grad_output, float *grad_weight,
                                      float *grad_bias, int accumulate, void *workspace, size_t workspace_bytes,
                                      void *stream) {
  Derived d;
  if (int rc = derive(s, d)) return rc;
  KGDET_CHECK_SHAPE(input && offset && grad_output && grad_weight, "null pointer");
  // v1 or v2, any weight / deformable groups: the split-operand kernel with natural-layout output -- small maps on the
  // LDS plane, larger ones gathering from a pixel-major copy of x
  if (!accumulate && g_options[KGDET_OPT_EXACT_BACKWARD] == 0 && (plane_ok(s, d) || gather_ok(s, d))) {
    const int rc = grad_weight_plane_grouped(1, &s, &input, &offset, mask ? &mask : nullptr, &grad_output, &grad_weight, workspace,
                                             workspace_bytes, stream, !plane_ok(s, d));
    if (rc == KGDET_OK) {
      if (grad_bias) {
        const int O_total = s->out_channels_total > 0 ? s->out_channels_total : s->O;
        hipLaunchKernelGGL(dcn_bias_grad, dim3(s->O), dim3(256), 0, (hipStream_t)stream,
                           grad_output + (size_t)s->out_channel_offset * d.Ho * d.Wo, grad_bias, s->N, O_total,
                           d.Ho * d.Wo, accumulate);
        KGDET_CHECK_LAUNCH("dcn_bias_grad");
      }
      return KGDET_OK;
    }
    if (rc != KGDET_E_UNSUPPORTED && rc != KGDET_E_WORKSPACE) return rc;   // else: the f32 kernel below
  }
  const int cpdg = s->C / s->deformable_groups;
  if (!(s->deformable_groups == 1 || cpdg % d.Cg == 0 || (d.Cg % cpdg == 0 && cpdg % kTileN == 0))) {
    set_error("deformable_groups=%d / groups=%d / C=%d: channel tiles straddle deformable groups (unsupported)",
              s->deformable_groups, s->groups, s->C);
    return KGDET_E_UNSUPPORTED;
  }
  const size_t need = slab_bytes() + (size_t)s->groups * d.fwd_image_floats() * sizeof(float);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  float *slabs = (float *)workspace;
  float *gpk = (float *)((char *)workspace + slab_bytes());  // packed gradient image
  const int G = grid_size();
  for (int g = 0; g < s->groups; ++g) {
    DcnProblem p;
    fill_problem(s, d, g, p);
    p.x = input; p.offset = offset; p.mask = mask;
    p.out = gpk + (size_t)g * d.fwd_image_floats();
    DcnBwdWeightArgs a{};
    a.grad_out = grad_output;
    a.n_ctiles = ceil_div(d.Cg_pad, kTileN);
    a.n_otiles = d.Og_pad / kTileM;
    a.stages_per_tile = ceil_div(p.P, kChunk);
    a.Cg_pad128 = a.n_ctiles * kTileN;
    const int n_tiles = d.K * a.n_otiles * a.n_ctiles;
    a.total_units = (long long)n_tiles * a.stages_per_tile;
    hipLaunchKernelGGL(dcn_bwd_weight_mfma, dim3(G), dim3(kThreads), 0, (hipStream_t)stream, p, a, slabs);
    hipLaunchKernelGGL(dcn_bwd_weight_fixup, dim3(n_tiles, 4), dim3(kThreads), 0, (hipStream_t)stream, p, a,
                       (const float *)slabs, G);
  }
  KGDET_CHECK_LAUNCH("dcn_bwd_weight_mfma");
  if (int rc = kgdet_dcn_unpack_weight_grad(s, gpk, grad_weight, accumulate, stream)) return rc;
  if (grad_bias) {
    const int O_total = s->out_channels_total > 0 ? s->out_channels_total : s->O;
    hipLaunchKernelGGL(dcn_bias_grad, dim3(s->O), dim3(256), 0, (hipStream_t)stream,
                       grad_output + (size_t)s->out_channel_offset * d.Ho * d.Wo, grad_bias, s->N, O_total,
                       d.Ho * d.Wo, accumulate);
    KGDET_CHECK_LAUNCH("dcn_bias_grad");
  }
  return KGDET_OK;
}

}  // extern "C"
