// Deformable convolution on gfx950: shared device-side pieces.
//
// The three kernels (forward, backward-input, backward-weight) are all "gather -> LDS -> f32 MFMA"
// pipelines with the same geometry:
//   workgroup = 512 threads = 8 waves arranged 4 (M) x 2 (N); each wave owns a 64x64 block of a
//   256 (M) x 128 (N) tile as 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 VGPRs);
//   the reduction dimension is consumed in chunks of 16, double-buffered in LDS as
//   A[k][256] / B[k][128] (k-major), which makes every MFMA operand fetch a conflict-free
//   ds_read_b32 (lanes 0-31 read 32 consecutive floats of row k, lanes 32-63 of row k+1).
// f32-input MFMA is bit-exact fp32 (an fma chain), so results track the fp32 reference closely.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kgdet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileM = 256;
constexpr int kTileN = 128;
constexpr int kChunk = 16;     // reduction elements per LDS stage
constexpr int kThreads = 512;  // 8 waves
constexpr int kTileElems = kTileM * kTileN;

// One sampling point of the plane forward kernel, precomputed by dcn_build_taps: where the four bilinear
// corners of (image, tap, output pixel) sit in the LDS feature plane and what they weigh (0 where a corner, or
// the whole tap, falls outside the image; x the modulation mask for v2).
struct DcnTapRec {
  unsigned off[4];  // LDS byte offset of channel quad 0 of the corner pixel (pixel * 16); quad c is one quad stride further
  float w[4];
};
// Transposed sampling (grad_input plane kernel): what input cell q collects for tap t.
//   DcnInvRec      a cell with at most 8 contributing (output pixel, corner) pairs: the (pixel, weight) contributions, pixels
//                  as LDS byte offsets like DcnTapRec (zero weight = unused slot); stored as [2 x off[4]][2 x w[4]] = 64 B.
//                  A cell with MORE than 8 contributions (round 4) carries eight zero weights and, in the upper bits of its
//                  last offset, bit 31 + a slot number (bits 17..30): its whole sum over the contributions
//                      Gov[image, tap, slot][o] = sum_e w_e * grad_out[o, p_e]     for ALL output channels o
//                  was computed beforehand by dcn_inv_overflow_sums and the kernel adds that vector.
//   DcnInvOvfCell  one such cell: its entries [start, start + n) of the (image, tap)'s entry list (unsorted; the summation
//                  order is made a fixed one -- ascending pixel -- by the kernel that sums)
// Why: a trained KGDet head samples tap t of EVERY location on an object at the same key point, so a few cells collect
// hundreds of contributions each.  Rounds 2-3 kept contributions 9.. in per-(tap, tile) overflow lists that ONE producer
// thread walked, per 16-channel chunk: the training step's grad_input went from 0.24 ms (random offsets) to 1.6 ms per
// launch within 500 steps on the synthetic batch (profiles/r04_steady_trace.md) -- the slow-down round 3 mistook for a
// thermal ramp.
struct DcnInvOvfCell {
  int start, n, cell, pad;
};
constexpr int kInvInline = 8;             // contributions a record holds
constexpr unsigned kInvFlag = 1u << 31;   // record's last offset: the cell's sum is pre-aggregated
__device__ __host__ __forceinline__ int dcn_inv_max_slots(int HW, int HoWo) {   // cells of one (image, tap) with > 8 contributions
  const int by_entries = 4 * HoWo / (kInvInline + 1);
  return (by_entries < HW ? by_entries : HW) + 1;
}

// The LDS feature plane of the plane kernels: a 16-channel slice of one image as FOUR quad planes
// [quad][pixel][4 channels] fp32 -- pixel q's channel quad c at c * stride + q * 16.  A bilinear corner of 4 channels is
// one ds_read_b128, and the quad is an IMMEDIATE offset of that instruction where the stride is a compile-time constant
// (kPlaneQuadStride: plane_role, the forward / grad_input kernels, whose plane starts at LDS address 0), so a tap
// record's four offsets are used as they are (the [pixel][16 channel] rows of rounds 1-2 needed an XOR swizzle per read
// to spread the quads over the banks: 12 VALU instructions per stage and SIMD beside the MFMA waves).  16 consecutive
// pixels of a quad cover all 64 banks.  grad_offset / grad_weight kernels use the same records with a run-time stride
// of pixels_padded * 16 (their LDS budgets depend on K).
constexpr int kPlaneMaxHW = 1344;                       // 3 * stride must fit the 16-bit offset field of ds_read
constexpr int kPlaneQuadStride = kPlaneMaxHW * 16;      // 21504 B
__device__ __host__ __forceinline__ int dcn_plane_offset(int q) { return q * 16; }
__device__ __host__ __forceinline__ int dcn_plane_padded_pixels(int HW) { return (HW + 63) & ~63; }

// Copy x[image, c0 .. c0 + 15, :, :] (xb = the image's first channel of the group, Cg channels, HW pixels each) into the
// quad planes.  The work is cut into UNITS = (quad, block of 64 pixels), one wave per unit: quad and block are wave-uniform,
// so the four loads of an item are buffer loads with ONE shared 32-bit lane offset and the 16-byte LDS store needs
// one address add -- 4 vector instructions per unit and lane (the item-per-thread copy of rounds 1-2 spent ~39 on the
// divisions by HW and 64-bit addresses, a third of all VALU instructions of the forward kernel).  Lanes past the end
// of the image re-read its last pixel and store into the padding of their quad plane (stride >= padded pixels * 16).
// This wave takes units u_first, u_first + u_step, ... < u_hi, ROUNDS of them in flight (all loads before the first
// store, unconditional from clamped addresses, so the counted s_waitcnt stay exact).  u_first, u_step, u_hi must be
// wave-uniform (readfirstlane).  ZERO_PAD: channels past Cg read as zero instead of repeating the last one.
// Buffer resource over a wave-uniform base pointer (pinned to SGPRs with readfirstlane): loads through it take the form
// buffer_load v, v_offset, s[rsrc], s_offset -- a 32-bit lane offset, a scalar offset computed on the scalar unit and an
// immediate -- so a load costs NO vector instruction beside the MFMA waves.  (Plain pointers read out of a dynamically
// indexed DcnProblem lose their address space: hipcc emits flat_load with a 64-bit vector add per load, and flat loads
// also count on lgkmcnt, the counter the LDS reads wait on.)  Offsets are bytes, < 4 GB from the base; no range check.
typedef __amdgpu_buffer_rsrc_t dcn_rsrc_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dcn_rsrc_t dcn_make_rsrc(const void *ptr) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, -1, 0x00020000);
}
__device__ __forceinline__ float dcn_buf_f32(dcn_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ u32x4_t dcn_buf_b128(dcn_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
// One 1 KiB piece of the blocked feature image straight into LDS (LDS-DMA: no registers, no ds_write): lane l's 16 bytes
// from buffer offset voff + soff land at lds + 16 l.  hipcc does not count this load in its vmcnt bookkeeping (its own
// counted waits only get stricter by it); the caller waits with dcn_wait_vm0() before the barrier that publishes the data.
__device__ __forceinline__ void dcn_dma_b128(u32x4_t rsrc, unsigned voff, unsigned soff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds)
               : "memory");
}
__device__ __forceinline__ void dcn_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int ROUNDS, bool ZERO_PAD = false>
__device__ __forceinline__ void dcn_plane_copy(const float *__restrict__ xb, int HW, int Cg, int c0, unsigned char *plane,
                                               unsigned stride, int u_first, int u_step, int u_hi, int lane) {
  const int nblk = (HW + 63) >> 6;
  // (u >= k * nblk) as 1 + ((u - k * nblk) >> 31): sums of compare results make hipcc leave the scalar unit
  auto quad_of = [&](int u) { return 3 + ((u - nblk) >> 31) + ((u - 2 * nblk) >> 31) + ((u - 3 * nblk) >> 31); };
  const dcn_rsrc_t xrs = dcn_make_rsrc(xb);
  for (int u0 = u_first; u0 < u_hi; u0 += ROUNDS * u_step) {
    f32x4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int u = min(u0 + r * u_step, u_hi - 1);
      const int quad = quad_of(u);
      const int blk = u - quad * nblk;
      const unsigned voff = (unsigned)min(blk * 64 + lane, HW - 1) * 4u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ch = c0 + quad * 4 + e;
        const unsigned soff = (unsigned)(min(ch, Cg - 1) * HW) * 4u;   // (a channel group of one image: < 4 GB)
        const float f = dcn_buf_f32(xrs, voff, soff);
        v[r][e] = (ZERO_PAD && ch >= Cg) ? f * 0.0f : f;
      }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int u = u0 + r * u_step;
      if (u < u_hi) {
        const int quad = quad_of(u);
        const int blk = u - quad * nblk;
        *reinterpret_cast<f32x4 *>(plane + quad * stride + (unsigned)(blk * 64 + lane) * 16u) = v[r];
      }
    }
  }
}
__device__ __forceinline__ int dcn_plane_units(int HW) { return 4 * ((HW + 63) >> 6); }

// The bf16 plane of the one-product (KGDET_DCN_BF16) forward kernel: two planes [oct][pixel][8 channels bf16] at the same
// stride -- a corner of EIGHT channels is one ds_read_b128, half the gathers and half the LDS bytes of the fp32 planes (the
// bf16 kernel's producers are what bounds it: one MFMA per product).  x is rounded to bf16 before the interpolation instead
// of after it; same error class.  Units = (oct, block of 64 pixels): eight buffer loads, four v_cvt_pk, one 16-byte store.
// KGDET_PLANE_F16 (round 5, default): the same planes hold FP16 -- three more mantissa bits than bf16, and the producers' fp32
// interpolation reads them through v_fma_mix_f32 (the fp16 -> fp32 conversion of a corner value is a source modifier of the
// FMA: eight instructions per corner of eight channels instead of eight shift / mask unpacks + eight FMAs).  Range: |x| <=
// 65504, beyond that the value saturates (the MODE register's FP16_OVFL bit, set by the kernel) where bf16 kept the exponent:
// activations behind GroupNorm / ReLU towers sit far inside.  -DKGDET_PLANE_F16=0: the bf16 planes of rounds 2-4 (A/B).
#ifndef KGDET_PLANE_F16
#define KGDET_PLANE_F16 1
#endif
typedef __bf16 dcn_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dcn_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dcn_f16x2 __attribute__((ext_vector_type(2)));
template <int ROUNDS>
__device__ __forceinline__ void dcn_plane_copy_bf16(const float *__restrict__ xb, int HW, int Cg, int c0, unsigned char *plane,
                                                    unsigned stride, int u_first, int u_step, int u_hi, int lane) {
  const int nblk = (HW + 63) >> 6;
  const dcn_rsrc_t xrs = dcn_make_rsrc(xb);
  for (int u0 = u_first; u0 < u_hi; u0 += ROUNDS * u_step) {
    float v[ROUNDS][8];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int u = min(u0 + r * u_step, u_hi - 1);
      const int oct = 1 + ((u - nblk) >> 31);            // (u >= nblk)
      const int blk = u - oct * nblk;
      const unsigned voff = (unsigned)min(blk * 64 + lane, HW - 1) * 4u;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ch = c0 + oct * 8 + e;
        v[r][e] = dcn_buf_f32(xrs, voff, (unsigned)(min(ch, Cg - 1) * HW) * 4u);
      }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int u = u0 + r * u_step;
      if (u < u_hi) {
        const int oct = 1 + ((u - nblk) >> 31);
        const int blk = u - oct * nblk;
#if KGDET_PLANE_F16
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        u32x4_ o;
#pragma unroll
        for (int e = 0; e < 4; ++e)     // (v_cvt_pk_f16_f32: round to nearest even, saturating with FP16_OVFL set)
          o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{v[r][2 * e], v[r][2 * e + 1]}, dcn_f16x2));
        *reinterpret_cast<u32x4_ *>(plane + oct * stride + (unsigned)(blk * 64 + lane) * 16u) = o;
#else
        dcn_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[r][e];
        *reinterpret_cast<dcn_bf16x8 *>(plane + oct * stride + (unsigned)(blk * 64 + lane) * 16u) = o;
#endif
      }
    }
  }
}
__device__ __forceinline__ int dcn_plane_units_bf16(int HW) { return 2 * ((HW + 63) >> 6); }

// One deformable convolution problem as the kernels see it (one weight group).
struct DcnProblem {
  const float *x;       // [N, C_total, H, W]
  const float *offset;  // [N, DG*2K, Ho, Wo]
  const float *mask;    // [N, DG*K, Ho, Wo] or nullptr
  const float *wpk;     // packed weight of this group: [K][Cg_pad][Og_pad]
  const void *wq;       // bf16 hi/lo image of this group for the plane kernel (dcn_forward_plane.hip)
  const DcnTapRec *taps;  // [N, DG, K, Ho*Wo] sampling records (plane kernel; MODE 1: DcnInvRec, 64 B each)
  const float *inv_gov;   // MODE 1: [N * K][gov_slots][gov_ld] pre-aggregated sums of the cells with > 8 contributions
  int gov_slots, gov_ld;  //         slots per (image, tap); floats per slot (all output channels of the convolution, padded to 16)
  int gov_c0;             //         first channel of this (sub-)problem's grad_out window inside a Gov vector
  int build_taps;         // this problem owns `taps` (others of the group may alias it: same offsets and geometry)
  const float *xblk;      // the blocked copy of x the LDS-DMA plane hand-overs read (forward chain, grad_offset pair kernel), or nullptr
  int build_xblk;         // this problem owns `xblk` (dcn_build_taps writes it)
  const float *bias;    // [O_total] or nullptr
  float *out;           // forward: [N, O_total, Ho, Wo]
  const float *gout;    // grad_offset kernel: grad_output [N, O_total, Ho, Wo]
  float *goff;          //                     grad_offset [N, 2K, Ho, Wo]
  float *gmask;         //                     grad_mask [N, K, Ho, Wo] (v2: `mask` set) or nullptr
  int N, C_total, c_base, Cg, Cg_pad;
  int O_total, o_base, Og, Og_pad;  // o_base: first channel of this group inside the output buffer
  int bias_base;                    // first channel of this group in the conv's own numbering
  int H, W, Ho, Wo, HoWo, P;  // P = N*Ho*Wo output pixels
  int kh, kw, K;
  int sh, sw, ph, pw, dh, dw;
  int DG, cpdg;  // deformable groups, channels (of C_total) per deformable group
  int dgi;       // backward plane kernels: the ONE deformable group this (sub-)problem's channels belong to
  int w_ld;      // grad_weight plane kernel: channels per row of the weight the tile is stored into (>= Cg: channel runs)
  int c16_base;  // grad_offset plane kernel: first 16-channel chunk of this channel run inside its weight group (rows of wqt)
  int mt_base, row0;  // grad_input plane kernel: first 256-row tile of wqt and first row inside it of this channel run
  int n_ntiles, n_mtiles, chunks_per_tap, chunks_per_tile;
  long long total_units;  // n_ntiles * n_mtiles * chunks_per_tile
  int tiles_per_image;    // > 0: pixel tiles never straddle images (plane kernel); 0: tiles run over N*Ho*Wo
  int kparts;             // plane kernel: the reduction range of every tile is cut into kparts parts (see DcnFwdGroup)
  // SUM GROUP (round 6; backward plane kernels, static ranges): problems of a launch whose results are one tensor -- grad_input
  // of the 3x3 / 5x5 / 7x7 convolutions of one feature map, grad_offset of the two maps that share an offset tensor.  Every
  // member writes ALL its ranges to slabs (none writes its output directly), and the fix-up block of the LEADER's tile adds the
  // members' parts in fixed order (members ascending, parts ascending) and writes the sum once: the autograd node's
  // stack + sum passes (5 cat + 5 reduce launches, 54 us per head stage) are gone.  sum_count <= 1: no group.
  int sum_count, sum_members[4];
  int seg_stages;         // stages per chunk of the reduction: K
  unsigned flags;
};

// Several independent forward problems run as ONE launch (the KGDet head runs a 3x3, a 5x5 and a 7x7
// deformable conv on each of two feature maps per stage): their (tile, stage) units are concatenated and
// dealt to the workgroups stream-K style, so slabs, fix-up and launch overhead are paid once per group.
constexpr int kMaxFwdGroup = 8;
struct DcnFwdGroup {
  int n;
  int xcd_slices;  // 1: workgroup g takes slice sk_slice_of_block(g) and numbers its slabs by range (plane kernel)
                   // 0: slice g, slab 0 = first partial tile, slab 1 = last partial tile (exact-fp32 kernel)
  int slots;       // slab slots per workgroup
  int dbl_plane;   // (unused: second feature-plane buffer, measured and dropped)
  int plane_bytes; // plane kernels: bytes of the largest feature plane of the group
  int wave_layout;    // accumulator layout of the slabs: 0 = 4 x 2 waves of 64 x 64 (2 x 2 MFMA blocks each),
                      // 1 = 8 x 1 waves of 32 x 128 (1 x 4 blocks each; plane kernels: every wave loads DISTINCT weight rows)
  int pair_mode;      // (unused since round 5: the tap-pair kernel left the library, tools/experiments/dcn_plane_pairs.h)
  int gather_mode;    // 1: no LDS plane (maps beyond kPlaneMaxHW): x is a pixel-major copy [N][H*W][C] and the tap records hold
                      //    byte offsets of its rows -- the producers gather 16-byte channel quads with buffer loads (plane_role MODE 2)
  int rounds;         // static schedule: the workgroup of slice r computes ranges r, r + G, ..., r + (rounds - 1) G
  int static_ranges;  // 1: workgroup of slice r computes exactly range r (problem, part, tile), r < range_begin[n];
                      //    the other workgroups exit.  Ranges of one (problem, part) are consecutive, so the 32
                      //    workgroups of an XCD walk the SAME weight stages at the same time and share them in L2.
  int tile_begin[kMaxFwdGroup + 1];        // prefix sums of n_ntiles * n_mtiles
  int range_begin[kMaxFwdGroup + 1];       // prefix sums of n_ntiles * n_mtiles * kparts
  long long unit_begin[kMaxFwdGroup + 1];  // prefix sums of total_units
  DcnProblem p[kMaxFwdGroup];
};

// The blocked copy of x the plane hand-overs' LDS-DMA reads: x[image, c_base + 16 c .. + 15, :, :] as
// [image][chunk c][quad][pixels padded to 64][4 channels] fp32 -- exactly the LDS plane's units, each 1 KiB contiguous.  Channels
// past Cg repeat the last one, pixels past H*W the last pixel (as dcn_plane_copy does).  One wave per unit; called from
// dcn_build_taps' blocks.
__host__ __device__ __forceinline__ size_t dcn_xblk_bytes(int N, int Cg_pad, int HW) {
  return (size_t)N * (Cg_pad / kChunk) * 4 * (((HW + 63) >> 6) * 64) * 16;
}
__device__ __forceinline__ void dcn_block_x_body(const DcnProblem &p, int first_unit, int unit_step) {
  const int HW = p.H * p.W, nblk = (HW + 63) >> 6, P64 = nblk * 64, n_c16 = p.chunks_per_tap;
  const int lane = threadIdx.x & 63;
  const long long total = (long long)p.N * n_c16 * 4 * nblk;
  for (long long u = first_unit; u < total; u += unit_step) {
    const int blk = (int)(u % nblk);
    const int quad = (int)((u / nblk) & 3);
    const int c = (int)((u / (4 * nblk)) % n_c16);
    const int b = (int)(u / ((long long)4 * nblk * n_c16));
    const int px = min(blk * 64 + lane, HW - 1);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = min(c * kChunk + quad * 4 + e, p.Cg - 1);
      v[e] = p.x[((long long)b * p.C_total + p.c_base + ch) * HW + px];
    }
    reinterpret_cast<f32x4 *>(const_cast<float *>(p.xblk))[(((long long)b * n_c16 + c) * 4 + quad) * P64 + blk * 64 + lane] = v;
  }
}
// Unit order of the plane kernel.  The weight image of a 7x7 conv is 25 MB (bf16 hi + lo) and every pixel tile
// streams all of it: with tile-major units the 256 workgroups pull ~0.8 GB per grouped launch through the
// fabric, which is what bounds the kernel (the per-XCD L2 is 4 MB).  So the reduction range of problem p is cut
// into p.kparts parts of <= ~1.7 MB of weights and the units are ordered (problem, part, tile, stage): the
// workgroups of one XCD (consecutive slices, sk_slice_of_block) then walk the tiles of ONE part together and
// share its weights through their L2.  A RANGE = (problem, part, tile) = a run of consecutive units; a
// workgroup's slice meets a few ranges, writes one partial tile (slab) per range it meets, in order, and
// dcn_fwd_fixup adds the slabs of a tile's ranges.
struct DcnUnitPos {
  int pi, part, tile;   // problem, reduction part, tile inside the problem
  int s, s_hi;          // first stage of the unit inside the tile, end of the part's stage range
  int range;            // global index of the range the unit lies in
};
// first stage of reduction part `part`: parts are whole channel chunks (a chunk = K consecutive stages sharing a plane)
__device__ __forceinline__ int dcn_part_lo(const DcnProblem &p, int part) {
  if (p.kparts == 1) return part ? p.chunks_per_tile : 0;   // (kernels whose stages are not (chunk, tap) pairs never split)
  return (int)((long long)p.chunks_per_tap * part / p.kparts) * p.seg_stages;
}
__device__ __forceinline__ DcnUnitPos dcn_unit_pos(const DcnFwdGroup &grp, long long u) {
  DcnUnitPos r;
  r.pi = 0;
  while (r.pi + 1 < grp.n && u >= grp.unit_begin[r.pi + 1]) ++r.pi;
  const DcnProblem &p = grp.p[r.pi];
  const int tiles = p.n_ntiles * p.n_mtiles;
  const long long v = u - grp.unit_begin[r.pi];
  r.part = 0;
  while (r.part + 1 < p.kparts && v >= (long long)tiles * dcn_part_lo(p, r.part + 1)) ++r.part;
  const int lo = dcn_part_lo(p, r.part);
  r.s_hi = dcn_part_lo(p, r.part + 1);
  const int len = r.s_hi - lo;
  const long long w = v - (long long)tiles * lo;
  r.tile = (int)(w / len);
  r.s = lo + (int)(w - (long long)r.tile * len);
  r.range = grp.range_begin[r.pi] + r.part * tiles + r.tile;
  return r;
}
// first unit of range (pi, part, tile)
__device__ __forceinline__ long long dcn_range_first_unit(const DcnFwdGroup &grp, int pi, int part, int tile) {
  const DcnProblem &p = grp.p[pi];
  const int tiles = p.n_ntiles * p.n_mtiles;
  const int lo = dcn_part_lo(p, part), len = dcn_part_lo(p, part + 1) - lo;
  return grp.unit_begin[pi] + (long long)tiles * lo + (long long)tile * len;
}

// Which slice of the stream-K unit space workgroup g takes.  Blocks are dealt to the 8 XCDs round-robin
// (block b -> XCD b % 8, each with a private 4 MB L2), so consecutive slices go to the workgroups of ONE
// XCD: neighbours in the unit space walk the same weight stages a few stages apart and share them through
// that L2 instead of each pulling them over the fabric.  Locality only; any bijection is correct.
__device__ __forceinline__ int sk_slice_of_block(int g, int G) {
  return (G % 8 == 0) ? (g % 8) * (G / 8) + g / 8 : g;
}
__device__ __forceinline__ int sk_block_of_slice(int r, int G) {
  return (G % 8 == 0) ? (r % (G / 8)) * 8 + r / (G / 8) : r;
}

// The unit interval [begin, end) of slice `slice`: stream-K share, or -- static_ranges -- exactly range `slice`
// (+ round * G in a multi-round static schedule; empty beyond the last range).
__device__ __forceinline__ void dcn_slice_bounds(const DcnFwdGroup &grp, long long slice, long long G, long long &begin,
                                                 long long &end, int round = 0) {
  const long long total = grp.unit_begin[grp.n];
  if (!grp.static_ranges) {
    begin = slice * total / G;
    end = (slice + 1) * total / G;
    return;
  }
  slice += (long long)round * G;
  if (slice >= grp.range_begin[grp.n]) { begin = end = 0; return; }
  int pi = 0;
  while (pi + 1 < grp.n && slice >= grp.range_begin[pi + 1]) ++pi;
  const DcnProblem &q = grp.p[pi];
  const int tiles = q.n_ntiles * q.n_mtiles;
  const int r = (int)slice - grp.range_begin[pi];
  const int part = r / tiles;
  begin = dcn_range_first_unit(grp, pi, part, r - part * tiles);
  end = begin + (dcn_part_lo(q, part + 1) - dcn_part_lo(q, part));
}

// Output pixel of column `col` of pixel tile `nt`: image b, position hw inside it; false if past the end.
__device__ __forceinline__ bool tile_pixel(const DcnProblem &p, int nt, int col, int &b, int &hw) {
  if (p.tiles_per_image > 0) {
    b = nt / p.tiles_per_image;
    hw = (nt - b * p.tiles_per_image) * kTileN + col;
    return hw < p.HoWo;
  }
  const int pix = nt * kTileN + col;
  if (pix >= p.P) { b = 0; hw = 0; return false; }
  b = pix / p.HoWo;
  hw = pix - b * p.HoWo;
  return true;
}

// stream-K: workgroup g of G owns units [unit_begin(g), unit_begin(g+1))
__device__ __forceinline__ long long unit_begin(long long g, long long total, long long G) {
  return g * total / G;
}

// Where one tap of one output pixel lands on the input, and how it is interpolated.
// Corner offsets are clamped into the plane and carry a zero weight when the corner (or the
// whole tap) is outside, so that gathers never branch: sample = sum_i w[i] * plane[o[i]].
// Mirrors deformable_im2col_bilinear + the (-1,H)x(-1,W) guard
// (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:84-114, :228).
struct Tap {
  int o[4];    // y0x0, y0x1, y1x0, y1x1 (element offsets inside an H*W plane)
  float w[4];  // bilinear weights (x mask for v2), 0 where invalid
};

struct TapGeom {  // also what the backward pass needs for d/dy, d/dx
  float ly, lx;   // fractional parts
  int in_range;   // tap inside (-1,H)x(-1,W)
  int va, vb, vc, vd;
};

__device__ __forceinline__ void make_tap(float y, float x, int H, int W, bool live, float m,
                                         Tap &t, TapGeom &g) {
  const bool in = live && (y > -1.0f) && (x > -1.0f) && (y < (float)H) && (x < (float)W);
  if (!in) { y = 0.0f; x = 0.0f; }
  const float fy = floorf(y), fx = floorf(x);
  const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
  const float ly = y - fy, lx = x - fx, hy = 1.0f - ly, hx = 1.0f - lx;
  const bool va = in && y0 >= 0 && x0 >= 0;
  const bool vb = in && y0 >= 0 && x1 <= W - 1;
  const bool vc = in && y1 <= H - 1 && x0 >= 0;
  const bool vd = in && y1 <= H - 1 && x1 <= W - 1;
  const int cy0 = max(y0, 0), cy1 = min(y1, H - 1), cx0 = max(x0, 0), cx1 = min(x1, W - 1);
  t.o[0] = cy0 * W + cx0;
  t.o[1] = cy0 * W + cx1;
  t.o[2] = cy1 * W + cx0;
  t.o[3] = cy1 * W + cx1;
  t.w[0] = va ? hy * hx * m : 0.0f;
  t.w[1] = vb ? hy * lx * m : 0.0f;
  t.w[2] = vc ? ly * hx * m : 0.0f;
  t.w[3] = vd ? ly * lx * m : 0.0f;
  g.ly = ly; g.lx = lx; g.in_range = in;
  g.va = va; g.vb = vb; g.vc = vc; g.vd = vd;
}

// Forward-only variant: the two corners of a row are adjacent floats, so one (4-byte aligned) 8-byte
// load fetches both -- half the vector-memory instructions, which matters because the gather path
// (texture addresser) and not the MFMA pipe is what saturates first (GRBM_TA_BUSY 90 % at 47 % MFMA).
// The pair always starts inside the row: for x0 == -1 it starts at column 0 and the first element
// takes the right-corner weight, for x0 == W-1 it starts at W-2 and the second element takes the
// left-corner weight; dead corners carry weight 0 as before.  Requires W >= 2.
struct TapPair {
  int o[2];    // start of the pair in row y0 / row y1 (element offsets inside an H*W plane)
  float w[4];  // weights of (row0.x, row0.y, row1.x, row1.y)
};

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ void make_tap_pair(float y, float x, int H, int W, bool live, float m, TapPair &t) {
  const bool in = live && (y > -1.0f) && (x > -1.0f) && (y < (float)H) && (x < (float)W);
  if (!in) { y = 0.0f; x = 0.0f; }
  const float fy = floorf(y), fx = floorf(x);
  const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1;
  const float ly = y - fy, lx = x - fx, hy = 1.0f - ly, hx = 1.0f - lx;
  const float wr0 = (in && y0 >= 0) ? hy * m : 0.0f;       // row y0 valid
  const float wr1 = (in && y1 <= H - 1) ? ly * m : 0.0f;   // row y1 valid
  float wx0, wx1;                                           // weights of the pair's two elements
  if (x0 < 0) { wx0 = lx; wx1 = 0.0f; }                    // only column 0 (the right corner) exists
  else if (x0 > W - 2) { wx0 = 0.0f; wx1 = hx; }           // only column W-1 (the left corner) exists
  else { wx0 = hx; wx1 = lx; }
  const int cx = min(max(x0, 0), W - 2);
  t.o[0] = max(y0, 0) * W + cx;
  t.o[1] = min(y1, H - 1) * W + cx;
  t.w[0] = wr0 * wx0; t.w[1] = wr0 * wx1; t.w[2] = wr1 * wx0; t.w[3] = wr1 * wx1;
}

// sampling position of tap t for output pixel (oy, ox) of image b; deformable group dgi
__device__ __forceinline__ void tap_position(const DcnProblem &p, int b, int dgi, int t, int hw, int oy,
                                             int ox, float &y, float &x, float &m) {
  const long long obase = ((long long)(b * p.DG + dgi) * 2 * p.K + 2 * t) * p.HoWo + hw;
  const float off_y = p.offset[obase];
  const float off_x = p.offset[obase + p.HoWo];
  const int i = t / p.kw, j = t - i * p.kw;
  y = (float)(oy * p.sh - p.ph + i * p.dh) + off_y;
  x = (float)(ox * p.sw - p.pw + j * p.dw) + off_x;
  m = p.mask ? p.mask[((long long)(b * p.DG + dgi) * p.K + t) * p.HoWo + hw] : 1.0f;
}

// 2x2 MFMA 32x32x2 steps over one 16-deep LDS stage.  A: [16][lda], B: [16][ldb].
__device__ __forceinline__ void mfma_stage(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                           int ldb, int a_off, int b_off, int lane, f32x16 (&acc)[2][2]) {
  const int kk = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int ks = 0; ks < kChunk / 2; ++ks) {
    const int k = 2 * ks + kk;
    const float a0 = A[k * lda + a_off + l31];
    const float a1 = A[k * lda + a_off + 32 + l31];
    const float b0 = B[k * ldb + b_off + l31];
    const float b1 = B[k * ldb + b_off + 32 + l31];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// row of accumulator register r inside a 32x32 MFMA block (C/D layout, col = lane & 31)
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Write one tile (register image) to out[N, O_total, Ho, Wo], with bias / ReLU fused.
__device__ __forceinline__ void store_output(const DcnProblem &p, int mt, int nt, int tid,
                                             const f32x16 (&acc)[2][2]) {
  const int lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    int b, hw;
    if (!tile_pixel(p, nt, wn * 64 + ni * 32 + (lane & 31), b, hw)) continue;
    float *obase = p.out + ((long long)b * p.O_total + p.o_base) * p.HoWo + hw;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = mt * kTileM + wm * 64 + mi * 32 + mfma_row(r, lane);
        if (o >= p.Og) continue;
        float v = acc[mi][ni][r];
        if (p.bias) v += p.bias[p.bias_base + o];
        if (p.flags & 1u /* KGDET_DCN_RELU */) v = fmaxf(v, 0.0f);
        obase[(long long)o * p.HoWo] = v;
      }
  }
}

// raw register image of a tile <-> slab (coalesced 16 B per lane)
__device__ __forceinline__ void store_slab(float *__restrict__ slab, int tid, const f32x16 (&acc)[2][2]) {
  f32x4 *s4 = reinterpret_cast<f32x4 *>(slab);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
        s4[((mi * 2 + ni) * 4 + q) * kThreads + tid] = v;
      }
}

__device__ __forceinline__ void add_slab(const float *__restrict__ slab, int tid, f32x16 (&acc)[2][2]) {
  const f32x4 *s4 = reinterpret_cast<const f32x4 *>(slab);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = s4[((mi * 2 + ni) * 4 + q) * kThreads + tid];
        acc[mi][ni][4 * q] += v[0];
        acc[mi][ni][4 * q + 1] += v[1];
        acc[mi][ni][4 * q + 2] += v[2];
        acc[mi][ni][4 * q + 3] += v[3];
      }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
}

// ---- layout 1: 8 waves x (32 rows x 128 columns), accumulators acc[ni] = MFMA block (rows wave*32.., cols ni*32..)
__device__ __forceinline__ void store_output_w8(const DcnProblem &p, int mt, int nt, int tid, const f32x16 (&acc)[4]) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    int b, hw;
    if (!tile_pixel(p, nt, ni * 32 + (lane & 31), b, hw)) continue;
    float *obase = p.out + ((long long)b * p.O_total + p.o_base) * p.HoWo + hw;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = mt * kTileM + wave * 32 + mfma_row(r, lane);
      if (o >= p.Og) continue;
      float v = acc[ni][r];
      if (p.bias) v += p.bias[p.bias_base + o];
      if (p.flags & 1u /* KGDET_DCN_RELU */) v = fmaxf(v, 0.0f);
      obase[(long long)o * p.HoWo] = v;
    }
  }
}
__device__ __forceinline__ void store_slab_w8(float *__restrict__ slab, int tid, const f32x16 (&acc)[4]) {
  f32x4 *s4 = reinterpret_cast<f32x4 *>(slab);
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]};
      s4[(ni * 4 + q) * kThreads + tid] = v;     // float4 column j = ni * 4 + q
    }
}
__device__ __forceinline__ void zero_acc_w8(f32x16 (&acc)[4]) {
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.0f;
}

// Which slab slot does workgroup g use for the segment of its range that starts at `seg_begin`?
// slot 0 = the segment its range starts with, slot 1 = any later (necessarily final) one.
__device__ __forceinline__ int slab_slot(long long seg_begin, long long my_begin) {
  return seg_begin == my_begin ? 0 : 1;
}

}  // namespace kgdet
