// Body of the "plane" kernels dcn_fwd_plane / dcn_bwd_input_plane, PHASED version: 8 identical waves (2 per SIMD), every
// wave alternates a SAMPLING phase (bilinear gathers from the LDS plane, hi / lo split, B stages of a whole group of up to
// 8 taps into LDS) and an MFMA phase (its 32 x 128 block of the 256 x 128 tile times the group's stages), one workgroup
// barrier after each.
//
// Why phases and not producer / consumer waves running side by side (dcn_plane.h, the previous design; measured with
// tools/microbench/mfma_valu.hip on gfx950): a SIMD issues MFMAs and ordinary VALU instructions through ONE port.  Beside two
// waves that keep the MFMA pipe full, a third wave gets about one VALU instruction per MFMA slot -- the sampling
// instruction mix ran 12x slower there (3340 instead of 274 cycles per half-stage, whatever its s_setprio) -- and every
// instruction it does get costs the MFMA waves ~10 cycles.  In the side-by-side kernel that was ~1500 cycles per stage
// for 768 cycles of MFMAs (126 VALU instructions per stage and SIMD).  Giving the producers a SIMD of their own leaves
// 8 MFMA waves on 3 SIMDs (3 + 3 + 2, the youngest wave of a 3-wave SIMD arrives last: 1700 cycles per stage,
// measured).  With phases the two instruction classes never meet: the sampling phase runs VALU at 4 cycles per
// instruction on all four SIMDs, the MFMA phase runs the pipes back to back (microbenchmark: 1030 - 1090 cycles per stage
// at 6 - 8 stages per group).
//
// Work split.  Sampling: wave w always samples pixel half (w >> 1) & 1 (64 of the tile's 128 pixel columns), channel
// half w & 1 (8 of the chunk's 16 channels) and, of every pair of stages of the group, stage (w >> 2): a UNIT = one
// ds_read_b128 per bilinear corner and channel quad (8 per lane), 16 packed fp32 FMAs, the split, two ds_write_b128.  Up to
// kPhUnits = 4 units per wave and group, pipelined over two corner register sets.  MFMA: wave w owns rows
// [32 w, 32 w + 32) of the tile (wave layout 1, as before): 12 MFMAs per stage (3 per 32 x 32 block: lo*hi, hi*lo,
// hi*hi), B fragments from LDS, weight fragments straight from the operand image in L2 two stages ahead.
//
// Segments (the stages of one channel chunk: they share the LDS plane) are cut into groups of 8, 8, ..., with the
// remainder arranged so that only the LAST group of a segment can be odd (the weight-fragment ring alternates two
// register sets by stage parity).  The next segment's plane is loaded into registers before the last group's MFMA phase
// and stored to LDS after it (nobody reads the plane then), the tap records of a group are loaded before the MFMA phase
// of the group before: none of the global latencies is exposed except in the first segment of a range.
#pragma once
#include "dcn_plane.h"
#define KGDET_PH_NOCARRY 1

namespace kgdet {
namespace {

constexpr int kPhThreads = 512;        // 8 waves
constexpr int kPhGroup = 8;            // stages per group (B buffer: kPhGroup x PARTS x 4 KB)
constexpr int kPhUnits = kPhGroup / 2; // sampling units per wave and group
constexpr int kPhPlaneRounds = 12;     // (pixel, quad) items a thread carries through an MFMA phase: 12 x 512 >= 4 x 1536

// stages of the next group when `rem` stages of the segment are left: 8 while that leaves at least 4, else 4 (which
// leaves 5..7 for the last group): every group but the last has 4 or 8 stages
__device__ __forceinline__ int ph_group_size(int rem) {
  if (rem <= kPhGroup) return rem;
  return rem - kPhGroup >= 4 ? kPhGroup : 4;
}

}  // namespace

template <int PARTS, int MODE>
__device__ __forceinline__ void plane_phased(const DcnFwdGroup &grp, float *__restrict__ slabs, unsigned char *smem) {
  unsigned char *Bs = smem;                                      // [kPhGroup][PARTS][kBPart]
  unsigned char *plane = smem + kPhGroup * PARTS * kBPart;       // [pixels][16 ch] fp32

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int chhalf = wave & 1, srow = wave >> 2;
  const int col = ((wave >> 1) & 1) * 64 + lane;      // pixel column of the tile this lane samples
  const long long G = gridDim.x, g = blockIdx.x;
  const long long slice = sk_slice_of_block((int)g, (int)G);
  long long my_begin, my_end;

#ifdef KGDET_PLANE_TRACE
  unsigned long long tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_t = KGDET_TR_NOW();
  const unsigned long long tr_start = tr_t;
#endif
  int slot = 0;  // slabs written so far (one per range met)
  for (int round = 0; round < (grp.static_ranges ? grp.rounds : 1); ++round) {   // (static schedule: one range per round)
  dcn_slice_bounds(grp, slice, G, my_begin, my_end, round);
  long long cur = my_begin;
  while (cur < my_end) {
    const DcnUnitPos pos = dcn_unit_pos(grp, cur);
    const DcnProblem &p = grp.p[pos.pi];
    const int HW = p.H * p.W;
    const int K = p.K;
    const int n_c16 = p.chunks_per_tap;
    const int cpt = p.chunks_per_tile;
    const int tile = pos.tile;
    const int s_begin = pos.s;
    const int s_end = (int)((my_end - cur) < (long long)(pos.s_hi - pos.s) ? pos.s + (my_end - cur) : pos.s_hi);
    const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
    const int tile_b = nt / p.tiles_per_image;  // image of the tile
    const int tile_in_img = nt - tile_b * p.tiles_per_image;
    const int HoWo = p.HoWo;
    // columns past the end of the image sample pixel 0 again: their results are never stored
    const int hw0 = tile_in_img * kTileN + col;
    const int hw_c = hw0 < HoWo ? hw0 : 0;

    f32x16 acc[4];
    zero_acc_w8(acc);

    typedef PlaneStageRegs<MODE> Regs;
    constexpr int NG = Regs::NG;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef f32x4 Corners[2][4];
    struct AFrag {
      bf16x8 a[PARTS];
    };

    // records of (image, deformable group) for channel chunk c: [K][pixels][NG] groups of 32 B
    auto seg_records = [&](int c) {
      const int dgi = p.DG == 1 ? 0 : (p.c_base + min(c * kChunk, p.Cg - 1)) / p.cpdg;   // (cpdg % 16 == 0)
      return reinterpret_cast<const uint4 *>(p.taps) + (((size_t)(tile_b * p.DG + dgi) * K) * HoWo + hw_c) * (2 * NG);
    };
    // the tap record of tap t (clamped by the caller) of the chunk whose records start at rb
    auto issue = [&](const uint4 *rb, int t, Regs &R) {
      const uint4 *rec = rb + (size_t)t * HoWo * (2 * NG);
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        R.off[gq] = rec[gq];
        R.w[gq] = *reinterpret_cast<const f32x4 *>(rec + NG + gq);
      }
      if constexpr (MODE == 1) {
        const DcnInvOvfSlots *sl = p.inv_ovf + ((size_t)(tile_b * K + t) * p.tiles_per_image + tile_in_img);
        R.ovf = *reinterpret_cast<const int2 *>(sl);   // (count, spill_start)
      }
    };
    // records of this wave's units of the group of taps [tg, tg + gn) (clamped to the chunk's last tap t_last)
    auto issue_group = [&](const uint4 *rb, int tg, int t_last, Regs (&R)[kPhUnits]) {
#pragma unroll
      for (int k = 0; k < kPhUnits; ++k) issue(rb, min(tg + 2 * k + srow, t_last), R[k]);
    };
    // x[tile_b, c_base + 16 c .. +15, :, :] -> LDS [pixel][16 channels] (64 B rows, quad slot ^ ((q >> 2) & 3): see
    // dcn_plane.h), in two halves: loads of (pixel, quad) items into registers (unconditional, clamped), stores later
    const float *xb = p.x + ((long long)tile_b * p.C_total + p.c_base) * HW;
    // round (quad, rr) moves pixel rr * 512 + tid of channel quad `quad`: no division, and the channel plane's base is
    // wave-uniform (scalar base + one 32-bit pixel offset per rr)
    constexpr int kPR = kPhPlaneRounds / 4;   // pixel rounds per quad
    auto plane_loads = [&](int c, f32x4 (&v)[kPhPlaneRounds]) {
      const int c0 = c * kChunk;
#pragma unroll
      for (int quad = 0; quad < 4; ++quad)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float *xc = xb + (long long)min(c0 + quad * 4 + e, p.Cg - 1) * HW;  // padded channels re-read the last real one
#pragma unroll
          for (int rr = 0; rr < kPR; ++rr) v[quad * kPR + rr][e] = xc[min(rr * kPhThreads + tid, HW - 1)];
        }
    };
    auto plane_stores = [&](const f32x4 (&v)[kPhPlaneRounds]) {
#pragma unroll
      for (int rr = 0; rr < kPR; ++rr) {
        const int q = rr * kPhThreads + tid;
        if (q < HW) {
#pragma unroll
          for (int quad = 0; quad < 4; ++quad)
            *reinterpret_cast<f32x4 *>(plane + dcn_plane_offset(q) + ((quad ^ ((q >> 2) & 3)) << 4)) = v[quad * kPR + rr];
        }
      }
    };

    // ---- sampling: corner offsets in the record are for quad 0; quad c of the same pixel is at offset ^ (c << 4)
    auto corner_reads = [&](const Regs &R, int gq, Corners &v) {
      const unsigned o[4] = {R.off[gq].x, R.off[gq].y, R.off[gq].z,
                             (MODE == 1 && gq == NG - 1) ? (R.off[gq].w & 0x1ffffu) : R.off[gq].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          v[c][e] = *reinterpret_cast<const f32x4 *>(plane + (o[e] ^ (unsigned)((chhalf * 2 + c) << 4)));
    };
    auto corner_fma = [&](const Regs &R, int gq, const Corners &v, f32x2 (&sv)[2][2], bool first) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 ve = {v[c][e][2 * h2], v[c][e][2 * h2 + 1]};
            const f32x2 we = {R.w[gq][e], R.w[gq][e]};
            sv[c][h2] = (first && e == 0) ? we * ve : __builtin_elementwise_fma(we, ve, sv[c][h2]);
          }
    };
    auto split_store = [&](int gi, const f32x2 (&sv)[2][2]) {   // 8 channels of stage gi of the group -> hi (and lo) image
      bf16x8 hi, lo;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int q = c * 4 + h2 * 2;
          hi[q] = (__bf16)sv[c][h2][0];
          hi[q + 1] = (__bf16)sv[c][h2][1];
          if constexpr (PARTS == 2) {
            const f32x2 hf = {(float)hi[q], (float)hi[q + 1]};
            const f32x2 lf = sv[c][h2] - hf;
            lo[q] = (__bf16)lf[0];
            lo[q + 1] = (__bf16)lf[1];
          }
        }
      unsigned char *dst = Bs + gi * PARTS * kBPart + chhalf * (kTileN * 16) + col * 16;
      *reinterpret_cast<bf16x8 *>(dst) = hi;
      if constexpr (PARTS == 2) *reinterpret_cast<bf16x8 *>(dst + kBPart) = lo;
    };
    // one unit, start to end (MODE 1: records of 8 contributions + the (tile, tap)'s overflow list)
    auto sample_unit = [&](int gi, const Regs &R) {
      f32x2 sv[2][2];
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        Corners v;
        corner_reads(R, gq, v);
        corner_fma(R, gq, v, sv, gq == 0);
      }
      if constexpr (MODE == 1) {
        // contributions beyond the 8 inline ones (a cell with more than 8 contributing (pixel, corner) pairs for one
        // tap: rare); every thread scans the list (uniform trip count) and adds the entries of its own cell
        const uint2 *spill = p.inv_spill + R.ovf.y;
        for (int i = 0; i < (R.ovf.x < 0 ? -R.ovf.x : R.ovf.x); ++i) {   // (|count|: the list scan serves both encodings)
          const uint2 e = spill[i];
          if ((int)(e.x & 127u) == col) {
            const float w = __uint_as_float(e.y);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const f32x4 v = *reinterpret_cast<const f32x4 *>(plane + ((e.x >> 7) ^ (unsigned)((chhalf * 2 + c) << 4)));
              sv[c][0] += f32x2{w * v[0], w * v[1]};
              sv[c][1] += f32x2{w * v[2], w * v[3]};
            }
          }
        }
      }
      split_store(gi, sv);
    };
    // this wave's units of a group of gn stages: unit k = stage 2 k + srow of the group
    auto sample_group = [&](int gn, const Regs (&R)[kPhUnits]) {
      if constexpr (MODE == 0) {   // the corner reads of a unit are issued before the arithmetic of the one before
        Corners V0, V1;
        f32x2 sv[2][2];
        corner_reads(R[0], 0, V0);
        corner_reads(R[1], 0, V1);   // (records past the end of the group are clamped copies: harmless reads)
        corner_fma(R[0], 0, V0, sv, true);
        if (srow < gn) split_store(srow, sv);
        corner_reads(R[2], 0, V0);
        corner_fma(R[1], 0, V1, sv, true);
        if (2 + srow < gn) split_store(2 + srow, sv);
        corner_reads(R[3], 0, V1);
        corner_fma(R[2], 0, V0, sv, true);
        if (4 + srow < gn) split_store(4 + srow, sv);
        corner_fma(R[3], 0, V1, sv, true);
        if (6 + srow < gn) split_store(6 + srow, sv);
      } else {
#pragma unroll
        for (int k = 0; k < kPhUnits; ++k)
          if (2 * k + srow < gn) sample_unit(2 * k + srow, R[k]);
      }
    };

    int s = s_begin;
    int c16 = s / K;
    int t0 = s - c16 * K;
    Regs R[kPhUnits];      // tap records of the group sampled next
    AFrag F0, F1;          // weight fragments of the next two stages (stage j of the segment uses F[j & 1]; groups start at
                           // even stages)
    bool primed = false;   // plane, first records and first fragments were loaded under the last group of the segment before
    while (s < s_end) {
      const int n = min(K - t0, s_end - s);  // stages of this segment: taps t0 .. t0+n-1 of chunk c16
      const bool has_next = s + n < s_end;
      const uint4 *rec_base = seg_records(c16);
      const int t_last = t0 + n - 1;
      // the wave's A (weight) fragments straight from the weight image (L2), 16 bytes per lane and fragment, coalesced
      // (wave-uniform stage base + one 32-bit lane offset: scalar-base loads, no vector address arithmetic in the MFMA
      // phase -- every VALU instruction there costs ~10 cycles of MFMA time)
      const unsigned a_lane = (unsigned)((lane >> 5) * (kTileM * 16) + (wave * 32 + (lane & 31)) * 16);
      auto a_frag = [&](int c, int t, AFrag &F) {
        const unsigned char *b = reinterpret_cast<const unsigned char *>(p.wq) +
                                 ((size_t)((mt * n_c16 + c) * K) + t) * (2 * kAPart);
#pragma unroll
        for (int part = 0; part < PARTS; ++part) F.a[part] = *reinterpret_cast<const bf16x8 *>(b + part * kAPart + a_lane);
      };
      auto multiply = [&](int gi, const AFrag &F) {
#ifdef KGDET_ABL_NOMFMA
        return;
#endif
        const unsigned char *B = Bs + gi * PARTS * kBPart + (lane >> 5) * (kTileN * 16) + (lane & 31) * 16;
        bf16x8 b[PARTS][4];
#pragma unroll
        for (int part = 0; part < PARTS; ++part)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) b[part][ni] = *reinterpret_cast<const bf16x8 *>(B + part * kBPart + ni * 32 * 16);
        if constexpr (PARTS == 2) {  // small terms first; four independent accumulators per pass
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[1], b[0][ni], acc[ni], 0, 0, 0);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0], b[1][ni], acc[ni], 0, 0, 0);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0], b[0][ni], acc[ni], 0, 0, 0);
      };

#ifdef KGDET_PLANE_TRACE
      tr_t = KGDET_TR_NOW();
      tr[9] += 1; tr[8] += n;
#endif
      if (!primed) {   // first segment of the range (every group ends with a barrier: B and the plane are free)
        issue_group(rec_base, t0, t_last, R);
        a_frag(c16, t0, F0);
        a_frag(c16, min(t0 + 1, t_last), F1);
        f32x4 pv[kPhPlaneRounds];
        plane_loads(c16, pv);
        plane_stores(pv);
        KGDET_TR_ADD(1, tr_t);
        __syncthreads();
        KGDET_TR_ADD(2, tr_t);
      }
      for (int j0 = 0; j0 < n;) {
        const int gn = ph_group_size(n - j0);
        const bool last = j0 + gn == n;
        // ---- sampling phase
#ifndef KGDET_ABL_NOSAMPLE
        sample_group(gn, R);
#endif
        // the loads that land under the MFMA phase are issued HERE, with their address arithmetic: the next group's
        // records, or the next segment's first records and plane
        f32x4 pv[kPhPlaneRounds];
        if (!last) {
          issue_group(rec_base, t0 + j0 + gn, t_last, R);
        } else if (has_next) {
          const int n2 = min(K, s_end - (s + n));
          issue_group(seg_records(c16 + 1), 0, n2 - 1, R);
#ifndef KGDET_PH_NOCARRY
          plane_loads(c16 + 1, pv);
#endif
        }
        KGDET_TR_ADD(3, tr_t);
        __syncthreads();
        KGDET_TR_ADD(0, tr_t);
        // ---- MFMA phase
        const int tj = t0 + j0;
        for (int i = 0; i + 1 < gn; i += 2) {
          multiply(i, F0);
          a_frag(c16, min(tj + i + 2, t_last), F0);
          multiply(i + 1, F1);
          a_frag(c16, min(tj + i + 3, t_last), F1);
        }
        if (gn & 1) multiply(gn - 1, F0);   // (only the last group of a segment is odd)
        if (last && has_next) {
          const int n2 = min(K, s_end - (s + n));
          a_frag(c16 + 1, 0, F0);
          a_frag(c16 + 1, min(1, n2 - 1), F1);
#ifdef KGDET_PH_NOCARRY
          plane_loads(c16 + 1, pv);
#endif
          plane_stores(pv);
        }
        KGDET_TR_ADD(4, tr_t);
        __syncthreads();
        KGDET_TR_ADD(5, tr_t);
        j0 += gn;
      }
      primed = has_next;
      s += n;
      ++c16;
      t0 = 0;
    }

    if (s_begin == 0 && s_end == cpt) {
      store_output_w8(p, mt, nt, tid, acc);
    } else {
      float *slab = slabs + ((long long)g * grp.slots + slot) * kTileElems;
      store_slab_w8(slab, tid, acc);
    }
    KGDET_TR_ADD(6, tr_t);
    ++slot;
    cur += s_end - s_begin;
  }
  }
#ifdef KGDET_PLANE_TRACE
  tr[7] = KGDET_TR_NOW() - tr_start;
  if (lane == 0 && (wave == 0 || wave == 7)) {
#pragma unroll
    for (int c = 0; c < 10; ++c) g_plane_trace[((int)blockIdx.x * 2 + (wave == 7 ? 1 : 0)) * 10 + c] = tr[c];
  }
#endif
}

}  // namespace kgdet
