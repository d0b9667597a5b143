// Body of dcn_fwd_plane_pairs: the forward plane kernel for K >= 7 taps with HALF-CHUNK planes and TAP-PAIR stages.
//
// Why.  In dcn_plane.h a stage is (16-channel chunk, one tap) and the 16-channel plane of a chunk fills the LDS next to the
// operand stages, so the plane can only be replaced once its last gather is done: at every segment switch the pipeline
// drains -- first group sampled with the consumers idle, then the consumers run dry for one producer group time, ~10 k
// cycles against ~1.2 k per stage (16 switches per range for a 3x3 problem).  Here a stage is
// (8-channel half-chunk, a PAIR of taps): the two k-halves of the MFMA reduction are two taps of the same 8 channels
// instead of two channel halves of one tap.  The plane of a segment is 8 channels (32 B per pixel, 33.6 KB at 25 x 42),
// THREE of them fit (ring), the next segment's plane is loaded while the current one is still being sampled, and the
// stage stream of a range never stops at a segment boundary: groups of three stages run straight across.
//   * weights: no new image -- the [khalf][o][8] block of (chunk c16, tap t, k-half) in the forward image `wq` is
//     exactly the block of (half-chunk 2 c16 + khalf, tap t); a consumer lane of k-half h reads tap 2u + h.
//   * odd K: the last pair of a segment has one tap; the producers store zeros for the missing half.
//   * tap records: as dcn_build_taps writes them for this layout (grp.pair_mode: offsets into 32-byte rows).
//   * schedule, slabs, fix-up: as dcn_plane.h (static ranges / rounds / stream-K; wave layout 1).
// Units: a problem's reduction has chunks_per_tap = number of HALF-chunks and seg_stages = ceil(K / 2) stages per
// half-chunk (set by the launcher).
#pragma once
#include "dcn_plane.h"

namespace kgdet {
namespace {

constexpr int kPairGroup = 3;        // stages between two workgroup barriers
constexpr int kPairPlanes = 3;       // half-plane ring
constexpr int kPairRow = 32;         // bytes per pixel of a half-plane: 8 channels fp32
constexpr int kPairCopy = 4;         // (pixel, quad) items a thread carries through an iteration: 2 quads x 2 x 768 pixels

}  // namespace

template <int PARTS, bool PRODUCER>
__device__ __forceinline__ void pair_role(const DcnFwdGroup &grp, float *__restrict__ slabs, unsigned char *smem) {
  unsigned char *Bs = smem;                                             // [2 groups][kPairGroup][PARTS][kBPart]
  unsigned char *planes = smem + 2 * kPairGroup * PARTS * kBPart;       // [kPairPlanes][plane_bytes]: [pixel][8 ch] fp32
  const int plane_stride = grp.plane_bytes;

  const int wtid = threadIdx.x;                               // 0 .. 767
  const int tid = PRODUCER ? wtid - kThreads : wtid;          // position inside the role
  const int lane = tid & 63, wave = tid >> 6;
  const int n_local = tid & (kTileN - 1);                     // producers: pixel column sampled
  const int pair = __builtin_amdgcn_readfirstlane((tid >> 7) & 1);   // producer wave pair: half-stages 3 pair .. 3 pair + 2 of a group
  const long long G = gridDim.x, g = blockIdx.x;
  const long long slice = sk_slice_of_block((int)g, (int)G);
  if constexpr (PRODUCER) __builtin_amdgcn_s_setprio(KGDET_PLANE_PRODUCER_PRIO);

  typedef PlaneStageRegs<0> Regs;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef f32x4 Corners[2][4];
  struct AFrag {
    bf16x8 a[PARTS];
  };

#ifdef KGDET_PLANE_TRACE
  unsigned long long tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_t = KGDET_TR_NOW();
  const unsigned long long tr_start = tr_t;
#endif
  int slot = 0;  // slabs written so far (one per range met)
  for (int round = 0; round < (grp.static_ranges ? grp.rounds : 1); ++round) {
    long long my_begin, my_end;
    dcn_slice_bounds(grp, slice, G, my_begin, my_end, round);
    long long cur = my_begin;
    while (cur < my_end) {
      const DcnUnitPos pos = dcn_unit_pos(grp, cur);
      const DcnProblem &p = grp.p[pos.pi];
      const int HW = p.H * p.W, HoWo = p.HoWo;
      const int K = p.K, Ks = p.seg_stages;
      const int n_c16 = p.chunks_per_tap >> 1;
      const int cpt = p.chunks_per_tile;
      const int tile = pos.tile;
      const int s_begin = pos.s;
      const int s_end = (int)((my_end - cur) < (long long)(pos.s_hi - pos.s) ? pos.s + (my_end - cur) : pos.s_hi);
      const int n_total = s_end - s_begin;
      const int n_groups = (n_total + kPairGroup - 1) / kPairGroup;
      const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
      const int tile_b = nt / p.tiles_per_image;
      const int hw0 = (nt - tile_b * p.tiles_per_image) * kTileN + n_local;
      const int hw_c = hw0 < HoWo ? hw0 : 0;   // columns past the end of the image sample pixel 0 again: never stored
      const int c8_0 = s_begin / Ks, u_0 = s_begin - c8_0 * Ks;   // coordinates of the range's first stage
      const int c8_last = (s_end - 1) / Ks;

      f32x16 acc[PRODUCER ? 1 : 4];
      if constexpr (!PRODUCER) zero_acc_w8(acc);

      // (c8, u) of stage j0 + d (d = 0 .. 4) given those of stage j0: at most one wrap (Ks >= 4); stages past the end of
      // the range are clamped to the last half-chunk's last pair (their results are never used)
      auto coords_at = [&](int c8g, int ug, int d, int &c8, int &u) {
        u = ug + d;
        c8 = c8g;
        if (u >= Ks) { u -= Ks; ++c8; }
        if (c8 > c8_last) { c8 = c8_last; u = Ks - 1; }
      };
      // ---- half-plane copy: x[tile_b, c_base + 8 c8 .. +7] -> ring buffer `buf` as [pixel][8 ch]; the two 16-byte quads
      // of pixel q sit at slot (quad ^ ((q >> 3) & 1)): 16 consecutive pixels of one quad then cover all 16 four-bank
      // groups.  Loads at the start of an iteration, stores at its end (the data arrived long before).
      const float *xb = p.x + ((long long)tile_b * p.C_total + p.c_base) * HW;
      auto copy_issue = [&](int c8, f32x4 (&v)[kPairCopy]) {
#pragma unroll
        for (int quad = 0; quad < 2; ++quad)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float *xc = xb + (long long)min(c8 * 8 + quad * 4 + e, p.Cg - 1) * HW;   // padded channels: the last real one
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) v[quad * 2 + rr][e] = xc[min(rr * kPlaneThreads + wtid, HW - 1)];
          }
      };
      auto copy_commit = [&](int buf, const f32x4 (&v)[kPairCopy]) {
        unsigned char *pl = planes + buf * plane_stride;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int q = rr * kPlaneThreads + wtid;
          if (q < HW) {
#pragma unroll
            for (int quad = 0; quad < 2; ++quad)
              *reinterpret_cast<f32x4 *>(pl + q * kPairRow + ((quad ^ ((q >> 3) & 1)) << 4)) = v[quad * 2 + rr];
          }
        }
      };
      // ---- tap record of (half-chunk c8, tap t) for this thread's pixel
      // (wave-uniform record-table base + one 32-bit lane offset: scalar-base loads, no vector address arithmetic)
      const unsigned rec_lane = (unsigned)hw_c * 32u;
      auto issue = [&](int c8, int t, Regs &R) {
        const int dgi = p.DG == 1 ? 0 : (p.c_base + min(c8 * 8, p.Cg - 1)) / p.cpdg;   // (cpdg % 16 == 0)
        const unsigned char *rec = reinterpret_cast<const unsigned char *>(p.taps) +
                                   (((size_t)(tile_b * p.DG + dgi) * K) + (size_t)min(t, K - 1)) * HoWo * 32;
        R.off[0] = *reinterpret_cast<const uint4 *>(rec + rec_lane);
        R.w[0] = *reinterpret_cast<const f32x4 *>(rec + rec_lane + 16);
      };
      // records of this wave pair's three half-stages of the group whose first stage has coordinates (c8g, ug)
      auto issue_group = [&](int c8g, int ug, Regs (&R)[3]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int hs = 3 * pair + k;
          int c8, u;
          coords_at(c8g, ug, hs >> 1, c8, u);
          issue(c8, 2 * u + (hs & 1), R[k]);
        }
      };
      // ---- sampling
      auto corner_reads = [&](const Regs &R, const unsigned char *pl, Corners &v) {
        const unsigned o[4] = {R.off[0].x, R.off[0].y, R.off[0].z, R.off[0].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < 2; ++c) v[c][e] = *reinterpret_cast<const f32x4 *>(pl + (o[e] ^ (unsigned)(c << 4)));
      };
      auto corner_fma = [&](const Regs &R, const Corners &v, f32x2 (&sv)[2][2]) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 ve = {v[c][e][2 * h2], v[c][e][2 * h2 + 1]};
              const f32x2 we = {R.w[0][e], R.w[0][e]};
              sv[c][h2] = e == 0 ? we * ve : __builtin_elementwise_fma(we, ve, sv[c][h2]);
            }
      };
      auto split_store = [&](int buf, int gi, int half, const f32x2 (&sv)[2][2], bool zero) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 hi_u, lo_u;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = c * 2 + h2;
            const f32x2 val = zero ? f32x2{0.f, 0.f} : sv[c][h2];
            const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(val, bf16x2));
            hi_u[q] = hu;
            if constexpr (PARTS == 2) {
              const f32x2 hf = {__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
              lo_u[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(val - hf, bf16x2));
            }
          }
        unsigned char *dst = Bs + (buf * kPairGroup + gi) * PARTS * kBPart + half * (kTileN * 16) + n_local * 16;
        *reinterpret_cast<u32x4 *>(dst) = hi_u;
        if constexpr (PARTS == 2) *reinterpret_cast<u32x4 *>(dst + kBPart) = lo_u;
      };
      // this wave pair's three half-stages of the group whose first stage is j0, into group buffer `buf`
      auto sample_group = [&](int j0, int c8g, int ug, int buf, const Regs (&R)[3]) {
        const unsigned char *pl[3];
        bool live[3], zero[3];
        int gi[3], hh[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int hs = 3 * pair + k;
          gi[k] = hs >> 1;
          hh[k] = hs & 1;
          int c8, u;
          coords_at(c8g, ug, gi[k], c8, u);
          live[k] = j0 + gi[k] < n_total;
          zero[k] = 2 * u + hh[k] >= K;          // the missing tap of an odd K's last pair
          pl[k] = planes + ((c8 - c8_0) % kPairPlanes) * plane_stride;
        }
        Corners V0, V1;
        f32x2 sv[2][2];
        corner_reads(R[0], pl[0], V0);
        corner_reads(R[1], pl[1], V1);   // (clamped records past the end of the range: harmless reads)
        corner_fma(R[0], V0, sv);
        if (live[0]) split_store(buf, gi[0], hh[0], sv, zero[0]);
        corner_reads(R[2], pl[2], V0);
        corner_fma(R[1], V1, sv);
        if (live[1]) split_store(buf, gi[1], hh[1], sv, zero[1]);
        corner_fma(R[2], V0, sv);
        if (live[2]) split_store(buf, gi[2], hh[2], sv, zero[2]);
      };
      // ---- consumers: weight fragments of stage j straight from the forward image: k-half h (lanes 32 h ..) = tap 2u + h
      // of half-chunk c8 = the (c8 & 1) k-half block of chunk c8 >> 1
      const unsigned a_lane = (unsigned)((wave * 32 + (lane & 31)) * 16);
      const unsigned a_lane_live = a_lane + (unsigned)(lane >> 5) * (unsigned)(2 * kAPart);
      auto a_issue = [&](int c8g, int ug, int d, AFrag &F) {   // stage (c8g, ug) + d
        int c8, u;
        coords_at(c8g, ug, d, c8, u);
        const bool dead = 2 * u + 1 >= K;     // no second tap (its B half is zero): both k-halves read tap 2u
        const unsigned char *b = reinterpret_cast<const unsigned char *>(p.wq) +
                                 ((size_t)((mt * n_c16 + (c8 >> 1)) * K) + 2 * u) * (2 * kAPart) + (c8 & 1) * (kTileM * 16);
        const unsigned off = dead ? a_lane : a_lane_live;
#pragma unroll
        for (int part = 0; part < PARTS; ++part) F.a[part] = *reinterpret_cast<const bf16x8 *>(b + part * kAPart + off);
      };
      auto multiply = [&](int buf, int gi, const AFrag &F) {
        if constexpr (!PRODUCER) {
          const unsigned char *B = Bs + (buf * kPairGroup + gi) * PARTS * kBPart + (lane >> 5) * (kTileN * 16) + (lane & 31) * 16;
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
        }
      };

      // ---- prologue: the planes of the segments that start within the first two groups (at most three), the first
      // records and fragments, group 0
      Regs RE[3], RO[3];     // producers: records of the even / odd groups (loaded two groups ahead)
      AFrag F0, F1;          // consumers: fragments of the next two stages (stage j uses F[j & 1])
#ifdef KGDET_PLANE_TRACE
      tr_t = KGDET_TR_NOW();
      tr[9] += 1; tr[8] += n_total;
#endif
      int segs_loaded = 0;                 // segments whose plane is (being) loaded; segment m sits in buffer m % 3
      int seg_next_start = 0;              // stage index (in the range) at which segment `segs_loaded` starts
      // coordinates of the first stage of the group the consumers multiply / the producers sample in the CURRENT iteration
      int c8_i = c8_0, u_i = u_0;
      auto advance = [&](int &c8, int &u) {   // by one group
        u += kPairGroup;
        if (u >= Ks) { u -= Ks; ++c8; }
      };
      if constexpr (PRODUCER) {
        int c8n = c8_0, un = u_0;
        issue_group(c8n, un, RE);
        advance(c8n, un);
        issue_group(c8n, un, RO);
      } else {
        a_issue(c8_0, u_0, 0, F0);
        a_issue(c8_0, u_0, 1, F1);
      }
      while (seg_next_start < n_total && seg_next_start <= 2 * kPairGroup - 1) {
        f32x4 v[kPairCopy];
        copy_issue(c8_0 + segs_loaded, v);
        copy_commit(segs_loaded % kPairPlanes, v);
        seg_next_start = (segs_loaded + 1) * Ks - u_0;
        ++segs_loaded;
      }
      KGDET_TR_ADD(1, tr_t);
      __syncthreads();
      KGDET_TR_ADD(2, tr_t);
      if constexpr (PRODUCER) sample_group(0, c8_0, u_0, 0, RE);
      __syncthreads();
      KGDET_TR_ADD(3, tr_t);

      // iteration i: the consumers multiply group i, the producers sample group i + 1 (records of group i + 2 loaded
      // into the set group i used), everybody carries a share of the plane of the segment that first appears in group i + 2
      auto iteration = [&](int i, auto BUF) {
        constexpr int buf = decltype(BUF)::value;
        const int j0 = i * kPairGroup;
        f32x4 cv[kPairCopy];
        const bool copy = seg_next_start < n_total && seg_next_start <= j0 + 3 * kPairGroup - 1;
        // (the loads are issued in every iteration, from a clamped half-chunk: loads under a condition make the number in
        // flight path-dependent and hipcc then waits vmcnt(0) before every use of a weight fragment)
        copy_issue(min(c8_0 + segs_loaded, c8_last), cv);
        if constexpr (PRODUCER) {
          if (i + 1 < n_groups) {
            int c8a = c8_i, ua = u_i;      // (c8_i, u_i): group i
            advance(c8a, ua);              // group i + 1: sampled now
            int c8b = c8a, ub = ua;
            advance(c8b, ub);              // group i + 2: its records are loaded now
            if constexpr (buf == 0) {
              issue_group(c8b, ub, RE);     // (RE was group i's set: sampled in iteration i - 1)
              sample_group(j0 + kPairGroup, c8a, ua, 1, RO);
            } else {
              issue_group(c8b, ub, RO);
              sample_group(j0 + kPairGroup, c8a, ua, 0, RE);
            }
          }
        } else {
          AFrag &Fa = buf ? F1 : F0, &Fb = buf ? F0 : F1;   // stage j uses F[j & 1]; group i starts at stage 3 i
          multiply(buf, 0, Fa);
          a_issue(c8_i, u_i, 2, Fa);
          __builtin_amdgcn_sched_barrier(0);
          if (j0 + 1 < n_total) multiply(buf, 1, Fb);
          a_issue(c8_i, u_i, 3, Fb);
          __builtin_amdgcn_sched_barrier(0);
          if (j0 + 2 < n_total) multiply(buf, 2, Fa);
          a_issue(c8_i, u_i, 4, Fa);
        }
        advance(c8_i, u_i);
        if (copy) {
          copy_commit(segs_loaded % kPairPlanes, cv);
          seg_next_start = (segs_loaded + 1) * Ks - u_0;
          ++segs_loaded;
        }
        KGDET_TR_ADD(4, tr_t);
        __syncthreads();
        KGDET_TR_ADD(5, tr_t);
      };
      for (int i = 0; i < n_groups; i += 2) {
        iteration(i, std::integral_constant<int, 0>{});
        if (i + 1 < n_groups) iteration(i + 1, std::integral_constant<int, 1>{});
      }

      if constexpr (!PRODUCER) {
        if (s_begin == 0 && s_end == cpt) {
          store_output_w8(p, mt, nt, tid, acc);
        } else {
          float *slab = slabs + ((long long)g * grp.slots + slot) * kTileElems;
          store_slab_w8(slab, tid, acc);
        }
      }
      KGDET_TR_ADD(6, tr_t);
      ++slot;
      cur += s_end - s_begin;
    }
  }
#ifdef KGDET_PLANE_TRACE
  tr[7] = KGDET_TR_NOW() - tr_start;
  if ((tid & 63) == 0 && (tid >> 6) == 0) {
#pragma unroll
    for (int c = 0; c < 10; ++c) g_plane_trace[((int)blockIdx.x * 16 + (PRODUCER ? 8 : 0)) * 10 + c] = tr[c];
  }
#endif
}

}  // namespace kgdet
