// Shared body of the "plane" kernels (dcn_forward_plane.hip, dcn_backward_plane.hip): LDS-resident feature
// plane, producer / consumer waves, bf16 hi/lo split MFMA, stream-K over (tile, stage) units.
// See dcn_forward_plane.hip for the design notes.
#pragma once
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kAPart = 2 * kTileM * 8 * 2;  // bytes of one part of an A stage
constexpr int kBPart = 2 * kTileN * 8 * 2;
constexpr int kProducers = 256;                          // 4 producer waves
constexpr int kPlaneThreads = kThreads + kProducers;     // 8 consumer + 4 producer waves
constexpr int kPlaneRounds = 3;                          // (pixel, quad) items a thread has in flight while copying a plane

// A (weight) fragments: true = every consumer wave loads its own 64 rows of the stage straight from the weight image
// (L2) into a two-stage register ring; false = the producers copy the stage into LDS and the consumers read it there.
#ifndef KGDET_PLANE_A_FROM_L2
#define KGDET_PLANE_A_FROM_L2 1
#endif
constexpr bool kAFromL2 = KGDET_PLANE_A_FROM_L2 != 0;
// Ablation switches for experiment builds (make VARIANT=... EXTRA=-DKGDET_ABL_...; results are WRONG by design):
//   KGDET_ABL_NOSAMPLE  producers skip the corner reads and the interpolation (B stage = garbage): consumer-bound time
//   KGDET_ABL_NOMFMA    consumers skip the B-fragment reads and the MFMAs: producer-bound time
//   KGDET_ABL_NOALOAD   consumers do not load their weight fragments (one load in the prologue): L2 / fabric share

constexpr int kOvfCap = DcnInvOvfSlots::kCap;   // overflow entries of one (tile, tap) staged through LDS

// MODE of the plane kernels
//   0  forward:     B stage = bilinear samples of x; 4 corners per (pixel, tap) from a DcnTapRec
//   1  grad_input:  B stage = transposed sampling of grad_output: input cell q collects, for tap t, every
//                   (output pixel, weight) pair whose corner is q -- the first 8 from a DcnInvRec, the rest from
//                   the (tile, tap)'s overflow list
template <int MODE>
struct PlaneModeTraits {
  static constexpr int kGroups = MODE == 1 ? 2 : 1;  // groups of 4 (pixel, weight) entries in the record
};

// what a producer thread has in flight for one stage
template <int PARTS, int MODE>
struct PlaneStageRegs {
  static constexpr int NG = PlaneModeTraits<MODE>::kGroups;
  f32x4 a[PARTS][2];  // this thread's 2 x 16 B of each part of the weight stage
  uint4 off[NG];      // LDS byte offsets of the sampled pixels (quad 0)
  f32x4 w[NG];        // and their weights
  uint2 ovf;          // MODE 1: overflow slot (tid & 31) of the stage's (tile, tap)
  int n_ovf;          //         and the list's length
};

}  // namespace

// Loop structure: a workgroup walks its stream-K slice range by range; inside a range the stages that share
// a feature plane (one channel chunk, consecutive taps) form a SEGMENT.  A segment starts with the plane copy
// and the priming of the producers' three-deep register pipeline (weight stage + tap record of stages
// j, j+1, j+2); its steady state is branch-free as far as vector-memory instructions go -- every body issues
// the loads of stage j+3 unconditionally (clamped to the last stage), so hipcc's counted s_waitcnt vmcnt(N)
// stay exact and a load has two full stages to land.  Stage coordinates are carried incrementally; there is no
// integer division in the loop (a runtime s / K costs ~35 dependent SALU ops).
// The two roles are two instantiations of this function (same loop structure, same barriers), so the
// accumulators exist only in the consumers' register allocation.
template <int PARTS, bool PRODUCER, int MODE>
__device__ __forceinline__ void plane_role(const DcnFwdGroup &grp, float *__restrict__ slabs, unsigned char *smem) {
  unsigned char *As = smem;                                                          // [2][PARTS][kAPart]
  unsigned char *Bs = smem + 2 * PARTS * kAPart;                                     // [2][PARTS][kBPart]
  uint2 *ovf_lds = reinterpret_cast<uint2 *>(smem + 2 * PARTS * (kAPart + kBPart));  // [2][kOvfCap + 1] (MODE 1)
  unsigned char *plane = smem + 2 * PARTS * (kAPart + kBPart) + (MODE == 1 ? 2 * (kOvfCap + 1) * 8 : 0);  // [pixels][16 ch] fp32

  const int wtid = threadIdx.x;                               // 0 .. 767 (plane copy)
  const int tid = PRODUCER ? wtid - kThreads : wtid;          // position inside the role
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;                    // consumers: 64 x 64 block of the tile
  const int n_local = tid & (kTileN - 1);                     // producers: pixel column sampled
  const int half = (tid >> 7) & 1;                            //            and which 8 of the chunk's 16 channels
  const long long G = gridDim.x, g = blockIdx.x;
  const long long total = grp.unit_begin[grp.n];
  const long long slice = sk_slice_of_block((int)g, (int)G);
  const long long my_begin = unit_begin(slice, total, G);
  const long long my_end = unit_begin(slice + 1, total, G);

  long long cur = my_begin;
  int slot = 0;  // slabs written so far (one per range met)
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
    // columns past the end of the image sample pixel 0 again: their results are never stored
    const int hw0 = (nt - tile_b * p.tiles_per_image) * kTileN + n_local;
    const int hw_c = hw0 < p.HoWo ? hw0 : 0;
    const int HoWo = p.HoWo;

    f32x16 acc[PRODUCER ? 1 : 2][PRODUCER ? 1 : 2];  // consumers only
    if constexpr (!PRODUCER) zero_acc(acc);

    typedef PlaneStageRegs<PARTS, MODE> Regs;
    constexpr int NG = Regs::NG;

    int s = s_begin;
    int c16 = s / K;
    int t0 = s - c16 * K;
    while (s < s_end) {
      const int n = min(K - t0, s_end - s);  // stages of this segment: taps t0 .. t0+n-1 of chunk c16
      const int dgi = p.DG == 1 ? 0 : (p.c_base + min(c16 * kChunk + half * 8, p.Cg - 1)) / p.cpdg;
      // records of (image, deformable group): [K][pixels][NG] groups of 32 B
      const uint4 *rec_base = reinterpret_cast<const uint4 *>(p.taps) +
                              (((size_t)(tile_b * p.DG + dgi) * K) * HoWo + hw_c) * (2 * NG);
      const int tile_in_img = nt - tile_b * p.tiles_per_image;
      const unsigned char *wq_base = reinterpret_cast<const unsigned char *>(p.wq) +
                                     (size_t)((mt * n_c16 + c16) * K) * (2 * kAPart) + tid * 16;

      auto issue = [&](int j, Regs &R) {  // loads of stage j (clamped): weight stage + tap record
        const unsigned t = (unsigned)(t0 + min(j, n - 1));
        const uint4 *rec = rec_base + (size_t)t * HoWo * (2 * NG);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          R.off[gq] = rec[gq];
          R.w[gq] = *reinterpret_cast<const f32x4 *>(rec + NG + gq);
        }
        if constexpr (MODE == 1) {
          const DcnInvOvfSlots *sl = p.inv_ovf + ((size_t)(tile_b * K + t) * p.tiles_per_image + tile_in_img);
          R.n_ovf = sl->count;
          R.ovf = sl->e[tid & (kOvfCap - 1)];
        }
        if constexpr (!kAFromL2) {
#pragma unroll
          for (int part = 0; part < PARTS; ++part)
#pragma unroll
            for (int r = 0; r < 2; ++r)
              R.a[part][r] = *reinterpret_cast<const f32x4 *>(wq_base + (size_t)t * (2 * kAPart) + part * kAPart +
                                                              r * (kProducers * 16));
        }
      };
      // consumers (kAFromL2): the wave's A fragments of stage j, 16 bytes per lane and fragment, coalesced
      struct AFrag {
        bf16x8 a[PARTS][2];
      };
      const unsigned char *wq_cons = reinterpret_cast<const unsigned char *>(p.wq) +
                                     (size_t)((mt * n_c16 + c16) * K) * (2 * kAPart) + (lane >> 5) * (kTileM * 16) +
                                     (wm * 64 + (lane & 31)) * 16;
      auto a_issue = [&](int j, AFrag &F) {
#ifdef KGDET_ABL_NOALOAD
        if (j > 1) return;
#endif
        const unsigned t = (unsigned)(t0 + min(j, n - 1));
        const unsigned char *b = wq_cons + (size_t)t * (2 * kAPart);
#pragma unroll
        for (int part = 0; part < PARTS; ++part)
#pragma unroll
          for (int i = 0; i < 2; ++i) F.a[part][i] = *reinterpret_cast<const bf16x8 *>(b + part * kAPart + i * 32 * 16);
      };
      auto commit_ovf = [&](int slot, const Regs &R) {  // MODE 1: the stage's overflow slots -> LDS
        if constexpr (MODE == 1) {
          if (tid < kOvfCap) ovf_lds[slot * (kOvfCap + 1) + tid] = R.ovf;
          if (tid == 0) ovf_lds[slot * (kOvfCap + 1) + kOvfCap] = make_uint2((unsigned)R.n_ovf, 0u);
        }
      };
      auto commit_weights = [&](int buf, const Regs &R) {
        if constexpr (kAFromL2) return;
#pragma unroll
        for (int part = 0; part < PARTS; ++part)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            *reinterpret_cast<f32x4 *>(As + (buf * PARTS + part) * kAPart + tid * 16 + r * (kProducers * 16)) =
                R.a[part][r];
      };
      // Copy x[tile_b, c_base + 16*c16 .. +15, :, :] into LDS as [pixel][16 channels] (64 B rows).  The four
      // 16-byte channel quads of pixel q sit at slot (quad ^ ((q >> 2) & 3)): with the row start (q & 3) * 16
      // banks this spreads any 16 consecutive pixels of one quad over all 16 four-bank groups (ds_read_b128 /
      // ds_write_b128 serve 16 / 8 lanes per LDS cycle) instead of the 4 groups of a plain row-major image.
      // A thread moves (pixel, quad) items: 4 coalesced dword loads (one per channel plane) -> one 16-byte
      // LDS store.  All loads are issued before the first store, unconditionally from clamped addresses
      // (a guarded load makes hipcc branch and drain the queue).
      auto load_plane = [&]() {
        const int c0 = c16 * kChunk;
        const float *xb = p.x + ((long long)tile_b * p.C_total + p.c_base) * HW;
        const int items = 4 * HW;  // (pixel, quad) pairs
        for (int i0 = 0; i0 < items; i0 += kPlaneRounds * kPlaneThreads) {
          f32x4 v[kPlaneRounds];
#pragma unroll
          for (int r = 0; r < kPlaneRounds; ++r) {
            const int i = min(i0 + r * kPlaneThreads + wtid, items - 1);
            const int q = i % HW, quad = i / HW;  // consecutive threads -> consecutive pixels
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ch = min(c0 + quad * 4 + e, p.Cg - 1);  // padded channels re-read the last real one
              v[r][e] = xb[(long long)ch * HW + q];
            }
          }
#pragma unroll
          for (int r = 0; r < kPlaneRounds; ++r) {
            const int i = i0 + r * kPlaneThreads + wtid;
            if (i < items) {
              const int q = i % HW, quad = i / HW;
              *reinterpret_cast<f32x4 *>(plane + dcn_plane_offset(q) + ((quad ^ ((q >> 2) & 3)) << 4)) = v[r];
            }
          }
        }
      };
      // B stage: sample this thread's 8 channels of its pixel at the record's four corners, split, store.
      // Corner offsets in the record are for quad 0; quad c of the same pixel is at offset ^ (c << 4).
      int ovf_tap = 0;  // tap of the stage being sampled (MODE 1 spill path)
      auto sample = [&](int buf, const Regs &R, int ovf_slot) {
#ifdef KGDET_ABL_NOSAMPLE
        {
          bf16x8 z;
#pragma unroll
          for (int q = 0; q < 8; ++q) z[q] = (__bf16)R.w[0][q & 3];
          unsigned char *dst0 = Bs + buf * PARTS * kBPart + half * (kTileN * 16) + n_local * 16;
          *reinterpret_cast<bf16x8 *>(dst0) = z;
          if constexpr (PARTS == 2) *reinterpret_cast<bf16x8 *>(dst0 + kBPart) = z;
          return;
        }
#endif
        // interpolation and hi/lo split on channel PAIRS: v_pk_fma_f32 / v_pk_add_f32 do two lanes' worth per
        // issue slot, and issue slots are what this kernel is short of (VALU and MFMA time add up on a SIMD)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 sv[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};  // [quad][pair]
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          const unsigned o[4] = {R.off[gq].x, R.off[gq].y, R.off[gq].z, R.off[gq].w};
          f32x4 v[2][4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              v[c][e] = *reinterpret_cast<const f32x4 *>(plane + (o[e] ^ (unsigned)((half * 2 + c) << 4)));
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const f32x2 ve = {v[c][e][2 * h2], v[c][e][2 * h2 + 1]};
                const f32x2 we = {R.w[gq][e], R.w[gq][e]};
                sv[c][h2] = (gq == 0 && e == 0) ? we * ve : __builtin_elementwise_fma(we, ve, sv[c][h2]);
              }
        }
        if constexpr (MODE == 1) {
          // contributions beyond the 8 inline ones: the (tile, tap)'s overflow list, staged in LDS one stage ago;
          // every thread scans it (uniform trip count) and adds the entries of its own cell
          const uint2 *ov = ovf_lds + ovf_slot * (kOvfCap + 1);
          const int n_all = (int)ov[kOvfCap].x;
          const int n_lds = min(n_all, kOvfCap);
          for (int i = 0; i < n_lds; ++i) {
            const uint2 e = ov[i];
            if ((int)(e.x & 127u) == n_local) {
              const float w = __uint_as_float(e.y);
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(plane + ((e.x >> 7) ^ (unsigned)((half * 2 + c) << 4)));
                sv[c][0] += f32x2{w * v[0], w * v[1]};
                sv[c][1] += f32x2{w * v[2], w * v[3]};
              }
            }
          }
          if (n_all > kOvfCap) {  // longer than the staged slots: the rest straight from the spill list (rare)
            const DcnInvOvfSlots *sl = p.inv_ovf + ((size_t)(tile_b * K + ovf_tap) * p.tiles_per_image + tile_in_img);
            const uint2 *spill = p.inv_spill + sl->spill_start;
            for (int i = kOvfCap; i < n_all; ++i) {
              const uint2 e = spill[i];
              if ((int)(e.x & 127u) == n_local) {
                const float w = __uint_as_float(e.y);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                  const f32x4 v = *reinterpret_cast<const f32x4 *>(plane + ((e.x >> 7) ^ (unsigned)((half * 2 + c) << 4)));
                  sv[c][0] += f32x2{w * v[0], w * v[1]};
                  sv[c][1] += f32x2{w * v[2], w * v[3]};
                }
              }
            }
          }
        }
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
        unsigned char *dst = Bs + buf * PARTS * kBPart + half * (kTileN * 16) + n_local * 16;
        *reinterpret_cast<bf16x8 *>(dst) = hi;
        if constexpr (PARTS == 2) *reinterpret_cast<bf16x8 *>(dst + kBPart) = lo;
      };
      auto multiply = [&](int buf, const AFrag &F) {
#ifdef KGDET_ABL_NOMFMA
        return;
#endif
        if constexpr (!PRODUCER) {
          const unsigned char *A = As + buf * PARTS * kAPart + (lane >> 5) * (kTileM * 16) + (wm * 64 + (lane & 31)) * 16;
          const unsigned char *B = Bs + buf * PARTS * kBPart + (lane >> 5) * (kTileN * 16) + (wn * 64 + (lane & 31)) * 16;
          bf16x8 a[PARTS][2], b[PARTS][2];
#pragma unroll
          for (int part = 0; part < PARTS; ++part)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              if constexpr (kAFromL2) a[part][i] = F.a[part][i];
              else a[part][i] = *reinterpret_cast<const bf16x8 *>(A + part * kAPart + i * 32 * 16);
              b[part][i] = *reinterpret_cast<const bf16x8 *>(B + part * kBPart + i * 32 * 16);
            }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              if constexpr (PARTS == 2) {  // small terms first
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[1][ni], acc[mi][ni], 0, 0, 0);
              }
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
            }
        }
      };

      // prologue: pipeline primed three deep, plane in LDS, stage 0 in buffer 0
      Regs R0, R1, R2;
      AFrag F0, F1;   // two stages ahead is enough for L2 latency at ~1.4 us per stage; a third set spills
      __syncthreads();  // the previous segment's readers of plane / A / B are done
      if constexpr (PRODUCER) {
        issue(0, R0);
        issue(1, R1);
        issue(2, R2);
      } else if constexpr (kAFromL2) {
        a_issue(0, F0);
        a_issue(1, F1);
      }
      load_plane();
      if constexpr (PRODUCER) {
        commit_weights(0, R0);
        commit_ovf(0, R0);
        commit_ovf(1, R1);
      }
      __syncthreads();
      if constexpr (PRODUCER) {
        ovf_tap = t0;
        sample(0, R0, 0);
      }
      __syncthreads();
      // stage j: producers put the loads of stage j+3 in flight (RI, consumed one body ago), move the weight
      // stage j+1 (RC, loaded two bodies ago) into LDS and sample B stage j+1; consumers multiply stage j.
      auto body = [&](int j, Regs &RI, Regs &RC, Regs &RN, AFrag &FI) {   // FI: the fragments of stage j
        const int buf = j & 1;
        if constexpr (PRODUCER) {
          issue(j + 3, RI);
          commit_ovf(buf, RN);  // stage j+2's overflow slots (RN, loaded one body ago): read by sample() next body
          if (j + 1 < n) {
            commit_weights(buf ^ 1, RC);
            ovf_tap = t0 + j + 1;
            sample(buf ^ 1, RC, buf ^ 1);
          }
        } else {
          if (j < n) multiply(buf, FI);
          if constexpr (kAFromL2) a_issue(j + 2, FI);   // unconditional, clamped: stage j + 2 into the freed set
        }
        __syncthreads();
      };
      for (int j = 0; j < n; j += 6) {  // 6 = lcm(3 register sets, 2 LDS buffers): static names in the body
        body(j, R0, R1, R2, F0);
        body(j + 1, R1, R2, R0, F1);
        body(j + 2, R2, R0, R1, F0);
        if (j + 3 < n) {
          body(j + 3, R0, R1, R2, F1);
          body(j + 4, R1, R2, R0, F0);
          body(j + 5, R2, R0, R1, F1);
        }
      }
      s += n;
      ++c16;
      t0 = 0;
    }

    if constexpr (!PRODUCER) {
      if (s_begin == 0 && s_end == cpt) {
        store_output(p, mt, nt, tid, acc);
      } else {
        float *slab = slabs + ((long long)g * grp.slots + slot) * kTileElems;
        store_slab(slab, tid, acc);
      }
    }
    ++slot;
    cur += s_end - s_begin;
  }
}

}  // namespace kgdet
