// Deformable convolution forward, "plane" variant for gfx950: feature planes resident in LDS, bilinear
// gathers served by LDS, bf16 MFMA with an fp32-accurate hi/lo operand split, producer / consumer waves.
//
// Same reference path as dcn_forward.hip (deformable_im2col + addmm_,
// mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:190-276, 570-632; deform_conv_cuda.cpp:221-245, 534-563)
// and the same GEMM view / stream-K split.  What differs from the exact-fp32 kernel:
//   * MFMA rate: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  Every fp32 operand v is split into
//     hi = bf16(v), lo = bf16(v - hi) and the product is taken as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with
//     v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 3 MFMAs of 32 cycles replace 8 of 64; the dropped
//     a_lo*b_lo term and the representation residuals are <= 2^-16 relative per product (against the f64
//     oracle: ~1e-6 of the output scale, tests/test_gpu_dcn.py).  PARTS = 1 keeps only the hi parts:
//     plain bf16 operands for autocast inference.
//   * gathers: a 16-channel slice of ONE image is copied into LDS as [pixel][16 channels] (67 KB at 25x42)
//     and reused by all K taps of that channel chunk, so a bilinear corner of 4 channels is one
//     ds_read_b128; the reduction runs chunk-major / tap-minor and pixel tiles never straddle images.
//   * sampling geometry is not recomputed per channel chunk: dcn_build_taps writes one 32-byte record per
//     (image, tap, output pixel) -- four LDS byte offsets and four bilinear weights -- and the kernel only
//     loads it.  With the split MFMA the kernel is VALU-issue bound (SQ counters: MFMA pipe 30 % busy,
//     waves stalled on issue 34 % of the time), so instructions per sample are what matters.
//   * 12 waves, two roles: waves 0-7 are CONSUMERS (they own the 256 x 128 accumulator tile as 4 x 2 waves
//     of 64 x 64 and do only operand-fragment reads and MFMAs), waves 8-11 are PRODUCERS (they stream the
//     weight stages and build the next B stage: one thread samples 8 channels of one pixel).  The SIMD
//     interleaves the roles' instruction streams; one barrier per stage hands the double-buffered stages over.
//
// Operand images (identical in global memory and LDS, so the weight stage is a lane-linear copy):
//   A stage (tap t, channel chunk c16, 256 output channels): [part][khalf][o 256][8 bf16]   8 KB / part
//   B stage (same reduction slice, 128 pixels):              [part][khalf][px 128][8 bf16]  4 KB / part
//   lane l of a wave reads row/col (l & 31) of k-half (l >> 5) as one 16-byte ds_read_b128; reduction
//   element k = khalf*8 + j is channel c16*16 + k for both operands.
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

// what a producer thread has in flight for one stage
template <int PARTS>
struct PlaneStageRegs {
  f32x4 a[PARTS][2];  // this thread's 2 x 16 B of each part of the weight stage
  uint4 off;          // DcnTapRec: LDS byte offsets of the four corners (quad 0)
  f32x4 w;            //            and their bilinear weights
};

}  // namespace

#ifndef KGDET_ABL
#define KGDET_ABL 0
#endif
// Loop structure: a workgroup walks its stream-K slice range by range; inside a range the stages that share
// a feature plane (one channel chunk, consecutive taps) form a SEGMENT.  A segment starts with the plane copy
// and the priming of the producers' three-deep register pipeline (weight stage + tap record of stages
// j, j+1, j+2); its steady state is branch-free as far as vector-memory instructions go -- every body issues
// the loads of stage j+3 unconditionally (clamped to the last stage), so hipcc's counted s_waitcnt vmcnt(N)
// stay exact and a load has two full stages to land.  Stage coordinates are carried incrementally; there is no
// integer division in the loop (a runtime s / K costs ~35 dependent SALU ops).
// The two roles are two instantiations of this function (same loop structure, same barriers), so the
// accumulators exist only in the consumers' register allocation.
template <int PARTS, bool PRODUCER>
__device__ __forceinline__ void plane_role(const DcnFwdGroup &grp, float *__restrict__ slabs, unsigned char *smem) {
  constexpr int ABL = KGDET_ABL;
  unsigned char *As = smem;                                                          // [2][PARTS][kAPart]
  unsigned char *Bs = smem + 2 * PARTS * kAPart;                                     // [2][PARTS][kBPart]
  unsigned char *plane = smem + 2 * PARTS * (kAPart + kBPart);                       // [H*W][16 channels] fp32, swizzled

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

    typedef PlaneStageRegs<PARTS> Regs;

    int s = s_begin;
    int c16 = s / K;
    int t0 = s - c16 * K;
    while (s < s_end) {
      const int n = min(K - t0, s_end - s);  // stages of this segment: taps t0 .. t0+n-1 of chunk c16
      const int dgi = p.DG == 1 ? 0 : (p.c_base + min(c16 * kChunk + half * 8, p.Cg - 1)) / p.cpdg;
      const DcnTapRec *rec_base = p.taps + ((size_t)(tile_b * p.DG + dgi) * K) * HoWo + hw_c;
      const unsigned char *wq_base = reinterpret_cast<const unsigned char *>(p.wq) +
                                     (size_t)((mt * n_c16 + c16) * K) * (2 * kAPart) + tid * 16;

      auto issue = [&](int j, Regs &R) {  // loads of stage j (clamped): weight stage + tap record
        const unsigned t = (unsigned)(t0 + min(j, n - 1));
        const uint4 *rec = reinterpret_cast<const uint4 *>(rec_base + (size_t)t * HoWo);
        R.off = rec[0];
        R.w = *reinterpret_cast<const f32x4 *>(rec + 1);
#pragma unroll
        for (int part = 0; part < PARTS; ++part)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            R.a[part][r] = *reinterpret_cast<const f32x4 *>(wq_base + (size_t)t * (2 * kAPart) + part * kAPart +
                                                            r * (kProducers * 16));
      };
      auto commit_weights = [&](int buf, const Regs &R) {
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
        if (ABL & 16) return;
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
      auto sample = [&](int buf, const Regs &R) {
        if (ABL & 8) return;
        const unsigned o[4] = {R.off.x, R.off.y, R.off.z, R.off.w};
        f32x4 v[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            v[c][e] = *reinterpret_cast<const f32x4 *>(plane + (o[e] ^ (unsigned)((half * 2 + c) << 4)));
        // interpolation and hi/lo split on channel PAIRS: v_pk_fma_f32 / v_pk_add_f32 do two lanes' worth per
        // issue slot, and issue slots are what this kernel is short of (VALU and MFMA time add up on a SIMD)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        bf16x8 hi, lo;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            f32x2 sv = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 ve = {v[c][e][2 * h2], v[c][e][2 * h2 + 1]};
              const f32x2 we = {R.w[e], R.w[e]};
              sv = e == 0 ? we * ve : __builtin_elementwise_fma(we, ve, sv);
            }
            const int q = c * 4 + h2 * 2;
            hi[q] = (__bf16)sv[0];
            hi[q + 1] = (__bf16)sv[1];
            if constexpr (PARTS == 2) {
              const f32x2 hf = {(float)hi[q], (float)hi[q + 1]};
              const f32x2 lf = sv - hf;
              lo[q] = (__bf16)lf[0];
              lo[q + 1] = (__bf16)lf[1];
            }
          }
        unsigned char *dst = Bs + buf * PARTS * kBPart + half * (kTileN * 16) + n_local * 16;
        *reinterpret_cast<bf16x8 *>(dst) = hi;
        if constexpr (PARTS == 2) *reinterpret_cast<bf16x8 *>(dst + kBPart) = lo;
      };
      auto multiply = [&](int buf) {
        if constexpr (!PRODUCER) {
          if (ABL & 2) return;
          const unsigned char *A = As + buf * PARTS * kAPart + (lane >> 5) * (kTileM * 16) + (wm * 64 + (lane & 31)) * 16;
          const unsigned char *B = Bs + buf * PARTS * kBPart + (lane >> 5) * (kTileN * 16) + (wn * 64 + (lane & 31)) * 16;
          bf16x8 a[PARTS][2], b[PARTS][2];
#pragma unroll
          for (int part = 0; part < PARTS; ++part)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              a[part][i] = *reinterpret_cast<const bf16x8 *>(A + part * kAPart + i * 32 * 16);
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
      __syncthreads();  // the previous segment's readers of plane / A / B are done
      if constexpr (PRODUCER) {
        issue(0, R0);
        issue(1, R1);
        issue(2, R2);
      }
      load_plane();
      if constexpr (PRODUCER) commit_weights(0, R0);
      __syncthreads();
      if constexpr (PRODUCER) sample(0, R0);
      __syncthreads();
      // stage j: producers put the loads of stage j+3 in flight (RI, consumed one body ago), move the weight
      // stage j+1 (RC, loaded two bodies ago) into LDS and sample B stage j+1; consumers multiply stage j.
      auto body = [&](int j, Regs &RI, Regs &RC) {
        const int buf = j & 1;
        if constexpr (PRODUCER) {
          issue(j + 3, RI);
          if (j + 1 < n) {
            commit_weights(buf ^ 1, RC);
            sample(buf ^ 1, RC);
          }
        } else {
          if (j < n) multiply(buf);
        }
        __syncthreads();
      };
      for (int j = 0; j < n; j += 6) {  // 6 = lcm(3 register sets, 2 LDS buffers): static names in the body
        body(j, R0, R1);
        body(j + 1, R1, R2);
        body(j + 2, R2, R0);
        if (j + 3 < n) {
          body(j + 3, R0, R1);
          body(j + 4, R1, R2);
          body(j + 5, R2, R0);
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

template <int PARTS>
__global__ __launch_bounds__(kPlaneThreads, 1) void dcn_fwd_plane(const DcnFwdGroup grp, float *__restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) plane_role<PARTS, true>(grp, slabs, smem);
  else plane_role<PARTS, false>(grp, slabs, smem);
}

template __global__ void dcn_fwd_plane<1>(const DcnFwdGroup grp, float *__restrict__ slabs);
template __global__ void dcn_fwd_plane<2>(const DcnFwdGroup grp, float *__restrict__ slabs);

int dcn_fwd_plane_threads() { return kPlaneThreads; }

// Tap records of every problem of a group that owns its table (p.build_taps): one thread per
// (image, deformable group, tap, output pixel).  grid = (blocks over the largest table, problems).
// Mirrors deformable_im2col_bilinear + the (-1,H)x(-1,W) guard (deform_conv_cuda_kernel.cu:84-114, :228);
// modulated (v2) problems fold the mask into the weights (:570-632).
__global__ __launch_bounds__(256) void dcn_build_taps(const DcnFwdGroup grp) {
  const DcnProblem &p = grp.p[blockIdx.y];
  if (!p.build_taps) return;
  const long long n_rec = (long long)p.N * p.DG * p.K * p.HoWo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_rec; i += (long long)gridDim.x * 256) {
    const int hw = (int)(i % p.HoWo);
    const int t = (int)((i / p.HoWo) % p.K);
    const int bd = (int)(i / ((long long)p.HoWo * p.K));  // b * DG + dgi
    const int oy = hw / p.Wo, ox = hw - oy * p.Wo;
    const int ti = t / p.kw, tj = t - ti * p.kw;
    const long long ob = ((long long)bd * 2 * p.K + 2 * t) * p.HoWo + hw;
    const float y = (float)(oy * p.sh - p.ph + ti * p.dh) + p.offset[ob];
    const float x = (float)(ox * p.sw - p.pw + tj * p.dw) + p.offset[ob + p.HoWo];
    const float m = p.mask ? p.mask[((long long)bd * p.K + t) * p.HoWo + hw] : 1.0f;
    Tap tap;
    TapGeom geom;
    make_tap(y, x, p.H, p.W, true, m, tap, geom);
    DcnTapRec r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = tap.o[e];
      r.off[e] = (unsigned)(dcn_plane_offset(q) + (((q >> 2) & 3) << 4));
      r.w[e] = tap.w[e];
    }
    const_cast<DcnTapRec *>(p.taps)[i] = r;
  }
}

size_t dcn_fwd_plane_lds_bytes(int parts, int HW) {
  return (size_t)2 * parts * (kAPart + kBPart) + (size_t)kChunk * HW * sizeof(float);
}

// ----------------------------------------------------------------------------------------------
// Weight packing, all three images in one pass over the weights.  [O, Cg, K] (one group) ->
//   wpk[t][c pad16][o pad256]                       fp32  (exact forward kernel, dcn_forward.hip)
//   wpt[t][o pad16][c pad256]                       fp32  (backward-input, dcn_backward*.hip)
//   wq [mt][c16][t][part (hi, lo)][khalf][o 256][8] bf16  (plane forward kernel; hi = bf16(w), lo = bf16(w - hi))
// all zero padded.  One workgroup per (8 channels = one k-half, 32 output channels): each wave reads
// whole runs of K contiguous floats into an LDS tile [8][33][K+1]; the three images are written from it
// in runs of 128 B (wpk), 32 B (wpt) and 512 B (wq).
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_pack_weight_all(const float *__restrict__ w, float *__restrict__ wpk,
                                                            float *__restrict__ wpt, void *__restrict__ wq,
                                                            int Og, int Cg, int K, int Cg_pad, int Og_pad,
                                                            int Og_pad16, int Cg_pad256) {
  extern __shared__ float tile[];  // [(cc * 33 + oo)][K+1]
  const int c8 = blockIdx.x, o0 = blockIdx.y * 32, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ld = K + 1;
  for (int t0 = 0; t0 < K; t0 += 64) {
    const int t = t0 + lane;
    for (int r0 = wave; r0 < 256; r0 += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // 8 independent (clamped, unconditional) loads in flight per lane
        const int r = r0 + 4 * u, cc = r >> 5, oo = r & 31;
        const int c = min(c8 * 8 + cc, Cg - 1), o = min(o0 + oo, Og - 1);
        v[u] = w[((long long)o * Cg + c) * K + min(t, K - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 4 * u, cc = r >> 5, oo = r & 31;
        const bool real = (c8 * 8 + cc < Cg) && (o0 + oo < Og);
        if (t < K) tile[(cc * 33 + oo) * ld + t] = real ? v[u] : 0.0f;
      }
    }
  }
  __syncthreads();
  const int cbase = c8 * 8;
  if (cbase < Cg_pad && o0 < Og_pad) {
    const int oo = tid & 31, cc = tid >> 5;
    for (int t = 0; t < K; ++t)
      wpk[((long long)t * Cg_pad + cbase + cc) * Og_pad + o0 + oo] = tile[(cc * 33 + oo) * ld + t];
    if (wq) {
      const int n_c16 = Cg_pad / kChunk;
      const int c16 = c8 >> 1, khalf = c8 & 1;
      const int mt = o0 / kTileM, o_in = o0 % kTileM;
      for (int e = tid; e < K * 64; e += 256) {
        const int o2 = e & 31, part = (e >> 5) & 1, t = e >> 6;
        bf16x8 v;
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          const float f = tile[(c2 * 33 + o2) * ld + t];
          const __bf16 hi = (__bf16)f;
          v[c2] = part == 0 ? hi : (__bf16)(f - (float)hi);
        }
        const size_t stage = (size_t)((mt * n_c16 + c16) * K + t) * (2 * kAPart);
        unsigned char *dst = reinterpret_cast<unsigned char *>(wq) + stage + part * kAPart + khalf * (kTileM * 16) +
                             (o_in + o2) * 16;
        *reinterpret_cast<bf16x8 *>(dst) = v;
      }
    }
  }
  if (o0 < Og_pad16) {
    const int cc = tid & 7, oo = tid >> 3;
    if (o0 + oo < Og_pad16)
      for (int t = 0; t < K; ++t)
        wpt[((long long)t * Og_pad16 + o0 + oo) * Cg_pad256 + cbase + cc] = tile[(cc * 33 + oo) * ld + t];
  }
}

}  // namespace kgdet
