// Deformable convolution forward for gfx950: bilinear gather -> LDS -> f32 MFMA, no column matrix.
//
// Reference path replaced: deformable_im2col (+ modulated) and the addmm_ that follows it
// (mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:190-276, 570-632; deform_conv_cuda.cpp:221-245,
// 534-563).  The reference writes a [C*K, N*Ho*Wo] fp32 column matrix to HBM and reads it back
// for the GEMM; here a 16-channel x 128-pixel slice of it lives only in LDS.
//
// GEMM view:  out[o, p] = sum_{t, c} Wpk[t][c][o] * sample(x[b(p), c], pos(p, t))
//   M = output channels (tile 256), N = output pixels p = (b, oy, ox) (tile 128),
//   reduction = (tap t, channel c) in stages of (one tap, 16 channels).
// Parallelism: M*N has only ~17 tiles at KGDet sizes (256 x 2100), so the reduction is split
// stream-K style: the n_tiles * stages units are dealt evenly to the G resident workgroups;
// a workgroup whose range does not cover a whole tile writes its partial tile to a slab and
// dcn_fwd_fixup sums the slabs in workgroup order (deterministic, no atomics).
#include "common.h"
#include "dcn_common.h"

namespace kgdet {

namespace {

constexpr int kLdsA = kChunk * kTileM;  // floats per A stage
constexpr int kLdsB = kChunk * kTileN;

// A stage = Wpk[t][c0 .. c0+15][m0 .. m0+255]: 16 rows of 1 KiB, copied straight into LDS by the
// LDS-DMA path (no VGPR round trip).  The LDS image is lane-linear, which is exactly the
// k-major [16][256] layout mfma_stage reads.
__device__ __forceinline__ void stage_weights(const DcnProblem &p, int t, int c0, int m0, float *As, int tid) {
  // the LDS-DMA base must be provably wave-uniform (it goes to M0), or hipcc emits a waterfall loop
  const int wave_base = __builtin_amdgcn_readfirstlane(tid >> 6) << 6;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int q = tid + kThreads * r;  // float4 index inside the stage
    const int k = q >> 6, col4 = q & 63;
    const float *src = p.wpk + ((long long)(t * p.Cg_pad + c0 + k) * p.Og_pad + m0 + col4 * 4);
    float *dst = As + (wave_base + kThreads * r) * 4;  // wave-uniform base; HW adds lane*16 B
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  }
}

}  // namespace

__global__ __launch_bounds__(kThreads, 2) void dcn_fwd_mfma(const DcnProblem p, float *__restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kLdsA + 2 * kLdsB];
  float *As = lds;
  float *Bs = lds + 2 * kLdsA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int n_local = tid & (kTileN - 1);  // pixel column this thread gathers for
  const int cq = tid >> 7;                 // which 4 of the stage's 16 channels
  const long long G = gridDim.x, g = blockIdx.x;
  const long long my_begin = unit_begin(g, p.total_units, G);
  const long long my_end = unit_begin(g + 1, p.total_units, G);
  const int HW = p.H * p.W;

  long long cur = my_begin;
  while (cur < my_end) {
    const int cpt = p.chunks_per_tile;
    const int tile = (int)(cur / cpt);
    const long long tile_begin = (long long)tile * cpt;
    const int s_begin = (int)(cur - tile_begin);
    const int s_end = (int)((my_end - tile_begin) < cpt ? (my_end - tile_begin) : cpt);
    const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
    const int m0 = mt * kTileM;

    // the output pixel this thread samples for
    const int pix = nt * kTileN + n_local;
    const bool live = pix < p.P;
    const int pb = live ? pix / p.HoWo : 0;
    const int hw = live ? pix - pb * p.HoWo : 0;
    const int oy = hw / p.Wo, ox = hw - oy * p.Wo;
    const unsigned xb_off = ((unsigned)pb * (unsigned)p.C_total + (unsigned)p.c_base) * (unsigned)HW;  // elements

    f32x16 acc[2][2];
    zero_acc(acc);

    // ---- software-pipelined stage loop -------------------------------------------------------------
    // Two stages deep, entirely with compiler-visible loads (so hipcc's counted s_waitcnt vmcnt(N) stay
    // exact and nothing ever drains the queue): iteration s multiplies stage s out of LDS buffer s&1,
    //   - k-steps 0-3: ISSUE the loads of stage s+2 (4 x 2 corner-pair gathers + the 16 KiB weight stage
    //     as 2 x 16 B per thread) into one register set,
    //   - k-steps 4-7: COMMIT stage s+1 from the other register set (interpolate -> B, weights -> A of
    //     LDS buffer (s+1)&1), loaded one iteration earlier,
    // so every load has ~1.5 stage times (> 3 us) to land, which covers the L2-miss latency of the
    // weight stream (a one-stage-deep LDS-DMA version measured 1.7 us per stage with nothing else to do).
    // Every slice issues in the shadow of the MFMAs before it; operand fragments of the next k-step are
    // fetched before the current MFMAs; raw offsets of an upcoming tap are fetched a stage early.
    struct StageRegs {
      f32x2u v[4][2];  // gathered corner pairs [channel][row]
      f32x4 a[2];      // this thread's 2 x 16 B of the weight stage
      float w[4];      // bilinear weights the gathers belong to
    };
    StageRegs R0, R1;
    TapPair tap;       // tap used by the loads being ISSUED
    int tap_key = -1;
    float raw_y = 0.f, raw_x = 0.f, raw_m = 0.f;  // prefetched position of an upcoming tap
    int raw_key = -1;

    auto stage_key = [&](int s, int &t, int &c0, int &dgi) {
      t = s / p.chunks_per_tap;
      c0 = (s - t * p.chunks_per_tap) * kChunk + cq * 4;
      dgi = (p.c_base + min(c0, p.Cg - 1)) / p.cpdg;
      return t * p.DG + dgi;
    };
    auto fetch_raw = [&](int s) {
      int t, c0, dgi;
      raw_key = stage_key(s, t, c0, dgi);
      if (live) tap_position(p, pb, dgi, t, hw, oy, ox, raw_y, raw_x, raw_m);
      else { raw_y = raw_x = 0.f; raw_m = 0.f; }
    };
    auto retarget_tap = [&](int s) {  // make `tap` the tap of stage s (VALU only: raw_* already hold it)
      int t, c0, dgi;
      const int key = stage_key(s, t, c0, dgi);
      if (key != tap_key) {
        if (raw_key != key) fetch_raw(s);  // prologue only
        make_tap_pair(raw_y, raw_x, p.H, p.W, live, raw_m, tap);
        tap_key = key;
      }
    };
    auto issue_gather = [&](int s, int j, StageRegs &R) {
      int t, c0, dgi;
      stage_key(s, t, c0, dgi);
      const int c = min(c0 + j, p.Cg - 1);  // padded channels read a valid plane; their weights are 0
      // uniform base + 32-bit byte offset => saddr addressing, one VGPR per address (tensors are < 2 GiB)
      const unsigned plane_off = xb_off + (unsigned)c * (unsigned)HW;
      const char *base = reinterpret_cast<const char *>(p.x);
      R.v[j][0] = *reinterpret_cast<const f32x2u *>(base + (size_t)((plane_off + (unsigned)tap.o[0]) * 4u));
      R.v[j][1] = *reinterpret_cast<const f32x2u *>(base + (size_t)((plane_off + (unsigned)tap.o[1]) * 4u));
    };
    auto issue_weights = [&](int s, StageRegs &R) {
      const int t = s / p.chunks_per_tap;
      const int c0 = (s - t * p.chunks_per_tap) * kChunk;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int q = tid + kThreads * r;  // float4 index inside the [16][256] stage
        const int k = q >> 6, col4 = q & 63;
        const unsigned woff = ((unsigned)(t * p.Cg_pad + c0 + k) * (unsigned)p.Og_pad + (unsigned)(m0 + col4 * 4)) * 4u;
        R.a[r] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p.wpk) + (size_t)woff);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) R.w[q] = tap.w[q];
    };
    auto commit_gather = [&](float *Bdst, int j, const StageRegs &R) {
      const float sv = R.w[0] * R.v[j][0][0] + R.w[1] * R.v[j][0][1] + R.w[2] * R.v[j][1][0] + R.w[3] * R.v[j][1][1];
      Bdst[(cq * 4 + j) * kTileN + n_local] = sv;
    };
    auto commit_weights = [&](float *Adst, const StageRegs &R) {
#pragma unroll
      for (int r = 0; r < 2; ++r) *reinterpret_cast<f32x4 *>(Adst + (tid + kThreads * r) * 4) = R.a[r];
    };

    // prologue: stage s_begin straight into buffer 0, loads of stage s_begin+1 in flight in R1
    retarget_tap(s_begin);
    issue_weights(s_begin, R0);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_gather(s_begin, j, R0);
    if (s_begin + 1 < s_end) {
      retarget_tap(s_begin + 1);
      issue_weights(s_begin + 1, R1);
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_gather(s_begin + 1, j, R1);
    }
    commit_weights(As, R0);
#pragma unroll
    for (int j = 0; j < 4; ++j) commit_gather(Bs, j, R0);
    if (s_begin + 2 < s_end) fetch_raw(s_begin + 2);
    __syncthreads();

    const int kk = lane >> 5, l31 = lane & 31;
    // one stage: MFMAs of stage s from buffer s&1; ISSUE stage s+2 into RI; COMMIT stage s+1 from RC
    auto run_stage = [&](int s, StageRegs &RI, StageRegs &RC) {
      const int buf = (s - s_begin) & 1;
      const bool do_issue = (s + 2) < s_end, do_commit = (s + 1) < s_end;
      const float *A = As + buf * kLdsA + wm * 64 + l31;
      const float *Bv = Bs + buf * kLdsB + wn * 64 + l31;
      float *Anext = As + (buf ^ 1) * kLdsA, *Bnext = Bs + (buf ^ 1) * kLdsB;
      if (do_issue) retarget_tap(s + 2);
      float a0 = A[kk * kTileM], a1 = A[kk * kTileM + 32], b0 = Bv[kk * kTileN], b1 = Bv[kk * kTileN + 32];
#pragma unroll
      for (int ks = 0; ks < kChunk / 2; ++ks) {
        if (ks < 4) {
          if (do_issue) {
            if (ks == 0) issue_weights(s + 2, RI);
            issue_gather(s + 2, ks, RI);
          }
        } else if (do_commit) {
          if (ks == 4) commit_weights(Anext, RC);
          commit_gather(Bnext, ks - 4, RC);
        }
        float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
        if (ks + 1 < kChunk / 2) {
          const int k = 2 * (ks + 1) + kk;
          na0 = A[k * kTileM]; na1 = A[k * kTileM + 32]; nb0 = Bv[k * kTileN]; nb1 = Bv[k * kTileN + 32];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        __builtin_amdgcn_sched_barrier(0);  // keep the slices where they are written
      }
      if ((s + 3) < s_end) {  // raw position of the tap three stages ahead, only if it is a new one
        int t3, c3, d3;
        if (stage_key(s + 3, t3, c3, d3) != tap_key) fetch_raw(s + 3);
      }
      __syncthreads();
    };
    for (int s = s_begin; s < s_end; s += 2) {
      run_stage(s, R0, R1);
      if (s + 1 < s_end) run_stage(s + 1, R1, R0);
    }

    if (s_begin == 0 && s_end == cpt) {
      store_output(p, mt, nt, tid, acc);
    } else {
      float *slab = slabs + ((long long)g * 2 + slab_slot(cur, my_begin)) * kTileElems;
      store_slab(slab, tid, acc);
    }
    cur = tile_begin + s_end;
  }
}

// Split tiles: add the workgroups' slabs, ranges in order, workgroups in order (deterministic).
// grid = (tiles, 16): block (tile, j) owns one float4 column j of the 16-float4 register image, so the
// additions are spread over 16x more workgroups than tiles (17 tiles alone would leave 93 % of the chip idle).
__global__ __launch_bounds__(kThreads) void dcn_fwd_fixup(const DcnFwdGroup grp, const float *__restrict__ slabs,
                                                         int G) {
  const int gtile = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
  int pi = 0;
  while (pi + 1 < grp.n && gtile >= grp.tile_begin[pi + 1]) ++pi;
  const DcnProblem &p = grp.p[pi];
  const int tile = gtile - grp.tile_begin[pi];
  const int tiles = p.n_ntiles * p.n_mtiles;
  const long long total = grp.unit_begin[grp.n];

  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int part = 0; part < p.kparts; ++part) {
    const long long tb = dcn_range_first_unit(grp, pi, part, tile);
    const long long te = tb + (dcn_part_lo(p, part + 1) - dcn_part_lo(p, part));
    if (te == tb) continue;
    const int range = grp.range_begin[pi] + part * tiles + tile;
    if (grp.static_ranges) {   // range r was computed whole by the workgroup of slice r (slab slot 0)
      if (p.kparts == 1) return;  // written directly
      const long long slab = (long long)sk_block_of_slice(range, G) * grp.slots;
      const f32x4 v = reinterpret_cast<const f32x4 *>(slabs + slab * kTileElems)[j * kThreads + tid];
      sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
      continue;
    }
    long long g = tb * G / total;  // slice that holds unit tb
    while (unit_begin(g + 1, total, G) <= tb) ++g;
    while (unit_begin(g, total, G) > tb) --g;
    if (p.kparts == 1 && unit_begin(g, total, G) <= tb && unit_begin(g + 1, total, G) >= te) return;  // written directly
    for (; g < G; ++g) {
      const long long b0 = unit_begin(g, total, G);
      if (b0 >= te) break;
      if (unit_begin(g + 1, total, G) == b0) continue;  // slice with an empty range
      const long long seg_begin = b0 > tb ? b0 : tb;
      long long slab;
      if (grp.xcd_slices) slab = (long long)sk_block_of_slice((int)g, G) * grp.slots + (range - dcn_unit_pos(grp, b0).range);
      else slab = g * 2 + slab_slot(seg_begin, b0);
      const f32x4 v = reinterpret_cast<const f32x4 *>(slabs + slab * kTileElems)[j * kThreads + tid];
      sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
    }
  }
  // layout 0: float4 column j = (mi, ni, q): accumulator registers 4q .. 4q+3 of block (mi, ni) of wave (wm, wn)
  // layout 1: float4 column j = (ni, q) of wave `wave`: rows wave*32.., columns ni*32..
  const int q = j & 3;
  const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
  const int lane = tid & 63, wave = tid >> 6;
  const int row0 = grp.wave_layout ? wave * 32 : (wave & 3) * 64 + (j >> 3) * 32;
  const int col0 = grp.wave_layout ? (j >> 2) * 32 : (wave >> 2) * 64 + ((j >> 2) & 1) * 32;
  int b, hw;
  if (!tile_pixel(p, nt, col0 + (lane & 31), b, hw)) return;
  float *obase = p.out + ((long long)b * p.O_total + p.o_base) * p.HoWo + hw;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int o = mt * kTileM + row0 + mfma_row(4 * q + e, lane);
    if (o >= p.Og) continue;
    float v = sum[e];
    if (p.bias) v += p.bias[p.bias_base + o];
    if (p.flags & KGDET_DCN_RELU) v = fmaxf(v, 0.0f);
    obase[(long long)o * p.HoWo] = v;
  }
}

// Fix-up of the static one-range-per-workgroup schedule (DcnFwdGroup::static_ranges): tile `gtile` = the sum of its
// kparts slabs (slot 0 of the workgroup of each (part, tile) range), parts in order.  grid = (tiles, 2): a block
// sums 8 of the 16 float4 columns of a tile -- per part all eight loads are in flight before the adds.  (The generic
// fix-up above spends one block per column and walks the stream-K slice arithmetic: 20 us for the head stage against
// ~8 us here.)
__global__ __launch_bounds__(kThreads) void dcn_fwd_fixup_static(const DcnFwdGroup grp, const float *__restrict__ slabs,
                                                                int G) {
  const int gtile = blockIdx.x, tid = threadIdx.x;
  int pi = 0;
  while (pi + 1 < grp.n && gtile >= grp.tile_begin[pi + 1]) ++pi;
  const DcnProblem &p = grp.p[pi];
  const bool grouped = p.sum_count > 1;   // (sum group, DcnProblem::sum_count: the leader's block adds every member's parts)
  if (grouped ? p.sum_members[0] != pi : p.kparts == 1) return;  // (no group, one part: written directly by the tile's only workgroup)
  const int tile = gtile - grp.tile_begin[pi];
  const int tiles = p.n_ntiles * p.n_mtiles;
  constexpr int kCols = 8;
  const int j0 = blockIdx.y * kCols;
  f32x4 sum[kCols];
#pragma unroll
  for (int c = 0; c < kCols; ++c) sum[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int m = 0; m < (grouped ? p.sum_count : 1); ++m) {
    const int mi = grouped ? p.sum_members[m] : pi;
    for (int part = 0; part < grp.p[mi].kparts; ++part) {
      const int range = grp.range_begin[mi] + part * tiles + tile;
      // (range r is computed by the workgroup of slice r % G in its round r / G: slab slot r / G)
      const f32x4 *s4 = reinterpret_cast<const f32x4 *>(slabs + ((long long)sk_block_of_slice(range % G, G) * grp.slots + range / G) * kTileElems);
      f32x4 v[kCols];
#pragma unroll
      for (int c = 0; c < kCols; ++c) v[c] = s4[(j0 + c) * kThreads + tid];
#pragma unroll
      for (int c = 0; c < kCols; ++c) sum[c] += v[c];
    }
  }
  const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < kCols; ++c) {
    const int j = j0 + c, q = j & 3;
    const int row0 = grp.wave_layout ? wave * 32 : (wave & 3) * 64 + (j >> 3) * 32;
    const int col0 = grp.wave_layout ? (j >> 2) * 32 : (wave >> 2) * 64 + ((j >> 2) & 1) * 32;
    int b, hw;
    if (!tile_pixel(p, nt, col0 + (lane & 31), b, hw)) continue;
    float *obase = p.out + ((long long)b * p.O_total + p.o_base) * p.HoWo + hw;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = mt * kTileM + row0 + mfma_row(4 * q + e, lane);
      if (o >= p.Og) continue;
      float v = sum[c][e];
      if (p.bias) v += p.bias[p.bias_base + o];
      if (p.flags & KGDET_DCN_RELU) v = fmaxf(v, 0.0f);
      obase[(long long)o * p.HoWo] = v;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Weight packing: [O, Cg, K] (one group) -> [K][Cg_pad][Og_pad], zero padded.
// One workgroup per (c, 64 output channels): reads 64 runs of K contiguous floats, transposes
// through LDS, writes K runs of 64 contiguous floats.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_pack_weight(const float *__restrict__ w, float *__restrict__ wpk,
                                                        int Og, int Cg, int K, int Cg_pad, int Og_pad) {
  extern __shared__ float tile[];  // [64][K+1]
  const int c = blockIdx.x, o0 = blockIdx.y * 64, tid = threadIdx.x;
  const int ld = K + 1;
  for (int e = tid; e < 64 * K; e += 256) {
    const int oo = e / K, t = e - oo * K;
    const int o = o0 + oo;
    tile[oo * ld + t] = (o < Og && c < Cg) ? w[((long long)o * Cg + c) * K + t] : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * K; e += 256) {
    const int t = e >> 6, oo = e & 63;
    wpk[((long long)t * Cg_pad + c) * Og_pad + o0 + oo] = tile[oo * ld + t];
  }
}

// Transposed image for backward-input: [O, Cg, K] -> Wt[K][Og_pad16][Cg_pad256], zero padded.
// One workgroup per (o, 64 channels): reads 64*K contiguous floats, writes K runs of 64 floats.
__global__ __launch_bounds__(256) void dcn_pack_weight_t(const float *__restrict__ w, float *__restrict__ wpt,
                                                          int Og, int Cg, int K, int Og_pad16, int Cg_pad256) {
  extern __shared__ float tile[];  // [64][K+1]
  const int o = blockIdx.x, c0 = blockIdx.y * 64, tid = threadIdx.x;
  const int ld = K + 1;
  for (int e = tid; e < 64 * K; e += 256) {
    const int cc = e / K, t = e - cc * K;
    const int c = c0 + cc;
    tile[cc * ld + t] = (o < Og && c < Cg) ? w[((long long)o * Cg + c) * K + t] : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * K; e += 256) {
    const int t = e >> 6, cc = e & 63;
    wpt[((long long)t * Og_pad16 + o) * Cg_pad256 + c0 + cc] = tile[cc * ld + t];
  }
}

// inverse: packed gradient [K][Cg_pad][Og_pad] -> [O, Cg, K]
__global__ __launch_bounds__(256) void dcn_unpack_weight(const float *__restrict__ wpk, float *__restrict__ w,
                                                          int Og, int Cg, int K, int Cg_pad, int Og_pad,
                                                          int accumulate) {
  extern __shared__ float tile[];
  const int c = blockIdx.x, o0 = blockIdx.y * 64, tid = threadIdx.x;
  const int ld = K + 1;
  for (int e = tid; e < 64 * K; e += 256) {
    const int t = e >> 6, oo = e & 63;
    tile[oo * ld + t] = wpk[((long long)t * Cg_pad + c) * Og_pad + o0 + oo];
  }
  __syncthreads();
  for (int e = tid; e < 64 * K; e += 256) {
    const int oo = e / K, t = e - oo * K;
    const int o = o0 + oo;
    if (o < Og) {
      float *dst = w + ((long long)o * Cg + c) * K + t;
      *dst = accumulate ? *dst + tile[oo * ld + t] : tile[oo * ld + t];
    }
  }
}

}  // namespace kgdet
