// Deformable convolution backward w.r.t. the input on the plane kernel (gfx950).
//
// Reference path replaced: the col2im half of deform_conv_backward_input_cuda / modulated_..._backward --
// columns = W^T grad_out (deform_conv_cuda.cpp:329-332, 634-643) scattered into grad_input with one float
// atomicAdd per (channel, tap, pixel, corner) (deformable_col2im, deform_conv_cuda_kernel.cu:279-334, 634-700).
//
// Here the scatter is turned around:
//     grad_input[b, c, q] = sum_t sum_o W[o, c, t] * G_t[b, o, q],
//     G_t[b, o, q]        = sum over the (output pixel p, corner) pairs of tap t that land on cell q of
//                           w_corner(p, t) * grad_out[b, o, p]
// which is the forward GEMM with the roles of input and output channels swapped and "sampling" replaced by
// TRANSPOSED sampling: cell q gathers from a short list of (pixel, weight) pairs instead of from 4 corners.
// dcn_build_inverse_taps inverts the sampling map per (image, tap) -- counting sort in LDS, lists sorted by
// pixel so every sum has a fixed order -- and stores, per cell with at most 8 pairs, the pairs inline (DcnInvRec); the sum
// of a cell with more is formed beforehand for all output channels by dcn_inv_overflow_sums (below; round 4 -- the cost of
// the whole backward no longer depends on where the sampling points fall).  dcn_bwd_input_plane is then
// plane_role<MODE = 1>: grad_out planes in LDS, weights from the transposed operand image `wqt`, bf16 hi/lo
// split MFMA, stream-K with the forward's fix-up.  No atomics, deterministic, no pre-zeroing of grad_input.
// Deformable groups > 1 (different cells lists per channel group inside one M tile) stay on the older kernels.
#include <type_traits>
#include "dcn_plane.h"

namespace kgdet {

template <int PARTS>
__global__ __launch_bounds__(kRoleThreads, 1) void dcn_bwd_input_plane(const DcnFwdGroup grp, float *__restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) plane_role<PARTS, true, 1>(grp, slabs, smem);
  else plane_role<PARTS, false, 1>(grp, slabs, smem);
}

template __global__ void dcn_bwd_input_plane<1>(const DcnFwdGroup grp, float *__restrict__ slabs);
template __global__ void dcn_bwd_input_plane<2>(const DcnFwdGroup grp, float *__restrict__ slabs);

size_t dcn_bwd_input_plane_fixed_lds_bytes(int parts) { return (size_t)2 * kGroupTaps * parts * kBPart; }
size_t dcn_bwd_input_plane_lds_bytes(int parts, int plane_pixels) {
  return plane_pixels <= kPlaneMaxHW ? dcn_bwd_input_plane_fixed_lds_bytes(parts) + (size_t)4 * kPlaneQuadStride : (size_t)1 << 30;
}

// ------------------------------------------------------------------------------------------------
// Inverse sampling records.  One workgroup per (image, tap); p describes the FORWARD problem
// (x [N, C, H, W], offsets over Ho x Wo) and names the deformable group (p.dgi) whose offsets are read.
//   inv   [N][K][H*W] DcnInvRec       cells with <= 8 contributions: inline, sorted by pixel; others: flag + slot
//   hdr   [N][K] int                  cells of the (image, tap) with > 8 contributions
//   cells [N][K][max_slots]           DcnInvOvfCell of those cells, in cell order (slot = rank among them)
//   spill [N][K][4 * Ho*Wo] int2      every (pixel, weight) entry of the (image, tap), grouped by cell (order inside a
//                                     cell arbitrary: dcn_inv_overflow_sums sums in ascending pixel order)
// LDS: cnt [HW + 1], cursor [HW], ent [4 * HoWo] (pixel, weight) pairs, flag / slot scans [2 * HW].
// Cost independent of how the sampling points are distributed: no per-cell loop is longer than 8 entries (rounds 2-3
// insertion-sorted every list -- quadratic in a cell's length -- and the builder of a head stage went from 37 to 118 us
// while the training step's offsets concentrated).
// ------------------------------------------------------------------------------------------------
#ifndef KGDET_HOT_MIN
#define KGDET_HOT_MIN 64   // contributions above which a cell becomes a column of dcn_hot_gemm
#endif
template <int THREADS>
__device__ __forceinline__ void build_inverse_taps_body(const DcnProblem &p, uint4 *__restrict__ inv, int *__restrict__ hdr,
                                                        DcnInvOvfCell *__restrict__ cells, int2 *__restrict__ spill,
                                                        int block, int *sm, int4 *__restrict__ hot_cols = nullptr,
                                                        int *__restrict__ hot_count = nullptr, int hot_max = 0) {
  const int HW = p.H * p.W;
  int *cnt = sm;                                          // [HW + 1]
  int *cursor = sm + (HW + 1);                            // [HW]
  int2 *ent = reinterpret_cast<int2 *>(sm + 2 * HW + 2);  // [4 * HoWo]
  __shared__ int wave_tot[THREADS / 64];

  const int t = block % p.K, b = block / p.K;
  const int tid = threadIdx.x;
  const int max_slots = dcn_inv_max_slots(HW, p.HoWo);

  for (int i = tid; i <= HW; i += THREADS) cnt[i] = 0;
  __syncthreads();
  // pass 1: count the corners landing in each cell
  for (int px = tid; px < p.HoWo; px += THREADS) {
    const int oy = px / p.Wo, ox = px - oy * p.Wo;
    float y, x, m;
    tap_position(p, b, p.dgi, t, px, oy, ox, y, x, m);
    Tap tap;
    TapGeom geo;
    make_tap(y, x, p.H, p.W, true, m, tap, geo);
    if (geo.va) atomicAdd(&cnt[tap.o[0]], 1);
    if (geo.vb) atomicAdd(&cnt[tap.o[1]], 1);
    if (geo.vc) atomicAdd(&cnt[tap.o[2]], 1);
    if (geo.vd) atomicAdd(&cnt[tap.o[3]], 1);
  }
  __syncthreads();
  // exclusive scan of an int array a[0..HW) into out[] (each thread owns a contiguous run); returns the total
  auto scan = [&](const int *a, int *out) -> int {
    const int per = (HW + THREADS - 1) / THREADS;
    const int lo = min(HW, tid * per), hi = min(HW, lo + per);
    int local = 0;
    for (int i = lo; i < hi; ++i) local += a[i];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int nb = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += nb;
    }
    __syncthreads();
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
    int run = base + incl - local;
    for (int i = lo; i < hi; ++i) {
      const int c = a[i];
      out[i] = run;
      run += c;
    }
    int total = 0;
    for (int w = 0; w < THREADS / 64; ++w) total += wave_tot[w];
    __syncthreads();
    return total;
  };
  const int n_entries = scan(cnt, cursor);
  // pass 2: fill (slot order inside a cell is arbitrary here ...)
  for (int px = tid; px < p.HoWo; px += THREADS) {
    const int oy = px / p.Wo, ox = px - oy * p.Wo;
    float y, x, m;
    tap_position(p, b, p.dgi, t, px, oy, ox, y, x, m);
    Tap tap;
    TapGeom geo;
    make_tap(y, x, p.H, p.W, true, m, tap, geo);
    const int valid[4] = {geo.va, geo.vb, geo.vc, geo.vd};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (valid[q]) {
        const int slot = atomicAdd(&cursor[tap.o[q]], 1);
        ent[slot] = make_int2(px, __float_as_int(tap.w[q]));
      }
  }
  __syncthreads();
  // ... so the SHORT lists (the ones that stay inline) are sorted by pixel here; two corners of one pixel never share a
  // cell.  Longer lists are summed in pixel order by dcn_inv_overflow_sums without being sorted.
  for (int cell = tid; cell < HW; cell += THREADS) {
    const int n = cnt[cell];
    if (n > kInvInline) continue;
    const int e1 = cursor[cell];  // end (cursor advanced by the fill)
    const int e0 = e1 - n;
    for (int i = e0 + 1; i < e1; ++i) {
      const int2 key = ent[i];
      int j = i - 1;
      while (j >= e0 && ent[j].x > key.x) { ent[j + 1] = ent[j]; --j; }
      ent[j + 1] = key;
    }
  }
  // pixel -> LDS byte offset of its quad 0 (< 2^17: the upper 15 bits of a record's last offset are free)
  auto plane_off = [](int px) { return (unsigned)dcn_plane_offset(px); };
  uint4 *inv_bt = inv + (size_t)(b * p.K + t) * HW * 4;
  int *extra = reinterpret_cast<int *>(ent + (size_t)4 * p.HoWo);  // [HW] ints right behind the entries (LDS sized for it)
  int *epos = extra + HW;                                          // [HW]
  for (int cell = tid; cell < HW; cell += THREADS) extra[cell] = cnt[cell] > kInvInline ? 1 : 0;
  __syncthreads();
  const int n_flagged = scan(extra, epos);     // epos[cell] = the cell's slot among the (image, tap)'s long cells
  const size_t bt = (size_t)(b * p.K + t);
  if (tid == 0) hdr[bt] = n_flagged < max_slots ? n_flagged : max_slots;   // (cannot exceed the bound: 9 entries per long cell)
  for (int cell = tid; cell < HW; cell += THREADS) {
    const int n = cnt[cell], e0 = cursor[cell] - n;
    const bool long_cell = n > kInvInline;
    unsigned off[8];
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int2 e = ent[e0 + min(i, max(n - 1, 0))];
      const bool live = !long_cell && i < n;
      off[i] = live ? plane_off(e.x) : plane_off(0);   // (weight 0; a plane address all the same: 0 x garbage could be NaN)
      w[i] = live ? __int_as_float(e.y) : 0.0f;
    }
    if (long_cell) {
      const int slot = epos[cell];
      off[7] |= kInvFlag | ((unsigned)slot << 17);
      if (slot < max_slots) {
        // hot cells (> 64 contributions) join the (offset tensor, image)'s column list of dcn_hot_gemm: .pad = 1 (the counter was
        // zeroed by the launch's memset; the ORDER of the columns is arbitrary, a column's value is not)
        int handled = 0;
        if (hot_count != nullptr && n > KGDET_HOT_MIN) {
          const int idx = atomicAdd(&hot_count[b], 1);
          if (idx < hot_max) {
            hot_cols[(size_t)b * kHotMaxCols + idx] = make_int4(t, slot, e0, n);
            handled = 1;
          }
        }
        cells[bt * max_slots + slot] = DcnInvOvfCell{e0, n, cell, handled};
      }
    }
    uint4 *r = inv_bt + (size_t)cell * 4;
    r[0] = make_uint4(off[0], off[1], off[2], off[3]);
    r[1] = make_uint4(off[4], off[5], off[6], off[7]);
    r[2] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
    r[3] = make_uint4(__float_as_uint(w[4]), __float_as_uint(w[5]), __float_as_uint(w[6]), __float_as_uint(w[7]));
  }
  // the entry list leaves as it lies (coalesced); only the long cells' ranges of it are ever read
  if (n_flagged > 0) {
    int2 *spill_bt = spill + bt * 4 * p.HoWo;
    for (int i = tid; i < n_entries; i += THREADS) spill_bt[i] = ent[i];
  }
}

__global__ __launch_bounds__(256) void dcn_build_inverse_taps(const DcnProblem p, uint4 *__restrict__ inv, int *__restrict__ hdr,
                                                              DcnInvOvfCell *__restrict__ cells, int2 *__restrict__ spill) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  build_inverse_taps_body<256>(p, inv, hdr, cells, spill, (int)blockIdx.x, sm);
}

// the inverse tables of several problems in ONE launch: block (x, y) = (image, tap) x of problem y.  (A problem has N * K
// blocks -- 18 / 50 / 98 for the three kernel sizes of a KGDet head stage at B = 2: launched one after the other they left
// most of the chip idle three times over, 3 x 33 us.)
__global__ __launch_bounds__(256) void dcn_build_inverse_taps_multi(const DcnInvBuildGroup grp) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  const DcnInvBuild &e = grp.e[blockIdx.y];
  if ((int)blockIdx.x >= e.p.N * e.p.K) return;
  build_inverse_taps_body<256>(e.p, e.inv, e.hdr, e.cells, e.spill, (int)blockIdx.x, sm, e.hot_cols, e.hot_count, e.hot_max);
}

// unit (problem uz, (image, tap) ubt, split uy) of a workgroup of the sums kernels: grid = (N * K, kInvSumSplit, problems), or with
// XCD-local units (DcnInvSumSched) linear workgroup id L -> XCD L % 8, unit r = L / 8 of that XCD's segments.  false: no unit.
__device__ __forceinline__ bool inv_sum_unit(const DcnInvSumGroup &grp, int &uz, int &ubt, int &uy) {
  uz = blockIdx.z; ubt = blockIdx.x; uy = blockIdx.y;
  if (grp.sched.on) {
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    int si = 0;
    const int ns = grp.sched.n_seg[xcd];
    while (si < ns && r >= grp.sched.seg[xcd][si].first + grp.sched.seg[xcd][si].n) ++si;
    if (si >= ns) return false;
    const DcnInvSumSeg sg = grp.sched.seg[xcd][si];
    const int u = r - sg.first + sg.u0;
    uz = sg.z;
    ubt = sg.b * grp.e[sg.z].K + u / kInvSumSplit;
    uy = u % kInvSumSplit;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// Gov[image, tap, slot][o] = sum over the contributions e of a long cell of w_e * grad_out[image, o, p_e], all output
// channels o of the convolution at once (the plane kernel would walk the list once per 16-channel chunk).
// grid = (N * K, kInvSumSplit, problems): workgroup (bt, y) takes the slots y, y + kInvSumSplit, ... of (image, tap) bt.
// Per cell: the entries are scattered into a dense by-pixel weight array + a bitmap (a pixel contributes at most once to a
// cell), the set bits are ranked (popcount prefix) into an ascending pixel list, sixteen 16-lane groups sum sixteenths of
// that list over all channels (a lane: 4 x 4 channels, 16-byte loads from the PIXEL-MAJOR copy of grad_out: 1 KB
// contiguous per pixel at 256 channels), the sixteen partial vectors are added in group order.  Fixed order => bit-
// repeatable; work proportional to the number of contributions, wherever they fall.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_inv_overflow_sums(const DcnInvSumGroup grp) {
  __shared__ float wd[4 * kPlaneMaxHW];
  __shared__ unsigned short sorted[kPlaneMaxHW];
  __shared__ unsigned bitmap[64];
  __shared__ int prefix[64];
  __shared__ float part[16][256];
  int uz, ubt, uy;
  if (!inv_sum_unit(grp, uz, ubt, uy)) return;
  const DcnInvSum &e = grp.e[uz];
  const int bt = ubt;
  if (bt >= e.NK) return;
  const int count = e.hdr[bt];
  if (uy >= count) return;
  const int tid = threadIdx.x;
  const int b = bt / e.K;
  const int HoWo = e.HoWo, O = e.O, O_ld = e.O_ld;
  const int words = (HoWo + 31) >> 5;
  const int2 *spill_bt = e.spill + (size_t)bt * 4 * HoWo;
  const float *gt = e.gout_t + (size_t)b * HoWo * O;
  const int g16 = tid >> 4, l = tid & 15;
  const bool vec = (O & 3) == 0;
  // the (image, tap)'s long cells, once, into LDS (round 6: the loops below read them one dependent global load at a time -- the
  // medium-cell loop in every wave, the cluster rule in ONE thread: with converged key points that serial chain, not the row reads,
  // was the kernel's time: 134 us per head stage whatever XCD the rows came from)
  // (an image with more hot cells than the column list holds goes through the cluster path WHOLE: which cells made the list depends on
  // the order the builder's workgroups ran in, and the two paths sum in different orders)
  const bool hot_on = grp.hot_gemm && e.hot_count != nullptr && e.hot_count[b] <= e.hot_max;
  if (hot_on) return;       // (every cell above 64 contributions of this image is a column of the GEMM: nothing left for this kernel)
  constexpr int kMaxCells = 4 * kPlaneMaxHW / (kInvInline + 1) + 8;
  __shared__ int c_start[kMaxCells];
  __shared__ short c_n[kMaxCells], c_cell[kMaxCells];   // (c_n < 0: the cell's sum is formed by dcn_hot_gemm)
  const int n_cells = count < kMaxCells ? count : kMaxCells;
  for (int i = tid; i < n_cells; i += 256) {
    const DcnInvOvfCell c = e.cells[(size_t)bt * e.max_slots + i];
    c_start[i] = c.start;
    c_n[i] = (short)((hot_on && c.pad) ? -1 : c.n);
    c_cell[i] = (short)c.cell;
  }
  __syncthreads();
  // (cells with at most 64 contributions: dcn_inv_medium_sums, below)
  // ---- longer lists: the whole workgroup per CLUSTER of up to four such cells (q, q + 1, q + W, q + W + 1: the four bilinear
  // corners of a sample).  With converged key points those four cells list (nearly) the same pixels, each with its own corner
  // weight: the cluster is summed in ONE pass over the union of the pixels -- a pixel's 1 KB row of grad_out is loaded once and
  // multiplied into up to four accumulator sets (cell by cell the hot cells read 1.4 GB of rows per head stage: this kernel's
  // time is those reads).  Clusters are formed greedily in cell order by one thread, identically in every workgroup of the
  // (image, tap); at most 4 * HoWo / 65 < 96 cells can have more than 64 contributions.
  __shared__ short big[96];
  __shared__ short members[96][4];
  __shared__ int n_clusters_s;
  if (tid == 0) {
    int nb = 0;
    for (int sl = 0; sl < n_cells && nb < 96; ++sl)
      if (c_n[sl] > 64) big[nb++] = (short)sl;
    unsigned long long taken0 = 0ull, taken1 = 0ull;
    auto is_taken = [&](int i) { return i < 64 ? (taken0 >> i) & 1ull : (taken1 >> (i - 64)) & 1ull; };
    auto take = [&](int i) { if (i < 64) taken0 |= 1ull << i; else taken1 |= 1ull << (i - 64); };
    int nc = 0;
    for (int i = 0; i < nb; ++i) {
      if (is_taken(i)) continue;
      take(i);
      const int q = c_cell[big[i]], x = q % e.W;
      short m[4] = {big[i], -1, -1, -1};
      int k = 1;
      const int want[3] = {x + 1 < e.W ? q + 1 : -1, q + e.W, x + 1 < e.W ? q + e.W + 1 : -1};
      for (int j = i + 1; j < nb && k < 4; ++j) {
        if (is_taken(j)) continue;
        const int cj = c_cell[big[j]];
        if (cj > q + e.W + 1) break;
        if (cj == want[0] || cj == want[1] || cj == want[2]) { m[k++] = big[j]; take(j); }
      }
      members[nc][0] = m[0]; members[nc][1] = m[1]; members[nc][2] = m[2]; members[nc][3] = m[3];
      ++nc;
    }
    n_clusters_s = nc;
  }
  __syncthreads();
#ifdef KGDET_SUMS_ABL_NOCLUSTER
  const int n_clusters = 0;
#else
  const int n_clusters = n_clusters_s;
#endif
  float *wd4 = wd;                  // [4][HoWo]: by-pixel weights of the members
  for (int cl = uy; cl < n_clusters; cl += kInvSumSplit) {
    int slot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) slot[k] = members[cl][k];
    const int nm = 1 + (slot[1] >= 0) + (slot[2] >= 0) + (slot[3] >= 0);    // (members are packed to the front)
    __syncthreads();
    if (tid < 64) bitmap[tid] = 0;
    for (int i = tid; i < nm * HoWo; i += 256) wd4[i] = 0.0f;
    __syncthreads();
    for (int k = 0; k < nm; ++k) {
      const DcnInvOvfCell c = {c_start[slot[k]], (int)c_n[slot[k]], (int)c_cell[slot[k]], 0};
      for (int i = tid; i < c.n; i += 256) {
        const int2 en = spill_bt[c.start + i];
        wd4[k * HoWo + en.x] = __int_as_float(en.y);
        atomicOr(&bitmap[en.x >> 5], 1u << (en.x & 31));
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int pc = tid < words ? __popc(bitmap[tid]) : 0;
      int incl = pc;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int nb = __shfl_up(incl, d);
        if (tid >= d) incl += nb;
      }
      prefix[tid] = incl - pc;
    }
    __syncthreads();
    const int n_px = prefix[words - 1] + __popc(bitmap[words - 1]);
    for (int px = tid; px < HoWo; px += 256) {
      const unsigned m = bitmap[px >> 5];
      if ((m >> (px & 31)) & 1u) sorted[prefix[px >> 5] + __popc(m & ((1u << (px & 31)) - 1u))] = (unsigned short)px;
    }
    __syncthreads();
#ifdef KGDET_SUMS_ABL_NOLOOP
    const int r0 = 0, r1 = n_px > 0 ? 1 : 0;
#else
    const int r0 = (int)((long long)g16 * n_px / 16), r1 = (int)((long long)(g16 + 1) * n_px / 16);
#endif
    for (int c0 = 0; c0 < O; c0 += 256) {
      f32x4 acc[4][4];   // [member][channel block j]
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int r = r0; r < r1; ++r) {
        const int px = sorted[r];
        const float *src = gt + (size_t)px * O;
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ch = c0 + j * 64 + l * 4;
          v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef KGDET_SUMS_ABL_NOLOAD
          v[j][0] = (float)px;
          continue;
#endif
          if (vec) {
            if (ch < O) v[j] = *reinterpret_cast<const f32x4 *>(src + ch);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (ch + i < O) v[j][i] = src[ch + i];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k >= nm) continue;
          const float w = wd4[k * HoWo + px];
          if (w == 0.0f) continue;          // (the pixel does not feed this member)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[k][j][0] = __builtin_fmaf(w, v[j][0], acc[k][j][0]);
            acc[k][j][1] = __builtin_fmaf(w, v[j][1], acc[k][j][1]);
            acc[k][j][2] = __builtin_fmaf(w, v[j][2], acc[k][j][2]);
            acc[k][j][3] = __builtin_fmaf(w, v[j][3], acc[k][j][3]);
          }
        }
      }
      for (int k = 0; k < nm; ++k) {      // the members' sixteen partial vectors, one member at a time through `part`
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          if (kk == k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(&part[g16][j * 64 + l * 4]) = acc[kk][j];
          }
        __syncthreads();
        // a Gov vector holds the convolution's weight groups one after the other, each padded to whole 16-channel chunks
        // (zeros: the chunk's padding channels meet zero weights, but must be finite)
        float *dst = e.gov + ((size_t)bt * e.max_slots + slot[k]) * O_ld;
        if (c0 == 0)
          for (int d = tid; d < O_ld; d += 256)
            if (d % e.Og_pad16 >= e.Og) dst[d] = 0.0f;
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) sum += part[g][tid];
        if (c0 + tid < O) dst[((c0 + tid) / e.Og) * e.Og_pad16 + (c0 + tid) % e.Og] = sum;
        __syncthreads();
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Gov of the cells with 9 .. 64 contributions: ONE WAVE per cell, no workgroup barrier in the loop.  In the training step's own
// steady state (tools/dump_step_offsets.py: the offsets after 700 steps on the bench's batch) no cell is hot, but a quarter of all
// contributions sit in ~76 such cells per (image, tap) -- 25 000 cell sums and 360 MB of grad_output rows per head stage.  As a
// section of dcn_inv_overflow_sums (45 KB of LDS for the cluster path: 3 workgroups per CU) this took 71 us of that kernel's 91: a
// chain of three dependent memory latencies per cell at 12 waves per CU.  Here: 3 KB of LDS (8 workgroups per CU), the workgroup's
// medium cells compacted into a list first, the NEXT cell's entries requested before the current cell's rows, up to 16 rows in flight.
// Lane i holds entry i; its rank = the number of entries with a smaller pixel (pixels of a cell are distinct); the wave walks the
// entries in rank order, lanes = 64 channel quads (one 16-byte load per entry and lane).  Same units as dcn_inv_overflow_sums
// (XCD-local: the rows come from the L2 that holds the image's pixel-major grad_output).  Deterministic.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcn_inv_medium_sums(const DcnInvSumGroup grp) {
  constexpr int kMaxCells = 4 * kPlaneMaxHW / (kInvInline + 1) + 8;
  constexpr int kMine = (kMaxCells + kInvSumSplit - 1) / kInvSumSplit;     // cells of one split (<= 256: one per thread)
  static_assert(kMine <= 256, "one thread per cell of the split");
  __shared__ int m_start[kMine];
  __shared__ unsigned short m_n[kMine], m_slot[kMine];
  __shared__ unsigned short w_px[4][64];
  __shared__ float w_w[4][64];
  __shared__ int wave_count[4];
  int uz, ubt, uy;
  if (!inv_sum_unit(grp, uz, ubt, uy)) return;
  const DcnInvSum &e = grp.e[uz];
  const int bt = ubt;
  if (bt >= e.NK) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this split's cells, one per thread -- requested together with the header that says how many of them are real (two dependent
  // round trips to memory the builder wrote from another XCD would be ~4 us before the first useful load)
  const int slot = uy + tid * kInvSumSplit;
  DcnInvOvfCell c = {0, 0, 0, 0};
  if (slot < e.max_slots && slot < kMaxCells) c = e.cells[(size_t)bt * e.max_slots + slot];
  const int count = e.hdr[bt];
  if (uy >= count) return;
  const int b = bt / e.K;
  const int HoWo = e.HoWo, O = e.O, O_ld = e.O_ld;
  const int2 *spill_bt = e.spill + (size_t)bt * 4 * HoWo;
  const float *gt = e.gout_t + (size_t)b * HoWo * O;
  const bool vec = (O & 3) == 0;
  const int n_cells = count < kMaxCells ? count : kMaxCells;
  // the medium ones compacted in slot order (ballot ranks inside a wave, wave totals across)
  const bool medium = slot < n_cells && c.n > 0 && c.n <= 64;
  const unsigned long long mask = __ballot(medium);
  if (lane == 0) wave_count[wave] = __popcll(mask);
  __syncthreads();
  int base = 0, n_medium = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    base += w < wave ? wave_count[w] : 0;
    n_medium += wave_count[w];
  }
  if (medium) {
    const int at = base + __popcll(mask & ((1ull << lane) - 1ull));
    m_start[at] = c.start;
    m_n[at] = (unsigned short)c.n;
    m_slot[at] = (unsigned short)slot;
  }
  __syncthreads();
#ifdef KGDET_SUMS_ABL_NOMEDIUM
  return;
#endif
  auto entries = [&](int i) {
    int2 en = make_int2(0x7fffffff, 0);
#ifdef KGDET_MED_ABL_NOENTRIES
    if (i < n_medium && lane < (int)m_n[i]) en = make_int2(lane * 16 + i, 0x3f800000);
    return en;
#endif
    if (i < n_medium && lane < (int)m_n[i]) en = spill_bt[m_start[i] + lane];
    return en;
  };
  int2 en_next = entries(wave);
  if (vec && O_ld == O && O <= 256) {
    // the head's shapes (<= 256 channels, no padding inside a Gov vector).  Branch-free: every lane loads 16 bytes of every row from
    // a clamped channel offset (lanes past O compute and drop).  Results wait in LDS and leave four cells at a time: with loads AND
    // stores in flight the memory counter cannot tell them apart, so a wait for a cell's rows right after the previous cell's store
    // waited for that store's whole round trip too (~4 us per cell measured: more than the rows themselves).
    __shared__ f32x4 res[4][4][64];
    __shared__ unsigned short res_slot[4][4];
    // (buffer loads: one multiply-add per row address instead of a 64-bit pointer computation -- a third of the kernel's vector
    // instructions were address arithmetic)
    const dcn_rsrc_t g_rsrc = dcn_make_rsrc(gt);
    const unsigned row_bytes = (unsigned)O * 4u, ch_bytes = (unsigned)min(lane * 4, O - 4) * 4u;
#ifndef KGDET_MED_ROWS
#define KGDET_MED_ROWS 12
#endif
    constexpr int kRows = KGDET_MED_ROWS;              // rows in flight (48 registers: six waves per SIMD stay resident; 10: 37.3, 12: 36.4, 16: 44.8 us)
    int pend = 0;
    auto flush = [&]() __attribute__((always_inline)) {
      for (int q = 0; q < pend; ++q)
        if (lane * 4 < O)
          *reinterpret_cast<f32x4 *>(e.gov + ((size_t)bt * e.max_slots + res_slot[wave][q]) * O_ld + lane * 4) = res[wave][q][lane];
      pend = 0;
    };
    for (int i = wave; i < n_medium; i += 4) {
      const int2 en = en_next;
      const int n = m_n[i];
      en_next = entries(i + 4);                        // (in flight under this cell's rows)
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += __builtin_amdgcn_readlane(en.x, j) < en.x ? 1 : 0;
      __builtin_amdgcn_wave_barrier();                 // (the previous cell's reads of the slots are done: same wave, in order)
      if (lane < n) {
        w_px[wave][rank] = (unsigned short)en.x;
        w_w[wave][rank] = __int_as_float(en.y);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int r0 = 0; r0 < n; r0 += kRows) {
        f32x4 v[kRows];
        float w[kRows];
#pragma unroll
        for (int u = 0; u < kRows; ++u) {              // (a batch's surplus rows re-read the cell's last row with weight 0)
          const int idx = min(r0 + u, n - 1);
          w[u] = r0 + u < n ? w_w[wave][idx] : 0.0f;
#ifdef KGDET_MED_ABL_NOROWS
          v[u] = f32x4{(float)idx, 0.f, 0.f, 0.f}; continue;
#endif
          v[u] = __builtin_bit_cast(f32x4, dcn_buf_b128(g_rsrc, (unsigned)w_px[wave][idx] * row_bytes + ch_bytes, 0));
        }
#pragma unroll
        for (int u = 0; u < kRows; ++u) {
          acc[0] = __builtin_fmaf(w[u], v[u][0], acc[0]);
          acc[1] = __builtin_fmaf(w[u], v[u][1], acc[1]);
          acc[2] = __builtin_fmaf(w[u], v[u][2], acc[2]);
          acc[3] = __builtin_fmaf(w[u], v[u][3], acc[3]);
        }
      }
      res[wave][pend][lane] = acc;
      if (lane == 0) res_slot[wave][pend] = m_slot[i];
      if (++pend == 4) flush();
    }
    flush();
    return;
  }
  for (int i = wave; i < n_medium; i += 4) {           // any other shape: a cell at a time
    const int2 en = en_next;
    const int n = m_n[i];
    en_next = entries(i + 4);
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += __builtin_amdgcn_readlane(en.x, j) < en.x ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
      w_px[wave][rank] = (unsigned short)en.x;
      w_w[wave][rank] = __int_as_float(en.y);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    // a Gov vector holds the convolution's weight groups one after the other, each padded to whole 16-channel chunks (zeros: the
    // chunk's padding channels meet zero weights, but must be finite)
    float *dst = e.gov + ((size_t)bt * e.max_slots + m_slot[i]) * O_ld;
    if (O_ld != O)
      for (int d = lane; d < O_ld; d += 64)
        if (d % e.Og_pad16 >= e.Og) dst[d] = 0.0f;
    for (int c0 = 0; c0 < O; c0 += 256) {
      const int ch = c0 + lane * 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int r = 0; r < n; ++r) {
        const int px = w_px[wave][r];
        const float w = w_w[wave][r];
        const float *src = gt + (size_t)px * O;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
          if (ch < O) v = *reinterpret_cast<const f32x4 *>(src + ch);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ch + k < O) v[k] = src[ch + k];
        }
        acc[0] = __builtin_fmaf(w, v[0], acc[0]);
        acc[1] = __builtin_fmaf(w, v[1], acc[1]);
        acc[2] = __builtin_fmaf(w, v[2], acc[2]);
        acc[3] = __builtin_fmaf(w, v[3], acc[3]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ch + k < O) dst[((ch + k) / e.Og) * e.Og_pad16 + (ch + k) % e.Og] = acc[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Hot cells as a GEMM (round 6).  With converged key points every (image, tap) has a handful of cells that collect hundreds of
// contributions each, and the cluster path above reads every contributing pixel's 1 KB row of grad_out once per cluster: the same
// ~1050 rows for every tap of an image, 349 MB per head stage at ~8 TB/s of L2 hits + 357 M lane-FMAs (tools/experiments/README.md,
// round 6).  Here the hot cells of ALL taps of an (offset tensor, image) are columns of one product over the pixels,
//     Gov[(t, slot)][o] = sum_px Wd[col][px] * grad_out[o][px],       Wd[col][px] = the cell's weight for pixel px (0: none),
// rows read once per 32 columns: v_mfma_f32_32x32x16_bf16 on hi / lo parts split on the fly (3 products, fp32 accumulate: the
// arithmetic of the plane kernels).  The inverse-record builder lists the hot cells as it meets them (one atomic counter per (offset
// tensor, image): the ORDER of the columns is arbitrary, every column's value is not) and marks them handled.  A cell is hot above 64
// contributions, so a tap has at most 4 HoWo / 65 of them: kHotMaxCols = 4096 holds every image of a 7x7 kernel on a plane-sized map;
// an image that overflows the list (larger kernels) takes the cluster path whole.  dcn_hot_gemm: workgroup = (32-column tile,
// 128-channel part) of a (problem, image), placed on the XCD that the sums kernel reads that image's grad_output on; the workgroup
// scatters its columns' weights into a dense LDS tile [32][pixels] (the A operand), its eight waves split the k-steps, and the eight
// partial tiles are summed in a fixed order.  Deterministic.  (A dense product multiplies every pixel of the image with a cell's
// weight, zero for the pixels that do not contribute: an Inf / NaN anywhere in an image's grad_output reaches all hot cells of that image
// as NaN, where the reference's scatter would confine it to the cells its pixel feeds -- a gradient that is lost either way.)
// ------------------------------------------------------------------------------------------------
size_t dcn_hot_gemm_lds_bytes() { return (size_t)32 * kHotRowLd * sizeof(float); }

__global__ __launch_bounds__(512) void dcn_hot_gemm(const DcnHotGemmGroup grp, const DcnInvSumSched sched) {
  typedef __bf16 bf16x8h __attribute__((ext_vector_type(8)));
  extern __shared__ __attribute__((aligned(16))) float a_tile[];    // [32 columns][kHotRowLd]: by-pixel weights of the tile's cells
  __shared__ int s_bt[32], s_slot[32];
  // workgroup L runs on XCD L % 8: it takes the units (32-column tile, 128-channel part) r, r + 32, ... of that XCD's (problem, image)
  // groups laid end to end (how many units a group has is known only here: the builder counted its hot cells)
  const int xcd = (int)blockIdx.x & 7, r = (int)blockIdx.x >> 3, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, kb = lane >> 5;
  // a wave: all 128 channels of the unit x an eighth of the k-steps.  Lane (r32, kb) reads, per pixel, the FOUR channels 4 r32 .. 4 r32 + 3
  // with one 16-byte load (a wave load instruction costs the texture addresser 16 cycles whatever its width: with 4-byte loads that
  // rate, not the arithmetic, set the kernel's time), so the unit's four 32-wide MFMA blocks are the channels {4 n + c}, c = 0 .. 3.
  auto split = [](const float *v, bf16x8h &hi, bf16x8h &lo) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hi[j] = (__bf16)v[j];
      lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
  };
  int counts[kInvSumSegs];                            // (all of the XCD's groups at once: one memory latency, not one per group)
#pragma unroll
  for (int si = 0; si < kInvSumSegs; ++si)
    counts[si] = (si < sched.n_seg[xcd] && sched.seg[xcd][si].u0 == 0)     // (a pair's second segment: its columns are the first one's)
                     ? grp.e[sched.seg[xcd][si].z].count[sched.seg[xcd][si].b] : 0;
  int base = 0;
  for (int si = 0; si < sched.n_seg[xcd]; ++si) {
    const DcnHotGemm &e = grp.e[sched.seg[xcd][si].z];
    const int b = sched.seg[xcd][si].b;
    const int count = counts[si] <= e.max_cols ? counts[si] : 0;     // (too many for the list: the image takes the cluster path whole)
    const int o_parts = (e.O + 127) / 128;
    const int units = (count + 31) / 32 * o_parts;
    const int HoWo = e.HoWo, O = e.O;
    const dcn_rsrc_t g_rsrc = dcn_make_rsrc(e.gout_t + (size_t)b * HoWo * O);
    for (int u = ((r - base) % kHotBlocksPerXcd + kHotBlocksPerXcd) % kHotBlocksPerXcd; u < units; u += kHotBlocksPerXcd) {
      const int tile = u / o_parts, oh = u - tile * o_parts;
      const int col0 = tile * 32;
      const int oc = oh * 128 + 4 * r32;               // the lane's four channels
      // the tile's cells (for the scatter: 16 threads per column)
      const int4 my_col = (col0 + (tid >> 4) < count) ? e.cols[(size_t)b * kHotMaxCols + col0 + (tid >> 4)] : make_int4(0, 0, 0, 0);
      __syncthreads();                                 // (the previous unit's reads of the tile and of s_bt are done)
      if (tid < 32) {
        const int4 c = e.cols[(size_t)b * kHotMaxCols + min(col0 + tid, count - 1)];
        s_bt[tid] = b * e.K + c.x;
        s_slot[tid] = c.y;
      }
      f32x16 acc[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
      const unsigned row_bytes = (unsigned)O * 4u;
      const unsigned ch = (unsigned)min(oc, O - 4) * 4u;   // (channels past O: computed, not stored)
      const unsigned voff = (unsigned)(8 * kb) * row_bytes + ch;
      for (int pk0 = 0; pk0 < HoWo; pk0 += kHotRange) {   // (maps of up to kHotRange pixels: one pass)
        const int len = min(HoWo - pk0, kHotRange);
        const int nsteps = (len + 15) >> 4;
        // k-steps of 16 pixels: A from the LDS tile (lane = column r32, pixels 8 kb .. 8 kb + 7), B = the pixel-major grad_output
        // through buffer loads (lane offset + a scalar offset per pixel row: no vector address arithmetic); the loads of kHotAhead
        // steps are in flight while the previous kHotAhead steps multiply.  Only the last step of a map can hold pixels past its
        // end: that one clamps per lane and reads them as zeros; steps past the range are skipped.
#ifndef KGDET_HOT_AHEAD
#define KGDET_HOT_AHEAD 1
#endif
        constexpr int kHotAhead = KGDET_HOT_AHEAD;
        // (the first steps' loads go out before the tile is built: they do not depend on it)
        const float *arow = a_tile + r32 * kHotRowLd + 8 * kb;
        const int full_steps = len >> 4;
        auto load_b = [&](int ks, u32x4_t *v) __attribute__((always_inline)) {
          if (ks < full_steps) {
            const unsigned s0 = (unsigned)(pk0 + ks * 16) * row_bytes;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = dcn_buf_b128(g_rsrc, voff, s0 + j * row_bytes);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int px = pk0 + ks * 16 + 8 * kb + j;
              const bool in = px < HoWo && ks < nsteps;
              const u32x4_t x = dcn_buf_b128(g_rsrc, (unsigned)min(px, HoWo - 1) * row_bytes + ch, 0);
              v[j] = in ? x : u32x4_t{0u, 0u, 0u, 0u};
            }
          }
        };
        u32x4_t cur[kHotAhead][8], nxt[kHotAhead][8];
#pragma unroll
        for (int q = 0; q < kHotAhead; ++q) load_b(wave + 8 * q, cur[q]);
        if (pk0 > 0) __syncthreads();
        {
          float4 *z = reinterpret_cast<float4 *>(a_tile);
          for (int i = tid; i < 32 * kHotRowLd / 4; i += 512) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        {   // scatter: a pixel contributes at most once to a cell, so no two entries of a column share a slot
          const int2 *sp = e.spill + ((size_t)b * e.K + my_col.x) * 4 * HoWo + my_col.z;
          float *row = a_tile + (tid >> 4) * kHotRowLd;
          const int sub = tid & 15, n = my_col.w;
#ifdef KGDET_HOT_ABL_NOSCATTER
          if (n > 100000)
#endif
          for (int i0 = sub; i0 < n; i0 += 16 * 8) {
            int2 en[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) en[q] = (i0 + 16 * q < n) ? sp[i0 + 16 * q] : make_int2(-1, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int px = en[q].x - pk0;
              if (en[q].x >= 0 && px >= 0 && px < len) row[px] = __int_as_float(en[q].y);
            }
          }
        }
        __syncthreads();
#ifdef KGDET_HOT_ABL_NOGEMM
        if (nsteps > 100000)
#endif
        for (int ks = wave; ks < nsteps; ks += 8 * kHotAhead) {
#pragma unroll
          for (int q = 0; q < kHotAhead; ++q) load_b(ks + 8 * (kHotAhead + q), nxt[q]);
#pragma unroll
          for (int q = 0; q < kHotAhead; ++q) {
            const int k = ks + 8 * q;
            if (k >= nsteps) break;                        // (wave-uniform)
            float a[8];
            const float4 a_lo = *reinterpret_cast<const float4 *>(arow + k * 16);
            const float4 a_hi = *reinterpret_cast<const float4 *>(arow + k * 16 + 4);
            a[0] = a_lo.x; a[1] = a_lo.y; a[2] = a_lo.z; a[3] = a_lo.w; a[4] = a_hi.x; a[5] = a_hi.y; a[6] = a_hi.z; a[7] = a_hi.w;
            bf16x8h ah, al, bh, bl;
            split(a, ah, al);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = __uint_as_float(cur[q][j][c]);
              split(v, bh, bl);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[c], 0, 0, 0);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[c], 0, 0, 0);
              acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[c], 0, 0, 0);
            }
          }
#pragma unroll
          for (int q = 0; q < kHotAhead; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[q][j] = nxt[q][j];
        }
      }
      // the eight k-parts, summed in a fixed order through the (now free) tile: part[w][c][i][lane]; wave w sums and stores the
      // accumulator rows 2 w, 2 w + 1 of every block
      __syncthreads();
      float *part = a_tile;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[((wave * 4 + c) * 16 + i) * 64 + lane] = acc[c][i];
      __syncthreads();
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * wave + ii;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = part[((0 * 4 + c) * 16 + i) * 64 + lane];
#pragma unroll
          for (int w = 1; w < 8; ++w) v[c] += part[((w * 4 + c) * 16 + i) * 64 + lane];
        }
        const int row = mfma_row(i, lane);
        if (col0 + row < count && oc < O)
          *reinterpret_cast<float4 *>(e.gov + ((size_t)s_bt[row] * e.max_slots + s_slot[row]) * e.O_ld + oc) =
              make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    base += units;
  }
}

// grad_out windows [n][O of O_total][P] -> pixel-major copies [n][P][O], several problems in one launch
// (blockIdx.z = problem * N + image): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void dcn_gout_pixel_major_multi(const DcnPixelMajorGroup grp) {
  __shared__ float tile[32][33];
  int z = blockIdx.z, pi = 0;
  while (pi + 1 < grp.n && z >= grp.e[pi].N) { z -= grp.e[pi].N; ++pi; }
  const DcnPixelMajorItem &it = grp.e[pi];
  const int n = z;
  const int P = it.P, C = it.C;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  if (n >= it.N || p0 >= P || c0 >= C) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float *s = it.src + (size_t)n * it.src_image_stride;
  float *d = it.dst + (size_t)n * P * C;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, px = p0 + tx;
    tile[r][tx] = (c < C && px < P) ? s[(size_t)c * P + px] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int px = p0 + r, c = c0 + tx;
    if (px < P && c < C) d[(size_t)px * C + c] = tile[tx][r];
  }
}

// Everything grad_input needs before its sums and its plane kernel, ONE launch (round 6): workgroups [0, build_blocks * builds.n)
// build the inverse records of the distinct offset tensors (512 threads per (image, tap): the builder is a chain of short
// block-wide phases), the rest transpose the grad_output windows into their pixel-major copies -- two launches of round 5, of which
// the first kept 166 workgroups busy for 30 us and left the rest of the chip idle.
__global__ __launch_bounds__(512) void dcn_bwd_input_prepare(const DcnInvBuildGroup grp, const DcnPixelMajorGroup pm, int build_blocks,
                                                            int pm_bx, int pm_by) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  __shared__ float tile[2][32][33];
  const int n_build = build_blocks * grp.n;
  if ((int)blockIdx.x < n_build) {
    const int y = (int)blockIdx.x / build_blocks, x = (int)blockIdx.x - y * build_blocks;
    const DcnInvBuild &e = grp.e[y];
    if (x >= e.p.N * e.p.K) return;
    build_inverse_taps_body<512>(e.p, e.inv, e.hdr, e.cells, e.spill, x, sm, e.hot_cols, e.hot_count, e.hot_max);
    return;
  }
  // two 32 x 32 tiles per workgroup (one per 256 threads)
  const int half = threadIdx.x >> 8, t256 = threadIdx.x & 255;
  const int u = ((int)blockIdx.x - n_build) * 2 + half;
  int z = u / (pm_bx * pm_by);
  const int rem = u - z * (pm_bx * pm_by);
  const int by = rem / pm_bx, bx = rem - by * pm_bx;
  int pi = 0;
  while (pi + 1 < pm.n && z >= pm.e[pi].N) { z -= pm.e[pi].N; ++pi; }
  const DcnPixelMajorItem &it = pm.e[pi];
  const int P = it.P, C = it.C;
  const int p0 = bx * 32, c0 = by * 32;
  const bool live = z < it.N && p0 < P && c0 < C;     // (uniform per half; both halves reach the barrier)
  const int tx = t256 & 31, ty = t256 >> 5;  // 32 x 8
  const float *src = it.src + (size_t)z * it.src_image_stride;
  float *dst = it.dst + (size_t)z * P * C;
  if (live)
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r, px = p0 + tx;
      tile[half][r][tx] = (c < C && px < P) ? src[(size_t)c * P + px] : 0.0f;
    }
  __syncthreads();
  if (live)
    for (int r = ty; r < 32; r += 8) {
      const int px = p0 + r, c = c0 + tx;
      if (px < P && c < C) dst[(size_t)px * C + c] = tile[half][tx][r];
    }
}

size_t dcn_build_inverse_taps_lds_bytes(int HW, int HoWo) {
  return ((size_t)2 * HW + 2) * sizeof(int) + (size_t)4 * HoWo * 8 + (size_t)2 * HW * sizeof(int) + 16 * sizeof(int);
}

}  // namespace kgdet
