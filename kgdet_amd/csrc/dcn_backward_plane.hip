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
// pixel so every sum has a fixed order -- and stores, per cell, the first 8 pairs inline (DcnInvRec) and the rest
// in a per-(tap, 128-cell tile) overflow list (DcnInvOvfSlots + spill array).  dcn_bwd_input_plane is then
// plane_role<MODE = 1>: grad_out planes in LDS, weights from the transposed operand image `wqt`, bf16 hi/lo
// split MFMA, stream-K with the forward's fix-up.  No atomics, deterministic, no pre-zeroing of grad_input.
// Deformable groups > 1 (different cells lists per channel group inside one M tile) stay on the older kernels.
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
//   inv   [N][K][H*W] DcnInvRec     first 8 contributions of every input cell
//   slots [N][K][tiles] DcnInvOvfSlots  (tiles = ceil(H*W / 128))
//   spill [N][K][4 * Ho*Wo] entries  all overflow entries, tile after tile
// LDS: cnt [HW + 1], cursor [HW], ent [4 * HoWo] (pixel, weight) pairs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void build_inverse_taps_body(const DcnProblem &p, uint4 *__restrict__ inv,
                                                        DcnInvOvfSlots *__restrict__ slots, uint2 *__restrict__ spill,
                                                        int block, int *sm) {
  const int HW = p.H * p.W;
  int *cnt = sm;                                          // [HW + 1]
  int *cursor = sm + (HW + 1);                            // [HW]
  int2 *ent = reinterpret_cast<int2 *>(sm + 2 * HW + 2);  // [4 * HoWo]
  __shared__ int wave_tot[4];

  const int t = block % p.K, b = block / p.K;
  const int tid = threadIdx.x;
  const int n_tiles = (HW + kTileN - 1) / kTileN;

  for (int i = tid; i <= HW; i += 256) cnt[i] = 0;
  __syncthreads();
  // pass 1: count the corners landing in each cell
  for (int px = tid; px < p.HoWo; px += 256) {
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
    const int per = (HW + 255) / 256;
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
    const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return total;
  };
  scan(cnt, cursor);
  // pass 2: fill (slot order inside a cell is arbitrary here ...)
  for (int px = tid; px < p.HoWo; px += 256) {
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
  // ... so sort every cell's (short) list by pixel; two corners of one pixel never share a cell
  for (int cell = tid; cell < HW; cell += 256) {
    const int e1 = cursor[cell];  // end (cursor advanced by the fill)
    const int e0 = e1 - cnt[cell];
    for (int i = e0 + 1; i < e1; ++i) {
      const int2 key = ent[i];
      int j = i - 1;
      while (j >= e0 && ent[j].x > key.x) { ent[j + 1] = ent[j]; --j; }
      ent[j + 1] = key;
    }
  }
  __syncthreads();
  // pixel -> LDS byte offset of its quad 0 (< 2^17: the upper 15 bits of a record's last offset are free)
  auto plane_off = [](int px) { return (unsigned)dcn_plane_offset(px); };
  uint4 *inv_bt = inv + (size_t)(b * p.K + t) * HW * 4;
  // overflow: contributions 8.. of every cell, compacted in cell order; `cursor` becomes their start positions
  for (int cell = tid; cell < HW; cell += 256) {
    const int n = cnt[cell];
    cursor[cell] -= n;               // back to the list start (the records above are done with the end)
    cnt[cell] = n | (max(n - 8, 0) << 16);  // low half: list length, high half: overflow length
  }
  __syncthreads();
  // scan the overflow lengths (kept in a scratch view: reuse wave-private registers through a lambda on a copy)
  int *extra = reinterpret_cast<int *>(ent + (size_t)4 * p.HoWo);  // [HW] ints right behind the entries (LDS sized for it)
  int *epos = extra + HW;                                          // [HW]
  for (int cell = tid; cell < HW; cell += 256) extra[cell] = cnt[cell] >> 16;
  __syncthreads();
  scan(extra, epos);
  // inline records: the first 8 contributions of every cell.  The last offset also carries the cell's overflow range
  // -- count (5 bits) << 27 | start inside the (tile, tap)'s list (10 bits) << 17 -- so that a producer thread of the
  // grad_input kernel walks ITS OWN overflow entries instead of scanning the whole list (one dependent scalar load +
  // a divergent body per entry: 360 us against 210 for a head stage with N(0, 2^2)-pixel offsets).  A tile whose ranges
  // do not fit the fields gets a negative count: list scan, as before.
  int *tile_scan = epos + HW;     // [16] tiles whose ranges do not fit the fields (H*W <= 1536: at most 12 tiles)
  if (tid < 16) tile_scan[tid] = 0;
  __syncthreads();
  for (int cell = tid; cell < HW; cell += 256) {
    const int n = cnt[cell] & 0xffff, ne = cnt[cell] >> 16, e0 = cursor[cell];
    const int st = epos[cell] - epos[(cell / kTileN) * kTileN];
    if (ne > 31 || st + ne > 1023) tile_scan[cell / kTileN] = 1;
    unsigned off[8];
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int2 e = ent[e0 + min(i, max(n - 1, 0))];
      off[i] = i < n ? plane_off(e.x) : plane_off(0);   // (weight 0; a plane address all the same: 0 x garbage could be NaN)
      w[i] = i < n ? __int_as_float(e.y) : 0.0f;
    }
    off[7] |= ((unsigned)min(ne, 31) << 27) | ((unsigned)min(st, 1023) << 17);
    uint4 *r = inv_bt + (size_t)cell * 4;
    r[0] = make_uint4(off[0], off[1], off[2], off[3]);
    r[1] = make_uint4(off[4], off[5], off[6], off[7]);
    r[2] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
    r[3] = make_uint4(__float_as_uint(w[4]), __float_as_uint(w[5]), __float_as_uint(w[6]), __float_as_uint(w[7]));
  }
  uint2 *spill_bt = spill + (size_t)(b * p.K + t) * 4 * p.HoWo;
  DcnInvOvfSlots *slots_bt = slots + (size_t)(b * p.K + t) * n_tiles;
  for (int cell = tid; cell < HW; cell += 256) {
    const int n = cnt[cell] & 0xffff, ne = cnt[cell] >> 16;
    if (ne == 0) continue;
    const int tile = cell / kTileN, tile_start = epos[tile * kTileN];
    const int e0 = cursor[cell] + 8;
    for (int i = 0; i < ne; ++i) {
      const int2 e = ent[e0 + i];
      const uint2 v = make_uint2((plane_off(e.x) << 7) | (unsigned)(cell - tile * kTileN), (unsigned)e.y);
      const int pos = epos[cell] + i;
      spill_bt[pos] = v;
      if (pos - tile_start < DcnInvOvfSlots::kCap) slots_bt[tile].e[pos - tile_start] = v;
    }
    (void)n;
  }
  for (int tile = tid; tile < n_tiles; tile += 256) {
    const int s0 = epos[tile * kTileN];
    const int last = min((tile + 1) * kTileN, HW) - 1;
    const int s1 = epos[last] + extra[last];
    slots_bt[tile].count = tile_scan[tile] ? -(s1 - s0) : s1 - s0;   // < 0: ranges not in the records, scan the list
    slots_bt[tile].spill_start = (int)((size_t)(b * p.K + t) * 4 * p.HoWo) + s0;
  }
}

__global__ __launch_bounds__(256) void dcn_build_inverse_taps(const DcnProblem p, uint4 *__restrict__ inv,
                                                              DcnInvOvfSlots *__restrict__ slots,
                                                              uint2 *__restrict__ spill) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  build_inverse_taps_body(p, inv, slots, spill, (int)blockIdx.x, sm);
}

// the inverse tables of several problems in ONE launch: block (x, y) = (image, tap) x of problem y.  (A problem has N * K
// blocks -- 18 / 50 / 98 for the three kernel sizes of a KGDet head stage at B = 2: launched one after the other they left
// most of the chip idle three times over, 3 x 33 us.)
__global__ __launch_bounds__(256) void dcn_build_inverse_taps_multi(const DcnInvBuildGroup grp) {
  extern __shared__ __attribute__((aligned(16))) int sm[];
  const DcnInvBuild &e = grp.e[blockIdx.y];
  if ((int)blockIdx.x >= e.p.N * e.p.K) return;
  build_inverse_taps_body(e.p, e.inv, e.slots, e.spill, (int)blockIdx.x, sm);
}

size_t dcn_build_inverse_taps_lds_bytes(int HW, int HoWo) {
  return ((size_t)2 * HW + 2) * sizeof(int) + (size_t)4 * HoWo * 8 + (size_t)2 * HW * sizeof(int) + 16 * sizeof(int);
}

}  // namespace kgdet
