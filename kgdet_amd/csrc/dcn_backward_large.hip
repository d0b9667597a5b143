// Deformable convolution v1 / v2 backward (grad_input + grad_offset [+ grad_mask]) for feature maps too large for the LDS-plane kernels
// (config 5's stride-8 / stride-16 levels: [2, 256, 100, 168], [2, 256, 50, 84]).  No atomics, deterministic.
//
// Reference path replaced: deform_conv_backward_input_cuda (deform_conv_cuda.cpp:260-371): columns = W^T grad_out
// (:329-332), deformable_col2im_coord -> grad_offset (deform_conv_cuda_kernel.cu:337-435), deformable_col2im ->
// grad_input with one float atomicAdd per (channel, tap, pixel, corner) (:279-334).  Round 1 ran these maps on
// dcn_bwd_input_mfma, which keeps the atomic scatter; it was 7.9 ms of config 5's 51 ms of kernels per step.
//
// Here, with 288 GB of HBM, the column gradient IS materialised once -- transposed, channels innermost:
//     colT[b][p][t*C + c] = sum_o grad_out[b][o][p] * W[o][c][t]                                  (310 MB at 2x100x168)
// as ONE split-bf16 MFMA GEMM per image on the backbone's 1x1-convolution kernel (conv_nn<1>, csrc/conv1x1.hip) with
// the pixels in the role of output channels (A = grad_out^T packed as a weight image, "image" = the permuted weight),
// and both consumers read it in 1 KB runs:
//   * grad_offset: one workgroup per output pixel, a thread per channel: colT row x the four corner rows of x^T
//     (NHWC copy) x the derivative weights, block-reduced per tap;
//   * grad_input: the scatter turned around with an inverse index.  Every (image, tap, pixel, valid corner) becomes an
//     entry of its input cell's list: counted with integer atomics, storage handed out by a bump allocation, entries dropped
//     in (slot order arbitrary) and every list then RANKED by its entries' (tap, pixel) numbers -- unique inside a cell -- so
//     that every cell's sum has the enumeration order whatever order the atomics served the slots in (rounds 2-3: a stable
//     hipCUB radix sort of all entries; round 4: these four small kernels, no third-party primitive on the path); one
//     workgroup per 16 cells, a thread per channel, adds w * colT[p][t*C + c].
#include <limits.h>
#include <type_traits>

#include "common.h"
#include "dcn_common.h"

extern "C" int kgdet_conv_pack(const float *w, int32_t O, int32_t C, int32_t taps, int32_t transpose, void *packed,
                               void *stream);
extern "C" size_t kgdet_conv_packed_bytes(int32_t M, int32_t K, int32_t taps);
extern "C" size_t kgdet_conv_apply_workspace_bytes(int64_t B, int32_t M, int32_t K, int32_t H, int32_t W, int32_t taps,
                                                    int32_t stride);
extern "C" int kgdet_conv_apply_epilogue(const void *packed, const float *x, float *y, const float *bias,
                                         const float *residual, int32_t relu, int64_t B, int32_t M, int32_t K,
                                         int32_t H, int32_t W, int32_t taps, int32_t stride, void *workspace,
                                         size_t workspace_bytes, void *stream);

namespace kgdet {

namespace {

// src [n][C][P] -> dst [n][P][C]  (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void large_transpose(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                       long long P, long long src_image_stride) {
  __shared__ float tile[32][33];
  const long long p0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32, n = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float *s = src + n * src_image_stride;
  float *d = dst + (long long)n * P * C;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const long long p = p0 + tx;
    tile[r][tx] = (c < C && p < P) ? s[(long long)c * P + p] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long long p = p0 + r;
    const int c = c0 + tx;
    if (p < P && c < C) d[p * C + c] = tile[tx][r];
  }
}

// wpk [K][C_pad][O_pad] (the exact-fp32 forward kernel's image, part of the packed weight) -> wp [O][K*C] with
// wp[o][t*C + c] = W[o][c][t]
__global__ __launch_bounds__(256) void large_permute_weight(const float *__restrict__ wpk, float *__restrict__ wp, int O,
                                                            int C, int K, int C_pad, int O_pad) {
  const long long n = (long long)O * C * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int t = (int)((i / C) % K);
    const int o = (int)(i / ((long long)C * K));
    wp[i] = wpk[((long long)t * C_pad + c) * O_pad + o];
  }
}

// one thread per (image, tap, output pixel): its four corners are entries (((tap*P + pixel) * 4 + corner) << 32 | weight) of
// their cells' lists.
// FILL = false: count (cnt[cell]); FILL = true: drop the entries into the lists (cnt serves as the cursor; slot order arbitrary)
// This lane's slot in the list of `cell` (live lanes only): lanes of a wave that name the same cell share ONE atomic and take
// consecutive slots (up to four distinct cells per call that way, the remaining lanes one atomic each) -- offsets trained onto a
// few points put thousands of corners on one counter.  Called by ALL lanes of the wave.
__device__ __forceinline__ int large_take(int *__restrict__ cnt, int cell, bool live) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  int slot = 0;
  bool pending = live;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned long long act = __ballot(pending);
    if (act == 0ull) break;                                   // (uniform)
    const int leader = __ffsll((long long)act) - 1;
    const int cf = __shfl(cell, leader);
    const bool mine = pending && cell == cf;
    const unsigned long long m = __ballot(mine);
    int base = 0;
    if (lane == leader) base = atomicAdd(cnt + cf, __popcll(m));
    base = __shfl(base, leader);
    if (mine) {
      slot = base + __popcll(m & below);
      pending = false;
    }
  }
  if (pending) slot = atomicAdd(cnt + cell, 1);
  return slot;
}

template <bool FILL>
__global__ __launch_bounds__(256) void large_cell_entries(const DcnProblem p, int *__restrict__ cnt,
                                                          const int *__restrict__ start,
                                                          unsigned long long *__restrict__ vals) {
  const long long n = (long long)p.N * p.K * p.HoWo;
  const int HW = p.H * p.W;
  for (long long i0 = (long long)blockIdx.x * 256; i0 < n; i0 += (long long)gridDim.x * 256) {   // (uniform trip count per wave)
    const long long i_raw = i0 + threadIdx.x, i = i_raw < n ? i_raw : n - 1;
    const int hw = (int)(i % p.HoWo);
    const int t = (int)((i / p.HoWo) % p.K);
    const int b = (int)(i / ((long long)p.HoWo * p.K));
    const int oy = hw / p.Wo, ox = hw - oy * p.Wo;
    float y, x, m;
    tap_position(p, b, p.dgi, t, hw, oy, ox, y, x, m);
    Tap tap;
    TapGeom geo;
    make_tap(y, x, p.H, p.W, true, m, tap, geo);   // (v2: the modulation mask rides on the corner weights; m = 1 for v1)
    const int valid[4] = {geo.va, geo.vb, geo.vc, geo.vd};
    const unsigned long long src = (unsigned long long)((unsigned)(t * p.HoWo + hw)) << 34;   // key = (tap, pixel, corner): unique
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = i_raw < n && valid[e];
      const int cell = live ? b * HW + tap.o[e] : 0;
      const int slot = large_take(cnt, cell, live);
      if (FILL && live) vals[start[cell] + slot] = src | ((unsigned long long)e << 32) | __float_as_uint(tap.w[e]);
    }
  }
}

// storage for every cell's list (wave-aggregated bump allocation: where a list lies is arbitrary, what it will hold is not);
// the counters are reset to serve as fill cursors
__global__ __launch_bounds__(256) void large_cell_alloc(int *__restrict__ cnt, int *__restrict__ start, int *__restrict__ total,
                                                        int n_cells) {
  const int cell = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int c = cell < n_cells ? cnt[cell] : 0;
  int incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const int wave_sum = __shfl(incl, 63);
  int base = 0;
  if (lane == 63 && wave_sum > 0) base = atomicAdd(total, wave_sum);
  base = __shfl(base, 63);
  if (cell < n_cells) {
    start[cell] = base + incl - c;
    cnt[cell] = 0;
  }
}

// wave = cell: its entries in slot order -> in (tap, pixel, corner) order.  The keys (upper halves) are unique, so an entry's
// place is the number of smaller keys: by lane reads up to 64
// entries, beyond through an LDS copy of the keys read as broadcasts.
constexpr int kLargeSortChunk = 1024;
constexpr int kLargeWaveSortMax = 512;     // longest list the per-wave rank sort (quadratic) takes
constexpr int kLargeLongSortLds = 8192;    // entries large_cell_sort_long sorts in LDS (64 KB); longer lists in place
__global__ __launch_bounds__(256) void large_cell_sort(const int *__restrict__ start, const int *__restrict__ len,
                                                       const unsigned long long *__restrict__ in,
                                                       unsigned long long *__restrict__ out, int n_cells,
                                                       int *__restrict__ long_q, int *__restrict__ long_n) {
  __shared__ __attribute__((aligned(16))) unsigned keys[4][kLargeSortChunk];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int cell = blockIdx.x * 4 + wave;
  if (cell >= n_cells) return;
  const int n = __builtin_amdgcn_readfirstlane(len[cell]);
  if (n == 0) return;
  if (n > kLargeWaveSortMax) {     // (offsets trained onto a few cells: thousands of entries) -> large_cell_sort_long
    if (lane == 0) long_q[atomicAdd(long_n, 1)] = cell;
    return;
  }
  const int base = __builtin_amdgcn_readfirstlane(start[cell]);
  unsigned *kw = keys[wave];
  constexpr unsigned long long kNone = ~0ull;
  if (n <= 64) {
    const unsigned long long e = lane < n ? in[base + lane] : kNone;
    const unsigned key = (unsigned)(e >> 32);
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (unsigned)__shfl((int)key, j) < key ? 1 : 0;
    if (lane < n) out[base + rank] = e;
    return;
  }
  for (int i0 = 0; i0 < n; i0 += 256) {
    unsigned long long e[4];
    int rank[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = i0 + u * 64 + lane;
      e[u] = idx < n ? in[base + idx] : kNone;
    }
    for (int c0 = 0; c0 < n; c0 += kLargeSortChunk) {
      if (n > kLargeSortChunk || i0 == 0) {
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < kLargeSortChunk; j += 64) kw[j] = c0 + j < n ? (unsigned)(in[base + c0 + j] >> 32) : 0xffffffffu;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      }
      const int m4 = (min(kLargeSortChunk, n - c0) + 3) & ~3;
      for (int j = 0; j < m4; j += 4) {
        const uint4 k4 = *reinterpret_cast<const uint4 *>(kw + j);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned key = (unsigned)(e[u] >> 32);
          rank[u] += (k4.x < key ? 1 : 0) + (k4.y < key ? 1 : 0) + (k4.z < key ? 1 : 0) + (k4.w < key ? 1 : 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * 64 + lane < n) out[base + rank[u]] = e[u];
  }
}

// workgroup = one long list: a bitonic network whose every compare-exchange puts the smaller key at the lower index (first
// step of a merge: i with its mirror image inside the block, then i with i + j), so the power-of-two padding stays virtual -- a
// pair whose upper index lies beyond the list does nothing.  O(n log^2 n); in LDS up to kLargeLongSortLds entries, longer lists
// in place in the slot-order array.  (The entries are 64-bit with the unique key in the upper half: compared whole.)
__global__ __launch_bounds__(256) void large_cell_sort_long(const int *__restrict__ start, const int *__restrict__ len,
                                                            unsigned long long *__restrict__ in,
                                                            unsigned long long *__restrict__ out,
                                                            const int *__restrict__ long_q, const int *__restrict__ long_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char large_smem[];
  unsigned long long *buf = reinterpret_cast<unsigned long long *>(large_smem);
  const int tid = threadIdx.x, count = *long_n;
  for (int qi = blockIdx.x; qi < count; qi += gridDim.x) {
    const int cell = long_q[qi], n = len[cell], base = start[cell];
    const bool in_lds = n <= kLargeLongSortLds;
    unsigned long long *a = in_lds ? buf : in + base;
    if (in_lds)
      for (int t = tid; t < n; t += 256) buf[t] = in[base + t];
    __syncthreads();
    int n_pad = 1;
    while (n_pad < n) n_pad <<= 1;
    for (int k = 2; k <= n_pad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        const bool flip = j == (k >> 1);
        for (int t = tid; t < (n_pad >> 1); t += 256) {
          const int i = 2 * t - (t & (j - 1));
          const int p2 = flip ? (i ^ (k - 1)) : (i + j);
          if (p2 < n) {
            const unsigned long long x = a[i], y = a[p2];
            if (x > y) { a[i] = y; a[p2] = x; }
          }
        }
        __syncthreads();
      }
    for (int t = tid; t < n; t += 256) out[base + t] = a[t];
    __syncthreads();
  }
}

// grad_input[b][c][q] = sum over the entries of cell (b, q), in sorted (= enumeration) order, of w * colT[b][p][t*C + c]
// grid = (ceil(HW / kGatherCells), N, ceil(C / 256)), 256 threads = channels.
// The kernel is latency-bound (a cell's ~36 rows are 36 dependent-free but sequentially issued 1 KB reads per wave): 16 cells
// per workgroup keep the LDS tile at 16 KB, so eight workgroups share a CU (32 cells: four), and eight loads are in flight
// per thread -- 319 -> ~190 us on [2, 256, 100, 168] 3x3.  The adds keep the sorted order whatever the batch size.
constexpr int kGatherCells = 16;
__global__ __launch_bounds__(256) void large_gather_input(const float *__restrict__ colT, const int *__restrict__ start,
                                                          const int *__restrict__ len,
                                                          const unsigned long long *__restrict__ vals,
                                                          float *__restrict__ grad_input, int C, int K, int HW, int P,
                                                          int c_base, int C_all) {
  // (grad_input: channels c_base .. c_base + C - 1 of a [N, C_all, HW] tensor -- a channel run of a grouped convolution)
  __shared__ float tile[kGatherCells][257];
  const int q0 = blockIdx.x * kGatherCells, b = blockIdx.y, c = blockIdx.z * 256 + threadIdx.x;
  const bool live = c < C;
  const int cc = live ? c : 0;
  const long long KC = (long long)K * C;
  const float *base = colT + (long long)b * P * KC + cc;
  auto batch = [&](int e, auto N_, float &acc) __attribute__((always_inline)) {
    constexpr int N = decltype(N_)::value;
    float v[N], w[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const unsigned long long val = vals[e + u];
      const unsigned tp = (unsigned)(val >> 34);
      const unsigned t = tp / (unsigned)P, px = tp - t * (unsigned)P;
      w[u] = __uint_as_float((unsigned)val);
      v[u] = base[(long long)px * KC + (long long)t * C];
    }
#pragma unroll
    for (int u = 0; u < N; ++u) acc += w[u] * v[u];
  };
  for (int r = 0; r < kGatherCells; ++r) {
    const int q = q0 + r;
    float acc = 0.0f;
    if (q < HW && len[b * HW + q] <= kLargeWaveSortMax) {     // (longer lists: large_gather_long, chunked over many workgroups)
      const int e0 = start[b * HW + q], e1 = e0 + len[b * HW + q];
      int e = e0;
      for (; e + 8 <= e1; e += 8) batch(e, std::integral_constant<int, 8>{}, acc);
      if (e + 4 <= e1) { batch(e, std::integral_constant<int, 4>{}, acc); e += 4; }
      for (; e < e1; ++e) batch(e, std::integral_constant<int, 1>{}, acc);
    }
    tile[r][threadIdx.x] = acc;
  }
  __syncthreads();
  // [cells][256 channels] -> grad_input[b][c][q0 .. q0 + kGatherCells - 1]: 16 lanes share a channel's run
  const int lane_q = threadIdx.x & (kGatherCells - 1), ch_sub = threadIdx.x / kGatherCells;
  for (int c2 = ch_sub; c2 < 256; c2 += 256 / kGatherCells) {
    const int ch = blockIdx.z * 256 + c2, q = q0 + lane_q;
    if (ch < C && q < HW && len[b * HW + q] <= kLargeWaveSortMax) grad_input[((long long)b * C_all + c_base + ch) * HW + q] = tile[lane_q][c2];
  }
}

// Cells with more than kLargeWaveSortMax entries (offsets trained onto a few points: tens of thousands of corners on one cell) --
// in the kernel above one workgroup would walk such a list alone, 4.4 ms per launch at [2, 256, 100, 168] with every sample next
// to one of 17 points per image.  Their lists are cut into chunks of kLongChunk entries: large_long_chunks numbers the chunks,
// large_gather_long sums one chunk per workgroup (thread = channel, the same batches of eight loads), large_gather_long_sum adds
// a cell's chunk sums in chunk order: deterministic, and the cost follows the number of entries, not their distribution.
constexpr int kLongChunk = 512;
__global__ __launch_bounds__(256) void large_long_chunks(const int *__restrict__ len, const int *__restrict__ long_q,
                                                         const int *__restrict__ long_n, int *__restrict__ chunk_total,
                                                         int *__restrict__ chunk_first, int2 *__restrict__ rec) {
  const int count = *long_n;
  for (int qi = blockIdx.x * 256 + threadIdx.x; qi < count; qi += gridDim.x * 256) {
    const int n = len[long_q[qi]], nch = (n + kLongChunk - 1) / kLongChunk;
    const int first = atomicAdd(chunk_total, nch);      // (which records a cell gets is arbitrary, what they hold is not)
    chunk_first[qi] = first;
    for (int j = 0; j < nch; ++j) rec[first + j] = make_int2(qi, j);
  }
}

__global__ __launch_bounds__(256) void large_gather_long(const float *__restrict__ colT, const int *__restrict__ start,
                                                         const int *__restrict__ len,
                                                         const unsigned long long *__restrict__ vals,
                                                         const int *__restrict__ long_q, const int *__restrict__ chunk_total,
                                                         const int2 *__restrict__ rec, float *__restrict__ part, int C, int K,
                                                         int HW, int P) {
  const int total = *chunk_total;
  const long long KC = (long long)K * C;
  for (int r = blockIdx.x; r < total; r += gridDim.x) {
    const int2 rc = rec[r];
    const int cell = long_q[rc.x], b = cell / HW;
    const int e0 = start[cell] + rc.y * kLongChunk, e1 = min(start[cell] + len[cell], e0 + kLongChunk);
    for (int c = threadIdx.x; c < C; c += 256) {
      const float *base = colT + (long long)b * P * KC + c;
      float acc = 0.0f;
      int e = e0;
      for (; e + 8 <= e1; e += 8) {
        float v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned long long val = vals[e + u];
          const unsigned tp = (unsigned)(val >> 34);
          const unsigned t = tp / (unsigned)P, px = tp - t * (unsigned)P;
          w[u] = __uint_as_float((unsigned)val);
          v[u] = base[(long long)px * KC + (long long)t * C];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += w[u] * v[u];
      }
      for (; e < e1; ++e) {
        const unsigned long long val = vals[e];
        const unsigned tp = (unsigned)(val >> 34);
        const unsigned t = tp / (unsigned)P, px = tp - t * (unsigned)P;
        acc += __uint_as_float((unsigned)val) * base[(long long)px * KC + (long long)t * C];
      }
      part[(long long)r * C + c] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void large_gather_long_sum(const int *__restrict__ len, const int *__restrict__ long_q,
                                                             const int *__restrict__ long_n,
                                                             const int *__restrict__ chunk_first,
                                                             const float *__restrict__ part, float *__restrict__ grad_input,
                                                             int C, int HW, int c_base, int C_all) {
  const int count = *long_n;
  for (int qi = blockIdx.x; qi < count; qi += gridDim.x) {
    const int cell = long_q[qi], b = cell / HW, q = cell - b * HW;
    const int nch = (len[cell] + kLongChunk - 1) / kLongChunk, first = chunk_first[qi];
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.0f;
      for (int j = 0; j < nch; ++j) s += part[(long long)(first + j) * C + c];      // chunk order
      grad_input[((long long)b * C_all + c_base + c) * HW + q] = s;
    }
  }
}

// grad_offset[b][2t + dir][p] = sum_c colT[b][p][t*C + c] * (sum_corner dw_dir[corner] * x^T[b][corner][c])
// (deformable_col2im_coord + get_coordinate_weight, deform_conv_cuda_kernel.cu:144-187, 337-435; out of range -> 0)
// grid = (P, N), 256 threads = channels (looped for C > 256); kMaxK taps accumulate in registers
constexpr int kLargeMaxK = 49;
__global__ __launch_bounds__(256) void large_grad_offset(const DcnProblem p, const float *__restrict__ colT,
                                                         const float *__restrict__ xT, float *__restrict__ grad_offset,
                                                         float *__restrict__ grad_mask, int accumulate) {
  // (p.C_total channels = one channel run; p.dgi: its deformable group; accumulate: a later run of the same deformable group
  //  adds to the sums of the earlier ones -- launches run one after the other, so the order of the sum is fixed)
  __shared__ float red[4][3];
  const int hw = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int oy = hw / p.Wo, ox = hw - oy * p.Wo;
  const int C = p.C_total, K = p.K, HW = p.H * p.W;
  const long long KC = (long long)K * C;
  const float *crow = colT + ((long long)b * p.HoWo + hw) * KC;
  const float *ximg = xT + (long long)b * HW * C;
  for (int t = 0; t < K; ++t) {
    float y, x, m;
    tap_position(p, b, p.dgi, t, hw, oy, ox, y, x, m);
    Tap tap;
    TapGeom geo;
    make_tap(y, x, p.H, p.W, true, 1.0f, tap, geo);   // tap.w: the plain bilinear weights (d out / d mask = the sample)
    const float hy = 1.0f - geo.ly, hx = 1.0f - geo.lx;
    // v2: the offset derivatives carry the mask value (deform_conv_cuda_kernel.cu:636-766)
    const float ka = geo.va ? m : 0.f, kb = geo.vb ? m : 0.f, kc = geo.vc ? m : 0.f, kd = geo.vd ? m : 0.f;
    const float wy[4] = {-hx * ka, -geo.lx * kb, hx * kc, geo.lx * kd};
    const float wx[4] = {-hy * ka, hy * kb, -geo.ly * kc, geo.ly * kd};
    float dy = 0.0f, dx = 0.0f, dm = 0.0f;
    for (int c = tid; c < C; c += 256) {
      const float g = crow[(long long)t * C + c];
      float sy = 0.0f, sx = 0.0f, sm = 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = ximg[(long long)tap.o[e] * C + c];
        sy += wy[e] * v;
        sx += wx[e] * v;
        sm += tap.w[e] * v;
      }
      dy += g * sy;
      dx += g * sx;
      dm += g * sm;
    }
    // block reduction in a fixed order: lanes by shuffle, then the four waves in order
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      dy += __shfl_down(dy, d);
      dx += __shfl_down(dx, d);
      dm += __shfl_down(dm, d);
    }
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6][0] = dy; red[tid >> 6][1] = dx; red[tid >> 6][2] = dm; }
    __syncthreads();
    if (tid == 0) {
      const float ty = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
      const float tx = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
      float *dst = grad_offset + ((long long)(b * p.DG + p.dgi) * 2 * K + 2 * t) * p.HoWo + hw;
      dst[0] = accumulate ? dst[0] + ty : ty;
      dst[p.HoWo] = accumulate ? dst[p.HoWo] + tx : tx;
      if (grad_mask) {
        float *dm = grad_mask + ((long long)(b * p.DG + p.dgi) * K + t) * p.HoWo + hw;
        const float tm = ((red[0][2] + red[1][2]) + red[2][2]) + red[3][2];
        *dm = accumulate ? *dm + tm : tm;
      }
    }
  }
}

struct LargePlan {
  size_t gT, xT, wp, colT, packed, conv_ws, vals, cells, long_q, rec, part, total;
  long long n_entries;
};

LargePlan large_plan(const DcnProblem &p) {
  LargePlan L;
  const long long P = p.HoWo, HW = (long long)p.H * p.W, KC = (long long)p.K * p.C_total;
  L.n_entries = (long long)p.N * p.K * P * 4;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.gT = al((size_t)p.N * P * p.Og * 4);
  L.xT = al((size_t)p.N * HW * p.C_total * 4);
  L.wp = al((size_t)p.Og * KC * 4);
  L.colT = al((size_t)p.N * P * KC * 4);
  L.packed = al(kgdet_conv_packed_bytes((int)P, p.Og, 1));
  L.conv_ws = al(kgdet_conv_apply_workspace_bytes(1, (int)P, p.Og, 1, (int)KC, 1, 1));
  L.vals = al((size_t)L.n_entries * 8);                       // twice: slot order, sorted
  L.cells = al((size_t)(p.N * HW + 64) * 4);                  // twice: counters / cursors (+ the allocation counter), starts
  L.long_q = al((size_t)(L.n_entries / kLargeWaveSortMax + 64) * 4);      // twice: the queue, the cells' first chunk records
  L.rec = al((size_t)(2 * (L.n_entries / kLongChunk) + 64) * 8);             // (long cell, chunk) records
  L.part = al((size_t)(2 * (L.n_entries / kLongChunk) + 64) * p.C_total * 4);   // chunk sums
  L.total = L.gT + L.xT + L.wp + L.colT + L.packed + L.conv_ws + 2 * L.vals + 2 * L.cells + 2 * L.long_q + L.rec + L.part;
  return L;
}

}  // namespace

// eligible (p describes ONE channel run: C_total = the run's channels, Og = the output channels of its weight group):
// v1 or v2 (mask), O % 16 == 0, entry count within int range
bool dcn_bwd_large_ok(const DcnProblem &p, bool has_mask, int groups) {
  const long long n_entries = (long long)p.N * p.K * p.HoWo * 4;
  (void)has_mask; (void)groups;
  return p.Og % 16 == 0 && ((long long)p.K * p.C_total) % 2 == 0 &&
         p.K <= kLargeMaxK && n_entries < (1LL << 31) && (long long)p.N * p.H * p.W < (1LL << 31) - 2 &&
         (long long)p.K * p.HoWo < (1LL << 30);
}

size_t dcn_bwd_large_workspace_bytes(const DcnProblem &p) { return large_plan(p).total; }

// p: the FORWARD problem of ONE CHANNEL RUN (round 5: weight groups and deformable groups run as channel runs, so the float-atomic
// scatter kernel of rounds 1-4 is gone): x = the input's first channel of the run, C_total = the run's channel count, Og = the
// output channels of the run's weight group, wpk = that group's [K][Cg_pad][Og_pad] fp32 weight image at the run's first channel,
// DG / dgi = deformable groups of the convolution / the run's; x_channels = channels of the whole input tensor, c_base = the
// run's first channel in it (grad_input is written there), accumulate_offset: an earlier run of the same deformable group
// has written grad_offset / grad_mask already.  grad_output: the convolution's gradient buffer; out_channel_offset: first
// channel of the run's weight group in it.
int dcn_bwd_large(const DcnProblem &p, const float *grad_output, int out_channels_total,
                  int out_channel_offset, float *grad_input, float *grad_offset, float *grad_mask, void *workspace,
                  size_t workspace_bytes, void *stream, int x_channels, int c_base, int accumulate_offset) {
  const LargePlan L = large_plan(p);
  if (workspace == nullptr || workspace_bytes < L.total) {
    set_error("workspace too small for the large-map backward: need %zu bytes, got %zu", L.total, workspace_bytes);
    return KGDET_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned char *w8 = (unsigned char *)workspace;
  float *gT = (float *)w8; w8 += L.gT;
  float *xT = (float *)w8; w8 += L.xT;
  float *wp = (float *)w8; w8 += L.wp;
  float *colT = (float *)w8; w8 += L.colT;
  void *packed = w8; w8 += L.packed;
  void *conv_ws = w8; w8 += L.conv_ws;
  unsigned long long *vals_a = (unsigned long long *)w8; w8 += L.vals;
  unsigned long long *vals_b = (unsigned long long *)w8; w8 += L.vals;
  int *cell_cnt = (int *)w8; w8 += L.cells;
  int *cell_start = (int *)w8; w8 += L.cells;
  int *long_q = (int *)w8; w8 += L.long_q;
  int *chunk_first = (int *)w8; w8 += L.long_q;
  int2 *chunk_rec = (int2 *)w8; w8 += L.rec;
  float *chunk_part = (float *)w8;
  const int C = p.C_total, O = p.Og, K = p.K;
  const long long P = p.HoWo, HW = (long long)p.H * p.W, KC = (long long)K * C;
  const int O_total = out_channels_total > 0 ? out_channels_total : O;

  // grad_out window [N][O][P] (channels out_channel_offset.. of an O_total-wide buffer) -> gT [N][P][O]; x -> xT
  hipLaunchKernelGGL(large_transpose, dim3((unsigned)((P + 31) / 32), (O + 31) / 32, p.N), dim3(256), 0, st,
                     grad_output + (long long)out_channel_offset * P, gT, O, P, (long long)O_total * P);
  hipLaunchKernelGGL(large_transpose, dim3((unsigned)((HW + 31) / 32), (C + 31) / 32, p.N), dim3(256), 0, st, p.x, xT, C,
                     HW, (long long)x_channels * HW);
  hipLaunchKernelGGL(large_permute_weight, dim3(1024), dim3(256), 0, st, p.wpk, wp, O, C, K, p.Cg_pad, p.Og_pad);
  // colT[b] = gT[b] (P x O) * wp (O x KC): the pixels are the "output channels" of a 1x1 convolution over the KC-pixel
  // "image" wp
  for (int b = 0; b < p.N; ++b) {
    if (int rc = kgdet_conv_pack(gT + (long long)b * P * O, (int)P, O, 1, 0, packed, stream)) return rc;
    if (int rc = kgdet_conv_apply_epilogue(packed, wp, colT + (long long)b * P * KC, nullptr, nullptr, 0, 1, (int)P, O, 1,
                                           (int)KC, 1, 1, conv_ws, L.conv_ws, stream))
      return rc;
  }
  // inverse index: per-cell lists, ranked into enumeration order
  const int n_cells = (int)(p.N * HW);
  int *cell_total = cell_cnt + n_cells;
  KGDET_HIP_TRY(hipMemsetAsync(cell_cnt, 0, (size_t)(n_cells + 64) * sizeof(int), st));
  hipLaunchKernelGGL(large_cell_entries<false>, dim3(2048), dim3(256), 0, st, p, cell_cnt, (const int *)nullptr,
                     (unsigned long long *)nullptr);
  hipLaunchKernelGGL(large_cell_alloc, dim3((n_cells + 255) / 256), dim3(256), 0, st, cell_cnt, cell_start, cell_total, n_cells);
  hipLaunchKernelGGL(large_cell_entries<true>, dim3(2048), dim3(256), 0, st, p, cell_cnt, (const int *)cell_start, vals_a);
  int *long_n = cell_total + 1;      // (zeroed with the counters)
  hipLaunchKernelGGL(large_cell_sort, dim3((n_cells + 3) / 4), dim3(256), 0, st, (const int *)cell_start, (const int *)cell_cnt,
                     (const unsigned long long *)vals_a, vals_b, n_cells, long_q, long_n);
  static thread_local bool long_attr = false;
  if (!long_attr) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)large_cell_sort_long, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kLargeLongSortLds * 8));
    long_attr = true;
  }
  hipLaunchKernelGGL(large_cell_sort_long, dim3(512), dim3(256), kLargeLongSortLds * 8, st, (const int *)cell_start,
                     (const int *)cell_cnt, vals_a, vals_b, (const int *)long_q, (const int *)long_n);
  hipLaunchKernelGGL(large_gather_input, dim3((unsigned)((HW + kGatherCells - 1) / kGatherCells), p.N, (C + 255) / 256), dim3(256), 0, st, colT,
                     (const int *)cell_start, (const int *)cell_cnt, vals_b, grad_input, C, K, (int)HW, (int)P, c_base, x_channels);
  int *chunk_total = cell_total + 2;     // (zeroed with the counters)
  hipLaunchKernelGGL(large_long_chunks, dim3(64), dim3(256), 0, st, (const int *)cell_cnt, (const int *)long_q, (const int *)long_n,
                     chunk_total, chunk_first, chunk_rec);
  hipLaunchKernelGGL(large_gather_long, dim3(2048), dim3(256), 0, st, colT, (const int *)cell_start, (const int *)cell_cnt, vals_b,
                     (const int *)long_q, (const int *)chunk_total, (const int2 *)chunk_rec, chunk_part, C, K, (int)HW, (int)P);
  hipLaunchKernelGGL(large_gather_long_sum, dim3(1024), dim3(256), 0, st, (const int *)cell_cnt, (const int *)long_q,
                     (const int *)long_n, (const int *)chunk_first, (const float *)chunk_part, grad_input, C, (int)HW, c_base, x_channels);
  hipLaunchKernelGGL(large_grad_offset, dim3((unsigned)P, p.N), dim3(256), 0, st, p, colT, xT, grad_offset,
                     p.mask ? grad_mask : nullptr, accumulate_offset);
  KGDET_CHECK_LAUNCH("dcn_bwd_large");
  return KGDET_OK;
}

}  // namespace kgdet
