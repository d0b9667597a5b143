// Hard NMS (batched, fully on device) and soft-NMS for gfx950.
//
// Reference semantics pinned (bit-exact index selection): mmdet/ops/nms/src/nms_cpu.cpp:5-59
//   areas (x2-x1+1)*(y2-y1+1); visit in descending score order; a later box is suppressed when
//   inter / (area_i + area_j - inter) >= thr; the result is the kept indices in ASCENDING index
//   order.  (The reference's CUDA kernel uses > instead of >= and ships an N x N/64 bit mask to the
//   host for the greedy sweep, nms_kernel.cu:60,99-123; here the sweep stays on the device.)
// Soft-NMS: mmdet/ops/nms/src/soft_nms_cpu.pyx:22-127, including its swap-with-last removal order.
//
// One workgroup per segment (= one (image, class) problem), everything in LDS:
//   1. 64-bit keys (descending score, ascending index) -> bitonic sort
//   2. sorted boxes + areas into LDS
//   3. greedy suppression in chunks of 64 sorted boxes: wave 0 resolves the chunk internally with
//      64 readlane steps on 64-bit masks, then all waves apply the chunk's survivors to every
//      later box in parallel
//   4. survivors flagged by original index, prefix-scanned, written ascending.
// The float expressions are written exactly as in the reference and compiled with contraction off,
// so every >= decision is the same as on the CPU.
#include "common.h"

#pragma clang fp contract(off)

namespace kgdet {

namespace {

constexpr int kNmsThreads = 512;
constexpr int kNmsMaxLen = 4096;  // boxes per segment that fit the LDS plan below

__device__ __forceinline__ unsigned long long score_key(float s, unsigned idx) {
  unsigned u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order-preserving map
  return ((unsigned long long)(~u) << 32) | idx;   // invert: larger score sorts first; ties: lower index
}

__device__ __forceinline__ bool iou_ge(float ix1, float iy1, float ix2, float iy2, float iarea, float jx1,
                                       float jy1, float jx2, float jy2, float jarea, float thr) {
  const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
  const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
  const float w = fmaxf(0.0f, xx2 - xx1 + 1), h = fmaxf(0.0f, yy2 - yy1 + 1);
  const float inter = w * h;
  const float ovr = inter / (iarea + jarea - inter);
  return ovr >= thr;
}

// block-wide exclusive scan of one int per thread (512 threads); returns the prefix, total via out
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds_wave_sums, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int n = __shfl_up(incl, d);
    if (lane >= d) incl += n;
  }
  if (lane == 63) lds_wave_sums[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kNmsThreads / 64; ++w) {
    const int s = lds_wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return base + incl - v;
}

}  // namespace

// dynamic LDS layout: keys[NP] u64 | x1,y1,x2,y2,area [n each] | alive[n] u8 (then reused as flags)
// One problem: n boxes at box[i * box_stride + 0..3], scores at score[i * score_stride]; with `filter` only the
// boxes whose score is > score_thr take part (the per-class candidate filter of multiclass_nms_kp,
// bbox_nms_kp.py:28-33, done in place instead of by compaction); kept ORIGINAL indices, ascending, go to out[].
struct NmsProblem {
  const float *box;
  int box_stride;
  const float *score;
  int score_stride;
  bool filter;
  float score_thr;
  int n;
  long long *out;
  long long *num_out;
};

// Working arrays of one problem.  On-chip: carved out of dynamic LDS (<= kNmsMaxLen boxes).  Large segments: the same
// arrays in a global scratch area (one workgroup per segment still, so workgroup barriers order every access) plus a
// 64-box LDS cache of the chunk being resolved; same decisions, same order, any length.
struct NmsStore {
  unsigned long long *keys;   // [NP]
  float *bx1, *by1, *bx2, *by2, *bar;   // [n] sorted boxes + areas
  unsigned char *alive;       // [n]
  int *flag;                  // [n_all] keep flags by original index (on-chip: aliases by1)
  float *chunk;               // LDS [5][64] cache of the current chunk (large mode), nullptr on-chip
};

template <bool LARGE>
__device__ __forceinline__ void nms_block(const NmsProblem pr, float thr, const NmsStore st,
                                          int *wave_sums, unsigned long long *chunk_alive_p) {
  const int tid = threadIdx.x;
  const int n_all = pr.n;
  if (n_all <= 0) {
    if (tid == 0) *pr.num_out = 0;
    return;
  }
  int NP = 64;
  while (NP < n_all) NP <<= 1;
  unsigned long long &chunk_alive = *chunk_alive_p;

  unsigned long long *keys = st.keys;
  float *bx1 = st.bx1, *by1 = st.by1, *bx2 = st.bx2, *by2 = st.by2, *bar = st.bar;
  unsigned char *alive = st.alive;

  int mine = 0;
  for (int i = tid; i < NP; i += kNmsThreads) {
    unsigned long long k = ~0ull;
    if (i < n_all) {
      const float s = pr.score[(long long)i * pr.score_stride];
      if (!pr.filter || s > pr.score_thr) { k = score_key(s, (unsigned)i); ++mine; }
    }
    keys[i] = k;
  }
  int n;   // boxes taking part = the first n sorted keys
  block_exclusive_scan(mine, wave_sums, n);
  if (n == 0) {
    if (tid == 0) *pr.num_out = 0;
    return;
  }
  // bitonic sort, ascending keys
  for (int k = 2; k <= NP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < NP; i += kNmsThreads) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += kNmsThreads) {
    const unsigned src = (unsigned)(keys[i] & 0xffffffffu);
    const float *bp = pr.box + (long long)src * pr.box_stride;
    const float x1 = bp[0], y1 = bp[1], x2 = bp[2], y2 = bp[3];
    bx1[i] = x1; by1[i] = y1; bx2[i] = x2; by2[i] = y2;
    bar[i] = (x2 - x1 + 1) * (y2 - y1 + 1);
    alive[i] = 1;
  }
  __syncthreads();

  const int lane = tid & 63;
  for (int c0 = 0; c0 < n; c0 += 64) {
    // the chunk's boxes: on-chip they are read in place; large mode caches them in LDS first
    const float *cx1 = bx1 + c0, *cy1 = by1 + c0, *cx2 = bx2 + c0, *cy2 = by2 + c0, *car = bar + c0;
    if constexpr (LARGE) {
      if (tid < 64 && c0 + tid < n) {
        st.chunk[tid] = bx1[c0 + tid]; st.chunk[64 + tid] = by1[c0 + tid]; st.chunk[128 + tid] = bx2[c0 + tid];
        st.chunk[192 + tid] = by2[c0 + tid]; st.chunk[256 + tid] = bar[c0 + tid];
      }
      __syncthreads();
      cx1 = st.chunk; cy1 = st.chunk + 64; cx2 = st.chunk + 128; cy2 = st.chunk + 192; car = st.chunk + 256;
    }
    if (tid < 64) {  // wave 0: resolve the chunk internally
      const int i = c0 + lane;
      const bool have = i < n;
      const float x1 = have ? cx1[lane] : 0.f, y1 = have ? cy1[lane] : 0.f, x2 = have ? cx2[lane] : 0.f,
                  y2 = have ? cy2[lane] : 0.f, ar = have ? car[lane] : 0.f;
      unsigned long long mask = 0;  // bit j: this box suppresses chunk box j (j > lane)
      for (int j = lane + 1; j < 64 && c0 + j < n; ++j)
        if (iou_ge(x1, y1, x2, y2, ar, cx1[j], cy1[j], cx2[j], cy2[j], car[j], thr))
          mask |= 1ull << j;
      unsigned long long live = __ballot(have && alive[i]);
      const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
      for (int s = 0; s < 64; ++s) {
        if ((live >> s) & 1ull) {
          const unsigned lo = __builtin_amdgcn_readlane(mlo, s), hi = __builtin_amdgcn_readlane(mhi, s);
          live &= ~(((unsigned long long)hi << 32) | lo);
        }
      }
      if (have) alive[i] = (live >> lane) & 1ull;
      if (lane == 0) chunk_alive = live;
    }
    __syncthreads();
    const unsigned long long live = chunk_alive;
    if (live) {
      for (int j = c0 + 64 + tid; j < n; j += kNmsThreads) {
        if (!alive[j]) continue;
        const float x1 = bx1[j], y1 = by1[j], x2 = bx2[j], y2 = by2[j], ar = bar[j];
        unsigned long long rest = live;
        bool dead = false;
        while (rest && !dead) {
          const int s = __ffsll((long long)rest) - 1;
          rest &= rest - 1;
          dead = iou_ge(cx1[s], cy1[s], cx2[s], cy2[s], car[s], x1, y1, x2, y2, ar, thr);
        }
        if (dead) alive[j] = 0;
      }
    }
    __syncthreads();
  }

  // survivors -> flags by original index (on-chip the flags reuse the by1 array), ascending compaction
  int *flag = st.flag;
  if constexpr (!LARGE) {
    unsigned char keep_sorted_local[kNmsMaxLen / kNmsThreads];
    for (int i = tid, m = 0; i < n; i += kNmsThreads, ++m) keep_sorted_local[m] = alive[i];
    __syncthreads();
    for (int i = tid; i < n_all; i += kNmsThreads) flag[i] = 0;
    __syncthreads();
    for (int i = tid, m = 0; i < n; i += kNmsThreads, ++m)
      if (keep_sorted_local[m]) flag[(unsigned)(keys[i] & 0xffffffffu)] = 1;
  } else {
    for (int i = tid; i < n_all; i += kNmsThreads) flag[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kNmsThreads)
      if (alive[i]) flag[(unsigned)(keys[i] & 0xffffffffu)] = 1;
  }
  __syncthreads();
  // each thread owns a contiguous run of original indices so the scan preserves ascending order
  const int per = (n_all + kNmsThreads - 1) / kNmsThreads;
  const int lo = tid * per, hi = min(n_all, lo + per);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += flag[i];
  int total;
  int pos = block_exclusive_scan(cnt, wave_sums, total);
  for (int i = lo; i < hi; ++i)
    if (flag[i]) pr.out[pos++] = i;
  if (tid == 0) *pr.num_out = total;
}

__device__ __forceinline__ NmsStore nms_lds_store(unsigned char *smem, int max_np) {
  NmsStore st;
  st.keys = reinterpret_cast<unsigned long long *>(smem);
  st.bx1 = reinterpret_cast<float *>(smem + (size_t)max_np * 8);
  st.by1 = st.bx1 + max_np; st.bx2 = st.by1 + max_np; st.by2 = st.bx2 + max_np; st.bar = st.by2 + max_np;
  st.alive = reinterpret_cast<unsigned char *>(st.bar + max_np);
  st.flag = reinterpret_cast<int *>(st.by1);
  st.chunk = nullptr;
  return st;
}

// scratch bytes of one large segment of n boxes (NP = n rounded up to a power of two)
__host__ __device__ inline size_t nms_large_scratch_bytes(long long n) {
  long long np = 64;
  while (np < n) np <<= 1;
  return (size_t)np * 8 + (size_t)n * (5 * 4 + 4) + (((size_t)n + 15) & ~(size_t)15);
}

// Segments longer than the on-chip limit: same algorithm, arrays in global scratch at scratch + scratch_off[seg].
__global__ __launch_bounds__(kNmsThreads) void nms_segments_large(const float *__restrict__ dets,
                                                                  const long long *__restrict__ seg_offsets, float thr,
                                                                  long long *__restrict__ keep,
                                                                  long long *__restrict__ num_keep,
                                                                  unsigned char *__restrict__ scratch) {
  __shared__ int wave_sums[kNmsThreads / 64];
  __shared__ unsigned long long chunk_alive;
  __shared__ float chunk[5 * 64];
  const int seg = blockIdx.x;
  const long long seg_begin = seg_offsets[seg];
  const long long n = seg_offsets[seg + 1] - seg_begin;
  // this segment's scratch: every segment before it takes nms_large_scratch_bytes(its length), 256-byte aligned
  size_t off = 0;
  for (int q = 0; q < seg; ++q) off += (nms_large_scratch_bytes(seg_offsets[q + 1] - seg_offsets[q]) + 255) & ~(size_t)255;
  long long np = 64;
  while (np < n) np <<= 1;
  unsigned char *base = scratch + off;
  NmsStore st;
  st.keys = reinterpret_cast<unsigned long long *>(base);
  st.bx1 = reinterpret_cast<float *>(base + (size_t)np * 8);
  st.by1 = st.bx1 + n; st.bx2 = st.by1 + n; st.by2 = st.bx2 + n; st.bar = st.by2 + n;
  st.flag = reinterpret_cast<int *>(st.bar + n);
  st.alive = reinterpret_cast<unsigned char *>(st.flag + n);
  st.chunk = chunk;
  NmsProblem pr;
  pr.box = dets + seg_begin * 5;
  pr.box_stride = 5;
  pr.score = pr.box + 4;
  pr.score_stride = 5;
  pr.filter = false;
  pr.score_thr = 0.f;
  pr.n = (int)n;
  pr.out = keep + seg_begin;
  pr.num_out = num_keep + seg;
  nms_block<true>(pr, thr, st, wave_sums, &chunk_alive);
}

// dets [T, 5]; segment s = rows [seg_offsets[s], seg_offsets[s + 1])
__global__ __launch_bounds__(kNmsThreads) void nms_segments(const float *__restrict__ dets,
                                                            const long long *__restrict__ seg_offsets,
                                                            float thr, long long *__restrict__ keep,
                                                            long long *__restrict__ num_keep, int max_np) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int wave_sums[kNmsThreads / 64];
  __shared__ unsigned long long chunk_alive;
  const int seg = blockIdx.x;
  const long long seg_begin = seg_offsets[seg];
  NmsProblem pr;
  pr.box = dets + seg_begin * 5;
  pr.box_stride = 5;
  pr.score = pr.box + 4;
  pr.score_stride = 5;
  pr.filter = false;
  pr.score_thr = 0.f;
  pr.n = (int)(seg_offsets[seg + 1] - seg_begin);
  pr.out = keep + seg_begin;
  pr.num_out = num_keep + seg;
  nms_block<false>(pr, thr, nms_lds_store(smem, max_np), wave_sums, &chunk_alive);
}

// The per-class loop of multiclass_nms_kp (mmdet/core/post_processing/bbox_nms_kp.py:25-50) for a whole batch in
// one launch: workgroup (b, c) filters class c's candidates of image b (score > score_thr) and suppresses them.
// boxes [B, N, 4]; scores [B, N, S] with class c in column col0 + c; keep [B, C, N]; num_keep [B, C].
__global__ __launch_bounds__(kNmsThreads) void multiclass_nms_segments(const float *__restrict__ boxes,
                                                                       const float *__restrict__ scores, int N, int C,
                                                                       int S, int col0, float score_thr, float thr,
                                                                       long long *__restrict__ keep,
                                                                       long long *__restrict__ num_keep, int max_np) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int wave_sums[kNmsThreads / 64];
  __shared__ unsigned long long chunk_alive;
  const int seg = blockIdx.x, b = seg / C, c = seg - b * C;
  NmsProblem pr;
  pr.box = boxes + (long long)b * N * 4;
  pr.box_stride = 4;
  pr.score = scores + (long long)b * N * S + col0 + c;
  pr.score_stride = S;
  pr.filter = true;
  pr.score_thr = score_thr;
  pr.n = N;
  pr.out = keep + (long long)seg * N;
  pr.num_out = num_keep + seg;
  nms_block<false>(pr, thr, nms_lds_store(smem, max_np), wave_sums, &chunk_alive);
}

// The tail of multiclass_nms_kp (bbox_nms_kp.py:52-70) per image: concatenate the classes' survivors (class order,
// ascending candidate index inside a class); more than max_num -> the max_num best by score (ties: earlier in the
// concatenation first).  One workgroup per image; keys sorted in LDS.
// out_det [B, max_num, 5]; out_label [B, max_num] (0-based class); out_src [B, max_num] (row n of the image's
// candidate arrays, for gathering landmarks); out_count [B].  Rows past the count are zero.
__global__ __launch_bounds__(kNmsThreads) void multiclass_select(const float *__restrict__ boxes,
                                                                 const float *__restrict__ scores, int N, int C, int S,
                                                                 int col0, const long long *__restrict__ keep,
                                                                 const long long *__restrict__ num_keep, int max_num,
                                                                 float *__restrict__ out_det,
                                                                 long long *__restrict__ out_label,
                                                                 long long *__restrict__ out_src,
                                                                 long long *__restrict__ out_count, int max_np) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int prefix[65];
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int run = 0;
    for (int c = 0; c < C; ++c) {
      prefix[c] = run;
      run += (int)num_keep[b * C + c];
    }
    prefix[C] = run;
  }
  __syncthreads();
  const int T = prefix[C];
  const int out_n = min(T, max_num);
  auto entry = [&](int p, int &c, int &n) {   // position in the concatenation -> (class, candidate row)
    int lo = 0, hi = C - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= p) lo = mid; else hi = mid - 1;
    }
    c = lo;
    n = (int)keep[((long long)b * C + c) * N + (p - prefix[c])];
  };
  const float *sc = scores + (long long)b * N * S + col0;
  if (T > max_num) {
    int NP = 64;
    while (NP < T) NP <<= 1;
    for (int p = tid; p < NP; p += kNmsThreads) {
      unsigned long long k = ~0ull;
      if (p < T) {
        int c, n;
        entry(p, c, n);
        k = score_key(sc[(long long)n * S + c], (unsigned)p);
      }
      keys[p] = k;
    }
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < NP; i += kNmsThreads) {
          const int l = i ^ j;
          if (l > i) {
            const unsigned long long a = keys[i], bb = keys[l];
            const bool up = (i & k) == 0;
            if ((a > bb) == up) { keys[i] = bb; keys[l] = a; }
          }
        }
        __syncthreads();
      }
  }
  for (int r = tid; r < max_num; r += kNmsThreads) {
    float *od = out_det + ((long long)b * max_num + r) * 5;
    if (r < out_n) {
      const int p = T > max_num ? (int)(keys[r] & 0xffffffffu) : r;
      int c, n;
      entry(p, c, n);
      const float *bp = boxes + ((long long)b * N + n) * 4;
      od[0] = bp[0]; od[1] = bp[1]; od[2] = bp[2]; od[3] = bp[3];
      od[4] = sc[(long long)n * S + c];
      out_label[(long long)b * max_num + r] = c;
      out_src[(long long)b * max_num + r] = n;
    } else {
      od[0] = od[1] = od[2] = od[3] = od[4] = 0.f;
      out_label[(long long)b * max_num + r] = 0;
      out_src[(long long)b * max_num + r] = 0;
    }
  }
  if (tid == 0) out_count[b] = out_n;
}


// ----------------------------------------------------------------------------------------------
// soft-NMS: one workgroup, boxes in LDS, the reference's selection-sort loop with every inner
// sweep done in parallel.  The reference removes a box whose decayed score drops below min_score
// by overwriting it with the LAST box and re-examining that slot; the resulting arrangement is
// reproduced exactly: survivors in front of the new end stay, holes there are filled (left to
// right) by the surviving boxes behind the new end (taken right to left).
// ----------------------------------------------------------------------------------------------
// the LDS arrays of one soft-NMS problem of capacity `cap` boxes (9 words per box)
struct SoftNmsStore {
  float *bx1, *by1, *bx2, *by2, *bsc;
  int *bid, *hole, *mover, *flag;
};
__device__ __forceinline__ SoftNmsStore soft_nms_store(unsigned char *smem, int cap) {
  SoftNmsStore st;
  st.bx1 = reinterpret_cast<float *>(smem);
  st.by1 = st.bx1 + cap; st.bx2 = st.by1 + cap; st.by2 = st.bx2 + cap; st.bsc = st.by2 + cap;
  st.bid = reinterpret_cast<int *>(st.bsc + cap);
  st.hole = st.bid + cap;    // scratch: positions of removed boxes in front of the new end
  st.mover = st.hole + cap;  // scratch: positions of surviving boxes behind the new end
  st.flag = st.mover + cap;  // scratch: survive flags
  return st;
}

// the reference's loop (soft_nms_cpu.pyx:44-125) over the n boxes already in `st` (bid = their original rows); returns
// the number of boxes left, arranged as the reference leaves them.  All threads of the workgroup call it.
__device__ __forceinline__ int soft_nms_block(const SoftNmsStore &st, int n, float iou_thr, int method, float sigma,
                                              float min_score, int *wave_sums, unsigned long long *wave_best,
                                              int *n_front_surv_p) {
  float *bx1 = st.bx1, *by1 = st.by1, *bx2 = st.bx2, *by2 = st.by2, *bsc = st.bsc;
  int *bid = st.bid, *hole = st.hole, *mover = st.mover, *flag = st.flag;
  int &n_front_surv = *n_front_surv_p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int N = n;
  for (int i = 0; i < N; ++i) {
    // first maximum of scores in [i, N): key = (score order-preserving bits, inverted position)
    unsigned long long best = 0;
    for (int j = i + tid; j < N; j += kNmsThreads) {
      unsigned u = __float_as_uint(bsc[j]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      const unsigned long long k = ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - j);
      best = k > best ? k : best;
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      const unsigned long long o = __shfl_xor(best, dlt);
      best = o > best ? o : best;
    }
    if (lane == 0) wave_best[wave] = best;
    __syncthreads();
    best = wave_best[0];
#pragma unroll
    for (int w = 1; w < kNmsThreads / 64; ++w) best = wave_best[w] > best ? wave_best[w] : best;
    const int maxpos = 0x7fffffff - (int)(best & 0xffffffffu);
    __syncthreads();
    if (tid == 0 && maxpos != i) {  // swap rows i and maxpos (pyx:53-67)
      float t;
      t = bx1[i]; bx1[i] = bx1[maxpos]; bx1[maxpos] = t;
      t = by1[i]; by1[i] = by1[maxpos]; by1[maxpos] = t;
      t = bx2[i]; bx2[i] = bx2[maxpos]; bx2[maxpos] = t;
      t = by2[i]; by2[i] = by2[maxpos]; by2[maxpos] = t;
      t = bsc[i]; bsc[i] = bsc[maxpos]; bsc[maxpos] = t;
      const int ti = bid[i]; bid[i] = bid[maxpos]; bid[maxpos] = ti;
    }
    __syncthreads();
    const float tx1 = bx1[i], ty1 = by1[i], tx2 = bx2[i], ty2 = by2[i];

    // decay every box behind i (pyx:80-112); arithmetic promotions as cython emits them
    for (int pos = i + 1 + tid; pos < N; pos += kNmsThreads) {
      const float x1 = bx1[pos], y1 = by1[pos], x2 = bx2[pos], y2 = by2[pos];
      const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
      const float iw = (float)((double)(fminf(tx2, x2) - fmaxf(tx1, x1)) + 1.0);
      if (iw > 0.0f) {
        const float ih = (float)((double)(fminf(ty2, y2) - fmaxf(ty1, y1)) + 1.0);
        if (ih > 0.0f) {
          const float ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0)) + (double)area) -
                                   (double)(iw * ih));
          const float ov = (iw * ih) / ua;
          float weight;
          if (method == 1) weight = ov > iou_thr ? (float)(1.0 - (double)ov) : 1.0f;
          else if (method == 2) weight = (float)exp((double)((-(ov * ov)) / sigma));
          else weight = ov > iou_thr ? 0.0f : 1.0f;
          bsc[pos] = weight * bsc[pos];
        }
      }
    }
    __syncthreads();

    // removal (pyx:113-123).  A box is tested against min_score only inside the iw>0 && ih>0 branch,
    // so only "touched" boxes can be dropped in this round.
    const int M = N - (i + 1);
    const int per = (M + kNmsThreads - 1) / kNmsThreads;
    const int lo = min(N, i + 1 + tid * per), hi = min(N, lo + per);
    int surv = 0;
    for (int pos = lo; pos < hi; ++pos) {
      const float x1 = bx1[pos], y1 = by1[pos], x2 = bx2[pos], y2 = by2[pos];
      const float iw = (float)((double)(fminf(tx2, x2) - fmaxf(tx1, x1)) + 1.0);
      bool touched = false;
      if (iw > 0.0f) {
        const float ih = (float)((double)(fminf(ty2, y2) - fmaxf(ty1, y1)) + 1.0);
        touched = ih > 0.0f;
      }
      const int sv = !(touched && bsc[pos] < min_score);
      flag[pos] = sv;
      surv += sv;
    }
    int S;
    const int before = block_exclusive_scan(surv, wave_sums, S);  // survivors in front of this thread's run
    const int newN = i + 1 + S;
    if (tid == 0) n_front_surv = S;  // value when newN == N (nothing removed)
    __syncthreads();
    {
      int sb = before;
      for (int pos = lo; pos < hi; ++pos) {
        const int sv = flag[pos];
        if (pos == newN) n_front_surv = sb;                          // survivors inside [i+1, newN)
        if (!sv && pos < newN) hole[(pos - (i + 1)) - sb] = pos;     // k-th hole, left to right
        if (sv && pos >= newN) mover[S - sb - 1] = pos;              // k-th surviving box from the right
        sb += sv;
      }
    }
    __syncthreads();
    const int H = S - n_front_surv;  // holes in front of the new end == survivors behind it
    for (int k = tid; k < H; k += kNmsThreads) {
      const int dst = hole[k], src = mover[k];
      bx1[dst] = bx1[src]; by1[dst] = by1[src]; bx2[dst] = bx2[src]; by2[dst] = by2[src];
      bsc[dst] = bsc[src]; bid[dst] = bid[src];
    }
    __syncthreads();
    N = newN;
  }
  return N;
}

__global__ __launch_bounds__(kNmsThreads) void soft_nms_kernel(const float *__restrict__ dets, int n, float iou_thr,
                                                               int method, float sigma, float min_score,
                                                               float *__restrict__ out_dets,
                                                               long long *__restrict__ out_inds,
                                                               long long *__restrict__ num_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int wave_sums[kNmsThreads / 64];
  __shared__ unsigned long long wave_best[kNmsThreads / 64];
  __shared__ int n_front_surv;
  const SoftNmsStore st = soft_nms_store(smem, n);
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += kNmsThreads) {
    st.bx1[i] = dets[5 * i]; st.by1[i] = dets[5 * i + 1]; st.bx2[i] = dets[5 * i + 2]; st.by2[i] = dets[5 * i + 3];
    st.bsc[i] = dets[5 * i + 4];
    st.bid[i] = i;
  }
  __syncthreads();
  const int N = soft_nms_block(st, n, iou_thr, method, sigma, min_score, wave_sums, wave_best, &n_front_surv);
  for (int i = tid; i < N; i += kNmsThreads) {
    out_dets[5 * i] = st.bx1[i]; out_dets[5 * i + 1] = st.by1[i]; out_dets[5 * i + 2] = st.bx2[i];
    out_dets[5 * i + 3] = st.by2[i]; out_dets[5 * i + 4] = st.bsc[i];
    out_inds[i] = st.bid[i];
  }
  if (tid == 0) *num_out = N;
}

// multiclass_nms_kp with nms type 'soft_nms' (bbox_nms_kp.py:25-50 -> nms_wrapper.soft_nms -> soft_nms_cpu.pyx) for a whole
// batch in one launch: workgroup (b, c) collects class c's candidates of image b (score > score_thr, ascending candidate
// row: what `multi_bboxes[cls_inds]` hands the op) and runs the reference's loop on them in LDS.
// boxes [B, N, 4]; scores [B, N, S], class c in column col0 + c.
// seg_dets [B, C, N, 5] (box + decayed score), seg_src [B, C, N] (candidate row), seg_count [B, C].
__global__ __launch_bounds__(kNmsThreads) void multiclass_soft_nms_segments(
    const float *__restrict__ boxes, const float *__restrict__ scores, int N, int C, int S, int col0, float score_thr,
    float iou_thr, int method, float sigma, float min_score, float *__restrict__ seg_dets, int *__restrict__ seg_src,
    int *__restrict__ seg_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int wave_sums[kNmsThreads / 64];
  __shared__ unsigned long long wave_best[kNmsThreads / 64];
  __shared__ int n_front_surv;
  const SoftNmsStore st = soft_nms_store(smem, N);
  const int seg = blockIdx.x, b = seg / C, c = seg - b * C, tid = threadIdx.x;
  const float *bp = boxes + (long long)b * N * 4;
  const float *sp = scores + (long long)b * N * S + col0 + c;
  // candidates in ascending row order: each thread owns a contiguous run of rows
  const int per = (N + kNmsThreads - 1) / kNmsThreads;
  const int lo = min(N, tid * per), hi = min(N, lo + per);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += sp[(long long)i * S] > score_thr ? 1 : 0;
  int n;
  int pos = block_exclusive_scan(cnt, wave_sums, n);
  for (int i = lo; i < hi; ++i) {
    const float sc = sp[(long long)i * S];
    if (sc > score_thr) {
      st.bx1[pos] = bp[4 * i]; st.by1[pos] = bp[4 * i + 1]; st.bx2[pos] = bp[4 * i + 2]; st.by2[pos] = bp[4 * i + 3];
      st.bsc[pos] = sc;
      st.bid[pos] = i;
      ++pos;
    }
  }
  __syncthreads();
  const int M = n > 0 ? soft_nms_block(st, n, iou_thr, method, sigma, min_score, wave_sums, wave_best, &n_front_surv) : 0;
  float *od = seg_dets + (long long)seg * N * 5;
  int *os = seg_src + (long long)seg * N;
  for (int i = tid; i < M; i += kNmsThreads) {
    od[5 * i] = st.bx1[i]; od[5 * i + 1] = st.by1[i]; od[5 * i + 2] = st.bx2[i]; od[5 * i + 3] = st.by2[i];
    od[5 * i + 4] = st.bsc[i];
    os[i] = st.bid[i];
  }
  if (tid == 0) seg_count[seg] = M;
}

// The tail of multiclass_nms_kp (bbox_nms_kp.py:52-70) behind the soft-NMS segments: concatenation in class order; more
// than max_num -> the max_num best by (decayed) score, ties: earlier in the concatenation first.  A class's survivors
// leave soft-NMS in non-increasing score order (a box's score is final when it is selected as the maximum of what is left,
// and what is left only decays), so a class contributes at most its FIRST max_num entries to the top max_num: at most
// C * max_num keys are sorted, whatever N is.  One workgroup per image.
__global__ __launch_bounds__(kNmsThreads) void multiclass_soft_select(const float *__restrict__ seg_dets,
                                                                      const int *__restrict__ seg_src,
                                                                      const int *__restrict__ seg_count, int N, int C,
                                                                      int max_num, float *__restrict__ out_det,
                                                                      long long *__restrict__ out_label,
                                                                      long long *__restrict__ out_src,
                                                                      long long *__restrict__ out_count, int max_np) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int prefix[65], tprefix[65];   // full concatenation / truncated (first max_num per class)
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int run = 0, trun = 0;
    for (int c = 0; c < C; ++c) {
      prefix[c] = run; tprefix[c] = trun;
      const int k = seg_count[b * C + c];
      run += k; trun += min(k, max_num);
    }
    prefix[C] = run; tprefix[C] = trun;
  }
  __syncthreads();
  const int T = prefix[C], TT = tprefix[C];
  const int out_n = min(T, max_num);
  auto find = [&](const int *pf, int p) {
    int lo = 0, hi = C - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pf[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  if (T > max_num) {
    int NP = 64;
    while (NP < TT) NP <<= 1;
    for (int q = tid; q < NP; q += kNmsThreads) {
      unsigned long long k = ~0ull;
      if (q < TT) {
        const int c = find(tprefix, q), i = q - tprefix[c];
        k = score_key(seg_dets[(((long long)b * C + c) * N + i) * 5 + 4], (unsigned)(prefix[c] + i));
      }
      keys[q] = k;
    }
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < NP; i += kNmsThreads) {
          const int l = i ^ j;
          if (l > i) {
            const unsigned long long a = keys[i], bb = keys[l];
            const bool up = (i & k) == 0;
            if ((a > bb) == up) { keys[i] = bb; keys[l] = a; }
          }
        }
        __syncthreads();
      }
  }
  (void)max_np;
  for (int r = tid; r < max_num; r += kNmsThreads) {
    float *od = out_det + ((long long)b * max_num + r) * 5;
    if (r < out_n) {
      const int p = T > max_num ? (int)(keys[r] & 0xffffffffu) : r;
      const int c = find(prefix, p), i = p - prefix[c];
      const float *sd = seg_dets + (((long long)b * C + c) * N + i) * 5;
      od[0] = sd[0]; od[1] = sd[1]; od[2] = sd[2]; od[3] = sd[3]; od[4] = sd[4];
      out_label[(long long)b * max_num + r] = c;
      out_src[(long long)b * max_num + r] = seg_src[((long long)b * C + c) * N + i];
    } else {
      od[0] = od[1] = od[2] = od[3] = od[4] = 0.f;
      out_label[(long long)b * max_num + r] = 0;
      out_src[(long long)b * max_num + r] = 0;
    }
  }
  if (tid == 0) out_count[b] = out_n;
}

}  // namespace kgdet

using namespace kgdet;

extern "C" {

size_t kgdet_nms_workspace_bytes(int64_t total_n, int32_t num_segments) {
  // segments of up to 4096 boxes live entirely in LDS; when ANY segment is longer, nms_segments_large gives EVERY
  // segment (empty and tiny ones too) a 256-byte-aligned slice of nms_large_scratch_bytes(n) = np * 8 + 24 n +
  // roundup16(n) bytes with np = max(64, next power of two >= n) < max(64, 2 n).  Per segment that is at most
  // 512 + 16 n + 24 n + n + 15 + 255, so for any split of total_n boxes into num_segments segments:
  const size_t t = (size_t)(total_n > 0 ? total_n : 1), s = (size_t)(num_segments > 0 ? num_segments : 1);
  return 16 + 41 * t + 800 * s;
}

static int nms_launch(const float *dets, const int64_t *seg_offsets, int32_t num_segments, int64_t total_n,
                      int64_t max_seg_len, float iou_thr, int64_t *keep, int64_t *num_keep, void *workspace,
                      size_t workspace_bytes, void *stream) {
  KGDET_CHECK_SHAPE(num_segments >= 0 && max_seg_len >= 0, "negative size");
  if (num_segments == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(dets && seg_offsets && keep && num_keep, "null pointer");
  if (max_seg_len > kNmsMaxLen) {   // no size cliff: the same algorithm on global scratch (nms_wrapper.py:8-49 takes any N)
    const size_t need = kgdet_nms_workspace_bytes(total_n, num_segments);
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("nms: segments beyond %d boxes need %zu bytes of workspace (kgdet_nms_workspace_bytes), got %zu",
                kNmsMaxLen, need, workspace_bytes);
      return KGDET_E_WORKSPACE;
    }
    hipLaunchKernelGGL(nms_segments_large, dim3(num_segments), dim3(kNmsThreads), 0, (hipStream_t)stream, dets,
                       (const long long *)seg_offsets, iou_thr, (long long *)keep, (long long *)num_keep,
                       (unsigned char *)workspace + 16);
    KGDET_CHECK_LAUNCH("nms_segments_large");
    return KGDET_OK;
  }
  int np = 64;
  while (np < max_seg_len) np <<= 1;
  const size_t lds = (size_t)np * 8 + (size_t)np * 5 * 4 + (size_t)np;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)nms_segments, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 256));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)soft_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 256));
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_segments, dim3(num_segments), dim3(kNmsThreads), lds, (hipStream_t)stream, dets,
                     (const long long *)seg_offsets, iou_thr, (long long *)keep, (long long *)num_keep, np);
  KGDET_CHECK_LAUNCH("nms_segments");
  return KGDET_OK;
}

int kgdet_nms_batched(const float *dets, const int64_t *seg_offsets, int32_t num_segments, int64_t total_n,
                      int64_t max_seg_len, float iou_thr, int64_t *keep, int64_t *num_keep, void *workspace,
                      size_t workspace_bytes, void *stream) {
  return nms_launch(dets, seg_offsets, num_segments, total_n, max_seg_len, iou_thr, keep, num_keep, workspace,
                    workspace_bytes, stream);
}

size_t kgdet_multiclass_nms_workspace_bytes(int32_t B, int32_t N, int32_t C) {
  return ((size_t)B * C * N + (size_t)B * C) * sizeof(int64_t);
}

int kgdet_multiclass_nms(const float *boxes, const float *scores, int32_t B, int32_t N, int32_t C,
                         int32_t score_stride, int32_t score_col0, float score_thr, float iou_thr, int32_t max_num,
                         float *out_det, int64_t *out_label, int64_t *out_src, int64_t *out_count, void *workspace,
                         size_t workspace_bytes, void *stream) {
  KGDET_CHECK_SHAPE(B >= 0 && N >= 0 && C > 0 && C <= 64 && max_num > 0, "bad sizes (1 <= C <= 64)");
  KGDET_CHECK_SHAPE(score_col0 >= 0 && score_col0 + C <= score_stride, "score columns outside the row");
  if (B == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(out_det && out_label && out_src && out_count, "null pointer");
  if (N == 0) {
    KGDET_HIP_TRY(hipMemsetAsync(out_det, 0, (size_t)B * max_num * 5 * 4, (hipStream_t)stream));
    KGDET_HIP_TRY(hipMemsetAsync(out_label, 0, (size_t)B * max_num * 8, (hipStream_t)stream));
    KGDET_HIP_TRY(hipMemsetAsync(out_src, 0, (size_t)B * max_num * 8, (hipStream_t)stream));
    KGDET_HIP_TRY(hipMemsetAsync(out_count, 0, (size_t)B * 8, (hipStream_t)stream));
    return KGDET_OK;
  }
  KGDET_CHECK_SHAPE(boxes && scores, "null pointer");
  KGDET_CHECK_SHAPE(workspace && workspace_bytes >= kgdet_multiclass_nms_workspace_bytes(B, N, C), "workspace too small");
  if (N > kNmsMaxLen || (long long)N * C > 16384) {
    set_error("multiclass_nms: %d candidates x %d classes exceed the on-chip limits (%d per class, 16384 per image)",
              N, C, kNmsMaxLen);
    return KGDET_E_UNSUPPORTED;
  }
  int64_t *keep = (int64_t *)workspace, *num_keep = keep + (size_t)B * C * N;
  int np = 64;
  while (np < N) np <<= 1;
  int np2 = 64;
  while (np2 < N * C) np2 <<= 1;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)multiclass_nms_segments,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)multiclass_select, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      16384 * 8));
    attr_set = true;
  }
  const size_t lds = (size_t)np * 8 + (size_t)np * 5 * 4 + (size_t)np;
  hipLaunchKernelGGL(multiclass_nms_segments, dim3(B * C), dim3(kNmsThreads), lds, (hipStream_t)stream, boxes, scores,
                     N, C, score_stride, score_col0, score_thr, iou_thr, (long long *)keep, (long long *)num_keep, np);
  KGDET_CHECK_LAUNCH("multiclass_nms_segments");
  hipLaunchKernelGGL(multiclass_select, dim3(B), dim3(kNmsThreads), (size_t)np2 * 8, (hipStream_t)stream, boxes, scores,
                     N, C, score_stride, score_col0, (const long long *)keep, (const long long *)num_keep, max_num,
                     out_det, (long long *)out_label, (long long *)out_src, (long long *)out_count, np2);
  KGDET_CHECK_LAUNCH("multiclass_select");
  return KGDET_OK;
}

size_t kgdet_multiclass_soft_nms_workspace_bytes(int32_t B, int32_t N, int32_t C) {
  return (size_t)B * C * N * (5 * sizeof(float) + sizeof(int)) + (size_t)B * C * sizeof(int) + 64;
}

// the on-chip limits of kgdet_multiclass_soft_nms in ONE place (callers that must decide before a graph capture ask here)
int kgdet_multiclass_soft_nms_supported(int32_t B, int32_t N, int32_t C, int32_t max_num) {
  if (B < 0 || N < 0 || C <= 0 || C > 64 || max_num <= 0) return 0;
  int np2 = 64;
  while (np2 < C * max_num && np2 <= 16384) np2 <<= 1;
  return ((size_t)N * 9 * 4 <= 160 * 1024 - 256 && np2 <= 16384) ? 1 : 0;
}

int kgdet_multiclass_soft_nms(const float *boxes, const float *scores, int32_t B, int32_t N, int32_t C,
                              int32_t score_stride, int32_t score_col0, float score_thr, float iou_thr, int32_t method,
                              float sigma, float min_score, int32_t max_num, float *out_det, int64_t *out_label,
                              int64_t *out_src, int64_t *out_count, void *workspace, size_t workspace_bytes, void *stream) {
  KGDET_CHECK_SHAPE(B >= 0 && N >= 0 && C > 0 && C <= 64 && max_num > 0, "bad sizes (1 <= C <= 64)");
  KGDET_CHECK_SHAPE(score_col0 >= 0 && score_col0 + C <= score_stride, "score columns outside the row");
  KGDET_CHECK_SHAPE(method >= 0 && method <= 2, "method: 0 hard, 1 linear, 2 gaussian");
  if (B == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(out_det && out_label && out_src && out_count, "null pointer");
  if (N == 0) {
    KGDET_HIP_TRY(hipMemsetAsync(out_det, 0, (size_t)B * max_num * 5 * 4, (hipStream_t)stream));
    KGDET_HIP_TRY(hipMemsetAsync(out_label, 0, (size_t)B * max_num * 8, (hipStream_t)stream));
    KGDET_HIP_TRY(hipMemsetAsync(out_src, 0, (size_t)B * max_num * 8, (hipStream_t)stream));
    KGDET_HIP_TRY(hipMemsetAsync(out_count, 0, (size_t)B * 8, (hipStream_t)stream));
    return KGDET_OK;
  }
  KGDET_CHECK_SHAPE(boxes && scores, "null pointer");
  KGDET_CHECK_SHAPE(workspace && workspace_bytes >= kgdet_multiclass_soft_nms_workspace_bytes(B, N, C), "workspace too small");
  const size_t lds = (size_t)N * 9 * 4;
  int np2 = 64;
  while (np2 < C * max_num) np2 <<= 1;
  if (!kgdet_multiclass_soft_nms_supported(B, N, C, max_num)) {
    set_error("multiclass_soft_nms: %d candidates (limit %d) / %d classes x %d detections (limit 16384 keys) exceed the "
              "on-chip limits", N, (160 * 1024 - 256) / 36, C, max_num);
    return KGDET_E_UNSUPPORTED;
  }
  float *seg_dets = (float *)workspace;
  int *seg_src = (int *)(seg_dets + (size_t)B * C * N * 5);
  int *seg_count = seg_src + (size_t)B * C * N;
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)multiclass_soft_nms_segments,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)multiclass_soft_select, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      16384 * 8));
    attr_set = true;
  }
  hipLaunchKernelGGL(multiclass_soft_nms_segments, dim3(B * C), dim3(kNmsThreads), lds, (hipStream_t)stream, boxes, scores,
                     N, C, score_stride, score_col0, score_thr, iou_thr, (int)method, sigma, min_score, seg_dets, seg_src,
                     seg_count);
  KGDET_CHECK_LAUNCH("multiclass_soft_nms_segments");
  hipLaunchKernelGGL(multiclass_soft_select, dim3(B), dim3(kNmsThreads), (size_t)np2 * 8, (hipStream_t)stream,
                     (const float *)seg_dets, (const int *)seg_src, (const int *)seg_count, N, C, max_num, out_det,
                     (long long *)out_label, (long long *)out_src, (long long *)out_count, np2);
  KGDET_CHECK_LAUNCH("multiclass_soft_select");
  return KGDET_OK;
}

int kgdet_nms(const float *dets, int64_t n, float iou_thr, int64_t *keep, int64_t *num_keep, void *workspace,
              size_t workspace_bytes, void *stream) {
  KGDET_CHECK_SHAPE(n >= 0, "negative size");
  KGDET_CHECK_SHAPE(workspace && workspace_bytes >= 16, "nms workspace must hold 2 x int64");
  // segment table {0, n} built on the device side of the stream
  const int64_t host_offsets[2] = {0, n};
  KGDET_HIP_TRY(hipMemcpyAsync(workspace, host_offsets, sizeof(host_offsets), hipMemcpyHostToDevice,
                               (hipStream_t)stream));
  if (n == 0) {
    KGDET_HIP_TRY(hipMemsetAsync(num_keep, 0, sizeof(int64_t), (hipStream_t)stream));
    return KGDET_OK;
  }
  return nms_launch(dets, (const int64_t *)workspace, 1, n, n, iou_thr, keep, num_keep, workspace, workspace_bytes, stream);
}

int kgdet_soft_nms(const float *dets, int64_t n, float iou_thr, int32_t method, float sigma, float min_score,
                   float *out_dets, int64_t *out_inds, int64_t *num_out, void *stream) {
  KGDET_CHECK_SHAPE(n >= 0, "negative size");
  KGDET_CHECK_SHAPE(num_out, "null pointer");
  if (n == 0) {
    KGDET_HIP_TRY(hipMemsetAsync(num_out, 0, sizeof(int64_t), (hipStream_t)stream));
    return KGDET_OK;
  }
  KGDET_CHECK_SHAPE(dets && out_dets && out_inds, "null pointer");
  const size_t lds = (size_t)n * 9 * 4;
  if (lds > 160 * 1024 - 256) {
    set_error("soft_nms: %lld boxes exceed the on-chip limit", (long long)n);
    return KGDET_E_UNSUPPORTED;
  }
  static thread_local bool attr_set = false;
  if (!attr_set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)soft_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - 256));
    attr_set = true;
  }
  hipLaunchKernelGGL(soft_nms_kernel, dim3(1), dim3(kNmsThreads), lds, (hipStream_t)stream, dets, (int)n, iou_thr,
                     (int)method, sigma, min_score, out_dets, (long long *)out_inds, (long long *)num_out);
  KGDET_CHECK_LAUNCH("soft_nms_kernel");
  return KGDET_OK;
}

}  // extern "C"
