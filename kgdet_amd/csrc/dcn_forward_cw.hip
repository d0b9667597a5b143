// Deformable convolution forward, "column-wave" variant for gfx950 (round 5): FOUR waves per workgroup, one per SIMD, each
// wave owning 32 pixel columns x all 256 output rows of the 256 x 128 tile and sampling exactly the B operand it multiplies.
//
// Same reference path and arithmetic as dcn_forward_plane.hip (deformable_im2col + addmm_,
// mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:190-276, deform_conv_cuda.cpp:221-245; bf16 hi/lo split products, fp32
// accumulation), same schedule objects (DcnFwdGroup: static (problem, part, tile) ranges, slabs + dcn_fwd_fixup_static),
// same operand images, same tap records, same LDS feature plane.  What differs is who does what:
//
//   dcn_fwd_plane:  8 consumer waves (32 rows x 128 columns each: every wave reads the whole B stage from LDS) + 8 producer
//                   waves that sample B into LDS; the roles share the SIMDs' single vector issue port -- the per-wave phase
//                   trace of round 5 shows the younger consumer wave of every SIMD needing ~1070 cycles per stage for 768
//                   cycles of MFMAs (two waves) while every producer instruction issued beside them costs MFMA slots, and
//                   ~20 % of a workgroup's time going to barrier waits and segment hand-overs.
//   dcn_fwd_cw:     a wave's lane l samples channels (l >> 5) * 8 .. + 7 of pixel (l & 31) of the wave's column block --
//                   which IS its B fragment of v_mfma_f32_32x32x16_bf16: the sampled operand never goes through LDS, there
//                   is no producer / consumer hand-over, and the sampling instructions of stage j + 1 are fillers between
//                   the wave's own MFMAs of stage j (same-wave fillers issue in the shadow of a running MFMA; instructions of
//                   another wave do not -- MI355X_MICROARCH.md).  The A (weight) stage, 16 KB, is what the four waves
//                   share: each wave copies a quarter of it global -> registers (two stages ahead) -> LDS ring of three
//                   stages, every wave reads all of it as 16 ds_read_b128 per stage.  One workgroup barrier per stage (four
//                   waves).  512 registers per wave: 128 accumulators, two corner sets, two fragment half-sets, a whole
//                   plane share (17 units) in flight for the segment switch.
//
// Pipeline of iteration i (stage i multiplies):  gathers of stage i + 2 are issued (they complete before the barrier that
// ends the iteration), the corners gathered in iteration i - 1 are interpolated and split into the B fragment of stage
// i + 1 between the MFMAs, tap records and weight pieces of stage i + 4 are requested.  A new channel chunk (segment):
// its plane share is requested four stages ahead into registers and stored once the last gathers of the old plane are
// back -- one iteration without gathers, the next one issues two.
#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

namespace {

#ifdef KGDET_CW_ABL_NOMFMA
#define CW_MFMA(a, b, c) (c)
#else
#define CW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cw_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kCwThreads = 256;
constexpr int kCwAPart = 2 * kTileM * 8 * 2;     // bytes of one part (hi or lo) of an A stage: [khalf 2][o 256][8 bf16]
constexpr int kCwRing = 3;
#ifndef KGDET_CW_SCHED
#define KGDET_CW_SCHED 1
#endif                       // A stages in LDS: being read, complete, being written

#if defined(__HIP_DEVICE_COMPILE__)
typedef const f32x4 __attribute__((address_space(3))) *CwLdsF4;
typedef const cw_bf16x8 __attribute__((address_space(3))) *CwLdsB8;
typedef u32x4_t __attribute__((address_space(3))) *CwLdsU4W;
__device__ __forceinline__ f32x4 cw_lds_f4(unsigned addr) { return *(CwLdsF4)(addr); }
__device__ __forceinline__ cw_bf16x8 cw_lds_b8(unsigned addr) { return *(CwLdsB8)(addr); }
__device__ __forceinline__ void cw_lds_store(unsigned addr, u32x4_t v) { *(CwLdsU4W)(addr) = v; }
#else
__device__ __forceinline__ f32x4 cw_lds_f4(unsigned) { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ cw_bf16x8 cw_lds_b8(unsigned) { return cw_bf16x8{}; }
__device__ __forceinline__ void cw_lds_store(unsigned, u32x4_t) {}
#endif

}  // namespace

#ifdef KGDET_CW_TRACE
// experiment build (make VARIANT=cwtrace EXTRA=-DKGDET_CW_TRACE, tools/cw_trace.py): cycles per workgroup and wave by phase
//   0 top of the iteration (conditional blocks, LDS / memory issue)   1 main block (MFMAs + fillers)   2 barrier wait
//   3 prologue   4 epilogue   5 iterations   6 total
static __device__ unsigned long long g_cw_trace[256 * 4 * 12];
#define CW_TR(cat) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n__ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tr[cat] += n__ - tr_t; tr_t = n__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CW_TR(cat) do { } while (0)
#endif

template <int PARTS>
__global__ __launch_bounds__(kCwThreads, 1) void dcn_fwd_cw(const DcnFwdGroup grp, float *__restrict__ slabs) {
  static_assert(PARTS == 2, "split operands only (the one-product kernel keeps dcn_fwd_plane<1>)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((unsigned)(unsigned long long)(const unsigned char __attribute__((address_space(3))) *)smem != 0u) __builtin_trap();
  constexpr unsigned kRingBase = 4u * kPlaneQuadStride;          // the plane sits at LDS address 0
  constexpr unsigned kAStage = PARTS * kCwAPart;                  // 16 KB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = lane >> 5, l31 = lane & 31;
  const long long G = gridDim.x, g = blockIdx.x;
  const long long slice = sk_slice_of_block((int)g, (int)G);
  int slot = 0;
#ifdef KGDET_CW_TRACE
  unsigned long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_t = __builtin_amdgcn_s_memtime();
  const unsigned long long tr_start = tr_t;
#endif

  for (int round = 0; round < grp.rounds; ++round) {
    long long my_begin, my_end;
    dcn_slice_bounds(grp, slice, G, my_begin, my_end, round);
    if (my_begin >= my_end) continue;
    const DcnUnitPos pos = dcn_unit_pos(grp, my_begin);
    const DcnProblem &p = grp.p[pos.pi];
    const int HW = p.H * p.W, K = p.K, HoWo = p.HoWo;
    const int n_c16 = p.chunks_per_tap, cpt = p.chunks_per_tile;
    const int tile = pos.tile;
    const int s_begin = pos.s, s_end = pos.s_hi;           // static ranges: whole channel chunks
    const int n = s_end - s_begin;
    const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
    const int tile_b = nt / p.tiles_per_image;
    const int hw0 = (nt - tile_b * p.tiles_per_image) * kTileN + wave * 32 + l31;
    const int hw_c = hw0 < HoWo ? hw0 : 0;                 // columns past the end of the image sample pixel 0: never stored
    const int c_first = s_begin / K;                       // (s_begin % K == 0)

    // ---- operand sources
    const dcn_rsrc_t wq_rs = dcn_make_rsrc(p.wq);
    const unsigned a_base = (unsigned)((mt + p.mt_base) * n_c16 * K + s_begin) * kAStage;   // stage k: + k * kAStage
    const unsigned a_lane = (unsigned)(wave * 4096 + lane * 16);                               // this wave's quarter of a stage
    const dcn_rsrc_t rec_rs = dcn_make_rsrc(p.taps);
    const unsigned rec_lane = (unsigned)hw_c * 32u;
    const unsigned rec_tap = (unsigned)HoWo * 32u;
    auto seg_records = [&](int c) {
      const int dgi = p.DG == 1 ? 0 : (p.c_base + min(c * kChunk, p.Cg - 1)) / p.cpdg;
      return (unsigned)((tile_b * p.DG + dgi) * K) * rec_tap;
    };
    // the blocked copy of x (dcn_build_taps): [image][chunk][quad][pixels padded to 64][4 channels] -- a plane is contiguous
    const int nblk = (HW + 63) >> 6, n_units = 4 * nblk;
    const unsigned long long xa = reinterpret_cast<unsigned long long>(p.xblk);
    const u32x4_t xb_rs = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)xa),
                           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(xa >> 32)), 0xffffffffu, 0x00020000u};
    const unsigned plane_bytes = (unsigned)(n_units * 1024);
    const unsigned xb_img = (unsigned)(tile_b * n_c16) * plane_bytes;     // (an image group: < 4 GB)
    auto plane_dma = [&](int c) {   // units wave, wave + 4, ...: (quad, 64 pixels) = 1 KiB each
      const unsigned so = __builtin_amdgcn_readfirstlane(xb_img + (unsigned)c * plane_bytes);
      for (int u = wave; u < n_units; u += 4) {
        const int quad = u / nblk, blk = u - quad * nblk;
        dcn_dma_b128(xb_rs, (unsigned)(lane * 16), so + (unsigned)(u * 1024),
                    (unsigned)quad * (unsigned)kPlaneQuadStride + (unsigned)(blk * 1024));
      }
    };

    // ---- LDS addresses of this lane
    const unsigned a_wr = kRingBase + a_lane;                                      // + ring slot * kAStage + piece * 1024
    const unsigned a_rd = kRingBase + (unsigned)(khalf * (kTileM * 16) + l31 * 16);  // + slot * kAStage + part * kCwAPart + rb * 512
    const unsigned q_base = (unsigned)(2 * khalf) * (unsigned)kPlaneQuadStride;      // this lane's first channel quad

    // ---- register sets
    f32x16 acc[8];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    u32x4_t Roff[4];           // tap records: LDS offsets of the four corners (ring by stage & 3)
    f32x4 Rw[4];               //              and their weights
    f32x4 Cn[2][2][4];         // corner values: [stage & 1][quad][corner]
    cw_bf16x8 Bh[2], Bl[2];    // B fragments (hi, lo parts): [stage & 1]
    u32x4_t Ap[4][4];          // this wave's quarter of a weight stage on its way global -> LDS: [stage & 3][piece]
    cw_bf16x8 H[2][4][PARTS];  // A fragments: [half: row blocks 0-3 / 4-7][row block][part]

    auto rec_issue = [&](unsigned seg_off, int t, u32x4_t &off, f32x4 &w) {
      const unsigned so = __builtin_amdgcn_readfirstlane(seg_off + (unsigned)t * rec_tap);
      off = dcn_buf_b128(rec_rs, rec_lane, so);
      w = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 16u, so));
    };
    auto piece_issue = [&](int k, u32x4_t (&A)[4]) {
      const unsigned so = __builtin_amdgcn_readfirstlane(a_base + (unsigned)min(k, n - 1) * kAStage);
#pragma unroll
      for (int q = 0; q < 4; ++q) A[q] = dcn_buf_b128(wq_rs, a_lane, so + (unsigned)(q * 1024));
    };
    auto piece_store = [&](int ring_slot, const u32x4_t (&A)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) cw_lds_store(a_wr + (unsigned)ring_slot * kAStage + (unsigned)(q * 1024), A[q]);
    };
    auto gathers = [&](const u32x4_t &off, f32x4 (&V)[2][4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned a = off[e] + q_base;
        V[0][e] = cw_lds_f4(a);
        V[1][e] = cw_lds_f4(a + (unsigned)kPlaneQuadStride);
      }
    };
    // channel pair `cp` (0 .. 3) of the lane's eight: interpolate, split into hi / lo bf16
    auto sample_pair = [&](int cp, const f32x4 (&V)[2][4], const f32x4 &w, unsigned (&hi_u)[4], unsigned (&lo_u)[4]) {
      const int q = cp >> 1, c0 = (cp & 1) * 2;
      float s0 = w[0] * V[q][0][c0], s1 = w[0] * V[q][0][c0 + 1];
#pragma unroll
      for (int e = 1; e < 4; ++e) {
        s0 = __builtin_fmaf(w[e], V[q][e][c0], s0);
        s1 = __builtin_fmaf(w[e], V[q][e][c0 + 1], s1);
      }
      const cw_f32x2 sv = {s0, s1};
      const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(sv, cw_bf16x2));
      const cw_f32x2 df = {s0 - __uint_as_float(hu << 16), s1 - __uint_as_float(hu & 0xffff0000u)};
      hi_u[cp] = hu;
      lo_u[cp] = __builtin_bit_cast(unsigned, __builtin_convertvector(df, cw_bf16x2));
    };
    auto a_frags = [&](int ring_slot, int half, cw_bf16x8 (&F)[4][PARTS]) {
      const unsigned base = a_rd + (unsigned)ring_slot * kAStage + (unsigned)(half * 4 * 512);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int part = 0; part < PARTS; ++part) F[rb][part] = cw_lds_b8(base + (unsigned)(part * kCwAPart + rb * 512));
    };

    // ---- prologue: plane of the first chunk and weight stages 0, 1 into LDS; pieces 2 .. 5 and records 0 .. 3 in flight
    __syncthreads();                       // the previous range's readers of the plane and of the ring are done
    plane_dma(c_first);
    piece_issue(0, Ap[0]);
    piece_issue(1, Ap[1]);
    piece_issue(2, Ap[2]);
    piece_issue(3, Ap[3]);
    int tf = 0, cf = c_first;              // (chunk, tap) of the front stage
    unsigned segf = seg_records(cf);
    auto front_next = [&]() { if (++tf == K) { tf = 0; ++cf; segf = seg_records(cf); } };
#pragma unroll
    for (int k = 0; k < 4; ++k) {          // (stages past the end of the range: clamped chunks, harmless loads)
      rec_issue(segf, tf, Roff[k], Rw[k]);
      front_next();
    }
    piece_store(0, Ap[0]);
    piece_store(1, Ap[1]);
    piece_issue(4, Ap[0]);
    piece_issue(5, Ap[1]);
    dcn_wait_vm0();                         // (the plane pieces; hipcc's own waits cover its loads)
    __syncthreads();
    gathers(Roff[0], Cn[0]);
    gathers(Roff[1], Cn[1]);
    a_frags(0, 0, H[0]);
    {
      unsigned hi_u[4], lo_u[4];
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) sample_pair(cp, Cn[0], Rw[0], hi_u, lo_u);
      Bh[0] = __builtin_bit_cast(cw_bf16x8, hi_u);
      Bl[0] = __builtin_bit_cast(cw_bf16x8, lo_u);
    }
    CW_TR(3);

    // ---- stage loop.  tk = tap of stage i (carried); first-of-chunk tests for the stages ahead from it
    int tk = 0;            // tap of stage i
    int c_plane = c_first; // chunk whose plane is in LDS
    auto is_first = [&](int ahead, int i) {   // does stage i + ahead start a new chunk inside the range?
      int t = tk + ahead;
      t = t >= K ? t - K : t;
      return t == 0 && i + ahead < n;          // (ahead <= 2 < K)
    };
    auto iteration = [&](int i, auto PH) {
      constexpr int ph = decltype(PH)::value;      // i & 3
      constexpr int e0 = ph & 1, e1 = e0 ^ 1;
      const int ring_i = i % kCwRing;              // (scalar)
      const int ring_1 = ring_i + 1 == kCwRing ? 0 : ring_i + 1;
      const int ring_2 = ring_1 + 1 == kCwRing ? 0 : ring_1 + 1;
      // A new chunk at stage i + 2: its plane replaces the old one now (the last gathers of the old plane were issued in
      // iteration i - 1 and waited for at the barrier); the gathers of stage i + 2 issued below read garbage and are
      // repeated at the top of the next iteration (sw1), FIRST among that iteration's LDS operations so that the counted
      // lgkmcnt before the interpolation is exact on both paths.
      const bool sw2 = is_first(2, i), sw1 = is_first(1, i);
      if (sw1) gathers(Roff[(ph + 1) & 3], Cn[e1]);
      if (sw2) plane_dma(++c_plane);
      CW_TR(7);
      // ---- The iteration's body in ISSUE ORDER: 24 units of one MFMA and its fillers -- at most five vector / LDS / memory
      // instructions of the wave's own, which issue in the shadow of the running MFMA -- fenced by sched_barrier so that hipcc keeps
      // the order (left to itself it issues the LDS operations in one burst: a wave holds at most 15 of them in flight --
      // lgkmcnt is four bits -- and the sixteenth waited for the first gathers, ~1000 cycles per stage in the first version;
      // sched_group_barrier pipelines over this block were not honoured).
      //   units  0 ..  7: MFMAs of row blocks 0-3 (lo x hi, hi x lo) | the A fragments of row blocks 4-7 | channels 0 .. 7 interpolated
      //   units  8 .. 11: row blocks 0-3 (hi x hi)                   | gathers of stage i + 2           | the four hi / lo splits
      //   units 12 .. 15: row blocks 4-7 (lo x hi)                   | weight piece i + 2 -> LDS, piece i + 6 requested
      //   units 16 .. 23: row blocks 4-7 (hi x lo, hi x hi)          | A fragments of stage i + 1, row blocks 0-3 | records
      const cw_bf16x8 bh = Bh[e0], bl = Bl[e0];
      const f32x4 wv = Rw[(ph + 1) & 3];
      float sv[8];
      unsigned hi_u[4], lo_u[4];
      const unsigned a_rd_i = a_rd + (unsigned)ring_i * kAStage + 4u * 512u, a_rd_1 = a_rd + (unsigned)ring_1 * kAStage;
      const unsigned a_wr_2 = a_wr + (unsigned)ring_2 * kAStage;
      const unsigned a_so = __builtin_amdgcn_readfirstlane(a_base + (unsigned)min(i + 6, n - 1) * kAStage);
      const unsigned r_so = __builtin_amdgcn_readfirstlane(segf + (unsigned)tf * rec_tap);
      auto interp = [&](int ch) {      // channel ch of the lane's eight from the corners of stage i + 1
        const int q = ch >> 2, c = ch & 3;
        float v = wv[0] * Cn[e1][q][0][c];
#pragma unroll
        for (int e = 1; e < 4; ++e) v = __builtin_fmaf(wv[e], Cn[e1][q][e][c], v);
        sv[ch] = v;
      };
      auto split = [&](int cp) {       // channels 2 cp, 2 cp + 1 -> hi / lo bf16 pairs
        const cw_f32x2 pv = {sv[2 * cp], sv[2 * cp + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, cw_bf16x2));
        const cw_f32x2 df = {pv[0] - __uint_as_float(hu << 16), pv[1] - __uint_as_float(hu & 0xffff0000u)};
        hi_u[cp] = hu;
        lo_u[cp] = __builtin_bit_cast(unsigned, __builtin_convertvector(df, cw_bf16x2));
      };
#define CW_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef KGDET_CW_ABL_NOGATHER
#define CW_GATHER(addr) f32x4{__uint_as_float(addr), 1.f, 2.f, 3.f}
#else
#define CW_GATHER(addr) cw_lds_f4(addr)
#endif
#ifdef KGDET_CW_ABL_NOAFRAG
#define CW_AFRAG(dst, addr) do { if ((addr) == 0xffffffffu) dst = cw_lds_b8(addr); } while (0)
#else
#define CW_AFRAG(dst, addr) dst = cw_lds_b8(addr)
#endif
#pragma unroll
      for (int u = 0; u < 8; ++u) {            // rb = u & 3: pass u >> 2 (0: A lo x B hi, 1: A hi x B lo)
        const int rb = u & 3;
        acc[rb] = (u >> 2) == 0 ? CW_MFMA(H[0][rb][1], bh, acc[rb]) : CW_MFMA(H[0][rb][0], bl, acc[rb]);
        CW_AFRAG(H[1][u >> 1][u & 1], a_rd_i + (unsigned)((u & 1) * kCwAPart + (u >> 1) * 512));
#ifndef KGDET_CW_ABL_NOVALU
        interp(u);
#else
        sv[u] = wv[u & 3];
#endif
        CW_FENCE();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {            // row blocks 0-3, hi x hi | gather of corner u, first quad | split
        acc[u] = CW_MFMA(H[0][u][0], bh, acc[u]);
        Cn[e0][0][u] = CW_GATHER(Roff[(ph + 2) & 3][u] + q_base);
#ifndef KGDET_CW_ABL_NOVALU
        split(u);
#else
        hi_u[u] = __float_as_uint(sv[2 * u]); lo_u[u] = __float_as_uint(sv[2 * u + 1]);
#endif
        CW_FENCE();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {            // row blocks 4-7, lo x hi | gather of corner u, second quad | piece i + 6 requested
        acc[4 + u] = CW_MFMA(H[1][u][1], bh, acc[4 + u]);
        Cn[e0][1][u] = CW_GATHER(Roff[(ph + 2) & 3][u] + q_base + (unsigned)kPlaneQuadStride);
        CW_FENCE();
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {            // row blocks 4-7, hi x lo, hi x hi | A fragments of stage i + 1 | piece i + 2 -> LDS
        const int rb = u & 3;
        acc[4 + rb] = (u >> 2) == 0 ? CW_MFMA(H[1][rb][0], bl, acc[4 + rb]) : CW_MFMA(H[1][rb][0], bh, acc[4 + rb]);
        CW_AFRAG(H[0][u >> 1][u & 1], a_rd_1 + (unsigned)((u & 1) * kCwAPart + (u >> 1) * 512));
#ifndef KGDET_CW_ABL_NOAPIECE
        if (u < 4) {
          cw_lds_store(a_wr_2 + (unsigned)(u * 1024), Ap[(ph + 2) & 3][u]);
          Ap[(ph + 2) & 3][u] = dcn_buf_b128(wq_rs, a_lane, a_so + (unsigned)(u * 1024));
        }
#endif
        // (the record of stage i + 4 replaces the one of stage i, whose weights were last used in iteration i - 1)
        if (u == 4) Roff[ph] = dcn_buf_b128(rec_rs, rec_lane, r_so);
        if (u == 5) Rw[ph] = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 16u, r_so));
        CW_FENCE();
      }
#undef CW_FENCE
#undef CW_GATHER
#undef CW_AFRAG
      Bh[e1] = __builtin_bit_cast(cw_bf16x8, hi_u);
      Bl[e1] = __builtin_bit_cast(cw_bf16x8, lo_u);
      // (pinned: hipcc otherwise sinks the interpolation -- first used in the next iteration -- behind the MFMAs and the
      // branch below, where nothing overlaps it)
      asm volatile("" : "+v"(Bh[e1]), "+v"(Bl[e1]));
      if (++tk == K) tk = 0;
      front_next();
      CW_TR(1);
      if (sw2) dcn_wait_vm0();                      // the new plane is in LDS before anybody gathers from it
#ifndef KGDET_CW_ABL_NOBARRIER
      __syncthreads();
#endif
      CW_TR(2);
#ifdef KGDET_CW_TRACE
      tr[5] += 1;
#endif
    };
    for (int i = 0; i < n; i += 4) {
      iteration(i, std::integral_constant<int, 0>{});
      if (i + 1 < n) iteration(i + 1, std::integral_constant<int, 1>{});
      if (i + 2 < n) iteration(i + 2, std::integral_constant<int, 2>{});
      if (i + 3 < n) iteration(i + 3, std::integral_constant<int, 3>{});
    }

    // ---- epilogue: the tile, or this range's partial tile in the slab format of wave layout 1 (dcn_fwd_fixup_static reads
    // float4 column j = ni * 4 + q of "wave" rb: here the wave is the column block ni and rb the row block)
    if (s_begin == 0 && s_end == cpt) {
      int b, hw;
      if (tile_pixel(p, nt, wave * 32 + l31, b, hw)) {
        float *obase = p.out + ((long long)b * p.O_total + p.o_base) * HoWo + hw;
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = mt * kTileM + rb * 32 + mfma_row(r, lane);
            if (o >= p.Og) continue;
            float v = acc[rb][r];
            if (p.bias) v += p.bias[p.bias_base + o];
            if (p.flags & 1u /* KGDET_DCN_RELU */) v = fmaxf(v, 0.0f);
            obase[(long long)o * HoWo] = v;
          }
      }
    } else {
      f32x4 *s4 = reinterpret_cast<f32x4 *>(slabs + ((long long)g * grp.slots + slot) * kTileElems);
#pragma unroll
      for (int rb = 0; rb < 8; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[rb][4 * q], acc[rb][4 * q + 1], acc[rb][4 * q + 2], acc[rb][4 * q + 3]};
          s4[(wave * 4 + q) * kThreads + rb * 64 + lane] = v;
        }
    }
    ++slot;
    CW_TR(4);
  }
#ifdef KGDET_CW_TRACE
  tr[6] = __builtin_amdgcn_s_memtime() - tr_start;
  if (lane == 0)
    for (int c = 0; c < 12; ++c) g_cw_trace[((int)blockIdx.x * 4 + wave) * 12 + c] = tr[c];
#endif
}

template __global__ void dcn_fwd_cw<2>(const DcnFwdGroup grp, float *__restrict__ slabs);


#ifdef KGDET_CW_TRACE
}  // namespace kgdet
extern "C" int kgdet_debug_read_cw_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kgdet::g_cw_trace), sizeof(unsigned long long) * 256 * 4 * 12);
}
namespace kgdet {
#endif

int dcn_fwd_cw_threads() { return kCwThreads; }
size_t dcn_fwd_cw_lds_bytes(int parts) { return (size_t)4 * kPlaneQuadStride + (size_t)kCwRing * parts * kCwAPart; }

}  // namespace kgdet
