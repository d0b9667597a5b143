// Deformable convolution backward w.r.t. the offsets, round-6 decomposition: TAP PAIRS on 32x32x16 MFMA blocks, two wave
// halves that alternate between the matrix pipe and the LDS (gfx950).
//
// Reference path replaced: deformable_col2im_coord (deform_conv_cuda_kernel.cu:337-435, get_coordinate_weight :144-187)
// applied to columns = W^T grad_out (deform_conv_cuda.cpp:329-332); same product as dcn_backward_offset.hip:
//     grad_offset[b, 2t + dir, p] = sum_c colgrad[c, t, p] * d sample(x[b, c], pos(p, t)) / d dir,
//     colgrad[c, t, p]            = sum_o W[o, c, t] * grad_out[b, o, p].
//
// What bounded dcn_bwd_offset_plane (profiles/r05_dcn_bwd_plane_kernels.md): a stage was a 16 (channels) x 16 (pixels)
// block per wave, so each of its eight consumer waves read the WHOLE 16 KB W^T stage from LDS (128 KB of ds_read per
// stage and workgroup), every stage ended in a workgroup barrier, and the LDS array (1065 busy cycles per stage) and
// the matrix pipe (768) were never busy at the same time; the K x 128 x 2 tap accumulators took 50 KB of LDS at K = 49,
// which left no room for a second pair of W^T buffers.  Here:
//   * a wave keeps the grad_out fragment of 32 pixels -- [256 o][32 px], bf16 hi / lo, 128 VGPRs -- and a stage PAIR
//     (two taps of one 16-channel chunk) is ONE 32 x 32 block: rows = (tap, channel), 16 k-steps of
//     v_mfma_f32_32x32x16_bf16 x 3 products.  Twice the flops per W^T byte read from LDS: 64 KB of ds_read per stage
//     and workgroup instead of 128;
//   * the eight waves are 4 pixel groups x 2 HALVES; the two waves of a SIMD (w and w + 4) are the two halves of one
//     pixel group.  Pairs alternate between the halves: in slot q half (q & 1) multiplies pair q (48 MFMAs back to
//     back, W^T fragments from the pair buffer of its parity) while the other half runs the derivative tail of pair
//     q - 1 (corner reads from the x plane, dot products, per-tap sums) -- matrix work beside LDS / VALU work on every
//     SIMD, one workgroup barrier per PAIR of stages;
//   * W^T pairs (2 taps x (hi, lo) x 8 KB) arrive by LDS-DMA (buffer_load ... lds, no registers, no ds_write, no
//     producer waves: all eight waves have 256 registers) one slot ahead, into the buffer the other half read in
//     the slot before.  A 1 KiB piece costs its wave 100-185 cycles of issue in this loop (phase trace, tools/pair_trace.py);
//     the tail half issues five of a wave's eight behind its dot products, the matrix half three between its MFMAs;
//   * the accumulator layout gives a lane two channel quads of each tap for one pixel; after the dot products the two
//     lane halves are combined with ONE v_permlane32_swap (lower lanes end up with d/dy, upper lanes with d/dx);
//   * a tap occurs once per segment (chunk-major / tap-minor order), so its running sum needs no fast accumulator:
//     it lives in the range's slab (or in grad_offset itself when the range is the whole reduction) and is
//     read-modify-written with one coalesced load / store per pair and lane -- no LDS accumulators, any K.
// LDS: x plane [4 quads][1344 px][4 ch] fp32 at address 0 (86 016 B, tap-record offsets are absolute addresses and the
// quad is an immediate) | two pair buffers of 32 KB.  The plane of the next chunk arrives by LDS-DMA from a blocked copy of x
// (dcn_build_grad_taps) where the workspace has room for one: 2.7 k instead of 8.4 k cycles per switch.
// Static (problem, part, tile) ranges only, one range per workgroup; v1, split operands, K >= 3, Og % 32 == 0; everything else
// stays on dcn_bwd_offset_plane.  Deterministic (fixed summation orders, no atomics).
// Measured (one head stage at B = 2, kernel average of 30 under rocprofv3): 169-173 us against 254 for dcn_bwd_offset_plane<2>.
// What bounds it now: matrix phase + tail phase of a SIMD's two waves stay ~5.2 k cycles per pair of slots whatever is moved
// between them -- beside the wave whose next MFMA waits for the matrix pipe the other wave's vector instructions are not free
// (tools/experiments/README.md, round 6; a single-stream variant with one wave per SIMD is kept there).
#include "dcn_plane.h"

namespace kgdet {

namespace {

constexpr int kPairThreads = 512;                       // 8 waves: 4 pixel groups x 2 halves
constexpr int kPairBuf = 2 * 2 * kAPart;                // two taps x (hi, lo) x 8 KB
constexpr unsigned kPairA0 = 4u * kPlaneQuadStride;     // LDS address of pair buffer 0
constexpr int kPairKs = 16;                             // k-steps of 16 output channels: Og <= 256
#ifndef KGDET_PAIR_MATRIX_PRIO
#define KGDET_PAIR_MATRIX_PRIO 2                        // s_setprio of a wave while it multiplies (the tail half runs at 0)
#endif
#ifndef KGDET_PAIR_DMA_POS
#define KGDET_PAIR_DMA_POS 1                            // where the tail half issues its DMA pieces: 0 behind the corner reads, 1 behind the dot products
#endif
#ifndef KGDET_PAIR_M_PIECES
#define KGDET_PAIR_M_PIECES 3                           // DMA pieces (of a wave's eight) the MATRIX half issues between its MFMAs; the tail half issues the rest
#endif
#ifndef KGDET_PAIR_PACE
#define KGDET_PAIR_PACE 0                               // > 0: s_nop (PACE - 1) behind every MFMA (leaves the SIMD's issue port to the tail wave)
#endif
#ifndef KGDET_PAIR_DEPTH
#define KGDET_PAIR_DEPTH 3                              // k-steps of W^T fragments in flight ahead of the MFMAs
#endif

typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));

struct PairRec {          // what a lane needs for one (pixel, tap): the 48-byte record of dcn_build_grad_taps + the running sum
  u32x4_t off;
  f32x4 wy, wx;
  float prev;
};

#if KGDET_PAIR_PACE > 0
#define KGDET_PAIR_PACE_NOP() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(KGDET_PAIR_PACE - 1)); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define KGDET_PAIR_PACE_NOP() do { } while (0)
#endif
#ifdef KGDET_PAIR_TRACE
// Experiment build only (make VARIANT=pair_trace EXTRA=-DKGDET_PAIR_TRACE, tools/pair_trace.py): per-wave cycle sums by phase.
//   [block][wave][cat]: 0 prologue, 1 DMA issue, 2 matrix phase, 3 tail phase, 4 vmcnt(0) wait, 5 barrier wait, 6 plane switch,
//   7 whole kernel, 8 slots, 9 segments
static __device__ unsigned long long g_pair_trace[256 * 8 * 10];
#define KGDET_PT_ADD(cat) do { const unsigned long long n__ = __builtin_amdgcn_s_memtime(); tr[cat] += n__ - tr_t; tr_t = n__; } while (0)
#else
#define KGDET_PT_ADD(cat) do { } while (0)
#endif

}  // namespace

#ifdef KGDET_PAIR_TRACE
extern "C" int kgdet_debug_read_pair_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kgdet::g_pair_trace), sizeof(unsigned long long) * 256 * 8 * 10);
}
#endif

size_t dcn_bwd_offset_pair_lds_bytes() { return (size_t)kPairA0 + 2 * kPairBuf; }
int dcn_bwd_offset_pair_threads() { return kPairThreads; }

__global__ __launch_bounds__(kPairThreads, 1) void dcn_bwd_offset_pair(const DcnFwdGroup grp, float *__restrict__ slabs,
                                                                       int max_K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const f32x4 __attribute__((address_space(3))) *LdsQuadPtr;
  typedef const bf16x8p __attribute__((address_space(3))) *LdsFragPtr;
  auto lds_quad = [](unsigned addr) { return *(LdsQuadPtr)(addr); };
  auto lds_frag = [](unsigned addr) { return *(LdsFragPtr)(addr); };
#else
  auto lds_quad = [](unsigned) { return f32x4{0.f, 0.f, 0.f, 0.f}; };
  auto lds_frag = [](unsigned) { return bf16x8p{}; };
#endif
  if ((unsigned)(unsigned long long)(const unsigned char __attribute__((address_space(3))) *)smem != 0u) __builtin_trap();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, pg = wave & 3;           // waves w and w + 4 share a SIMD: the two halves of pixel group w
  const int px32 = lane & 31, hh = lane >> 5;
  const long long G = gridDim.x, g = blockIdx.x;
  const long long slice = sk_slice_of_block((int)g, (int)G);
  long long my_begin, my_end;
  dcn_slice_bounds(grp, slice, G, my_begin, my_end);
  if (my_begin >= my_end) return;

#ifdef KGDET_PAIR_TRACE
  unsigned long long tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_t = __builtin_amdgcn_s_memtime();
  const unsigned long long tr_start = tr_t;
#endif
  const DcnUnitPos pos = dcn_unit_pos(grp, my_begin);
  const DcnProblem &p = grp.p[pos.pi];
  const int HW = p.H * p.W, K = p.K, HoWo = p.HoWo, Og = p.Og, Cg = p.Cg;
  const int tile_b = pos.tile / p.tiles_per_image;
  const int tile_px0 = (pos.tile - tile_b * p.tiles_per_image) * kTileN;
  const int n_o16 = (Og + kChunk - 1) / kChunk;
  const int s_begin = pos.s, s_end = pos.s + (int)(my_end - my_begin);
  const bool whole = s_begin == 0 && s_end == p.chunks_per_tile && p.sum_count <= 1;   // (sum group: every range through its slab)
  const int seg0 = s_begin / K, nseg = (s_end - s_begin) / K;        // (static ranges are whole channel chunks)
  const int np = (K + 1) >> 1, Q = nseg * np;
  const int c16_base = p.c16_base;
  const float *xb = p.x + ((long long)tile_b * p.C_total + p.c_base) * HW;
  const int my_px = tile_px0 + pg * 32 + px32;
  const int my_px_c = my_px < HoWo ? my_px : 0;         // (columns past the end of the image redo pixel 0: never used)

  // ---- where the per-tap sums live: the range's slab [K][128][2], or grad_offset itself
  const float *acc_base;
  unsigned acc_voff, acc_tstride;
  bool store_ok = true;
  if (whole) {
    acc_base = p.goff + ((long long)(tile_b * p.DG + p.dgi) * 2 * K) * HoWo;
    acc_voff = (unsigned)(hh * HoWo + my_px_c) * 4u;
    acc_tstride = (unsigned)(2 * HoWo) * 4u;
    store_ok = my_px < HoWo;
  } else {
    acc_base = slabs + ((long long)g * grp.slots) * (size_t)(max_K * kTileN * 2);
    acc_voff = (unsigned)((pg * 32 + px32) * 2 + hh) * 4u;
    acc_tstride = (unsigned)(kTileN * 2) * 4u;
  }
  const dcn_rsrc_t acc_rs = dcn_make_rsrc(acc_base);
  const dcn_rsrc_t rec_rs = dcn_make_rsrc(p.taps);
  const unsigned rec_lane = (unsigned)my_px_c * 48u;
  const unsigned rec_img = (unsigned)(tile_b * K) * (unsigned)(HoWo * 48);
  const unsigned rec_tstride = (unsigned)(HoWo * 48);
  const unsigned long long wqa = reinterpret_cast<unsigned long long>(p.wq);
  const u32x4_t wq_rs_raw = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)wqa),
                             (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(wqa >> 32)), 0xffffffffu, 0x00020000u};
  // lane part of a DMA piece: [o16 parity = lane >> 5][khalf = (lane >> 4) & 1][c = lane & 15] inside wqt[ct][o16][t][part][khalf][c 256][8 o]
  const unsigned dma_voff = (unsigned)(((lane >> 4) & 1) * (kTileM * 16) + (lane & 15) * 16) + (unsigned)(lane >> 5) * (unsigned)K * (unsigned)(2 * kAPart);
  const unsigned o16_stride2 = (unsigned)(2 * K) * (unsigned)(2 * kAPart);   // two 16-o chunks further
  // this wave's four 1 KiB pieces of a pair: tap (wave >> 2), part ((wave >> 1) & 1), pieces 4 (wave & 1) .. + 3
  // (in the slots: the four waves of the TAIL half issue eight pieces each -- tap (pg >> 1), part (pg & 1); the matrix half issues none)
  // consumer side: lane (row = lane & 31 -> tap row >> 4, channel row & 15; k half = lane >> 5) of this half's pair buffer
  const unsigned a_lane = kPairA0 + (unsigned)half * kPairBuf + (unsigned)((lane & 31) >> 4) * (2 * kAPart) +
                          (unsigned)hh * 256u + (unsigned)(lane & 15) * 16u;
  const unsigned hh_add = (unsigned)hh * (unsigned)kPlaneQuadStride;   // this lane's quads: hh and 2 + hh

  // A pair's DMA = 32 pieces of 1 KiB, four per wave.  The pieces are NOT issued in one burst at the top of a slot: the CU's vector
  // memory path takes 64 B per clock, so 32 pieces keep it busy for >= 512 cycles and every wave sat in the issue of its own four
  // (800-1300 cycles per slot in the phase trace of the first version); they are spread over the slot instead, between the
  // MFMAs of the matrix half and behind the corner reads of the tail half.
  unsigned d_so = 0, d_lds = 0;
  bool d_on = false;
  auto plan_dma = [&](int seg, int j, int buf, bool on, int d_tapsel, int d_part) {
    const int cc = (seg0 + seg + c16_base) * kChunk;
    const int ct = cc / kTileM, c_in = cc % kTileM;
    const int t = d_tapsel ? min(2 * j + 1, K - 1) : 2 * j;
    d_so = (unsigned)(ct * n_o16 * K + t) * (unsigned)(2 * kAPart) + (unsigned)(d_part * kAPart + c_in * 16);
    d_lds = kPairA0 + (unsigned)buf * kPairBuf + (unsigned)d_tapsel * (2 * kAPart) + (unsigned)d_part * kAPart;
    d_on = on;
  };
  auto dma_piece = [&](int ii) __attribute__((always_inline)) {
#ifndef KGDET_PAIR_ABL_NODMA     // (ablation builds: results are wrong by design)
    if (d_on && 2 * ii < n_o16)    // (uniform) pieces past the last 16-o chunk are never loaded: those rows stay zero
      dcn_dma_b128(wq_rs_raw, dma_voff, d_so + (unsigned)ii * o16_stride2, d_lds + (unsigned)ii * 1024u);   // (n_o16 is even)
#endif
  };
  // The x plane of a segment: LDS-DMA from the blocked copy of x when the launch built one (dcn_build_grad_taps: 1 KiB units in
  // exactly the plane's order -- ~9 pieces per wave, no registers; the copy through registers took 8.4 k cycles per switch in
  // the phase trace, 16 % of a 5x5 workgroup), else through registers.  The caller waits (vmcnt) and publishes (barrier).
  const unsigned long long xba = reinterpret_cast<unsigned long long>(p.xblk);
  const u32x4_t xb_rs = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)xba),
                         (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(xba >> 32)), 0xffffffffu, 0x00020000u};
  const bool xb_dma = p.xblk != nullptr;
  const int plane_units = dcn_plane_units(HW), plane_nblk = plane_units >> 2;
  const unsigned xblk_img = (unsigned)(tile_b * p.chunks_per_tap) * (unsigned)(plane_units * 1024);
  auto copy_plane = [&](int seg) {
    if (xb_dma) {
      const unsigned so = xblk_img + (unsigned)(seg0 + seg) * (unsigned)(plane_units * 1024);
      for (int u = wave; u < plane_units; u += kPairThreads / 64) {
        const int quad = u / plane_nblk, blk = u - quad * plane_nblk;
        dcn_dma_b128(xb_rs, (unsigned)(lane * 16), so + (unsigned)(u * 1024),
                     (unsigned)quad * (unsigned)kPlaneQuadStride + (unsigned)(blk * 1024));
      }
      dcn_wait_vm0();
    } else {
      dcn_plane_copy<kPlaneRounds, true>(xb, HW, Cg, (seg0 + seg) * kChunk, smem, (unsigned)kPlaneQuadStride, wave, kPairThreads / 64,
                                         plane_units, lane);
    }
  };

  // ---- prologue: zero pair buffers (only when some 16-o chunks do not exist), first plane, grad_out fragment, pair 0
  if (n_o16 < kPairKs) {
    for (int i = tid; i < 2 * kPairBuf / 16; i += kPairThreads)
      *reinterpret_cast<f32x4 *>(smem + kPairA0 + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
  }
  plan_dma(0, 0, 0, true, wave >> 2, (wave >> 1) & 1);        // pair 0: all eight waves, four pieces each
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_piece((wave & 1) * 4 + i);
  copy_plane(0);
  bf16x8p gh[kPairKs], gl[kPairKs];
  {
    const float *gimg = p.gout + ((long long)tile_b * p.O_total + p.o_base) * HoWo + my_px_c;
#pragma unroll
    for (int ks = 0; ks < kPairKs; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = gimg[(long long)min(ks * 16 + hh * 8 + j, Og - 1) * HoWo];   // unconditional, clamped
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (ks * 16 + hh * 8 + j < Og) ? v[j] : 0.0f;
        gh[ks][j] = (__bf16)f;
        gl[ks][j] = (__bf16)(f - (float)gh[ks][j]);
      }
    }
  }
  dcn_wait_vm0();
  __syncthreads();
  KGDET_PT_ADD(0);

  f32x16 acc0, acc1;
  PairRec RA, RB;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  RA.prev = 0.f; RB.prev = 0.f;
  // The 48-byte records of a pair's two taps: loaded by the wave that will finish the pair, one whole pair of slots ahead --
  // at the end of its previous tail (the first ones here) -- so that the matrix phase issues nothing but fragment reads and MFMAs
  auto load_records = [&](int j_) __attribute__((always_inline)) {
    const int tA = 2 * j_, tB = min(2 * j_ + 1, K - 1);
    const unsigned soA = rec_img + (unsigned)tA * rec_tstride, soB = rec_img + (unsigned)tB * rec_tstride;
    RA.off = dcn_buf_b128(rec_rs, rec_lane, soA);
    RA.wy = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 16, soA));
    RA.wx = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 32, soA));
    RB.off = dcn_buf_b128(rec_rs, rec_lane, soB);
    RB.wy = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 16, soB));
    RB.wx = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 32, soB));
    // the taps' running sums: written one segment ago = np >= 2 slots before the slot this runs in (K >= 3); first segment: not used
    RA.prev = dcn_buf_f32(acc_rs, acc_voff, (unsigned)tA * acc_tstride);
    RB.prev = dcn_buf_f32(acc_rs, acc_voff, (unsigned)tB * acc_tstride);
  };

  // pair q = (segment seg, pair j inside it); slot q: half (q & 1) multiplies pair q, the other half finishes pair q - 1
  int seg = 0, j = 0;          // pair of this slot
  int pseg = 0, pj = 0;        // pair of the slot before
  for (int q = 0; q <= Q; ++q) {
    // the pair after this one into the buffer the OTHER half read one slot ago
    int nseg_ = seg, nj = j + 1;
    if (nj == np) { nj = 0; ++nseg_; }
    plan_dma(nseg_, nj, (q + 1) & 1, q + 1 < Q, pg >> 1, pg & 1);
    KGDET_PT_ADD(1);

    const bool matrix_half = ((q ^ half) & 1) == 0;
    if (matrix_half) {
      if (q < Q) {
        // ---- matrix phase: the pair's records and running sums requested (used one slot later), then 48 MFMAs on the pair
        // buffer of this half
        load_records(j);
        __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks these loads below the MFMAs, right in front of the slot's vmcnt(0))
#ifndef KGDET_PAIR_ABL_NOMFMA
        __builtin_amdgcn_s_setprio(KGDET_PAIR_MATRIX_PRIO);
        // W^T fragments kDepth k-steps ahead of the MFMAs that use them (the fragment ring shares its registers with the tail's
        // corner quads: a wave is in one phase or the other); hipcc's own schedule kept ONE k-step in flight and the MFMAs
        // waited out the LDS latency sixteen times per pair (matrix phase 2250 cycles for 1536 of MFMAs in the phase trace)
        constexpr int kDepth = KGDET_PAIR_DEPTH;
        bf16x8p fa[kDepth + 1][2];
#pragma unroll
        for (int i = 0; i < kDepth; ++i) {
          fa[i][0] = lds_frag(a_lane + i * 512);
          fa[i][1] = lds_frag(a_lane + kAPart + i * 512);
        }
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kPairKs; ++ks) {
          if (ks + kDepth < kPairKs) {
            fa[(ks + kDepth) % (kDepth + 1)][0] = lds_frag(a_lane + (ks + kDepth) * 512);
            fa[(ks + kDepth) % (kDepth + 1)][1] = lds_frag(a_lane + kAPart + (ks + kDepth) * 512);
          }
          if constexpr (KGDET_PAIR_M_PIECES > 0) {
            constexpr int kEvery = 12 / (KGDET_PAIR_M_PIECES > 0 ? KGDET_PAIR_M_PIECES : 1);
            if (ks % kEvery == kEvery - 1 && ks / kEvery < KGDET_PAIR_M_PIECES) dma_piece(8 - KGDET_PAIR_M_PIECES + ks / kEvery);
          }
          const bf16x8p ah = fa[ks % (kDepth + 1)][0], al = fa[ks % (kDepth + 1)][1];
          if (ks == 0) {   // (the chains start from the inline constant 0: no 32 v_mov per pair)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh[ks], zero16, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh[ks], zero16, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl[ks], acc0, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
          } else if (ks & 1) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh[ks], acc1, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh[ks], acc0, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl[ks], acc1, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
          } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gh[ks], acc0, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gh[ks], acc1, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gl[ks], acc0, 0, 0, 0);
            KGDET_PAIR_PACE_NOP();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
#else
        acc0[0] = RA.wy[0] + __builtin_bit_cast(float, (unsigned)gh[3][1] << 16);
#endif
      }
      KGDET_PT_ADD(2);
    } else if (q == 0) {
#pragma unroll
      for (int i = 0; i < 8 - KGDET_PAIR_M_PIECES; ++i) dma_piece(i);
    } else {
      // ---- tail of pair q - 1: the lane holds colgrad of channel quads hh and 2 + hh of both taps for its pixel
      const bool first = pseg == 0;
      const int tA = 2 * pj;
      const bool hasB = 2 * pj + 1 < K;
      // Everything this tail reads from registers was loaded one slot pair ago.  hipcc waits for a load at its first use with a
      // COUNTED vmcnt, and it does not count the LDS-DMA pieces: a first use behind the pieces would wait until they have landed.
      // So all of it is "used" here, in front of the pieces.
      asm volatile("" ::"v"(RA.off), "v"(RA.wy), "v"(RA.wx), "v"(RA.prev), "v"(RB.off), "v"(RB.wy), "v"(RB.wx), "v"(RB.prev));
      float cg[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cg[r] = acc0[r] + acc1[r];
      auto tail = [&](const PairRec &R, const float *c8, int t, int piece) __attribute__((always_inline)) {
        // all eight corner quads in flight before the first dot product (two at a time would be four LDS round trips per tap)
        f32x4 v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned a = R.off[e] + hh_add;
          v[2 * e] = lds_quad(a);
          v[2 * e + 1] = lds_quad(a + 2u * kPlaneQuadStride);
        }
        __builtin_amdgcn_sched_barrier(0);
#if KGDET_PAIR_DMA_POS == 0
#pragma unroll
        for (int i = 0; i < 4; ++i) if (piece + i < 8 - KGDET_PAIR_M_PIECES) dma_piece(piece + i);
        __builtin_amdgcn_sched_barrier(0);
#endif
        float gy = 0.f, gx = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 v0 = v[2 * e], v1 = v[2 * e + 1];
          float d = c8[0] * v0[0];
          d = fmaf(c8[1], v0[1], d); d = fmaf(c8[2], v0[2], d); d = fmaf(c8[3], v0[3], d);
          d = fmaf(c8[4], v1[0], d); d = fmaf(c8[5], v1[1], d); d = fmaf(c8[6], v1[2], d); d = fmaf(c8[7], v1[3], d);
          gy = fmaf(R.wy[e], d, gy);
          gx = fmaf(R.wx[e], d, gx);
        }
#if KGDET_PAIR_DMA_POS == 1     // behind the dot products: the LDS queue is empty here
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (piece + i < 8 - KGDET_PAIR_M_PIECES) dma_piece(piece + i);
        __builtin_amdgcn_sched_barrier(0);
#endif
        // lower lanes (channel quads 0, 2) + upper lanes (quads 1, 3): one v_permlane32_swap; lower lanes keep d/dy, upper d/dx
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(gy), __float_as_uint(gx), false, false);
        float s = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (!first) s += R.prev;
        if (store_ok)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s), acc_rs, (int)acc_voff, (int)((unsigned)t * acc_tstride), 0);
      };
#ifndef KGDET_PAIR_ABL_NOTAIL
      tail(RA, cg, tA, 0);
      if (hasB) tail(RB, cg + 8, tA + 1, 4);
      else {
#pragma unroll
        for (int i = 4; i < 8 - KGDET_PAIR_M_PIECES; ++i) dma_piece(i);
      }
#else
#pragma unroll
      for (int i = 0; i < 8 - KGDET_PAIR_M_PIECES; ++i) dma_piece(i);
      if (cg[0] == 1234.5f) tail(RA, cg, tA, 0);
#endif
    }
    KGDET_PT_ADD(3);         // (the matrix half arrives here with a fresh time stamp: nothing is added for it)
    // the tail half's DMA pieces have landed and its stores are out before the barrier publishes them (the matrix half's record
    // loads were requested a whole matrix phase ago)
    dcn_wait_vm0();
    KGDET_PT_ADD(4);
    __syncthreads();
    KGDET_PT_ADD(5);
    // plane switch: pair q - 1 was the last of its segment and its tail is done; the next tail needs the next chunk's plane
    if (q >= 1 && pj == np - 1 && pseg + 1 < nseg) {
#ifndef KGDET_PAIR_ABL_NOPLANE
      copy_plane(pseg + 1);
#endif
      __syncthreads();
      KGDET_PT_ADD(6);
    }
    pseg = seg; pj = j;
    seg = nseg_; j = nj;
  }
#ifdef KGDET_PAIR_TRACE
  if (lane == 0 && g < 256) {
    tr[7] = __builtin_amdgcn_s_memtime() - tr_start;
    tr[8] = (unsigned long long)Q;
    tr[9] = (unsigned long long)nseg;
    for (int c = 0; c < 10; ++c) g_pair_trace[((int)g * 8 + wave) * 10 + c] = tr[c];
  }
#endif
}

}  // namespace kgdet
