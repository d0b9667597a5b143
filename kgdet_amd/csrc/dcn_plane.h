// Shared body of the "plane" kernels (dcn_forward_plane.hip, dcn_backward_plane.hip): LDS-resident feature
// plane, producer / consumer waves, bf16 hi/lo split MFMA, static (problem, part, tile) ranges or stream-K over
// (tile, stage) units, one workgroup barrier per GROUP of four stages.  See dcn_forward_plane.hip for the design notes.
#pragma once
#include <type_traits>

#include "common.h"
#include "dcn_kernels.h"

namespace kgdet {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifdef KGDET_PLANE_TRACE
// Experiment build only (make VARIANT=trace EXTRA=-DKGDET_PLANE_TRACE, tools/plane_trace.py): per-workgroup cycle
// sums by phase, written by lane 0 of the first consumer / producer wave.
//   [block][role][cat]: 0 prologue barrier wait, 1 prologue loads + plane copy, 2 prologue second barrier wait,
//   3 prologue sample + third barrier, 4 group work (top of group .. before barrier), 5 group barrier wait,
//   6 epilogue (slab / output store), 7 whole kernel, 8 stages, 9 segments
static __device__ unsigned long long g_plane_trace[256 * 16 * 10];   // [block][wave][cat]; one copy per translation unit
#ifdef KGDET_PLANE_TRACE_REALTIME   // the constant 100 MHz counter instead of the core clock: cycles / ticks = the clock the kernel ran at
#define KGDET_TR_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define KGDET_TR_NOW() __builtin_amdgcn_s_memtime()
#endif
#define KGDET_TR_ADD(cat, t_from) do { const unsigned long long n__ = KGDET_TR_NOW(); tr[cat] += n__ - (t_from); (t_from) = n__; } while (0)
#else
#define KGDET_TR_ADD(cat, t_from) do { } while (0)
#endif

namespace {

constexpr int kAPart = 2 * kTileM * 8 * 2;  // bytes of one part of an A stage
constexpr int kBPart = 2 * kTileN * 8 * 2;
constexpr int kProducers = 256;                          // 4 producer waves
constexpr int kPlaneThreads = kThreads + kProducers;     // 8 consumer + 4 producer waves (grad_weight kernel, tap-pair kernel)
// plane_role (forward / grad_input): kRolePairs producer wave pairs; a pair (128 threads = the tile's 128 pixels) samples
// kGroupTaps / kRolePairs stages of every group
#ifndef KGDET_PLANE_PAIRS
#define KGDET_PLANE_PAIRS 4
#endif
constexpr int kRolePairs = KGDET_PLANE_PAIRS;
constexpr int kRoleProducers = kRolePairs * 128;
constexpr int kRoleThreads = kThreads + kRoleProducers;
static_assert(kRolePairs == 2 || kRolePairs == 4, "two or four producer wave pairs");
constexpr int kPlaneRounds = 6;                          // (pixel, quad) items a thread has in flight while copying a plane
constexpr int kGroupTaps = 4;                            // stages (taps) between two workgroup barriers
// KGDET_PLANE_STREAM (round 5, forward with split operands): a segment of 4 q + 1 stages (9 / 25 / 49 taps) runs as q FULL groups
// and ONE extra stage that has its own B slot behind the two group buffers: it is sampled together with the last full group
// and multiplied behind it, while the producers bring in the next plane and sample the next segment's first FULL group.  The
// short group of one stage used to be the segment's first: the consumers multiplied it while the producers needed a whole
// group time for the next four stages -- a bubble per segment (profiles/r05_dcn_fwd_plane_group_b2.md).  Built, correct (the 61
// forward tests), and MEASURED SLOWER: 163.2 / 162.5 against 161.0 / 160.1 us per launch sequence (1), 168.4 / 167.5 against
// 164.1 / 163.7 with the hand-over barrier one stage earlier (2) -- although the trace build counts 2.7 % FEWER cycles for the 5x5
// workgroups.  Default 0 (off); -DKGDET_PLANE_STREAM=1|2 builds the variants.
#ifndef KGDET_PLANE_STREAM
#define KGDET_PLANE_STREAM 0
#endif
constexpr bool kAFromL2 = true;   // (grad_weight kernel) consumers take their A fragments from the operand image in L2

// Ablation switches for experiment builds (make VARIANT=... EXTRA=-DKGDET_ABL_...; results are WRONG by design):
//   KGDET_ABL_NOSAMPLE  producers skip the corner reads and the interpolation (B stage = garbage): consumer-bound time
//   KGDET_ABL_NOMFMA    consumers skip the B-fragment reads and the MFMAs: producer-bound time
//   KGDET_ABL_NOALOAD   consumers do not load their weight fragments (one load in the prologue): L2 / fabric share
//   KGDET_ABL_NOGATHER  producers interpolate register values instead of LDS corner reads: the gathers' share
//   KGDET_ABL_NOBSTORE  producers compute the split B stage but do not store it: the LDS stores' share
// Pacing of the consumers' MFMAs (tools/microbench/mfma_valu.hip): a wave whose next MFMA queues behind the busy matrix
// pipe keeps the SIMD's vector issue port, and the producer wave of that SIMD gets ~one instruction per MFMA; an s_nop
// after each MFMA leaves the port to the producer.
#ifndef KGDET_PLANE_PACE
#define KGDET_PLANE_PACE 0
#endif
#if KGDET_PLANE_PACE == 0
#define KGDET_MFMA_PACE() do { } while (0)
#elif KGDET_PLANE_PACE == 100
#define KGDET_MFMA_PACE() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define KGDET_MFMA_PACE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(KGDET_PLANE_PACE)); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
// consumer wave layout: 0 = 8 x 1 waves of 32 x 128 (every wave reads the whole B stage, distinct weight rows),
// 1 = 4 x 2 waves of 64 x 64 (half the B reads per wave, every weight fragment fetched by two waves)
#ifndef KGDET_PLANE_WAVES42
#define KGDET_PLANE_WAVES42 0
#endif
#ifndef KGDET_PLANE_SCALAR_FMA
#define KGDET_PLANE_SCALAR_FMA 1
#endif
#ifndef KGDET_PLANE_PRODUCER_PRIO
#define KGDET_PLANE_PRODUCER_PRIO 0   // (with eight producer waves they have slack: prio 2 costs the consumers 5 %)
#endif
// Segment chaining (round 5): under the LAST group of a segment the producers copy the next segment's plane (all of it, one
// batch of loads per wave), a MID-GROUP barrier publishes it after the consumers' second stage, and the producers sample the
// next segment's FIRST group into the free B buffer while the consumers multiply stages three and four -- the next segment
// starts with its first group sampled and its first weight fragments in flight.  0: round 3's hand-over (plane under the last
// group, first group sampled with the consumers idle: ~5.7 k cycles per segment, phase trace of round 5's first run).
#ifndef KGDET_PLANE_CHAIN
#define KGDET_PLANE_CHAIN 5           // bit MODE: 1 forward, 2 grad_input, 4 forward on large maps (gather)
#endif
#ifndef KGDET_PLANE_ALTERNATE
#define KGDET_PLANE_ALTERNATE 0       // > 0: the consumer waves of a SIMD alternate between this priority and 0, stage by stage
#endif
#ifndef KGDET_PLANE_CHAIN_REGS
#define KGDET_PLANE_CHAIN_REGS 0      // 1: chain also where the plane copy goes through registers (measured slower: the producers'
#endif                                // 36 loads per wave beside the MFMA waves take ~5 k cycles)
#ifndef KGDET_PLANE_CHAIN_ROUNDS
#define KGDET_PLANE_CHAIN_ROUNDS 9    // plane units a producer wave has in flight in the chained copy (68 units / 8 waves)
#endif

// MODE of the plane kernels
//   0  forward:     B stage = bilinear samples of x; 4 corners per (pixel, tap) from a DcnTapRec
//   1  grad_input:  B stage = transposed sampling of grad_output: input cell q collects, for tap t, every
//                   (output pixel, weight) pair whose corner is q -- up to 8 from its DcnInvRec, or, for a cell with
//                   more, the pre-aggregated sum (dcn_inv_overflow_sums)
//   2  forward on a map beyond the LDS plane's capacity: like MODE 0, but the corners are 16-byte buffer loads from a
//                   pixel-major copy of x ([N][H*W][C]: a pixel's 16 channels of a chunk are 64 contiguous bytes), the
//                   tap records hold the rows' byte offsets; no plane, no plane copy, any map size
template <int MODE>
struct PlaneModeTraits {
  static constexpr int kGroups = MODE == 1 ? 2 : 1;  // groups of 4 (pixel, weight) entries in the record
};

// what a producer thread has in flight for one stage
template <int MODE>
struct PlaneStageRegs {
  static constexpr int NG = PlaneModeTraits<MODE>::kGroups;
  uint4 off[NG];      // LDS byte offsets of the sampled pixels (quad 0)
  f32x4 w[NG];        // and their weights
  int2 ovf;           // MODE 1: .x = the stage's tap
};

}  // namespace

// Loop structure: a workgroup walks its slice range by range (static schedule: exactly one range); inside a range the
// stages that share a feature plane (one channel chunk, consecutive taps) form a SEGMENT.  Only the first segment of a
// range starts with a plane copy by all sixteen waves: the next segment's plane and first tap records are loaded under
// the LAST group of the segment before (see the segment loop).  Loads are issued unconditionally from clamped addresses,
// so hipcc's counted s_waitcnt vmcnt(N) stay exact.
//
// Stages are handed from the producers to the consumers in GROUPS of kGroupTaps = 4.  The B buffer in LDS holds two
// groups: while the consumers multiply the four stages of group g (their weight fragments arrive from L2 two
// stages ahead, in a two-deep register ring), the producers sample the four stages of group g + 1 (each of the four wave
// pairs one of them); the tap records of a group are loaded one whole group before they are used.  ONE workgroup barrier
// per group.
// Why (phase trace of the one-barrier-per-stage version, tools/plane_trace.py: cycles per workgroup, 187 stages):
// both roles did ~1200 cycles of work per stage and each waited ~300 more at the barrier -- for the slowest of the
// waves, a different one every stage (random-gather bank conflicts, issue arbitration); a group averages
// that skew over four stages.  Inside a group a producer keeps the corner reads of both half-stages of its stage in flight
// (MODE 0 / 2), so the gather latency overlaps VALU work instead of heading a dependent chain.
// The SIMD has a single issue port for MFMA and other VALU instructions (tools/microbench/mfma_valu.hip,
// profiles/r02_dcn_fwd_plane_group_b2.md): beside the two MFMA waves of its SIMD a producer wave gets few slots and each of its
// instructions costs the MFMA waves cycles -- hence scalar role ids, buffer loads with scalar offsets, record sets that
// alternate instead of being copied, scalar (not packed) fp32 FMAs.  With eight producer waves at priority 0 the producers
// have slack and the consumers are the critical path (profiles/r03_dcn_fwd_plane_group_b2.md).
// Stage coordinates are carried incrementally; there is no integer division in the loop.
// The two roles are two instantiations of this function (same loop structure, same barriers), so the
// accumulators exist only in the consumers' register allocation.
template <int PARTS, bool PRODUCER, int MODE>
__device__ __forceinline__ void plane_role(const DcnFwdGroup &grp, float *__restrict__ slabs, unsigned char *smem) {
  unsigned char *plane = smem;                                       // [4 quads][kPlaneMaxHW pixels][4 ch] fp32, LDS address 0:
                                                                     // tap-record offsets + an immediate address a corner read
  unsigned char *Bs = smem + (MODE == 2 ? 0 : 4 * kPlaneQuadStride);   // [2 groups][kGroupTaps][PARTS][kBPart]
  // a corner read: the record's offset IS the LDS address (through `plane + offset` hipcc adds the symbol's 0 per read)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const f32x4 __attribute__((address_space(3))) *LdsQuadPtr;
  auto lds_quad = [](unsigned addr) { return *(LdsQuadPtr)(addr); };
#else
  auto lds_quad = [](unsigned) { return f32x4{0.f, 0.f, 0.f, 0.f}; };   // (host pass: pointers are 64-bit there)
#endif
  if ((unsigned)(unsigned long long)(const unsigned char __attribute__((address_space(3))) *)smem != 0u) __builtin_trap();   // (no static LDS in these kernels)

  const int wtid = threadIdx.x;                               // 0 .. 767 (plane copy)
  const int tid = PRODUCER ? wtid - kThreads : wtid;          // position inside the role
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave inside the role / among all twelve, in SGPRs
  const int wave_all = __builtin_amdgcn_readfirstlane(wtid >> 6);  // (plane copy: wave-uniform units)
  // consumers: wave w owns rows [32 w, 32 w + 32) x all 128 columns of the tile (wave layout 1): every wave loads
  // DISTINCT weight rows, 16 KB per stage and workgroup instead of 32 KB with 4 x 2 waves of 64 x 64 -- the vector
  // memory path of a CU takes 64 B per clock, and 32 KB of fragments per stage were 512 of its ~1000 cycles
  const int n_local = tid & (kTileN - 1);                     // producers: pixel column sampled
  // (wave-uniform role coordinates in SGPRs: scalar branches and no exec masking -- every VALU instruction of a producer
  // costs its SIMD's MFMA waves issue slots, tools/microbench/mfma_valu.hip)
  const int pair = __builtin_amdgcn_readfirstlane((tid >> 7) & (kRolePairs - 1));   // wave pair: samples stage pair (and, with two pairs, pair + 2) of a group
  const long long G = gridDim.x, g = blockIdx.x;
  const long long slice = sk_slice_of_block((int)g, (int)G);
  long long my_begin, my_end;

  if constexpr (PRODUCER) __builtin_amdgcn_s_setprio(KGDET_PLANE_PRODUCER_PRIO);
#ifdef KGDET_PLANE_CONSUMER_PRIO
  else __builtin_amdgcn_s_setprio(KGDET_PLANE_CONSUMER_PRIO);
#endif
#ifdef KGDET_PLANE_TRACE
  unsigned long long tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_t = KGDET_TR_NOW();
  const unsigned long long tr_start = tr_t;
#endif
  int slot = 0;  // slabs written so far (one per range met)
  // stream-K: one slice of units; static schedule: one range per round (ranges slice, slice + G, ...)
  for (int round = 0; round < (grp.static_ranges ? grp.rounds : 1); ++round) {
  dcn_slice_bounds(grp, slice, G, my_begin, my_end, round);
  long long cur = my_begin;
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

    constexpr bool k42 = (KGDET_PLANE_WAVES42 != 0) && (PARTS > 0);   // (value-dependent: the other layout's calls are discarded, not checked)
    typedef std::conditional_t<k42, f32x16[2][2], f32x16[4]> Acc;
    Acc acc;  // consumers only (dead in the producers' instantiation)
    if constexpr (!PRODUCER) {
      if constexpr (k42) zero_acc(acc);
      else zero_acc_w8(acc);
    }
    const int wm = wave_s & 3, wn = wave_s >> 2;   // (4 x 2 layout)

    typedef PlaneStageRegs<MODE> Regs;
    constexpr int NG = Regs::NG;

    int s = s_begin;
    int c16 = s / K;
    int t0 = s - c16 * K;
    // records of (image, deformable group) for channel chunk c: [K][pixels][NG] groups of 32 B; seg_records = byte offset
    // of the chunk's first tap (scalar), rec_lane = this thread's pixel (the one vector offset of every record load)
    const dcn_rsrc_t rec_rs = dcn_make_rsrc(p.taps);
    const unsigned rec_lane = (unsigned)hw_c * (unsigned)(32 * NG);
    auto seg_records = [&](int c) {
      const int dgi = p.DG == 1 ? 0 : (p.c_base + min(c * kChunk, p.Cg - 1)) / p.cpdg;   // (cpdg % 16 == 0)
      return (unsigned)((tile_b * p.DG + dgi) * K) * (unsigned)(HoWo * 32 * NG);
    };
    // loads of stage min(j, lim) of the segment (first tap tf) whose records start at rb: the tap record
    auto issue = [&](unsigned rb, int tf, int lim, int j, Regs &R) {
      const unsigned t = (unsigned)(tf + min(j, lim));
      const unsigned so = rb + t * (unsigned)(HoWo * 32 * NG);
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        R.off[gq] = __builtin_bit_cast(uint4, dcn_buf_b128(rec_rs, rec_lane + 16 * gq, so));
        R.w[gq] = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 16 * (NG + gq), so));
      }
      if constexpr (MODE == 1) R.ovf = int2{(int)t, 0};   // the stage's tap (row of the pre-aggregated sums)
    };
    // Copy x[tile_b, c_base + 16*c .. +15, :, :] into the LDS quad planes (dcn_plane_copy, dcn_common.h): units
    // [unit_lo, unit_hi) of the plane, this wave being number `w` of `NW_` waves that share them.
    const float *xb_img = p.x + ((long long)tile_b * p.C_total + p.c_base) * HW;
    // MODE 2: the pixel-major image; byte offset of this tile's image and channel window (the chunk is added per segment)
    const dcn_rsrc_t xg_rs = dcn_make_rsrc(p.x);
    const unsigned xg_img = (unsigned)(((long long)tile_b * HW * p.C_total + p.c_base) * 4);
    constexpr bool kBf16Plane = PARTS == 1 && MODE == 0;   // one-product forward: the plane holds bf16 (dcn_common.h)
    // chained hand-over by LDS-DMA: the blocked copy of x (p.xblk, written by dcn_build_taps; forward with split operands only)
    const bool xb_dma = MODE == 0 && !kBf16Plane && p.xblk != nullptr;
    const unsigned long long xba = reinterpret_cast<unsigned long long>(p.xblk);
    const u32x4_t xb_rs = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)xba),
                           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(xba >> 32)), 0xffffffffu, 0x00020000u};
    const unsigned xblk_img = (unsigned)(tile_b * n_c16) * (unsigned)((kBf16Plane ? dcn_plane_units_bf16(HW) : dcn_plane_units(HW)) * 1024);
    auto load_plane = [&](int c, int w, auto NW_, auto ROUNDS_, int unit_lo, int unit_hi) {
      constexpr int NW = decltype(NW_)::value;
      constexpr int ROUNDS = decltype(ROUNDS_)::value;
      if constexpr (kBf16Plane)
        dcn_plane_copy_bf16<(ROUNDS + 1) / 2>(xb_img, HW, p.Cg, c * kChunk, plane, (unsigned)kPlaneQuadStride, unit_lo + w, NW,
                                              unit_hi, lane);
      else
        dcn_plane_copy<ROUNDS>(xb_img, HW, p.Cg, c * kChunk, plane, (unsigned)kPlaneQuadStride, unit_lo + w, NW, unit_hi, lane);
    };
    // The next segment's plane is copied under the LAST group of a segment (the plane is not read any more once that
    // group's stages are sampled): the producers, who have nothing to sample then, take the first plane_split units,
    // the consumers the rest after their MFMAs.  Split operands: the consumers are busy for four stages of 12 MFMAs,
    // the producers take two full batches of loads; bf16: a third each way (equal shares per wave).
    const int plane_items = kBf16Plane ? dcn_plane_units_bf16(HW) : dcn_plane_units(HW);
#ifndef KGDET_PLANE_SPLIT_UNITS
#define KGDET_PLANE_SPLIT_UNITS (8 * kPlaneRounds)
#endif
    const int plane_split = PARTS == 2 ? min(plane_items, KGDET_PLANE_SPLIT_UNITS) : plane_items * (kRoleProducers / 64) / (kRoleThreads / 64);
    typedef std::integral_constant<int, kRoleThreads / 64> AllWaves;
    typedef std::integral_constant<int, kRoleProducers / 64> ProducerWaves;
    typedef std::integral_constant<int, kThreads / 64> ConsumerWaves;
    typedef std::integral_constant<int, kPlaneRounds> FullRounds;
    typedef std::integral_constant<int, kPlaneRounds / 2> HalfRounds;   // the consumers' share is the small one
    Regs X0;               // producers, stream mode: the record of the segment's extra stage
    Regs E0, E1, O0, O1;   // producers: this wave pair's two records of the even-numbered (E) and of the odd-numbered (O)
                           // groups of the segment.  Group g + 2's are loaded at the TOP of iteration g, which samples group
                           // g + 1 from the other set: a whole group to land, and no register moves (the iterations are
                           // unrolled by two)
    bool primed = false;   // the segment's plane and first records were loaded under the last group of the one before
    bool first_ready = false;   // ... and its first group sampled, its first two weight fragment sets requested (chained)
    unsigned bsel = 0;     // B buffer numbering of this segment: logical buffer b is physical buffer b ^ bsel (a chained
                           // segment finds its first group in the buffer the previous segment's last group did not use)
    constexpr unsigned kGroupBytes = (unsigned)(kGroupTaps * PARTS * kBPart);
    while (s < s_end) {
      const int n = min(K - t0, s_end - s);  // stages of this segment: taps t0 .. t0+n-1 of chunk c16
      // Groups of the segment: a FIRST group of r = 1..4 stages (n - r is a multiple of 4), then full groups.  The
      // consumers have nothing to multiply while the first group is sampled, so it is the short one (r = 1 for the
      // 9 / 25 / 49 taps of 3x3 / 5x5 / 7x7), and the LAST group is a full one: the plane is not read any more once its
      // stages are sampled, and the producers copy the next segment's plane under its four stages of MFMAs.
      constexpr bool kStreamable = (KGDET_PLANE_STREAM != 0) && MODE == 0 && PARTS == 2 && kRolePairs == 4;
      // (stream: see KGDET_PLANE_STREAM above -- q = n_groups full groups + the extra stage n - 1)
      const bool stream = kStreamable && xb_dma && (n & 3) == 1 && n >= 9;
      const int r = stream ? 4 : ((n - 1) & 3) + 1;
      const int n_groups = stream ? (n - 1) / kGroupTaps : 1 + (n - r) / kGroupTaps;
      const bool has_next = s + n < s_end;
      const unsigned rec_base = seg_records(c16);
      unsigned xg_seg = xg_img + (unsigned)(c16 * kChunk * 4);   // (MODE 2) this segment's channel chunk

      // consumers: the wave's A (weight) fragments of stage j straight from the weight image (L2), 16 bytes per lane
      // and fragment, coalesced
      struct AFrag {
        bf16x8 a[PARTS][k42 ? 2 : 1];
      };
      // (buffer loads: wave-uniform stage offset + one 32-bit lane offset, no vector address arithmetic between MFMAs)
      // (grad_input of a channel run that starts inside a 256-row tile of wqt: rows row0 .., clamped -- rows past the run
      // are computed from another run's weights and never stored)
      const dcn_rsrc_t wq_rs = dcn_make_rsrc(p.wq);
      const unsigned wq_seg = (unsigned)(((mt + p.mt_base) * n_c16 + c16) * K) * (unsigned)(2 * kAPart);
      unsigned a_lane[k42 ? 2 : 1];
#pragma unroll
      for (int mi = 0; mi < (k42 ? 2 : 1); ++mi) {
        const int row = k42 ? wm * 64 + mi * 32 + (lane & 31) : wave * 32 + (lane & 31);
        a_lane[mi] = (unsigned)((lane >> 5) * (kTileM * 16) + min(p.row0 + row, kTileM - 1) * 16);
      }
      // fragments of the stage at byte offset `so` of the weight image (scalar)
      auto a_issue_at = [&](unsigned so, AFrag &F) {
#pragma unroll
        for (int part = 0; part < PARTS; ++part)
#pragma unroll
          for (int mi = 0; mi < (k42 ? 2 : 1); ++mi)
            F.a[part][mi] = __builtin_bit_cast(bf16x8, dcn_buf_b128(wq_rs, a_lane[mi], so + part * kAPart));
      };
      auto a_issue = [&](int j, AFrag &F) {
#ifdef KGDET_ABL_NOALOAD
        if (j > 1) return;
#endif
        a_issue_at(wq_seg + (unsigned)(t0 + min(j, n - 1)) * (2 * kAPart), F);
      };
      // chained hand-over: what the NEXT segment (chunk c16 + 1, taps 0 .. n2 - 1, first group of r2 stages) needs
      const int n2 = has_next ? min(K, s_end - (s + n)) : 1;
      const bool stream2 = kStreamable && xb_dma && has_next && (n2 & 3) == 1 && n2 >= 9;
      const int r2 = stream2 ? 4 : ((n2 - 1) & 3) + 1;
      const unsigned wq_seg2 = wq_seg + (unsigned)K * (2 * kAPart);   // (chunk c16 + 1 of the same row tile)
      const bool chain = ((KGDET_PLANE_CHAIN >> MODE) & 1) && has_next && n_groups >= 2 && (MODE == 2 || xb_dma || KGDET_PLANE_CHAIN_REGS);   // the last group is a full one: chain under it
      // ---- B stage, producers.  A group has four stages; producer wave pair w (2 waves = 128 pixels) samples stages
      // w and w + 2 of it -- a thread does ALL 16 channels of its pixel for a stage, as two half-stages of 8 channels
      // (one ds_read_b128 per corner and channel quad).  So a producer wave has TWO stage times for one stage of work
      // and four independent half-stages to interleave: it is latency-bound (a lone wave per SIMD running dependent
      // packed-fp32 chains: ~1100 cycles per stage even with the MFMA pipes idle), and with one stage per stage time it
      // was the critical path of every stage (phase trace: 233 k of 261 k cycles).
      // Corner offsets in the record are for quad 0; quad c of the same pixel is kPlaneQuadStride * c further: an
      // immediate offset of the ds_read_b128.
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef f32x4 Corners[2][4];
      auto corner_reads = [&](const Regs &R, int gq, int half, Corners &v) {
        // (MODE 1: the upper bits of a record's last offset hold the cell's overflow range)
        const unsigned o[4] = {R.off[gq].x, R.off[gq].y, R.off[gq].z,
                               (MODE == 1 && gq == NG - 1) ? (R.off[gq].w & 0x1ffffu) : R.off[gq].w};   // (upper bits: flag + slot)
        if constexpr (kBf16Plane) {   // eight channels of a corner in ONE read; bf16 -> fp32 is a shift / a mask per value
#if KGDET_PLANE_F16
          // (fp16 plane: the four dwords stay as they are -- corner_fma converts inside its FMAs)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[0][e] = lds_quad(o[e] + (unsigned)(half * kPlaneQuadStride));
          return;
#endif
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const u32x4_t raw = __builtin_bit_cast(u32x4_t, lds_quad(o[e] + (unsigned)(half * kPlaneQuadStride)));
#pragma unroll
            for (int c = 0; c < 2; ++c)
              v[c][e] = f32x4{__uint_as_float(raw[2 * c] << 16), __uint_as_float(raw[2 * c] & 0xffff0000u),
                              __uint_as_float(raw[2 * c + 1] << 16), __uint_as_float(raw[2 * c + 1] & 0xffff0000u)};
          }
          return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#ifdef KGDET_ABL_NOGATHER
            v[c][e] = f32x4{__uint_as_float(o[e]), R.w[gq][c], R.w[gq][e], (float)half};
#else
            if constexpr (MODE == 2)
              v[c][e] = __builtin_bit_cast(f32x4, dcn_buf_b128(xg_rs, o[e], xg_seg + (unsigned)((half * 2 + c) * 16)));
            else
              v[c][e] = lds_quad(o[e] + (unsigned)((half * 2 + c) * kPlaneQuadStride));
#endif
      };
      auto corner_fma = [&](const Regs &R, int gq, const Corners &v, f32x2 (&sv)[2][2], bool first) {
#if KGDET_PLANE_F16
        if constexpr (kBf16Plane) {
          // dword q = c * 2 + h2 of a corner holds channels 2 q, 2 q + 1 as fp16: fma(w, (float)half, acc) = v_fma_mix_f32
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                // (through a scalar: __builtin_bit_cast applied to the vector-element expression itself reads element 0
                //  whatever the index -- clang 19 / ROCm 7, reproduced on the host)
                const float dw = v[0][e][c * 2 + h2];
                const dcn_f16x2 hv = __builtin_bit_cast(dcn_f16x2, dw);
                const float w1 = R.w[gq][e];
                if (first && e == 0) { sv[c][h2][0] = w1 * (float)hv[0]; sv[c][h2][1] = w1 * (float)hv[1]; }
                else { sv[c][h2][0] = __builtin_fmaf(w1, (float)hv[0], sv[c][h2][0]); sv[c][h2][1] = __builtin_fmaf(w1, (float)hv[1], sv[c][h2][1]); }
              }
          return;
        }
#endif
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#if KGDET_PLANE_SCALAR_FMA
              // two v_fma_f32 instead of one v_pk_fma_f32: beside MFMA waves a packed fp32 instruction costs ~20 cycles
              // more than the pair of scalar ones (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
              const float w1 = R.w[gq][e];
              const float a0 = v[c][e][2 * h2], a1 = v[c][e][2 * h2 + 1];
              if (first && e == 0) { sv[c][h2][0] = w1 * a0; sv[c][h2][1] = w1 * a1; }
              else { sv[c][h2][0] = __builtin_fmaf(w1, a0, sv[c][h2][0]); sv[c][h2][1] = __builtin_fmaf(w1, a1, sv[c][h2][1]); }
#else
              const f32x2 ve = {v[c][e][2 * h2], v[c][e][2 * h2 + 1]};
              const f32x2 we = {R.w[gq][e], R.w[gq][e]};
              sv[c][h2] = (first && e == 0) ? we * ve : __builtin_elementwise_fma(we, ve, sv[c][h2]);
#endif
            }
      };
      auto split_store = [&](int buf, int gi, int half, const f32x2 (&sv)[2][2]) {
        // hi = bf16(v) for a PAIR of values with one v_cvt_pk_bf16_f32; the two hi values back as floats are a shift and
        // a mask of that dword (20 instructions per 8 values instead of the 28 hipcc emits for scalar conversions)
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 hi_u, lo_u;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = c * 2 + h2;
            const bf16x2 hp = __builtin_convertvector(sv[c][h2], bf16x2);
            const unsigned hu = __builtin_bit_cast(unsigned, hp);
            hi_u[q] = hu;
            if constexpr (PARTS == 2) {
              const f32x2 hf = {__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
#if KGDET_PLANE_SCALAR_FMA
              const f32x2 df = {sv[c][h2][0] - hf[0], sv[c][h2][1] - hf[1]};
              const bf16x2 lp = __builtin_convertvector(df, bf16x2);
#else
              const bf16x2 lp = __builtin_convertvector(sv[c][h2] - hf, bf16x2);
#endif
              lo_u[q] = __builtin_bit_cast(unsigned, lp);
            }
          }
        const bf16x8 hi = __builtin_bit_cast(bf16x8, hi_u), lo = __builtin_bit_cast(bf16x8, lo_u);
        // (buf 2: the extra stage's slot behind the two group buffers)
        unsigned char *dst = Bs + (buf == 2 ? 2u * kGroupBytes : ((unsigned)buf ^ bsel) * kGroupBytes + gi * PARTS * kBPart) +
                             half * (kTileN * 16) + n_local * 16;
#ifdef KGDET_ABL_NOBSTORE
        if (sv[0][0][0] != 1234.56789f) return;
#endif
        *reinterpret_cast<bf16x8 *>(dst) = hi;
        if constexpr (PARTS == 2) *reinterpret_cast<bf16x8 *>(dst + kBPart) = lo;
      };
      // MODE 1 (and the unpipelined path): one half-stage in sequence
      auto sample_half = [&](int buf, int gi, int half, const Regs &R) {
#ifdef KGDET_ABL_NOSAMPLE
        {
          f32x2 z[2][2] = {{{R.w[0][0], R.w[0][1]}, {R.w[0][2], R.w[0][3]}}, {{R.w[0][0], R.w[0][1]}, {R.w[0][2], R.w[0][3]}}};
          split_store(buf, gi, half, z);
          return;
        }
#endif
        f32x2 sv[2][2];
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
          Corners v;
          corner_reads(R, gq, half, v);
          corner_fma(R, gq, v, sv, gq == 0);
        }
        if constexpr (MODE == 1) {
          // a cell with more than 8 contributing (pixel, corner) pairs for this tap: its record holds zero weights and a
          // slot number; the sum over ALL its contributions was formed for every output channel by dcn_inv_overflow_sums
          // (dcn_backward_plane.hip) -- 32 bytes of it are this half-stage's 8 channels.  Bounded work per cell whatever
          // the offsets look like (rounds 2-3: a list walk per cell and chunk -- hundreds of entries on a trained head).
          const unsigned pk = R.off[NG - 1].w;
          if (pk & kInvFlag) {
            const float *gv = p.inv_gov + ((size_t)(tile_b * K + R.ovf.x) * p.gov_slots + ((pk >> 17) & 0x3fffu)) * p.gov_ld +
                              (p.gov_c0 + c16 * kChunk + half * 8);
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gv), g1 = *reinterpret_cast<const f32x4 *>(gv + 4);
            sv[0][0] += f32x2{g0[0], g0[1]};
            sv[0][1] += f32x2{g0[2], g0[3]};
            sv[1][0] += f32x2{g1[0], g1[1]};
            sv[1][1] += f32x2{g1[2], g1[3]};
          }
        }
        split_store(buf, gi, half, sv);
      };
      // this wave pair's two stages of a group (stage pair of the group from record RA, pair + 2 from RB) into group
      // buffer `buf`.  MODE 0: four half-stages, the corner reads of one issued before the
      // arithmetic of the one before (two corner register sets).
      auto sample_group = [&](int buf, const Regs &RA, const Regs &RB, bool a_live, bool b_live, auto after_reads) {
#ifdef KGDET_ABL_NOSAMPLE
        constexpr bool pipelined = false;
#else
        constexpr bool pipelined = MODE == 0 || MODE == 2;
#endif
#ifdef KGDET_PLANE_TRACE_PRODUCER   // unpipelined, timed: slot 0 = issue of 8 corner reads -> data there, slot 6 = the rest
        if constexpr (pipelined) {
          auto probe = [&](int gi, int half, const Regs &R, bool live) {
            Corners V;
            f32x2 sv[2][2];
            const unsigned long long ta = KGDET_TR_NOW();
            corner_reads(R, 0, half, V);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const unsigned long long tb = KGDET_TR_NOW();
            corner_fma(R, 0, V, sv, true);
            if (live) split_store(buf, gi, half, sv);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const unsigned long long tc = KGDET_TR_NOW();
            tr[0] += tb - ta;
            tr[6] += tc - tb;
          };
          probe(pair, 0, RA, a_live); probe(pair, 1, RA, a_live);
          if constexpr (kRolePairs == 2) { probe(pair + 2, 0, RB, b_live); probe(pair + 2, 1, RB, b_live); }
          return;
        }
#endif
        if constexpr (kRolePairs == 4) {   // one stage per pair: both half-stages' corner reads in flight
          if constexpr (pipelined) {
            Corners V0, V1;
            f32x2 sv[2][2];
            corner_reads(RA, 0, 0, V0);
            corner_reads(RA, 0, 1, V1);
            after_reads();   // (MODE 2: the next records' loads go out BEHIND the corner loads -- vmcnt retires in order)
            corner_fma(RA, 0, V0, sv, true);
            if (a_live) split_store(buf, pair, 0, sv);
            corner_fma(RA, 0, V1, sv, true);
            if (a_live) split_store(buf, pair, 1, sv);
          } else {
            if (a_live) { sample_half(buf, pair, 0, RA); sample_half(buf, pair, 1, RA); }
          }
        } else if constexpr (pipelined) {
          // three corner register sets: the reads of three half-stages are in flight before the first arithmetic (a batch
          // of 8 random ds_read_b128 comes back after ~840 cycles while the consumers read their B fragments, ~420
          // alone -- in-kernel probe, tools/microbench/lds_gather.hip -- against ~190 cycles of arithmetic per half-stage)
          Corners V0, V1, V2;
          f32x2 sv[2][2];
          corner_reads(RA, 0, 0, V0);
          corner_reads(RA, 0, 1, V1);   // (clamped records past the end of the segment: harmless reads)
          corner_reads(RB, 0, 0, V2);
          corner_fma(RA, 0, V0, sv, true);
          if (a_live) split_store(buf, pair, 0, sv);
          corner_reads(RB, 0, 1, V0);
          corner_fma(RA, 0, V1, sv, true);
          if (a_live) split_store(buf, pair, 1, sv);
          corner_fma(RB, 0, V2, sv, true);
          if (b_live) split_store(buf, pair + 2, 0, sv);
          corner_fma(RB, 0, V0, sv, true);
          if (b_live) split_store(buf, pair + 2, 1, sv);
        } else {
          if (a_live) { sample_half(buf, pair, 0, RA); sample_half(buf, pair, 1, RA); }
          if (b_live) { sample_half(buf, pair + 2, 0, RB); sample_half(buf, pair + 2, 1, RB); }
        }
      };
      // ---- consumers: stage `gi` of group buffer `buf` times the fragment set F
      auto multiply = [&](int buf, int gi, const AFrag &F) {
#ifdef KGDET_ABL_NOMFMA
        return;
#endif
        if constexpr (!PRODUCER) {
          const unsigned char *B = Bs + (buf == 2 ? 2u * kGroupBytes : ((unsigned)buf ^ bsel) * kGroupBytes + gi * PARTS * kBPart) +
                                   (lane >> 5) * (kTileN * 16) +
                                   (lane & 31) * 16;
          if constexpr (k42) {
            bf16x8 b[PARTS][2];
#pragma unroll
            for (int part = 0; part < PARTS; ++part)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                b[part][ni] = *reinterpret_cast<const bf16x8 *>(B + part * kBPart + (wn * 2 + ni) * 32 * 16);
            if constexpr (PARTS == 2) {  // small terms first; four independent accumulators per pass
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[1][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0][mi], b[1][ni], acc[mi][ni], 0, 0, 0);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
          } else {
            bf16x8 b[PARTS][4];
#pragma unroll
            for (int part = 0; part < PARTS; ++part)
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) b[part][ni] = *reinterpret_cast<const bf16x8 *>(B + part * kBPart + ni * 32 * 16);
            if constexpr (PARTS == 2) {  // small terms first; four independent accumulators per pass
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) {
                acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[1][0], b[0][ni], acc[ni], 0, 0, 0);
                KGDET_MFMA_PACE();
              }
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) {
                acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0][0], b[1][ni], acc[ni], 0, 0, 0);
                KGDET_MFMA_PACE();
              }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0][0], b[0][ni], acc[ni], 0, 0, 0);
              KGDET_MFMA_PACE();
            }
          }
        }
      };

      // The two consumer waves of a SIMD (w and w + 4) take turns at s_setprio 1, stage by stage: at equal priority the
      // arbiter prefers the older wave, which then waits ~20 % of every group at the barrier for its younger partner
      // (per-wave phase trace, round 5: waves 0-3 work 1716 / wait 615, waves 4-7 work 2145 / wait 187 per 200 stages)
      auto lead = [&](int k) {
#if KGDET_PLANE_ALTERNATE
        if constexpr (!PRODUCER) {
          if (((wave_s >> 2) ^ k) & 1) __builtin_amdgcn_s_setprio(KGDET_PLANE_ALTERNATE);
          else __builtin_amdgcn_s_setprio(0);
        }
#endif
      };
      AFrag F0, F1;      // consumers: weight fragments two stages ahead, alternating (a third set in the loop spills): stage
                         // j of a segment sits in F0 when j - r is even, so that full groups find F0, F1, F0, F1
#ifdef KGDET_PLANE_TRACE
      tr_t = KGDET_TR_NOW();
      tr[9] += 1; tr[8] += n;
#endif
      // (every group ends with a workgroup barrier: the previous segment's -- or range's -- readers of B and of the plane
      // are done)
      if (!primed) {   // first segment of the range: records, then the plane copy by all twelve waves
        if constexpr (PRODUCER) {
          issue(rec_base, t0, r - 1, pair, E0);
          issue(rec_base, t0, n - 1, r + pair, O0);
          if constexpr (kRolePairs == 2) { issue(rec_base, t0, r - 1, pair + 2, E1); issue(rec_base, t0, n - 1, r + pair + 2, O1); }
          if constexpr (kStreamable)
            if (stream && n_groups == 2) issue(rec_base, t0, n - 1, n - 1, X0);
        }
        if constexpr (MODE != 2) load_plane(c16, wave_all, AllWaves{}, FullRounds{}, 0, plane_items);
        KGDET_TR_ADD(1, tr_t);
        __syncthreads();
        KGDET_TR_ADD(2, tr_t);
      }
      // first group (buffer 0).  One stage: each wave pair samples one 8-channel half of it (both hold its record)
      if (!first_ready) {
        if constexpr (PRODUCER) {
          if (r == 1) { if (pair < 2) sample_half(0, 0, pair, E0); }
          else sample_group(0, E0, E1, pair < r, pair + 2 < r, [] {});
        } else {   // full groups find their stages in F0, F1, F0, F1: an odd first group starts with F1
          if (r & 1) { a_issue(0, F1); a_issue(1, F0); }
          else { a_issue(0, F0); a_issue(1, F1); }
        }
        __syncthreads();
      }
      KGDET_TR_ADD(3, tr_t);
      // producers, while the consumers multiply group gi: sample group gi + 1 into the other buffer from (Sa, Sb) and
      // load the records of group gi + 2 into (Ia, Ib); under the last group, the next segment's records and plane instead
      auto produce = [&](int gi, int buf_next, const Regs &Sa, const Regs &Sb, Regs &Ia, Regs &Ib) {
        if (gi + 1 < n_groups) {
          const int jn = r + (gi + 1) * kGroupTaps;   // first stage of group gi + 2
          if constexpr (MODE == 2 && kRolePairs == 4) {
            sample_group(buf_next, Sa, Sb, true, true, [&] { issue(rec_base, t0, n - 1, jn + pair, Ia); });
          } else {
            issue(rec_base, t0, n - 1, jn + pair, Ia);
            if constexpr (kRolePairs == 2) issue(rec_base, t0, n - 1, jn + pair + 2, Ib);
            if constexpr (kStreamable)
              if (stream && gi + 3 == n_groups) issue(rec_base, t0, n - 1, n - 1, X0);   // (group gi + 2 is the last full one)
            sample_group(buf_next, Sa, Sb, true, true, [] {});
            if constexpr (kStreamable)
              if (stream && gi + 2 == n_groups && pair < 2) sample_half(2, 0, pair, X0);  // behind the last full group: the extra stage
          }
        } else if (has_next) {
          const unsigned rb2 = seg_records(c16 + 1);
          issue(rb2, 0, r2 - 1, pair, E0);
          issue(rb2, 0, n2 - 1, r2 + pair, O0);
          if constexpr (kRolePairs == 2) { issue(rb2, 0, r2 - 1, pair + 2, E1); issue(rb2, 0, n2 - 1, r2 + pair + 2, O1); }
          if constexpr (kStreamable)
            if (stream2 && (n2 - 1) / kGroupTaps == 2) issue(rb2, 0, n2 - 1, n2 - 1, X0);
          if (chain) {
            // the whole plane by the producers, one batch of loads per wave; then, behind the mid-group barrier that
            // publishes it, the next segment's first group into the B buffer this segment's last group does not use
            // (logical buffer n_groups & 1 of this segment = logical buffer 0 of the next one)
            if constexpr (MODE != 2) {
#ifndef KGDET_ABL_NOCHAINPLANE   // (ablation: what a free plane copy would buy -- results are wrong)
              if (xb_dma) {   // LDS-DMA from the blocked copy of x: no registers, no ds_write, ~nine instructions per wave
                const unsigned so = __builtin_amdgcn_readfirstlane(xblk_img + (unsigned)(c16 + 1) * (unsigned)(plane_items * 1024));
                const int nblk = plane_items >> 2;
                for (int u = wave_s; u < plane_items; u += kRoleProducers / 64) {
                  const int quad = u / nblk, blk = u - quad * nblk;
                  dcn_dma_b128(xb_rs, (unsigned)(lane * 16), so + (unsigned)(u * 1024),
                               (unsigned)quad * (unsigned)kPlaneQuadStride + (unsigned)(blk * 1024));
                }
                dcn_wait_vm0();
              } else {
                load_plane(c16 + 1, wave_s, ProducerWaves{}, std::integral_constant<int, KGDET_PLANE_CHAIN_ROUNDS>{}, 0, plane_items);
              }
#endif
              KGDET_TR_ADD(4, tr_t);
              __syncthreads();
              KGDET_TR_ADD(0, tr_t);   // (trace: category 0 = wait at the mid-group barrier)
            } else {
              xg_seg += (unsigned)(kChunk * 4);   // (gathered corners: the next segment's channel chunk)
            }
            const int fb = n_groups & 1;
            if (r2 == 1) { if (pair < 2) sample_half(fb, 0, pair, E0); }
            else sample_group(fb, E0, E1, pair < r2, pair + 2 < r2, [] {});
          } else {
            if constexpr (MODE != 2) load_plane(c16 + 1, wave_s, ProducerWaves{}, FullRounds{}, 0, plane_split);
          }
        }
      };
      // group 0
      if constexpr (PRODUCER) {
        produce(0, 1, O0, O1, E0, E1);
      } else {
        const int o = r & 1;   // (one-sided conditionals only: MFMAs on both sides of a branch make hipcc copy accumulators)
        if (o) { multiply(0, 0, F1); a_issue(2, F1); }
        if (r - o >= 2) { multiply(0, o, F0); a_issue(o + 2, F0); multiply(0, o + 1, F1); a_issue(o + 3, F1); }
        if (r - o >= 4) { multiply(0, o + 2, F0); a_issue(o + 4, F0); multiply(0, o + 3, F1); a_issue(o + 5, F1); }
        if constexpr (MODE != 2)
          if (n_groups == 1 && has_next) load_plane(c16 + 1, wave_s, ConsumerWaves{}, HalfRounds{}, plane_split, plane_items);
      }
      KGDET_TR_ADD(4, tr_t);
      __syncthreads();
      KGDET_TR_ADD(5, tr_t);
      // full group gi >= 1 (stages jg .. jg + 3) in buffer BUF.  Consumers: fragment sets F0, F1, F0, F1, each re-loaded
      // with the stage two ahead.
      auto group = [&](int gi, auto BUF) {
        constexpr int buf = decltype(BUF)::value;
        if constexpr (PRODUCER) {   // (gi odd <=> buf 1: group gi + 1 is even)
          if constexpr (buf == 1) produce(gi, 0, E0, E1, O0, O1);
          else produce(gi, 1, O0, O1, E0, E1);
        } else {
          const int jg = r + (gi - 1) * kGroupTaps;
          // (one scheduling region per stage: across all four, hipcc hoists the B reads of later stages and spills)
          const bool hand = chain && gi + 1 == n_groups;
          const bool extra = kStreamable && stream && gi + 1 == n_groups;      // the extra stage n - 1 follows this group
          lead(0);
          multiply(buf, 0, F0);
          a_issue(jg + 2, F0);
          __builtin_amdgcn_sched_barrier(0);
#if KGDET_PLANE_STREAM == 2   // (variant: the hand-over barrier one stage earlier -- the producers only wait for their DMA)
          if constexpr (MODE != 2)
            if (hand && extra) {
              KGDET_TR_ADD(4, tr_t);
              __syncthreads();
              KGDET_TR_ADD(0, tr_t);
            }
#endif
          lead(1);
          multiply(buf, 1, F1);
          a_issue(jg + 3, F1);
          __builtin_amdgcn_sched_barrier(0);
          // chained hand-over under the last group: the plane of the next segment is complete behind the mid-group barrier
          // (its sampling overlaps stages three and four); the fragment sets take the next segment's first two stages
          if constexpr (MODE != 2)
            if (hand && !((KGDET_PLANE_STREAM == 2) && extra)) {
              KGDET_TR_ADD(4, tr_t);
              __syncthreads();
              KGDET_TR_ADD(0, tr_t);
            }
          unsigned so0 = hand ? wq_seg2 + (unsigned)min((r2 & 1) ? 1 : 0, n2 - 1) * (2 * kAPart)
                              : wq_seg + (unsigned)(t0 + min(jg + 4, n - 1)) * (2 * kAPart);
          unsigned so1 = hand ? wq_seg2 + (unsigned)min((r2 & 1) ? 0 : 1, n2 - 1) * (2 * kAPart)
                              : wq_seg + (unsigned)(t0 + min(jg + 5, n - 1)) * (2 * kAPart);
          if (extra) {   // F0: the extra stage; F1: the next segment's stage 0 (no next segment: anything)
            so0 = wq_seg + (unsigned)(t0 + n - 1) * (2 * kAPart);
            so1 = hand ? wq_seg2 : so0;
          }
          lead(0);
          multiply(buf, 2, F0);
          a_issue_at(so0, F0);
          __builtin_amdgcn_sched_barrier(0);
          lead(1);
          multiply(buf, 3, F1);
          a_issue_at(so1, F1);
          if constexpr (kStreamable) {
            if (extra) {
              __builtin_amdgcn_sched_barrier(0);
              lead(0);
              multiply(2, 0, F0);
              a_issue_at(hand ? wq_seg2 + (unsigned)min(1, n2 - 1) * (2 * kAPart) : so0, F0);     // the next segment's stage 1
              // the next segment's first stages sit in (F1, F0); an even first group expects (F0, F1)
              if (hand && !(r2 & 1)) {
                const AFrag T = F0;
                F0 = F1;
                F1 = T;
              }
            }
          }
          if constexpr (MODE != 2)
            if (!chain && gi + 1 == n_groups && has_next)
              load_plane(c16 + 1, wave_s, ConsumerWaves{}, HalfRounds{}, plane_split, plane_items);
        }
        KGDET_TR_ADD(4, tr_t);
        __syncthreads();
        KGDET_TR_ADD(5, tr_t);
      };
      for (int gi = 1; gi < n_groups; gi += 2) {
        group(gi, std::integral_constant<int, 1>{});
        if (gi + 1 < n_groups) group(gi + 1, std::integral_constant<int, 0>{});
      }
      primed = has_next;
      first_ready = chain;
      if (chain && (n_groups & 1)) bsel ^= 1u;
      s += n;
      ++c16;
      t0 = 0;
    }

    if constexpr (!PRODUCER) {
      // (the thread index rebuilt from the lane counter and the wave's SGPR, behind an opaque barrier: hipcc otherwise hoists
      // the stores' lane-dependent addresses to the top of the range and, at 128 registers, spills them)
      int tid_e = wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(tid_e));
      if (s_begin == 0 && s_end == cpt && p.sum_count <= 1) {
        if constexpr (k42) store_output(p, mt, nt, tid_e, acc);
        else store_output_w8(p, mt, nt, tid_e, acc);
      } else {
        float *slab = slabs + ((long long)g * grp.slots + slot) * kTileElems;
        if constexpr (k42) store_slab(slab, tid_e, acc);
        else store_slab_w8(slab, tid_e, acc);
      }
    }
    KGDET_TR_ADD(6, tr_t);
    ++slot;
    cur += s_end - s_begin;
  }
  }
#ifdef KGDET_PLANE_TRACE
  tr[7] = KGDET_TR_NOW() - tr_start;
  if ((tid & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 10; ++c) g_plane_trace[((int)blockIdx.x * 16 + wave_all) * 10 + c] = tr[c];
  }
#endif
}

}  // namespace kgdet
