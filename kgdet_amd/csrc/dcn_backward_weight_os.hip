// Deformable convolution backward w.r.t. the weight, OUTPUT-STATIONARY (gfx950; round 4).
//
// Reference path replaced: deform_conv_backward_parameters_cuda (deform_conv_cuda.cpp:373-484) -- a second
// deformable_im2col into a [C*K, P] column matrix and grad_W += grad_out @ columns^T.
//
//     grad_W[o, c, t] = sum_{b, p} grad_out[b, o, p] * sample(x[b, c], pos(p, t))
// M = 256 output channels, N = (channel, tap) columns, REDUCTION over the B*H*W pixels only (2100 for the KGDet maps).
// Rounds 1-3 (dcn_backward_weight_plane.hip) cut this into 256 x 128 tiles (16 channels x 8 taps: 416 tiles for a head
// stage) and dealt (tile, 16-pixel stage) units stream-K over the 256 CUs: every workgroup left through 64 KB partial
// tiles -- 107 MB written per head stage, 143 MB moved again by a fix-up that gathers a channel chunk's tap groups and
// writes rows: 198 + 41 us, 0.20 of the split-bf16 MFMA roof with the packs, 8.1x the algorithmic traffic.
//
// Here a tile is 256 x (16 channels x 13 taps = 208 columns, 7 MFMA blocks of 32) and a workgroup owns ONE tile for the
// WHOLE reduction: 16 chunks x ceil(K / 13) tap groups = 64 / 32 / 16 tiles for a 7x7 / 5x5 / 3x3 weight, 224 for the
// six problems of a KGDet head stage -- one round over the 256 CUs.  No split over pixels, no partial tiles, no fix-up:
// the accumulators are written straight to grad_weight [O, C, kh, kw] (13-float runs; tile columns ordered (channel, tap)).
//   * consumers: 8 waves, wave w = rows [32 w, 32 w + 32) x all 7 column blocks (112 accumulator registers); its two
//     grad_out fragments per stage come straight from the bf16 hi/lo fragment image dcn_pack_grad_out writes
//     (gq[mt][b][px16][part][khalf][o 256][8 px]: 16 bytes per lane and part, two stages ahead);
//   * producers: 4 waves; a stage = 16 pixels of one image; a thread owns (tap, channel quad, FOUR pixels): four tap
//     records (through LDS, loaded coalesced two stages ahead), 16 ds_read_b128 corner reads from the quad planes of the
//     chunk all in flight together, 64 FMAs, hi/lo split, eight 8-byte B stores -- ONE LDS round trip for the records and one
//     for the corners per stage (a first version with two (tap, quad, pixel pair) tasks per thread in sequence paid four, with
//     a single producer wave per SIMD to hide them: 245 us for a head stage); 208 of the 256 threads have a tap;
//   * one workgroup barrier per stage; the chunk's plane of the NEXT image is loaded between two images' stage loops.
// Deterministic (fixed order), v1 and v2 (records carry mask x weight), weight / deformable groups as channel runs.
#include <type_traits>

#include "dcn_plane.h"

namespace kgdet {

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kOsTaps = 13;                     // taps per tile
constexpr int kOsBlocks = 7;                    // 32-column MFMA blocks per tile (13 x 16 = 208 <= 224)
constexpr int kOsCols = kOsBlocks * 32;         // 224
constexpr int kOsBPart = 2 * kOsCols * 8 * 2;   // one part of a B stage: [khalf][column][8 px] bf16 = 7168 B
constexpr int kOsRecSlot = 8 * 16 * 4;          // 16-byte record pieces of a stage: [8 piece kinds][16 taps][4 pixel quads]
constexpr int kOsPieces = kOsTaps * 32;         // pieces that exist: 13 taps x 16 px x 2 halves = 416
}  // namespace

int dcn_bwd_weight_os_threads() { return kPlaneThreads; }
int dcn_bwd_weight_os_taps() { return kOsTaps; }
size_t dcn_bwd_weight_os_lds_bytes(int parts, int HW) {
  return (size_t)2 * parts * kOsBPart + (size_t)2 * kOsRecSlot * 16 + (size_t)kChunk * dcn_plane_padded_pixels(HW) * sizeof(float);
}

template <int PARTS, bool PRODUCER>
__device__ __forceinline__ void wgrad_os_role(const DcnFwdGroup &grp, unsigned char *smem) {
  unsigned char *Bs = smem;                                                    // [2][PARTS][kOsBPart]
  u32x4 *Rs = reinterpret_cast<u32x4 *>(smem + 2 * PARTS * kOsBPart);          // [2][kOsRecSlot]
  unsigned char *plane = smem + 2 * PARTS * kOsBPart + 2 * kOsRecSlot * 16;    // [4 quads][H*W padded][4 ch] fp32

  const int wtid = threadIdx.x;
  const int tid = PRODUCER ? wtid - kThreads : wtid;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef KGDET_OS_PRIO
  if constexpr (PRODUCER) __builtin_amdgcn_s_setprio(KGDET_OS_PRIO);
#endif

  // tile of this workgroup: consecutive blocks of an XCD (b % 8) take consecutive tiles, so the tiles of one problem --
  // readers of the same grad_out image and tap records -- sit on as few L2s as possible
  const int n_tiles = grp.tile_begin[grp.n];
  const int g = sk_slice_of_block((int)blockIdx.x, (int)gridDim.x);
  if (g >= n_tiles) return;
  int pi = 0;
  while (pi + 1 < grp.n && g >= grp.tile_begin[pi + 1]) ++pi;
  const DcnProblem &p = grp.p[pi];
  const int tile = g - grp.tile_begin[pi];
  const int n_tg = p.tiles_per_image;          // tap groups per channel chunk
  const int n_px16 = p.chunks_per_tap;         // stages per image
  const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
  const int c16 = nt / n_tg, tg = nt - c16 * n_tg;
  const int K = p.K, HoWo = p.HoWo, HW = p.H * p.W;
  const int t0 = tg * kOsTaps;
  const int taps_here = min(kOsTaps, K - t0);
  const int nb_live = (taps_here * 16 + 31) >> 5;          // column blocks that hold a live tap
  const unsigned qstride = (unsigned)dcn_plane_padded_pixels(HW) * 16u;

  f32x16 acc[PRODUCER ? 1 : kOsBlocks];
  if constexpr (!PRODUCER) {
#pragma unroll
    for (int ni = 0; ni < kOsBlocks; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.0f;
  }
  // columns of taps that do not exist (and the 16 padding columns of block 6) stay zero for the whole kernel
  for (int i = wtid; i < 2 * PARTS * kOsBPart / 16; i += kPlaneThreads) reinterpret_cast<u32x4 *>(Bs)[i] = u32x4{0u, 0u, 0u, 0u};

  // producer thread -> (local tap tl = lane & 15, pixel quad pq: pixels 4 pq .. 4 pq + 3 of the stage, channel quad = the wave).
  // Columns of the tile are ordered (channel, tap): column = channel_in_chunk * taps_here + tl, so the 16 lanes of a store
  // group (one pixel quad, taps 0 .. 15) write 16 consecutive 16-byte rows -- conflict-free (with (tap, channel) columns and
  // taps across the waves the lanes of a store differed in multiples of 256 bytes only: 8-way bank conflicts on every B
  // store, ~1000 LDS cycles per stage).
  const int tl = tid & 15, pq = (tid >> 4) & 3, quad = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char *plane_q = plane + quad * qstride;
  // record pieces this thread loads: j = tid and j = tid + 256 -> (tap r_tl, pixel r_px, half r_h)
  const int r_px = (tid >> 1) & 15, r_h = tid & 1;
  const int r_slot0 = ((r_px & 3) * 2 + r_h) * 64 + (tid >> 5) * 4 + (r_px >> 2);
  const int r_slot1 = r_slot0 + 8 * 4;
  const bool rec1 = tid + 256 < kOsPieces;
  const int r_t0 = min(t0 + (tid >> 5), K - 1), r_t1 = min(t0 + 8 + (tid >> 5), K - 1);

  for (int b = 0; b < p.N; ++b) {
    const u32x4 *rec_b = reinterpret_cast<const u32x4 *>(p.taps) + (size_t)((b * p.DG + p.dgi) * K) * HoWo * 2;
    const u32x4 *rec_src0 = rec_b + (size_t)r_t0 * HoWo * 2 + r_h;
    const u32x4 *rec_src1 = rec_b + (size_t)r_t1 * HoWo * 2 + r_h;
    const unsigned char *gq_base = reinterpret_cast<const unsigned char *>(p.wq) +
                                   ((size_t)(mt * p.N + b) * n_px16) * (size_t)(PARTS * kAPart);
    const unsigned char *gq_cons = gq_base + (lane >> 5) * (kTileM * 16) + (wave * 32 + (lane & 31)) * 16;
    const int n = n_px16;

    struct Regs {
      u32x4 r0, r1;
    };
    struct AFrag {
      bf16x8 a[PARTS];
    };
    auto issue = [&](int j, Regs &R) __attribute__((always_inline)) {
      const int q = min(j, n - 1);
      const size_t px = (size_t)min(q * 16 + r_px, HoWo - 1) * 2;
      R.r0 = rec_src0[px];
      R.r1 = rec_src1[px];       // (threads without a second piece re-read a clamped one: unconditional loads)
    };
    auto commit = [&](int slot, const Regs &R) __attribute__((always_inline)) {
      Rs[slot * kOsRecSlot + r_slot0] = R.r0;
      if (rec1) Rs[slot * kOsRecSlot + r_slot1] = R.r1;
    };
    auto a_issue = [&](int j, AFrag &F) __attribute__((always_inline)) {
      const unsigned char *src = gq_cons + (size_t)min(j, n - 1) * (PARTS * kAPart);
#pragma unroll
      for (int part = 0; part < PARTS; ++part) F.a[part] = *reinterpret_cast<const bf16x8 *>(src + part * kAPart);
    };
    auto load_plane = [&]() __attribute__((always_inline)) {
      const float *xb = p.x + ((long long)b * p.C_total + p.c_base) * HW;
      dcn_plane_copy<kPlaneRounds>(xb, HW, p.Cg, c16 * kChunk, plane, qstride, __builtin_amdgcn_readfirstlane(wtid >> 6),
                                   kPlaneThreads / 64, dcn_plane_units(HW), wtid & 63);
    };
    // this thread's 4 pixels x 4 channels at its tap -> four 8-byte pieces of B rows (tap, channel), hi and lo
    auto sample = [&](int slot, int buf) __attribute__((always_inline)) {
      if (tl >= taps_here) return;    // (taps 13 .. 15 of every pixel quad, and the taps a short last group lacks)
#ifdef KGDET_OS_ABL_NOSAMPLE
      return;
#endif
      // record pieces of pixel px = 4 pq + i: kind (px & 3) * 2 + h = 2 i + h at [kind][tap][px >> 2 = pq]
      const u32x4 *rr = Rs + slot * kOsRecSlot + tl * 4 + pq;
      u32x4 ro[4], rw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ro[i] = rr[(2 * i) * 64];
        rw[i] = rr[(2 * i + 1) * 64];
      }
      f32x4 v[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = *reinterpret_cast<const f32x4 *>(plane_q + ro[i][e]);
      float sv[4][4];   // [pixel][channel]
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          float a = __uint_as_float(rw[i][0]) * v[i][0][ch];
#pragma unroll
          for (int e = 1; e < 4; ++e) a = __builtin_fmaf(__uint_as_float(rw[i][e]), v[i][e][ch], a);
          sv[i][ch] = a;
        }
      unsigned char *dstb = Bs + buf * PARTS * kOsBPart + (pq >> 1) * (kOsCols * 16) + (quad * 4 * taps_here + tl) * 16 + (pq & 1) * 8;
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        u32x2 hi, lo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 x2 = {sv[2 * h][ch], sv[2 * h + 1][ch]};
          const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(x2, bf16x2));
          hi[h] = hu;
          const f32x2 d = {x2[0] - __uint_as_float(hu << 16), x2[1] - __uint_as_float(hu & 0xffff0000u)};
          lo[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2));
        }
        *reinterpret_cast<u32x2 *>(dstb + ch * taps_here * 16) = hi;
        if constexpr (PARTS == 2) *reinterpret_cast<u32x2 *>(dstb + ch * taps_here * 16 + kOsBPart) = lo;
      }
    };
    auto multiply = [&](int buf, const AFrag &F) __attribute__((always_inline)) {
#ifdef KGDET_OS_ABL_NOMFMA
      return;
#endif
      if constexpr (!PRODUCER) {
        const unsigned char *B = Bs + buf * PARTS * kOsBPart + (lane >> 5) * (kOsCols * 16) + (lane & 31) * 16;
        // B fragments ONE block ahead of the MFMAs that use them (two alternating sets; read-wait-multiply per block left the
        // LDS latency exposed seven times per stage -- the consumers alone ran at 46 % of the MFMA rate; two blocks ahead
        // spills at the 168-register budget)
        bf16x8 bb[2][PARTS];
        auto fetch = [&](int ni, bf16x8 (&dst)[PARTS]) __attribute__((always_inline)) {
#pragma unroll
          for (int part = 0; part < PARTS; ++part)
            dst[part] = *reinterpret_cast<const bf16x8 *>(B + part * kOsBPart + ni * 32 * 16);
        };
        // (all seven blocks, also of a short tap group whose last blocks are zero columns: straight-line code -- with a
        //  uniform branch per block hipcc kept one accumulator in scratch memory; the full tiles set the time anyway)
        fetch(0, bb[0]);
#pragma unroll
        for (int ni = 0; ni < kOsBlocks; ++ni) {
          if (ni + 1 < kOsBlocks) fetch(ni + 1, bb[(ni + 1) & 1]);
          if constexpr (PARTS == 2) {       // small terms first
            acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[1], bb[ni & 1][0], acc[ni], 0, 0, 0);
            acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0], bb[ni & 1][1], acc[ni], 0, 0, 0);
          }
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0], bb[ni & 1][0], acc[ni], 0, 0, 0);
        }
      }
    };

    // Pipeline of stage s: record loads at body s-4, registers -> LDS record slot s & 1 at body s-2, sampled into
    // B[s & 1] at body s-1, multiplied at body s (grad_out fragments requested at the top of body s-1).  Two register sets alternate.
    Regs RA, RB;
    AFrag FA, FB;
    __syncthreads();     // the image before: its last B stage is multiplied, its plane no longer read
    if constexpr (PRODUCER) {
      issue(0, RA);
      issue(1, RB);
    } else {
      a_issue(0, FA);
    }
    load_plane();
    if constexpr (PRODUCER) {
      commit(0, RA);
      commit(1, RB);
      issue(2, RA);
      issue(3, RB);
    }
    __syncthreads();
    if constexpr (PRODUCER) sample(0, 0);
    __syncthreads();
    auto body = [&](auto I, int j, Regs &R, AFrag &F, AFrag &G) __attribute__((always_inline)) {   // R holds stage j + 2, F stage j, G takes stage j + 1
      constexpr int i = decltype(I)::value;
      if constexpr (PRODUCER) {
        commit(i, R);
        issue(j + 4, R);
        if (j + 1 < n) sample(i ^ 1, i ^ 1);
      } else {
        // the NEXT stage's grad_out fragments are requested at the top of the body, into the set the stage before released:
        // a whole stage time to arrive.  (Requested two stages ahead at the END of the body, the wait at the next body's
        // top -- vmcnt counts in order -- also waited for the loads just issued: an L2 round trip per stage.)
        a_issue(j + 1, G);
        if (j < n) multiply(i, F);
      }
      __syncthreads();
    };
    for (int j = 0; j < n; j += 2) {
      body(std::integral_constant<int, 0>{}, j, RA, FA, FB);
      body(std::integral_constant<int, 1>{}, j + 1, RB, FB, FA);
    }
  }

  if constexpr (!PRODUCER) {
    // straight to grad_weight [O, C, kh, kw]: a lane holds one (channel, tap) column of 16 rows per block
#pragma unroll
    for (int ni = 0; ni < kOsBlocks; ++ni) {
      if (ni >= nb_live) continue;
      const int col = ni * 32 + (lane & 31);
      const int cc = col / taps_here, tlc = col - cc * taps_here;      // column = channel_in_chunk * taps_here + tap
      const int c = c16 * kChunk + cc, t = t0 + tlc;
      if (cc >= kChunk || c >= p.Cg) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = mt * kTileM + wave * 32 + mfma_row(r, lane);
        if (o < p.Og) p.out[((long long)o * p.w_ld + c) * K + t] = acc[ni][r];
      }
    }
  }
}

template <int PARTS>
__global__ __launch_bounds__(kPlaneThreads, 1) void dcn_bwd_weight_os(const DcnFwdGroup grp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) wgrad_os_role<PARTS, true>(grp, smem);
  else wgrad_os_role<PARTS, false>(grp, smem);
}

template __global__ void dcn_bwd_weight_os<2>(const DcnFwdGroup grp);

}  // namespace kgdet
