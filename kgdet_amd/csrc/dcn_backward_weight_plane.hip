// Deformable convolution backward w.r.t. the weight on an LDS-resident feature plane (gfx950).
//
// Reference path replaced: deform_conv_backward_parameters_cuda (deform_conv_cuda.cpp:373-484): a second
// deformable_im2col into a [C*K, P] column matrix in HBM and grad_W += grad_out @ columns^T.
//
//     grad_W[o, c, t] = sum_{b, p} grad_out[b, o, p] * sample(x[b, c], pos(p, t))
// is a GEMM with M = output channels (tile 256), N = (channel, tap) pairs (tile 128 = one 16-channel chunk x 8
// taps) and the REDUCTION over pixels, in stages of 16 pixels of one image:
//   * A stage: grad_out[b, o 256, 16 px] -- dcn_pack_grad_out rewrites grad_out once per call into the bf16 hi/lo
//     fragment image gq[mt][b][px16][part][khalf][o 256][8 px], so a stage is a lane-linear 16 KB copy, exactly
//     like the weight stage of the forward kernel;
//   * B stage: samples of the chunk's 16 channels at 8 taps for 16 pixels.  MFMA fragments hold 8 consecutive
//     reduction indices = 8 consecutive PIXELS per (channel, tap) row, so a producer thread owns (tap, channel
//     quad, 4 consecutive pixels): 4 forward tap records (dcn_build_taps), 16 ds_read_b128 corner reads from the
//     [pixel][16 channel] plane, 16 samples, written as four 8-byte row pieces.  Geometry costs nothing here and
//     the sampling VALU work per sample is a third of the forward kernel's;
//   * the tap records of a stage (4 KB) are fetched coalesced by all producer threads and handed to the sampling
//     threads through LDS; consumers, plane copy, stream-K over (tile, stage) units, slabs: as dcn_plane.h.
// The fix-up (or the kernel itself for unsplit tiles) writes grad_W in its natural [O, C, kh, kw] layout: no
// packed intermediate, no unpack kernel.  v1 and v2 (the records carry mask x bilinear weight); weight groups and
// deformable groups run as channel-run sub-problems (dcn_api.hip).
// Maps beyond the LDS plane (GATHER): no plane -- the corner reads are 16-byte buffer loads from a pixel-major copy of x
// (dcn_to_pixel_major; the records hold row offsets of that copy), issued one stage ahead of the arithmetic that uses
// them; the record ring is one slot deeper for that.
#include <type_traits>

#include "dcn_plane.h"

namespace kgdet {

namespace {
// plain vector type for register-resident copies: HIP's uint4 is a struct whose assignment lowers to a memcpy
// between address spaces, which kept the pipeline's register sets in scratch memory
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kTapsPerTile = 8;  // x 16 channels = 128 columns
}

// grad_out [N, O_total, Ho*Wo] (window o_base .. o_base + Og) -> gq[mt][b][px16][part][khalf][o 256][8 px] bf16
// All grad_out images of a grouped call in ONE launch: blockIdx.z = image (a head stage packs six: one launch instead of six).
// grid = (max over images of N * n_px16, max of M tiles, images), 256 threads.
__global__ __launch_bounds__(256) void dcn_pack_grad_out(const DcnPackGradOut g, int parts) {
  const DcnPackGradOutItem &it = g.item[blockIdx.z];
  const int stage = blockIdx.x;               // (b, px16)
  const int mt = blockIdx.y;
  if (stage >= it.N * it.n_px16 || mt >= it.n_mtiles) return;
  const int N = it.N, O_total = it.O_total, o_base = it.o_base, Og = it.Og, HoWo = it.HoWo, n_px16 = it.n_px16;
  const float *__restrict__ gout = it.gout;
  const int b = stage / n_px16, q = stage - b * n_px16;
  const int o_in = threadIdx.x, o = mt * kTileM + o_in;
  const float *src = gout + ((long long)b * O_total + o_base + min(o, Og - 1)) * HoWo;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = src[min(q * 16 + i, HoWo - 1)];
  unsigned char *dst = reinterpret_cast<unsigned char *>(it.gq) +
                       ((size_t)(mt * N + b) * n_px16 + q) * (size_t)(parts * kAPart) + o_in * 16;
#pragma unroll
  for (int khalf = 0; khalf < 2; ++khalf) {
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int px = q * 16 + khalf * 8 + j;
      const float f = (px < HoWo && o < Og) ? v[khalf * 8 + j] : 0.0f;
      hi[j] = (__bf16)f;
      lo[j] = (__bf16)(f - (float)hi[j]);
    }
    *reinterpret_cast<bf16x8 *>(dst + khalf * (kTileM * 16)) = hi;
    if (parts == 2) *reinterpret_cast<bf16x8 *>(dst + kAPart + khalf * (kTileM * 16)) = lo;
  }
}

int dcn_bwd_weight_plane_threads() { return kPlaneThreads; }
size_t dcn_bwd_weight_plane_lds_bytes(int parts, int HW) {
  return (size_t)3 * parts * kAPart + (size_t)2 * parts * kBPart + 2 * 4096 +
         (size_t)kChunk * dcn_plane_padded_pixels(HW) * sizeof(float);
}
size_t dcn_bwd_weight_gather_lds_bytes(int parts) { return (size_t)3 * parts * kAPart + (size_t)2 * parts * kBPart + 3 * 4096; }

// element (row o, column n) of tile (mt, c16, tg) -> grad_weight[o][c][t]
__device__ __forceinline__ void wgrad_store_elem(const DcnProblem &p, int mt, int c16, int tg, int row, int col, float v) {
  const int o = mt * kTileM + row;
  const int c = c16 * kChunk + (col & 15), t = tg * kTapsPerTile + (col >> 4);
  if (o < p.Og && c < p.Cg && t < p.K) p.out[((long long)o * p.w_ld + c) * p.K + t] = v;
}

template <int PARTS, bool PRODUCER, bool GATHER>
__device__ __forceinline__ void wgrad_role(const DcnFwdGroup &grp, float *__restrict__ slabs, unsigned char *smem) {
  static_assert(kAFromL2 || !GATHER, "the gather pipeline has no slot timeline for grad_out stages in LDS");
  unsigned char *As = smem;                                       // [3][PARTS][kAPart]  (ring)
  unsigned char *Bs = smem + 3 * PARTS * kAPart;                  // [2][PARTS][kBPart]
  u32x4 *Rs = reinterpret_cast<u32x4 *>(smem + 3 * PARTS * kAPart + 2 * PARTS * kBPart);  // [2 | GATHER: 3][256] record pieces
  unsigned char *plane = smem + 3 * PARTS * kAPart + 2 * PARTS * kBPart + 2 * 4096;  // [4 quads][H*W padded][4 channels] fp32 (not GATHER)

  const int wtid = threadIdx.x;
  const int tid = PRODUCER ? wtid - kThreads : wtid;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;                        // consumers
  // producers: every thread samples 2 consecutive pixels x 4 channels at one tap
  const int pp = tid & 7, tl = (tid >> 3) & 7, quad = (tid >> 6) & 3;  // pixel pair, local tap, channel quad
  const long long G = gridDim.x, g = blockIdx.x;
  const long long total = grp.unit_begin[grp.n];
  const long long slice = sk_slice_of_block((int)g, (int)G);
  const long long my_begin = unit_begin(slice, total, G);
  const long long my_end = unit_begin(slice + 1, total, G);

  long long cur = my_begin;
  int slot = 0;
  while (cur < my_end) {
    const DcnUnitPos pos = dcn_unit_pos(grp, cur);
    const DcnProblem &p = grp.p[pos.pi];
    const int HW = p.H * p.W;
    const unsigned qstride = (unsigned)dcn_plane_padded_pixels(HW) * 16u;   // bytes between the plane's channel quads
    const unsigned char *plane_q = plane + quad * qstride;                    // producers: this thread's quad plane
    const dcn_rsrc_t x_rs = dcn_make_rsrc(p.x);                               // (GATHER) the pixel-major copy of x
    const int K = p.K, HoWo = p.HoWo;
    const int n_px16 = p.chunks_per_tap;      // stages per image
    const int n_tg = p.tiles_per_image;       // tap groups per channel chunk
    const int tile = pos.tile;
    const int s_begin = pos.s;
    const int s_end = (int)((my_end - cur) < (long long)(pos.s_hi - pos.s) ? pos.s + (my_end - cur) : pos.s_hi);
    const int mt = tile % p.n_mtiles, nt = tile / p.n_mtiles;
    const int c16 = nt / n_tg, tg = nt - c16 * n_tg;
    const int t_mine = tg * kTapsPerTile + tl;  // the tap this thread samples
    const bool tap_live = t_mine < K;

    f32x16 acc[PRODUCER ? 1 : 2][PRODUCER ? 1 : 2];
    if constexpr (!PRODUCER) zero_acc(acc);

    struct Regs {
      u32x4 rec;            // 16-byte piece tid of the stage's 4 KB of tap records (8 taps x 16 pixels x 32 B)
      u32x4 a0, a1, a2, a3; // this thread's share of the grad_out stage (a2, a3: the lo part)
    };

    int s = s_begin;
    int b = s / n_px16;
    int q0 = s - b * n_px16;
    while (s < s_end) {
      const int n = min(n_px16 - q0, s_end - s);  // stages of this segment: pixel groups q0 .. q0+n-1 of image b
      // (GATHER) image b, this wave's four channels of chunk c16: scalar part of the corner addresses
      const unsigned x_soff = __builtin_amdgcn_readfirstlane(
          (unsigned)(((size_t)b * HW * p.C_total + p.c_base + c16 * kChunk + quad * 4) * sizeof(float)));
      const unsigned char *gq_base = reinterpret_cast<const unsigned char *>(p.wq) +
                                     ((size_t)(mt * p.N + b) * n_px16) * (size_t)(PARTS * kAPart);

      // Every producer thread loads ONE 16-byte piece of the stage's tap records (coalesced: 512 contiguous bytes
      // per tap) and 2 * PARTS pieces of the grad_out stage; the records go through LDS to the sampling threads.
      // (Letting each sampling thread fetch its own four records cost 140 us per launch in the texture
      // addresser -- 32 cache lines per load instruction -- and loads issued under `if (sampler)` another 200 us
      // of drained-queue latency.)
      const int r_tl = tid >> 5, r_px = (tid >> 1) & 15, r_h = tid & 1;
      const int r_t = min(tg * kTapsPerTile + r_tl, K - 1);
      const u32x4 *rec_src = reinterpret_cast<const u32x4 *>(p.taps) + ((size_t)((b * p.DG + p.dgi) * K + r_t) * HoWo) * 2 + r_h;
      const int r_slot = ((r_px & 3) * 2 + r_h) * 32 + r_tl * 4 + (r_px >> 2);  // [piece][tap][pixel quad]
      auto issue = [&](int j, Regs &R) __attribute__((always_inline)) {
        const int q = q0 + min(j, n - 1);
        R.rec = rec_src[(size_t)min(q * 16 + r_px, HoWo - 1) * 2];
        if constexpr (!kAFromL2) {
          const u32x4 *src = reinterpret_cast<const u32x4 *>(gq_base + (size_t)q * (PARTS * kAPart)) + tid;
          R.a0 = src[0];
          R.a1 = src[kProducers];
          if constexpr (PARTS == 2) {
            R.a2 = src[2 * kProducers];
            R.a3 = src[3 * kProducers];
          }
        }
      };
      // consumers (kAFromL2, see dcn_plane.h): the wave's grad_out fragments of stage j straight from the fragment image
      struct AFrag {
        bf16x8 a[PARTS][2];
      };
      const unsigned char *gq_cons = gq_base + (lane >> 5) * (kTileM * 16) + (wm * 64 + (lane & 31)) * 16;
      auto a_issue = [&](int j, AFrag &F) __attribute__((always_inline)) {
        const unsigned char *bsrc = gq_cons + (size_t)(q0 + min(j, n - 1)) * (PARTS * kAPart);
#pragma unroll
        for (int part = 0; part < PARTS; ++part)
#pragma unroll
          for (int i = 0; i < 2; ++i) F.a[part][i] = *reinterpret_cast<const bf16x8 *>(bsrc + part * kAPart + i * 32 * 16);
      };
      auto commit = [&](int a_slot, int r_slot2, const Regs &R) __attribute__((always_inline)) {  // grad_out stage -> As[a_slot], records -> Rs[r_slot2]
        if constexpr (!kAFromL2) {
          u32x4 *dst = reinterpret_cast<u32x4 *>(As + a_slot * PARTS * kAPart) + tid;
          dst[0] = R.a0;
          dst[kProducers] = R.a1;
          if constexpr (PARTS == 2) {
            dst[2 * kProducers] = R.a2;
            dst[3 * kProducers] = R.a3;
          }
        }
        Rs[r_slot2 * 256 + r_slot] = R.rec;
      };
      auto load_plane = [&]() __attribute__((always_inline)) {  // x[b, chunk c16] -> the LDS quad planes (dcn_common.h)
        const float *xb = p.x + ((long long)b * p.C_total + p.c_base) * HW;
        dcn_plane_copy<kPlaneRounds>(xb, HW, p.Cg, c16 * kChunk, plane, qstride, __builtin_amdgcn_readfirstlane(wtid >> 6),
                                     kPlaneThreads / 64, dcn_plane_units(HW), wtid & 63);
      };
      // sampler: 4 pixels x 4 channels at one tap -> four 8-byte pieces of B rows (channel, tap)
      // (GATHER) the corner values of a stage, fetched one body ahead of the arithmetic: [pixel][corner]
      struct Corners {
        f32x4 v[2][4];
      };
      auto gather_issue = [&](int rslot, Corners &V) __attribute__((always_inline)) {
        const u32x4 *rr = Rs + rslot * 256 + ((pp & 1) * 4) * 32 + tl * 4 + (pp >> 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 ro = rr[(2 * i) * 32];
#pragma unroll
          for (int e = 0; e < 4; ++e) V.v[i][e] = __builtin_bit_cast(f32x4, dcn_buf_b128(x_rs, ro[e], x_soff));
        }
      };
      auto sample = [&](int rslot, int buf, const Corners &V) __attribute__((always_inline)) {  // B stage (slot buf) from the records in Rs[rslot] and the plane / V
        // record pieces of pixel px = 2 pp + i sit at [piece ((px & 3) * 2 + h)][tap][px >> 2]
        const u32x4 *rr = Rs + rslot * 256 + ((pp & 1) * 4) * 32 + tl * 4 + (pp >> 1);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x4 sv[2];  // [pixel][channel]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          u32x4 ro = {0u, 0u, 0u, 0u};
          if constexpr (!GATHER) ro = rr[(2 * i) * 32];
          const u32x4 rw = rr[(2 * i + 1) * 32];
          const unsigned o[4] = {ro[0], ro[1], ro[2], ro[3]};
          const float w[4] = {__uint_as_float(rw[0]), __uint_as_float(rw[1]), __uint_as_float(rw[2]), __uint_as_float(rw[3])};
          f32x2 lo2 = {0.f, 0.f}, hi2 = {0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x4 v;
            if constexpr (GATHER) v = V.v[i][e];
            else v = *reinterpret_cast<const f32x4 *>(plane_q + o[e]);
            const f32x2 we = {w[e], w[e]};
            lo2 = __builtin_elementwise_fma(we, f32x2{v[0], v[1]}, lo2);
            hi2 = __builtin_elementwise_fma(we, f32x2{v[2], v[3]}, hi2);
          }
          sv[i] = tap_live ? f32x4{lo2[0], lo2[1], hi2[0], hi2[1]} : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // B rows are stored at slot n ^ (n >> 4): the 8 taps of a channel then hit 8 different bank groups
        // (unswizzled, the lanes of one store differ only in bits that are multiples of 32 dwords)
        unsigned char *dstb = Bs + buf * PARTS * kBPart + (pp >> 2) * (kTileN * 16) + (pp & 3) * 4;
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          unsigned char *dst = dstb + (((tl * 16 + quad * 4 + ch) ^ tl) * 16);
          bf16x2 hi, lo;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            hi[i] = (__bf16)sv[i][ch];
            lo[i] = (__bf16)(sv[i][ch] - (float)hi[i]);
          }
          *reinterpret_cast<bf16x2 *>(dst) = hi;
          if constexpr (PARTS == 2) *reinterpret_cast<bf16x2 *>(dst + kBPart) = lo;
        }
      };
      auto multiply = [&](int a_slot, int buf, const AFrag &F) __attribute__((always_inline)) {
        if constexpr (!PRODUCER) {
          const unsigned char *A = As + a_slot * PARTS * kAPart + (lane >> 5) * (kTileM * 16) + (wm * 64 + (lane & 31)) * 16;
          const unsigned char *B = Bs + buf * PARTS * kBPart + (lane >> 5) * (kTileN * 16);
          bf16x8 a[PARTS][2], bb[PARTS][2];
#pragma unroll
          for (int part = 0; part < PARTS; ++part)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int col = wn * 64 + i * 32 + (lane & 31);
              if constexpr (kAFromL2) a[part][i] = F.a[part][i];
              else a[part][i] = *reinterpret_cast<const bf16x8 *>(A + part * kAPart + i * 32 * 16);
              bb[part][i] = *reinterpret_cast<const bf16x8 *>(B + part * kBPart + (col ^ (col >> 4)) * 16);
            }
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if (tg * kTapsPerTile + wn * 4 + ni * 2 >= K) continue;  // both taps of this 32-column block are padding
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              if constexpr (PARTS == 2) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], bb[0][ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], bb[1][ni], acc[mi][ni], 0, 0, 0);
              }
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], bb[0][ni], acc[mi][ni], 0, 0, 0);
            }
          }
        }
      };

      // Pipeline of stage s: global loads at body s-4, registers -> LDS (A ring slot s % 3, record slot s & 1) at
      // body s-2, sampled into B[s & 1] at body s-1, multiplied at body s.  Two register sets alternate.
      // GATHER: record loads at body s-5, -> record slot s % 3 at body s-3, corner loads issued at body s-2, samples
      // into B[s & 1] at body s-1.
      Regs RA, RB;
      AFrag FA, FB;
      Corners V;
      __syncthreads();
      if constexpr (PRODUCER) {
        issue(0, RA);
        issue(1, RB);
      } else if constexpr (kAFromL2) {
        a_issue(0, FA);
        a_issue(1, FB);
      }
      if constexpr (!GATHER) load_plane();
      if constexpr (PRODUCER) {
        commit(0, 0, RA);
        commit(1, 1, RB);
        issue(2, RA);
        issue(3, RB);
      }
      __syncthreads();
      if constexpr (PRODUCER) {
        if constexpr (GATHER) {
          gather_issue(0, V);
          commit(2, 2, RA);
          issue(4, RA);
          sample(0, 0, V);
          gather_issue(1, V);
        } else {
          sample(0, 0, V);
        }
      }
      __syncthreads();
      // I = body index inside the unrolled group of 6 (compile time: the LDS slots are immediates and the six
      // bodies stay distinct -- with run-time slot arithmetic hipcc merged them back into a loop and kept the
      // two register sets in scratch memory, stalling on every freshly issued load to spill it)
      auto body = [&](auto I, int j, Regs &R, AFrag &F) __attribute__((always_inline)) {  // R holds stage j+2 (GATHER: j+3), F stage j
        constexpr int i = decltype(I)::value;
        if constexpr (PRODUCER) {
          if constexpr (GATHER) {
            if (j + 1 < n) sample((i + 1) % 3, (i + 1) & 1, V);   // V: the corners of stage j+1
            gather_issue((i + 2) % 3, V);
            commit(i % 3, i % 3, R);                              // stage j+3 -> the slot stage j left
            issue(j + 5, R);
          } else {
            commit((i + 2) % 3, i & 1, R);
            issue(j + 4, R);
            if (j + 1 < n) sample((i + 1) & 1, (i + 1) & 1, V);
          }
        } else {
          if (j < n) multiply(i % 3, i & 1, F);
          if constexpr (kAFromL2) a_issue(j + 2, F);
        }
        __syncthreads();
      };
      // (GATHER: stage s lives in the register set of its parity, so body j takes the set of parity j+3)
      for (int j = 0; j < n; j += 6) {  // 6 = lcm(3 A slots, 2 B / record slots, 2 register sets)
        body(std::integral_constant<int, 0>{}, j, GATHER ? RB : RA, FA);
        body(std::integral_constant<int, 1>{}, j + 1, GATHER ? RA : RB, FB);
        body(std::integral_constant<int, 2>{}, j + 2, GATHER ? RB : RA, FA);
        if (j + 3 < n) {
          body(std::integral_constant<int, 3>{}, j + 3, GATHER ? RA : RB, FB);
          body(std::integral_constant<int, 4>{}, j + 4, GATHER ? RB : RA, FA);
          body(std::integral_constant<int, 5>{}, j + 5, GATHER ? RA : RB, FB);
        }
      }
      s += n;
      ++b;
      q0 = 0;
    }

    // every range leaves through a slab -- also a whole tile: a tile holds 8 of a channel's K taps, so straight from the
    // accumulators grad_weight would be written in scattered 4-byte pieces; the fix-up gathers the tap groups of a
    // channel chunk and writes whole rows
    if constexpr (!PRODUCER) store_slab(slabs + ((long long)g * grp.slots + slot) * kTileElems, tid, acc);
    ++slot;
    cur += s_end - s_begin;
  }
}

template <int PARTS>
__global__ __launch_bounds__(kPlaneThreads, 1) void dcn_bwd_weight_plane(const DcnFwdGroup grp, float *__restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) wgrad_role<PARTS, true, false>(grp, slabs, smem);
  else wgrad_role<PARTS, false, false>(grp, slabs, smem);
}

template __global__ void dcn_bwd_weight_plane<1>(const DcnFwdGroup grp, float *__restrict__ slabs);
template __global__ void dcn_bwd_weight_plane<2>(const DcnFwdGroup grp, float *__restrict__ slabs);

// maps beyond the LDS plane: grp.gather_mode records, p.x = the pixel-major copy of x
template <int PARTS>
__global__ __launch_bounds__(kPlaneThreads, 1) void dcn_bwd_weight_gather(const DcnFwdGroup grp, float *__restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) wgrad_role<PARTS, true, true>(grp, slabs, smem);
  else wgrad_role<PARTS, false, true>(grp, slabs, smem);
}

template __global__ void dcn_bwd_weight_gather<2>(const DcnFwdGroup grp, float *__restrict__ slabs);

// Fix-up: one workgroup per (problem, channel chunk c16, M tile, block of 8 output channels).  For every tap group tg of
// the chunk it adds the slabs of tile (c16, tg) -- the workgroups' partial tiles in slice order, a fixed order -- and drops the
// 8 x 128 sums into an LDS image [8 o][16 c][K taps]; the image then leaves as 8 contiguous runs of 16 K floats of
// grad_weight [O, C, kh, kw].  (Rounds 2-3 stored a tile's elements where they fell: 4-byte pieces K floats apart,
// 82 MB of fabric writes for 54.5 MB of gradient, 55 us per head stage.)  The slab list of every tap group's tile is worked
// out once, by one thread per tap group (64-bit divisions), and shared through LDS.
// grid = (sum over problems of n_c16 * n_mtiles * 32), 256 threads, LDS = 8 * 16 * K floats.
constexpr int kFixMaxTg = 16, kFixMaxSlabs = 32;
__global__ __launch_bounds__(256) void dcn_bwd_weight_plane_fixup(const DcnFwdGroup grp, const float *__restrict__ slabs,
                                                                  int G) {
  extern __shared__ float wimg[];   // [8][16 * K]
  __shared__ long long slab_of[kFixMaxTg][kFixMaxSlabs];
  __shared__ int n_slab[kFixMaxTg];
  int pi = 0, blk = blockIdx.x;
  while (pi + 1 < grp.n) {
    const int nb = (grp.p[pi].n_ntiles / grp.p[pi].tiles_per_image) * grp.p[pi].n_mtiles * 32;
    if (blk < nb) break;
    blk -= nb;
    ++pi;
  }
  const DcnProblem &p = grp.p[pi];
  const int n_tg = p.tiles_per_image, K = p.K;
  const int rb = blk & 31, mt = (blk >> 5) % p.n_mtiles, c16 = (blk >> 5) / p.n_mtiles;
  const int wm = rb >> 3, mi = (rb >> 2) & 1, q = rb & 3;    // rows wm * 64 + mi * 32 + 8 q + {0..3, 4..7 (upper lanes)}
  const int tid = threadIdx.x, lane = tid & 63, ni = (tid >> 6) & 1, wn = tid >> 7;
  const int stid = (wm + 4 * wn) * 64 + lane;   // the thread of the main kernel that held these accumulators
  const long long total = grp.unit_begin[grp.n];
  if (tid < n_tg) {
    const int tile = (c16 * n_tg + tid) * p.n_mtiles + mt;
    const long long tb = dcn_range_first_unit(grp, pi, 0, tile), te = tb + p.chunks_per_tile;
    long long g = tb * G / total;
    while (unit_begin(g + 1, total, G) <= tb) ++g;
    while (unit_begin(g, total, G) > tb) --g;
    const int range = grp.range_begin[pi] + tile;
    int n = 0;
    for (; g < G && n < kFixMaxSlabs; ++g) {
      const long long b0 = unit_begin(g, total, G);
      if (b0 >= te) break;
      if (unit_begin(g + 1, total, G) == b0) continue;
      slab_of[tid][n++] = (long long)sk_block_of_slice((int)g, G) * grp.slots + (range - dcn_unit_pos(grp, b0).range);
    }
    n_slab[tid] = n;
  }
  __syncthreads();
  const int col = wn * 64 + ni * 32 + (lane & 31);
  const int c_local = col & 15, t_local = col >> 4;
  const int row_local = 4 * (lane >> 5);    // + e: the row inside this block of 8
  for (int tg = 0; tg < n_tg; ++tg) {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int n = n_slab[tg];
    for (int i = 0; i < n; ++i) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(slabs + slab_of[tg][i] * kTileElems)[((mi * 2 + ni) * 4 + q) * kThreads + stid];
      sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
    }
    const int t = tg * 8 + t_local;
    if (t < K) {
#pragma unroll
      for (int e = 0; e < 4; ++e) wimg[((row_local + e) * 16 + c_local) * K + t] = sum[e];
    }
  }
  __syncthreads();
  const int n_c = min(16, p.Cg - c16 * kChunk);          // channels of the chunk that exist
  const int run = n_c * K;
  for (int r = 0; r < 8; ++r) {
    const int o = mt * kTileM + wm * 64 + mi * 32 + 8 * q + r;
    if (o >= p.Og) break;
    float *dst = p.out + ((long long)o * p.w_ld + c16 * kChunk) * K;
    for (int i = tid; i < run; i += 256) dst[i] = wimg[r * 16 * K + i];
  }
}

}  // namespace kgdet
