// Deformable convolution backward w.r.t. the offsets on an LDS-resident feature plane (gfx950).
//
// Reference path replaced: deformable_col2im_coord (deform_conv_cuda_kernel.cu:337-435, get_coordinate_weight
// :144-187) applied to columns = W^T grad_out (deform_conv_cuda.cpp:329-332):
//     grad_offset[b, 2t + dir, p] = sum_c colgrad[c, t, p] * d sample(x[b, c], pos(p, t)) / d dir,
//     colgrad[c, t, p]            = sum_o W[o, c, t] * grad_out[b, o, p].
// The reference materialises the [C*K, P] column-gradient matrix in HBM; here it never leaves registers:
//   * a workgroup owns a tile of 128 output pixels of one image; each of its 8 consumer waves keeps the
//     grad_out fragment of ITS 16 pixels -- [256 o][16 px], bf16 hi/lo, 64 VGPRs -- for the whole tile;
//   * a stage = (16-channel chunk, tap): colgrad[16 c][16 px] = W_t^T[16 c x 256 o] . g[256 o x 16 px] as
//     8 k-steps of v_mfma_f32_16x16x32_bf16 x 3 products (hi/lo split, fp32 accumulate); the W_t^T stage (16 KB)
//     is staged in LDS by the producer waves straight from the transposed operand image `wqt`;
//   * the accumulator layout hands every lane 4 consecutive channels of one pixel -- exactly one quad of the
//     [pixel][16 channel] x plane in LDS, so the four bilinear corners are four ds_read_b128; the lane forms
//     sum_c colgrad * corner, applies the record's derivative weights, the four quad lanes of a pixel are
//     reduced with two cross-lane adds and the tap's running sums live in an LDS accumulator [K][128][2];
//   * the reduction runs chunk-major / tap-minor like the forward kernel (the x plane is copied once per chunk),
//     stream-K over (tile, stage) units, partial accumulators to slabs, dcn_bwd_offset_plane_fixup adds them.
// v1 only (no modulation mask), output channels per group <= 256; other shapes stay on the gather kernel.
#include "dcn_plane.h"

namespace kgdet {

namespace {

constexpr int kOffThreads = kThreads + kProducers;  // 8 consumer + 4 producer waves
constexpr int kMaxKs = 8;                            // k-steps of 32 output channels: Og <= 256

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

}  // namespace

// One 48-byte record per (image, tap, output pixel): corner offsets as in DcnTapRec, and the weights that turn
// the four corner values into d sample / dy and d sample / dx (zero where the corner, or the tap, is outside).
// Modulated (v2) problems: the derivative weights carry the mask value, and a fourth 16 bytes hold the plain bilinear
// weights -- d out / d mask is the sampled value itself (deform_conv_cuda_kernel.cu:636-766) -- 64 bytes per record.
__global__ __launch_bounds__(256) void dcn_build_grad_taps(const DcnFwdGroup grp) {
  const DcnProblem &p = grp.p[blockIdx.y];
  // (tap-pair kernel: the blocked copy of x its plane switches read by LDS-DMA, as dcn_build_taps does for the forward)
  if (p.build_xblk) dcn_block_x_body(p, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), (int)gridDim.x * 4);
  if (!p.build_taps) return;
  const long long n_rec = (long long)p.N * p.K * p.HoWo;
  uint4 *out = reinterpret_cast<uint4 *>(const_cast<DcnTapRec *>(p.taps));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_rec; i += (long long)gridDim.x * 256) {
    const int hw = (int)(i % p.HoWo);
    const int t = (int)((i / p.HoWo) % p.K);
    const int b = (int)(i / ((long long)p.HoWo * p.K));
    const int oy = hw / p.Wo, ox = hw - oy * p.Wo;
    const int ti = t / p.kw, tj = t - ti * p.kw;
    const long long ob = ((long long)(b * p.DG + p.dgi) * 2 * p.K + 2 * t) * p.HoWo + hw;   // (this run's deformable group)
    const float y = (float)(oy * p.sh - p.ph + ti * p.dh) + p.offset[ob];
    const float x = (float)(ox * p.sw - p.pw + tj * p.dw) + p.offset[ob + p.HoWo];
    Tap tap;
    TapGeom geo;
    make_tap(y, x, p.H, p.W, true, 1.0f, tap, geo);
    const float hy = 1.0f - geo.ly, hx = 1.0f - geo.lx;
    const float m = p.mask ? p.mask[((long long)(b * p.DG + p.dgi) * p.K + t) * p.HoWo + hw] : 1.0f;
    const int rs = p.mask ? 4 : 3;   // 16-byte pieces per record
    const float ka = geo.va ? m : 0.f, kb = geo.vb ? m : 0.f, kc = geo.vc ? m : 0.f, kd = geo.vd ? m : 0.f;
    unsigned off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) off[e] = (unsigned)dcn_plane_offset(tap.o[e]);
    out[i * rs + 0] = make_uint4(off[0], off[1], off[2], off[3]);
    // d/dy = hx (v10 - v00) + lx (v11 - v01),  d/dx = hy (v01 - v00) + ly (v11 - v10)   (:144-187)
    out[i * rs + 1] = make_uint4(__float_as_uint(-hx * ka), __float_as_uint(-geo.lx * kb), __float_as_uint(hx * kc),
                                __float_as_uint(geo.lx * kd));
    out[i * rs + 2] = make_uint4(__float_as_uint(-hy * ka), __float_as_uint(hy * kb), __float_as_uint(-geo.ly * kc),
                                __float_as_uint(geo.ly * kd));
    if (p.mask)   // (make_tap was called with mask 1: tap.w = the bilinear weights, zero outside)
      out[i * rs + 3] = make_uint4(__float_as_uint(tap.w[0]), __float_as_uint(tap.w[1]), __float_as_uint(tap.w[2]),
                                  __float_as_uint(tap.w[3]));
  }
}

// LDS: A [2][PARTS][8 KB] | offs_acc [K][128][2 (v2: 4)] fp32 | x plane [4 quads][H*W padded to 64][4 ch] fp32
// (the quad planes of dcn_common.h at a run-time stride: what is left for the plane depends on K)
size_t dcn_bwd_offset_plane_lds_bytes(int parts, int K, int HW, int masked) {
  return (size_t)2 * parts * kAPart + (size_t)K * kTileN * (masked ? 4 : 2) * sizeof(float) +
         (size_t)kChunk * dcn_plane_padded_pixels(HW) * sizeof(float);
}
int dcn_bwd_offset_plane_threads() { return kOffThreads; }

// MASK: modulated (v2) problems -- a third sum per (tap, pixel), grad_mask
template <int PARTS, bool PRODUCER, bool MASK>
__device__ __forceinline__ void offset_role(const DcnFwdGroup &grp, float *__restrict__ slabs, unsigned char *smem,
                                            int max_K) {
  constexpr int ACC = MASK ? 4 : 2;   // floats per (tap, pixel) sum: (dy, dx) or (dy, dx, dmask, -)
  constexpr int RS = MASK ? 4 : 3;    // 16-byte pieces per record
  unsigned char *As = smem;                                             // [2][PARTS][kAPart]: [o16][khalf][c 16][8 o]
  float *offs_acc = reinterpret_cast<float *>(smem + 2 * PARTS * kAPart);  // [K][128][ACC]
  unsigned char *plane = smem + 2 * PARTS * kAPart + (size_t)max_K * kTileN * ACC * sizeof(float);

  const int wtid = threadIdx.x;
  const int tid = PRODUCER ? wtid - kThreads : wtid;
  const int lane = tid & 63, wave = tid >> 6;
  const int px16 = lane & 15, kg = lane >> 4;       // consumers: pixel inside the wave's 16, channel quad / k group
  const long long G = gridDim.x, g = blockIdx.x;
  const long long slice = sk_slice_of_block((int)g, (int)G);
  long long my_begin, my_end;
  dcn_slice_bounds(grp, slice, G, my_begin, my_end);   // (static ranges: exactly one range, or nothing)

  long long cur = my_begin;
  int slot = 0;
  while (cur < my_end) {
    const DcnUnitPos pos = dcn_unit_pos(grp, cur);
    const DcnProblem &p = grp.p[pos.pi];
    const int HW = p.H * p.W;
    const unsigned qstride = (unsigned)dcn_plane_padded_pixels(HW) * 16u;   // bytes between the plane's channel quads
    const unsigned char *plane_kg = plane + kg * qstride;                     // consumers: this lane's quad plane
    const int K = p.K;
    const int cpt = p.chunks_per_tile;
    const int nt = pos.tile;  // one M "tile": the channel chunks are part of the reduction here
    const int s_begin = pos.s;
    const int s_end = (int)((my_end - cur) < (long long)(pos.s_hi - pos.s) ? pos.s + (my_end - cur) : pos.s_hi);
    const int tile_b = nt / p.tiles_per_image;
    const int tile_px0 = (nt - tile_b * p.tiles_per_image) * kTileN;
    const int HoWo = p.HoWo;
    const int n_o16 = (p.Og + kChunk - 1) / kChunk;   // 16-channel chunks of output channels in `wqt`
    const int n_ks = (n_o16 + 1) / 2;                  // k-steps of 32 output channels
    // this lane's pixel (columns past the end of the image redo pixel 0: never stored)
    const int my_px = tile_px0 + wave * 16 + px16;
    const int my_px_c = my_px < HoWo ? my_px : 0;

    // ---- consumers: grad_out fragment of the wave's 16 pixels, [n_ks][8 o] per lane, bf16 hi / lo ----
    bf16x8v gh[PRODUCER ? 1 : kMaxKs], gl[PRODUCER ? 1 : kMaxKs];
    if constexpr (!PRODUCER) {
      const float *gimg = p.gout + ((long long)tile_b * p.O_total + p.o_base) * HoWo + my_px_c;
#pragma unroll
      for (int ks = 0; ks < kMaxKs; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = ks * 32 + kg * 8 + j;
          v[j] = gimg[(long long)min(o, p.Og - 1) * HoWo];  // unconditional, clamped
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = ks * 32 + kg * 8 + j;
          const float f = (o < p.Og && ks < n_ks) ? v[j] : 0.0f;
          gh[ks][j] = (__bf16)f;
          gl[ks][j] = (__bf16)(f - (float)gh[ks][j]);
        }
      }
    }
    // zero the tap accumulators of this range
    __syncthreads();
    for (int i = wtid; i < K * kTileN * ACC; i += kOffThreads) offs_acc[i] = 0.0f;

    struct Regs {
      f32x4 a[PARTS][2];     // producers: this thread's 2 x 16 B of each part of the W^T stage
      uint4 off;             // consumers: the record of (pixel, tap)
      f32x4 wy, wx, wm;
    };
    // The three register sets live across the segments (a segment = the taps of one 16-channel chunk): the last bodies of a
    // segment put the FIRST three stages of the next one in flight instead of re-reading their own last stage, so a segment
    // does not start with a bare round trip to the weight image (21.7 MB per map, streamed once per pixel tile: it is served
    // from beyond the XCD's L2) and the records -- 16 times per tile for a 3x3 problem with its nine stages per segment.
    Regs R0, R1, R2;
    bool prefetched = false;

    int s = s_begin;
    int c16 = s / K;
    int t0 = s - c16 * K;
    while (s < s_end) {
      const int n = min(K - t0, s_end - s);
      const int n_pad = (n + 2) / 3 * 3;                         // bodies the loop below runs (three per round)
      const bool has_next = s + n < s_end;
      const int n_next = has_next ? min(K, s_end - (s + n)) : 1;
      // (buffer loads -- SGPR resource, 32-bit lane offset, scalar stage offset: no vector address arithmetic per stage;
      // VALU instructions beside the MFMA waves cost MFMA issue slots, tools/microbench/mfma_valu.hip)
      const dcn_rsrc_t rec_rs = dcn_make_rsrc(p.taps);
      const unsigned rec_seg = (unsigned)(tile_b * K) * (unsigned)(HoWo * RS * 16);
      const unsigned rec_lane = (unsigned)my_px_c * (unsigned)(RS * 16);
      // W^T stage of (chunk c16, tap t): for every 16-o chunk o16 the rows c16*16 .. +15 of both k-halves:
      // 256-byte runs inside wqt[ct][o16][t][part][khalf][c 256][8 o]
      const int ct = ((c16 + p.c16_base) * kChunk) / kTileM, c_in = ((c16 + p.c16_base) * kChunk) % kTileM;
      const dcn_rsrc_t wq_rs = dcn_make_rsrc(p.wq);
      const unsigned wq_base = (unsigned)(ct * n_o16 * K) * (unsigned)(2 * kAPart);
      unsigned a_lane[2];   // producers: this thread's two 16-byte units of a stage image, relative to the stage of o16 = 0
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int idx = tid + r * kProducers;          // 16-byte unit of the stage image: [o16 16][khalf 2][c 16]
        const int c = idx & 15, khalf = (idx >> 4) & 1, o16 = idx >> 5;
        a_lane[r] = (unsigned)(min(o16, n_o16 - 1) * K) * (unsigned)(2 * kAPart) + khalf * (kTileM * 16) + (c_in + c) * 16;
      }
      // the same for the next segment's chunk (its stages start at tap 0)
      const int ct_n = ((c16 + 1 + p.c16_base) * kChunk) / kTileM, c_in_n = ((c16 + 1 + p.c16_base) * kChunk) % kTileM;
      const unsigned wq_base_n = (unsigned)(ct_n * n_o16 * K) * (unsigned)(2 * kAPart);
      const unsigned a_lane_d = (unsigned)((c_in_n - c_in) * 16);      // (added modulo 2^32: c_in_n may be below c_in)

      auto issue = [&](int j, Regs &R) {
        const bool nxt = has_next && j >= n_pad;                       // (uniform)
        const unsigned t = nxt ? (unsigned)min(j - n_pad, n_next - 1) : (unsigned)(t0 + min(j, n - 1));
        if constexpr (PRODUCER) {
          const unsigned base = nxt ? wq_base_n : wq_base, dl = nxt ? a_lane_d : 0u;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int part = 0; part < PARTS; ++part)
              R.a[part][r] = __builtin_bit_cast(f32x4, dcn_buf_b128(wq_rs, a_lane[r] + dl, base + t * (2 * kAPart) + part * kAPart));
          }
        } else {
          const unsigned so = rec_seg + t * (unsigned)(HoWo * RS * 16);
          R.off = __builtin_bit_cast(uint4, dcn_buf_b128(rec_rs, rec_lane, so));
          R.wy = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 16, so));
          R.wx = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 32, so));
          if constexpr (MASK) R.wm = __builtin_bit_cast(f32x4, dcn_buf_b128(rec_rs, rec_lane + 48, so));
        }
      };
      auto commit_weights = [&](int buf, const Regs &R) {
        if constexpr (PRODUCER) {
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int idx = tid + r * kProducers;
            const bool real = (idx >> 5) < n_o16;
#pragma unroll
            for (int part = 0; part < PARTS; ++part)
              *reinterpret_cast<f32x4 *>(As + (buf * PARTS + part) * kAPart + idx * 16) =
                  real ? R.a[part][r] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      };
      auto load_plane = [&]() {  // as in plane_role: x[tile_b, 16 channels of chunk c16] -> the LDS quad planes
        // (channels past the group's end must read as ZERO here: they would otherwise enter grad_offset through a
        // non-zero colgrad of padded weight rows... which are zero; keep both sides zero)
        const float *xb = p.x + ((long long)tile_b * p.C_total + p.c_base) * HW;
        dcn_plane_copy<kPlaneRounds, true>(xb, HW, p.Cg, c16 * kChunk, plane, qstride,
                                           __builtin_amdgcn_readfirstlane(wtid >> 6), kOffThreads / 64, dcn_plane_units(HW),
                                           wtid & 63);
      };
      // one stage on the consumer side: colgrad block, derivative dot products, tap accumulators
      auto consume = [&](int j, int buf, const Regs &R) {
        if constexpr (!PRODUCER) {
          const unsigned char *A = As + buf * PARTS * kAPart + kg * 256 + px16 * 16;  // lane's row c = px16 (reused name)
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#ifdef KGDET_OFF_ABL_NOMFMA
          acc0 = f32x4{R.wy[0], R.wy[1], R.wx[0], R.wx[1]};
#else
#pragma unroll
          for (int ks = 0; ks < kMaxKs; ++ks) {
            if (ks < n_ks) {
              const bf16x8v ah = *reinterpret_cast<const bf16x8v *>(A + ks * 1024);
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh[ks], acc0, 0, 0, 0);
              if constexpr (PARTS == 2) {
                const bf16x8v al = *reinterpret_cast<const bf16x8v *>(A + kAPart + ks * 1024);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh[ks], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl[ks], acc2, 0, 0, 0);
              }
            }
          }
#endif
          f32x4 cg = acc0;
          if constexpr (PARTS == 2) { cg[0] += acc1[0] + acc2[0]; cg[1] += acc1[1] + acc2[1]; cg[2] += acc1[2] + acc2[2]; cg[3] += acc1[3] + acc2[3]; }
          // cg[r] = colgrad of channel 4 * kg + r for pixel px16: the x quad kg of the four corners
#ifdef KGDET_OFF_ABL_NOTAIL
          if (cg[0] != 1234.5f) return;
#endif
          // (requesting the corner quads before the MFMAs -- they do not depend on the column gradient -- spills at the
          //  168-register budget: 64 registers of grad_out fragments, three record sets)
          const unsigned o[4] = {R.off.x, R.off.y, R.off.z, R.off.w};
          float gy = 0.f, gx = 0.f, gm = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(plane_kg + o[e]);
            const float d = cg[0] * v[0] + cg[1] * v[1] + cg[2] * v[2] + cg[3] * v[3];
            gy += R.wy[e] * d;
            gx += R.wx[e] * d;
            if constexpr (MASK) gm += R.wm[e] * d;
          }
          // sum over the four quad lanes of a pixel (lanes px16 + 16 kg): x + x[lane ^ 16], then + [lane ^ 32], as ONE vector
          // instruction each -- gfx950's v_permlane16_swap / v_permlane32_swap on two copies of the value (every lane ends up
          // with the sum of its row pair / its halves; a + b is commutative, so the bits equal the ds_bpermute shuffles' that
          // rounds 1-3 used -- four LDS-crossbar round trips per stage at the end of a dependent chain)
          auto quad_sum = [](float x) __attribute__((always_inline)) {
            const unsigned u = __float_as_uint(x);
            const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            const float s16 = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
            const unsigned w = __float_as_uint(s16);
            const auto r32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
            return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
          };
          gy = quad_sum(gy);
          gx = quad_sum(gx);
          if constexpr (MASK) gm = quad_sum(gm);
          if (kg == 0) {
            if constexpr (MASK) {
              f32x4 *dst = reinterpret_cast<f32x4 *>(offs_acc) + (size_t)(t0 + j) * kTileN + wave * 16 + px16;
              f32x4 cur4 = *dst;
              cur4[0] += gy;
              cur4[1] += gx;
              cur4[2] += gm;
              *dst = cur4;
            } else {
              float2 *dst = reinterpret_cast<float2 *>(offs_acc) + (size_t)(t0 + j) * kTileN + wave * 16 + px16;
              float2 cur2 = *dst;
              cur2.x += gy;
              cur2.y += gx;
              *dst = cur2;
            }
          }
        }
      };

      __syncthreads();  // the previous segment's readers of plane / A are done; offs_acc zeroing is visible
      if (!prefetched) {
        issue(0, R0);
        issue(1, R1);
        issue(2, R2);
      }
      prefetched = has_next;
      load_plane();
      commit_weights(0, R0);
      __syncthreads();
      // stage j: producers move W^T stage j+1 into the other buffer and put stage j+3 in flight; consumers
      // multiply stage j (their record registers run the same three-deep pipeline)
      auto body = [&](int j, Regs &RI, Regs &RC, Regs &RN) {
        const int buf = j & 1;
        if constexpr (PRODUCER) {
          if (j + 1 < n) commit_weights(buf ^ 1, RC);
          issue(j + 3, RI);
        } else {
          if (j < n) consume(j, buf, RI);  // RI still holds stage j for the consumers ...
          issue(j + 3, RI);                // ... until here
        }
        (void)RN;
        __syncthreads();
      };
      for (int j = 0; j < n; j += 6) {
        body(j, R0, R1, R2);
        body(j + 1, R1, R2, R0);
        body(j + 2, R2, R0, R1);
        if (j + 3 < n) {
          body(j + 3, R0, R1, R2);
          body(j + 4, R1, R2, R0);
          body(j + 5, R2, R0, R1);
        }
      }
      s += n;
      ++c16;
      t0 = 0;
    }

    // the range's tap sums: straight to grad_offset if the range is the whole tile, else to a slab
    __syncthreads();
    {
      const bool whole = s_begin == 0 && s_end == cpt && p.sum_count <= 1;
      float *slab = slabs + ((long long)g * grp.slots + slot) * (size_t)(max_K * kTileN * ACC);
      for (int i = wtid; i < K * kTileN; i += kOffThreads) {
        const int t = i / kTileN, col = i - t * kTileN;
        float vy, vx, vm = 0.f;
        if constexpr (MASK) {
          const f32x4 v = reinterpret_cast<const f32x4 *>(offs_acc)[i];
          vy = v[0]; vx = v[1]; vm = v[2];
        } else {
          const float2 v = reinterpret_cast<const float2 *>(offs_acc)[i];
          vy = v.x; vx = v.y;
        }
        if (whole) {
          const int px = tile_px0 + col;
          if (px < HoWo) {
            float *dst = p.goff + ((long long)(tile_b * p.DG + p.dgi) * 2 * K + 2 * t) * HoWo + px;
            dst[0] = vy;
            dst[HoWo] = vx;
            if constexpr (MASK) p.gmask[((long long)(tile_b * p.DG + p.dgi) * K + t) * HoWo + px] = vm;
          }
        } else if constexpr (MASK) {
          reinterpret_cast<f32x4 *>(slab)[i] = f32x4{vy, vx, vm, 0.f};
        } else {
          reinterpret_cast<float2 *>(slab)[i] = float2{vy, vx};
        }
      }
    }
    ++slot;
    cur += s_end - s_begin;
  }
}

template <int PARTS>
__global__ __launch_bounds__(kOffThreads, 1) void dcn_bwd_offset_plane(const DcnFwdGroup grp, float *__restrict__ slabs,
                                                                       int max_K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) offset_role<PARTS, true, false>(grp, slabs, smem, max_K);
  else offset_role<PARTS, false, false>(grp, slabs, smem, max_K);
}

// modulated (v2) problems: grad_offset and grad_mask (static ranges only; split operands only)
__global__ __launch_bounds__(kOffThreads, 1) void dcn_bwd_offset_plane_masked(const DcnFwdGroup grp, float *__restrict__ slabs,
                                                                              int max_K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) offset_role<2, true, true>(grp, slabs, smem, max_K);
  else offset_role<2, false, true>(grp, slabs, smem, max_K);
}

// v2, static ranges: add the parts' slabs (float4 per (tap, pixel)) in order.  grid = (tiles, 8)
__global__ __launch_bounds__(256) void dcn_bwd_offset_plane_fixup_masked(const DcnFwdGroup grp, const float *__restrict__ slabs,
                                                                         int G, int max_K) {
  const int gtile = blockIdx.x;
  int pi = 0;
  while (pi + 1 < grp.n && gtile >= grp.tile_begin[pi + 1]) ++pi;
  const DcnProblem &p = grp.p[pi];
  if (p.kparts == 1) return;  // written directly
  const int tile = gtile - grp.tile_begin[pi];
  const size_t slab_floats = (size_t)max_K * kTileN * 4;
  const int tiles = p.n_ntiles * p.n_mtiles;
  const int tb = tile / p.tiles_per_image, px0 = (tile - tb * p.tiles_per_image) * kTileN;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < p.K * kTileN; i += gridDim.y * 256) {
    float sy = 0.f, sx = 0.f, sm = 0.f;
    for (int part = 0; part < p.kparts; ++part) {
      const int range = grp.range_begin[pi] + part * tiles + tile;
      const f32x4 v = reinterpret_cast<const f32x4 *>(slabs + (size_t)sk_block_of_slice(range, G) * grp.slots * slab_floats)[i];
      sy += v[0];
      sx += v[1];
      sm += v[2];
    }
    const int t = i / kTileN, px = px0 + (i - t * kTileN);
    if (px < p.HoWo) {
      float *dst = p.goff + ((long long)(tb * p.DG + p.dgi) * 2 * p.K + 2 * t) * p.HoWo + px;
      dst[0] = sy;
      dst[p.HoWo] = sx;
      p.gmask[((long long)(tb * p.DG + p.dgi) * p.K + t) * p.HoWo + px] = sm;
    }
  }
}

template __global__ void dcn_bwd_offset_plane<1>(const DcnFwdGroup grp, float *__restrict__ slabs, int max_K);
template __global__ void dcn_bwd_offset_plane<2>(const DcnFwdGroup grp, float *__restrict__ slabs, int max_K);

// Split tiles: add the slabs of a tile's ranges, slices in order.  grid = (tiles, 8): block (tile, y) handles every
// 8th group of 256 (tap, pixel) elements; the contributing slabs are listed once per block.
__global__ __launch_bounds__(256) void dcn_bwd_offset_plane_fixup(const DcnFwdGroup grp, const float *__restrict__ slabs,
                                                                  int G, int max_K) {
  __shared__ long long contrib[64];
  __shared__ int n_contrib;
  __shared__ long long contrib_next;
  const int gtile = blockIdx.x;
  int pi = 0;
  while (pi + 1 < grp.n && gtile >= grp.tile_begin[pi + 1]) ++pi;
  const DcnProblem &p = grp.p[pi];
  const int tile = gtile - grp.tile_begin[pi];
  const long long total = grp.unit_begin[grp.n];
  const size_t slab_floats0 = (size_t)max_K * kTileN * 2;
  if (grp.static_ranges) {   // one workgroup per (part, tile) range: add the parts' slabs in order
    const bool grouped = p.sum_count > 1;      // (sum group: the leader's block adds every member's parts, DcnProblem::sum_count)
    if (grouped ? p.sum_members[0] != pi : p.kparts == 1) return;  // (no group, one part: written directly)
    const int tiles_ = p.n_ntiles * p.n_mtiles;
    const int tb_ = tile / p.tiles_per_image, px0_ = (tile - tb_ * p.tiles_per_image) * kTileN;
    for (int e = 0; e < 4; ++e) {
      const int i = (blockIdx.y + e * gridDim.y) * 256 + threadIdx.x;
      if (i >= p.K * kTileN) continue;
      float sy = 0.f, sx = 0.f;
      for (int m = 0; m < (grouped ? p.sum_count : 1); ++m) {
        const int mi = grouped ? p.sum_members[m] : pi;
        for (int part = 0; part < grp.p[mi].kparts; ++part) {
          const int range = grp.range_begin[mi] + part * tiles_ + tile;
          const float2 v = reinterpret_cast<const float2 *>(slabs + (size_t)sk_block_of_slice(range, G) * grp.slots * slab_floats0)[i];
          sy += v.x;
          sx += v.y;
        }
      }
      const int t = i / kTileN, px = px0_ + (i - t * kTileN);
      if (px < p.HoWo) {
        float *dst = p.goff + ((long long)(tb_ * p.DG + p.dgi) * 2 * p.K + 2 * t) * p.HoWo + px;
        dst[0] = sy;
        dst[p.HoWo] = sx;
      }
    }
    return;
  }
  const long long tb = dcn_range_first_unit(grp, pi, 0, tile), te = tb + p.chunks_per_tile;
  long long g0 = tb * G / total;
  while (unit_begin(g0 + 1, total, G) <= tb) ++g0;
  while (unit_begin(g0, total, G) > tb) --g0;
  if (unit_begin(g0, total, G) <= tb && unit_begin(g0 + 1, total, G) >= te) return;  // written directly
  const size_t slab_floats = (size_t)max_K * kTileN * 2;
  const int range = grp.range_begin[pi] + tile;
  const int tile_b = tile / p.tiles_per_image, tile_px0 = (tile - tile_b * p.tiles_per_image) * kTileN;
  constexpr int kMaxElems = 4;  // (tap, pixel) elements per thread: K * 128 / (256 * gridDim.y), K <= 64
  float sy[kMaxElems] = {0.f, 0.f, 0.f, 0.f}, sx[kMaxElems] = {0.f, 0.f, 0.f, 0.f};
  long long g = g0;
  bool more = true;
  while (more) {  // contributors in batches of 64, slices in order
    __syncthreads();
    if (threadIdx.x == 0) {
      int n = 0;
      for (; g < G && n < 64; ++g) {
        const long long b0 = unit_begin(g, total, G);
        if (b0 >= te) break;
        if (unit_begin(g + 1, total, G) == b0) continue;
        contrib[n++] = ((long long)sk_block_of_slice((int)g, G) * grp.slots + (range - dcn_unit_pos(grp, b0).range)) *
                       (long long)slab_floats;
      }
      n_contrib = n;
      contrib_next = (g < G && unit_begin(g, total, G) < te) ? g : -1;
    }
    __syncthreads();
    const int n = n_contrib;
    g = contrib_next;
    more = g >= 0;
#pragma unroll
    for (int e = 0; e < kMaxElems; ++e) {
      const int i = (blockIdx.y + e * gridDim.y) * 256 + threadIdx.x;
      if (i < p.K * kTileN)
        for (int q = 0; q < n; ++q) {
          const float2 v = reinterpret_cast<const float2 *>(slabs + contrib[q])[i];
          sy[e] += v.x;
          sx[e] += v.y;
        }
    }
  }
#pragma unroll
  for (int e = 0; e < kMaxElems; ++e) {
    const int i = (blockIdx.y + e * gridDim.y) * 256 + threadIdx.x;
    if (i >= p.K * kTileN) continue;
    const int t = i / kTileN, px = tile_px0 + (i - t * kTileN);
    if (px < p.HoWo) {
      float *dst = p.goff + ((long long)(tile_b * p.DG + p.dgi) * 2 * p.K + 2 * t) * p.HoWo + px;
      dst[0] = sy[e];
      dst[p.HoWo] = sx[e];
    }
  }
}

}  // namespace kgdet
