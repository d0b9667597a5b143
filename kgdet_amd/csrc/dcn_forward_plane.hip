// Deformable convolution forward, "plane" variant for gfx950: feature planes resident in LDS, bilinear
// gathers served by LDS, bf16 MFMA with an fp32-accurate hi/lo operand split, producer / consumer waves.
//
// Same reference path as dcn_forward.hip (deformable_im2col + addmm_,
// mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:190-276, 570-632; deform_conv_cuda.cpp:221-245, 534-563)
// and the same GEMM view / stream-K split.  What differs from the exact-fp32 kernel:
//   * MFMA rate: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate.  Every fp32 operand v is split into
//     hi = bf16(v), lo = bf16(v - hi) and the product is taken as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with
//     v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 3 MFMAs of 32 cycles replace 8 of 64; the dropped
//     a_lo*b_lo term and the representation residuals are <= 2^-16 relative per product (against the f64
//     oracle: ~1e-6 of the output scale, tests/test_gpu_dcn.py).  PARTS = 1 keeps only the hi parts:
//     plain bf16 operands for autocast inference.
//   * gathers: a 16-channel slice of ONE image is copied into LDS as four quad planes [quad][pixel][4 channels]
//     (dcn_common.h; 67 KB at 25x42) and reused by all K taps of that channel chunk, so a bilinear corner of 4
//     channels is one ds_read_b128 whose address is the record's offset + an immediate; the reduction runs
//     chunk-major / tap-minor and pixel tiles never straddle images.
//   * sampling geometry is not recomputed per channel chunk: dcn_build_taps writes one 32-byte record per
//     (image, tap, output pixel) -- four LDS byte offsets and four bilinear weights -- and the kernel only
//     loads it (buffer loads: SGPR resource + one 32-bit lane offset + a scalar offset).  Beside MFMA waves every vector
//     instruction of a producer is expensive (one issue port per SIMD), so instructions per sample are what matters.
//   * 16 waves of 128 registers, two roles: waves 0-7 are CONSUMERS (wave w owns rows [32 w, 32 w + 32) x 128 columns of
//     the 256 x 128 accumulator tile, reads the B fragments from LDS, takes its A (weight) fragments straight from the weight
//     image in L2 -- 16 bytes per lane and fragment, two stages ahead in registers -- and issues the MFMAs), waves 8-15 are
//     PRODUCERS in four wave pairs at s_setprio 0 (a thread samples all 16 channels of one pixel for ONE of a group's four
//     stages; round 2 had two pairs doing two stages each at priority 2: the producers were the critical path, LDS gather
//     latency).  One barrier per group of four stages hands the double-buffered groups over.
//   * KGDET_DCN_BF16 (PARTS = 1): the LDS plane holds bf16, eight channels of a corner per ds_read_b128.
//   * dcn_fwd_gather (plane_role MODE 2): maps beyond the LDS plane -- no plane, the corners are buffer loads from a
//     pixel-major copy of x.
//
// Operand images (the lane-linear fragment order a wave reads with one 16-byte load per lane):
//   A stage (tap t, channel chunk c16, 256 output channels): [part][khalf][o 256][8 bf16]   8 KB / part
//   B stage (same reduction slice, 128 pixels):              [part][khalf][px 128][8 bf16]  4 KB / part
//   lane l of a wave reads row/col (l & 31) of k-half (l >> 5) as one 16-byte ds_read_b128; reduction
//   element k = khalf*8 + j is channel c16*16 + k for both operands.
#include "dcn_plane.h"

namespace kgdet {

template <int PARTS>
__global__ __launch_bounds__(kRoleThreads, 1) void dcn_fwd_plane(const DcnFwdGroup grp, float *__restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#if KGDET_PLANE_F16
  // the one-product kernel's fp16 plane: conversions beyond 65504 saturate instead of producing inf (MODE.FP16_OVFL, see
  // csrc/conv1x1.hip f16_saturate_on)
  if constexpr (PARTS == 1) __builtin_amdgcn_s_setreg(1 /*HW_REG_MODE*/ | (23 << 6) | (0 << 11), 1u);
#endif
  if (threadIdx.x >= kThreads) plane_role<PARTS, true, 0>(grp, slabs, smem);
  else plane_role<PARTS, false, 0>(grp, slabs, smem);
}

template __global__ void dcn_fwd_plane<1>(const DcnFwdGroup grp, float *__restrict__ slabs);
template __global__ void dcn_fwd_plane<2>(const DcnFwdGroup grp, float *__restrict__ slabs);


// Maps beyond the LDS plane (config 5's stride-8 / stride-16 levels): the same roles, the producers' corner reads are
// buffer loads from a pixel-major copy of x (plane_role MODE 2, dcn_plane.h).  LDS = the B stages only.
template <int PARTS>
__global__ __launch_bounds__(kRoleThreads, 1) void dcn_fwd_gather(const DcnFwdGroup grp, float *__restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x >= kThreads) plane_role<PARTS, true, 2>(grp, slabs, smem);
  else plane_role<PARTS, false, 2>(grp, slabs, smem);
}
template __global__ void dcn_fwd_gather<1>(const DcnFwdGroup grp, float *__restrict__ slabs);
template __global__ void dcn_fwd_gather<2>(const DcnFwdGroup grp, float *__restrict__ slabs);

// x [n][C][P] (image stride src_image_stride floats) -> [n][P][C]: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void dcn_to_pixel_major(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                          long long P, long long src_image_stride) {
  __shared__ float tile[32][33];
  const long long p0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32, n = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float *s = src + n * src_image_stride;
  float *d = dst + (long long)n * P * C;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const long long px = p0 + tx;
    tile[r][tx] = (c < C && px < P) ? s[(long long)c * P + px] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long long px = p0 + r;
    const int c = c0 + tx;
    if (px < P && c < C) d[px * C + c] = tile[tx][r];
  }
}

int dcn_fwd_plane_threads() { return kRoleThreads; }
int dcn_plane_wave_layout() { return KGDET_PLANE_WAVES42 ? 0 : 1; }   // DcnFwdGroup::wave_layout of plane_role's slabs

#ifdef KGDET_PLANE_TRACE
}  // namespace kgdet
extern "C" int kgdet_debug_read_plane_trace(unsigned long long *out) {   // the forward kernel's copy
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kgdet::g_plane_trace), sizeof(unsigned long long) * 256 * 16 * 10);
}
namespace kgdet {
#endif

// Tap records of every problem of a group that owns its table (p.build_taps): one thread per
// (image, deformable group, tap, output pixel).  grid = (blocks over the largest table, problems).
// Mirrors deformable_im2col_bilinear + the (-1,H)x(-1,W) guard (deform_conv_cuda_kernel.cu:84-114, :228);
// modulated (v2) problems fold the mask into the weights (:570-632).
__global__ __launch_bounds__(256) void dcn_build_taps(const DcnFwdGroup grp) {
  const DcnProblem &p = grp.p[blockIdx.y];
  if (p.build_xblk) dcn_block_x_body(p, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), (int)gridDim.x * 4);
  if (!p.build_taps) return;
  const long long n_rec = (long long)p.N * p.DG * p.K * p.HoWo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_rec; i += (long long)gridDim.x * 256) {
    const int hw = (int)(i % p.HoWo);
    const int t = (int)((i / p.HoWo) % p.K);
    const int bd = (int)(i / ((long long)p.HoWo * p.K));  // b * DG + dgi
    const int oy = hw / p.Wo, ox = hw - oy * p.Wo;
    const int ti = t / p.kw, tj = t - ti * p.kw;
    const long long ob = ((long long)bd * 2 * p.K + 2 * t) * p.HoWo + hw;
    const float y = (float)(oy * p.sh - p.ph + ti * p.dh) + p.offset[ob];
    const float x = (float)(ox * p.sw - p.pw + tj * p.dw) + p.offset[ob + p.HoWo];
    const float m = p.mask ? p.mask[((long long)bd * p.K + t) * p.HoWo + hw] : 1.0f;
    Tap tap;
    TapGeom geom;
    make_tap(y, x, p.H, p.W, true, m, tap, geom);
    DcnTapRec r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = tap.o[e];
      r.off[e] = grp.gather_mode ? (unsigned)q * (unsigned)(p.C_total * 4)           // rows of the pixel-major image
                                 : (unsigned)dcn_plane_offset(q);
      r.w[e] = tap.w[e];
    }
    const_cast<DcnTapRec *>(p.taps)[i] = r;
  }
}

// LDS of dcn_fwd_plane: the four quad planes at their fixed stride (whatever the image size: the quad is an immediate
// offset of the corner reads) + two groups of B stages
// two group buffers (+ the extra stage's slot of the KGDET_PLANE_STREAM variants)
size_t dcn_fwd_plane_fixed_lds_bytes(int parts) { return (size_t)(2 * kGroupTaps + (KGDET_PLANE_STREAM ? 1 : 0)) * parts * kBPart; }
size_t dcn_fwd_plane_lds_bytes(int parts, int HW) {
  return HW <= kPlaneMaxHW ? dcn_fwd_plane_fixed_lds_bytes(parts) + (size_t)4 * kPlaneQuadStride : (size_t)1 << 30;
}

// ----------------------------------------------------------------------------------------------
// Weight packing, all three images in one pass over the weights.  [O, Cg, K] (one group) ->
//   wpk[t][c pad16][o pad256]                       fp32  (exact forward kernel, dcn_forward.hip)
//   wpt[t][o pad16][c pad256]                       fp32  (backward-input, dcn_backward*.hip)
//   wq [mt][c16][t][part (hi, lo)][khalf][o 256][8] bf16  (plane forward kernel; hi = bf16(w), lo = bf16(w - hi))
//   wqt[ct][o16][t][part (hi, lo)][khalf][c 256][8] bf16  (the same with the roles of c and o swapped: grad_input)
// all zero padded.  One workgroup per (8 channels = one k-half, 32 output channels): each wave reads
// whole runs of K contiguous floats into an LDS tile [8][33][K+1]; the three images are written from it
// in runs of 128 B (wpk), 32 B (wpt) and 512 B (wq).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_weight_all_body(const float *__restrict__ w, float *__restrict__ wpk,
                                                            float *__restrict__ wpt, void *__restrict__ wq,
                                                            void *__restrict__ wqt, int Og, int Cg, int K,
                                                            int Cg_pad, int Og_pad, int Og_pad16, int Cg_pad256, float *tile) {
  const int c8 = blockIdx.x, o0 = blockIdx.y * 32, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int ld = K + 1;
  for (int t0 = 0; t0 < K; t0 += 64) {
    const int t = t0 + lane;
    for (int r0 = wave; r0 < 256; r0 += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // 8 independent (clamped, unconditional) loads in flight per lane
        const int r = r0 + 4 * u, cc = r >> 5, oo = r & 31;
        const int c = min(c8 * 8 + cc, Cg - 1), o = min(o0 + oo, Og - 1);
        v[u] = w[((long long)o * Cg + c) * K + min(t, K - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 4 * u, cc = r >> 5, oo = r & 31;
        const bool real = (c8 * 8 + cc < Cg) && (o0 + oo < Og);
        if (t < K) tile[(cc * 33 + oo) * ld + t] = real ? v[u] : 0.0f;
      }
    }
  }
  __syncthreads();
  const int cbase = c8 * 8;
  if (cbase < Cg_pad && o0 < Og_pad) {
    const int oo = tid & 31, cc = tid >> 5;
    if (wpk)
      for (int t = 0; t < K; ++t)
        wpk[((long long)t * Cg_pad + cbase + cc) * Og_pad + o0 + oo] = tile[(cc * 33 + oo) * ld + t];
    if (wq) {
      const int n_c16 = Cg_pad / kChunk;
      const int c16 = c8 >> 1, khalf = c8 & 1;
      const int mt = o0 / kTileM, o_in = o0 % kTileM;
      for (int e = tid; e < K * 64; e += 256) {
        const int o2 = e & 31, part = (e >> 5) & 1, t = e >> 6;
        bf16x8 v;
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
          const float f = tile[(c2 * 33 + o2) * ld + t];
          const __bf16 hi = (__bf16)f;
          v[c2] = part == 0 ? hi : (__bf16)(f - (float)hi);
        }
        const size_t stage = (size_t)((mt * n_c16 + c16) * K + t) * (2 * kAPart);
        unsigned char *dst = reinterpret_cast<unsigned char *>(wq) + stage + part * kAPart + khalf * (kTileM * 16) +
                             (o_in + o2) * 16;
        *reinterpret_cast<bf16x8 *>(dst) = v;
      }
    }
  }
  if (o0 < Og_pad16) {
    const int cc = tid & 7, oo = tid >> 3;
    if (wpt && o0 + oo < Og_pad16)
      for (int t = 0; t < K; ++t)
        wpt[((long long)t * Og_pad16 + o0 + oo) * Cg_pad256 + cbase + cc] = tile[(cc * 33 + oo) * ld + t];
    if (wqt) {  // transposed operand image: rows = input channels, reduction = output channels
      const int n_o16 = Og_pad16 / kChunk;
      for (int e = tid; e < K * 64; e += 256) {
        const int og = e & 3, c2 = (e >> 2) & 7, part = (e >> 5) & 1, t = e >> 6;
        const int o = o0 + og * 8, c = cbase + c2;
        if (o >= Og_pad16) continue;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = tile[(c2 * 33 + og * 8 + j) * ld + t];
          const __bf16 hi = (__bf16)f;
          v[j] = part == 0 ? hi : (__bf16)(f - (float)hi);
        }
        const size_t stage = (size_t)(((c / kTileM) * n_o16 + o / kChunk) * K + t) * (2 * kAPart);
        unsigned char *dst = reinterpret_cast<unsigned char *>(wqt) + stage + part * kAPart +
                             ((o % kChunk) / 8) * (kTileM * 16) + (c % kTileM) * 16;
        *reinterpret_cast<bf16x8 *>(dst) = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void dcn_pack_weight_all(const float *__restrict__ w, float *__restrict__ wpk,
                                                            float *__restrict__ wpt, void *__restrict__ wq,
                                                            void *__restrict__ wqt, int Og, int Cg, int K,
                                                            int Cg_pad, int Og_pad, int Og_pad16, int Cg_pad256) {
  extern __shared__ float tile[];  // [(cc * 33 + oo)][K+1]
  pack_weight_all_body(w, wpk, wpt, wq, wqt, Og, Cg, K, Cg_pad, Og_pad, Og_pad16, Cg_pad256, tile);
}

// several weights in one launch (the six of a KGDet head stage: training re-packs them every step): blockIdx.z = weight
__global__ __launch_bounds__(256) void dcn_pack_weight_all_multi(const DcnPackGroup grp) {
  extern __shared__ float tile[];
  const DcnPackOne &e = grp.e[blockIdx.z];
  if ((int)blockIdx.x >= e.Cg_pad256 / 8 || (int)blockIdx.y >= e.Og_pad / 32) return;
  pack_weight_all_body(e.w, e.wpk, e.wpt, e.wq, e.wqt, e.Og, e.Cg, e.K, e.Cg_pad, e.Og_pad, e.Og_pad16, e.Cg_pad256, tile);
}

}  // namespace kgdet
