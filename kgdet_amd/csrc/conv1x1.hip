// 1x1 and 3x3 convolutions (NCHW, fp32 in / fp32 out) as bf16 hi/lo-split MFMA GEMMs for gfx950.
//
// The ResNet-50 bottleneck (mmdet/models/backbones/resnet.py:142-186, 231-262) is two 1x1 convolutions around a
// 3x3; with the reference's fp32 arithmetic MIOpen runs them at 60-110 TFLOP/s (MI355X fp32 MFMA peak 157; its
// strided fp32 kernels at 15-20).  A 1x1 convolution at these shapes has ~50 flop per byte of fp32 activation traffic,
// i.e. it sits at ~400 TFLOP/s on the HBM roofline, and a 3x3 is nine times denser: the matrix pipe, not memory, is
// what fp32 MFMA leaves on the table.  Same arithmetic as the deformable kernels (dcn_forward_plane.hip): every fp32
// operand v = hi + lo with hi = bf16(v), lo = bf16(v - hi); a product is a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- ~5e-6 of the output scale (fp32 kernels: ~7e-7) at a third of
// the bf16 rate.
//
//   forward      y[b] (O x HoWo) = W (O x C*taps) . patches(x[b])                conv_nn<1 | 9>, A = packed W
//   grad_input   gx[b] (C x HW)  = W^T, taps mirrored . patches(gy[b])           conv_nn<1 | 9>, A = packed W^T (stride 1)
//   grad_weight  gW (O x C*taps) = sum_b gy[b] (O x HW) . patches(x[b])^T        conv_nt8<1 | 9> (maps with W or H*W % 4 != 0: zero-padded copies)
//
// conv_nn: 128 x 128 output tile, 512 threads = 8 waves as 2 (M) x 4 (N); reduction in stages of (16 channels, tap).
//   A stage = 8 KB of the pre-split weight image [part][khalf][128 rows][8 bf16] (one 16-byte load per thread);
//   B stage = 16 activation rows x 128 pixels: a thread owns (pixel, 4 channels), dword loads coalesced along pixels
//   (tap shift = address offset, tap validity = bit of a per-thread mask, stride 2 = input index mapping), split on the
//   fly, 8-byte LDS writes.  Four stages of loads in flight; the main loop has no guarded loads (counted vmcnt); one
//   barrier per stage; XCD-contiguous tile order; K-split + deterministic sum below 200 tiles; optional inference
//   epilogue (bias, residual, ReLU) in the store.
// conv_nt8: both operands are activations with the reduction (pixels) contiguous: a thread loads 4 consecutive pixels
//   of one row of each operand (3x3 column shift: two aligned loads + static selection).  The pixel range is cut into
//   chunks, one workgroup per (tile, chunk) writes a partial; conv1x1_sum / conv3x3_wsum add them in fixed order.
#include "common.h"

namespace kgdet {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kTM = 128, kTN = 128, kTK = 16;
constexpr int kPart = 2 * kTM * 16;          // bytes of one part of one operand stage: [khalf][128][8 bf16]
constexpr int kStage = 2 * kPart;            // hi + lo

// hi = bf16(v), lo = bf16(v - hi) for a PAIR of values: one v_cvt_pk_bf16_f32 for the two hi parts, their float values
// back by a shift and a mask of that dword, one packed subtraction, one v_cvt_pk_bf16_f32 for the lo parts -- 5
// instructions per pair where scalar conversions take 7-8.  These kernels split operands on the fly beside their MFMAs,
// and a SIMD issues both through one port (tools/microbench/mfma_valu.hip): every VALU instruction costs MFMA time.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned &hi, unsigned &lo) {
  const f32x2 v = {v0, v1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - hf, bf16x2));
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned a, b;
    split_pair(v[2 * i], v[2 * i + 1], a, b);
    h[i] = a;
    l[i] = b;
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// FORWARD operands (activations x weights) can be split into two fp16 parts instead of two bf16 parts: 22 mantissa bits
// instead of 16 at the same MFMA rate (v_mfma_f32_32x32x16_f16) and the same five conversion instructions per pair
// (v_cvt_pkrtz_f16_f32: the truncation is exact for the hi part's remainder and leaves <= 2^-21 after the lo part).
// Round 3 (tests/golden/make_step_golden.py): against a float64 evaluation of the training step the bf16 split left
// 1.4e-3 on gradient norms and 1.2e-2 on single gradient elements of the classification tower -- all of it from the
// FORWARD features' 5e-6..9e-6 (tools/step_vs_f64.py KGDET_EXP=fwd_exact: 7e-6 with an fp32 forward and the split backward).
// Activations and weights sit inside fp16's range; gradients do not (1e-8 and below), so grad_input / grad_weight
// operands stay bf16 (8 exponent bits).  Values beyond 65504 saturate, below 6e-8 vanish.
// fp16's normal range ends at 6.1e-5: the lo part of a weight of 0.03 (7e-6) would be subnormal and keep ~7 of its 11 bits.
// The pack kernels therefore store fp16-format weights scaled by kF16WeightScale = 2^8 (exact; weights up to 255 in
// magnitude) and the fp16 kernels multiply their accumulators by 2^-8 before the epilogue (exact as well).
constexpr float kF16WeightScale = 256.0f;
// "Values beyond 65504 saturate" is made true by the MODE register's FP16_OVFL bit (bit 23: an overflowed FP16 VALU result
// is clamped to +/-MAX_FP16, true infinities stay): set once at the top of every kernel that produces fp16 parts, no
// instruction per element.  Without it v_cvt_pk_f16_f32 rounds |v| > 65504 to inf, hi = inf, lo = v - inf = -inf, and the
// MFMA sum inf + (-inf) is NaN for the whole output channel (round-3 ADVICE; a folded BatchNorm scale gamma / sqrt(var + eps)
// of a channel with a tiny running variance reaches that through w * s * 2^8).  With it: hi = 65504, lo = f16(v - 65504):
// an 11-bit representation up to 131008 (measured on the GPU: 2e-4 of the output scale) and clamped beyond -- finite; the
// fp32-class envelope still ends at 65504 (tests/test_gpu_backbone.py::test_fp16_forward_parts_saturate_instead_of_nan).
__device__ __forceinline__ void f16_saturate_on() {
  __builtin_amdgcn_s_setreg(1 /*HW_REG_MODE*/ | (23 << 6) /*offset*/ | (0 << 11) /*size - 1*/, 1u);
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ void split_pair_t(float v0, float v1, unsigned &hi, unsigned &lo) {
  if constexpr (!F16) {
    split_pair(v0, v1, hi, lo);
  } else {
    // both parts rounded to nearest even (v_cvt_pk_f16_f32 on gfx950); the remainder v - hi is exact in fp32.  Truncated
    // parts (v_cvt_pkrtz_f16_f32) have the sign of v, so the dropped lo * lo term always had the sign of the product: a
    // bias of -6e-8 per layer that grew to -1.4e-6 (both truncated) / -3e-7 (hi truncated) of the rms over the backbone
    const f32x2 v = {v0, v1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    const f32x2 hf = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2);
    const f32x2 r = v - hf;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  }
}
template <bool F16>
__device__ __forceinline__ void split8_t(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned a, b;
    split_pair_t<F16>(v[2 * i], v[2 * i + 1], a, b);
    h[i] = a;
    l[i] = b;
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
// (bf16x8 is the 16-byte container of either format)
template <bool F16>
__device__ __forceinline__ f32x16 mfma_t(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// one stage of MFMAs: wave (wm, wn) multiplies its 64 x 64 block; operands from LDS stage buffers
__device__ __forceinline__ void mma_stage(const unsigned char *As, const unsigned char *Bs, int lane, int wm, int wn,
                                          f32x16 (&acc)[2][2]) {
  const unsigned char *A = As + (lane >> 5) * (kTM * 16) + (wm * 64 + (lane & 31)) * 16;
  const unsigned char *B = Bs + (lane >> 5) * (kTN * 16) + (wn * 64 + (lane & 31)) * 16;
  bf16x8 a[2][2], b[2][2];
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[part][i] = *reinterpret_cast<const bf16x8 *>(A + part * kPart + i * 32 * 16);
      b[part][i] = *reinterpret_cast<const bf16x8 *>(B + part * kPart + i * 32 * 16);
    }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {  // small terms first
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[1][ni], acc[mi][ni], 0, 0, 0);
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
    }
}

// workgroup b of G -> tile index: the tiles are cut into 8 contiguous runs, one per XCD (block b runs on XCD b % 8)
__device__ __forceinline__ int xcd_tile(int b, int tiles) {
  const int per = (tiles + 7) / 8;
  return (b & 7) * per + (b >> 3);
}

}  // namespace

// Operand image of a [O, C, T] weight (T = 1 or 9 taps): stages ordered (chunk of 16 reduction channels, tap),
// image[mt][k16 * T + t][part][khalf][128][8].
//   transpose = 0 (forward):    rows = O, reduction = C:  A[o][(c, t)] = w[o][c][t]
//   transpose = 1 (grad_input): rows = C, reduction = O:  A[c][(o, t)] = w[o][c][T - 1 - t]   (taps mirrored)
// gridDim.y == 2: block row 0 writes the forward image to img, row 1 the grad_input image to img_t (one launch per
// convolution and step instead of two).
__global__ __launch_bounds__(256) void conv1x1_pack(const float *__restrict__ w, int O, int C, int T, int transpose,
                                                    unsigned char *__restrict__ img, unsigned char *__restrict__ img_t,
                                                    int f16_forward) {
  if (gridDim.y == 2) {
    transpose = blockIdx.y;
    img = blockIdx.y ? img_t : img;
  }
  const bool f16 = f16_forward && !transpose;     // only the forward image (activations x weights) takes fp16 parts
  f16_saturate_on();
  const int M = transpose ? C : O, K = transpose ? O : C;
  const int k16s = (K + kTK - 1) / kTK;     // (a 1x1 weight's reduction may end inside a chunk: zeros up to its end)
  const long long total = (long long)((M + kTM - 1) / kTM) * k16s * T * 2 * kTM;   // (mt, k16, t, khalf, row)
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const int row = (int)(i % kTM);
    const int khalf = (int)((i / kTM) & 1);
    const long long st = i / (2 * kTM);           // stage index (mt, k16, t)
    const int t = (int)(st % T);
    const int k16 = (int)((st / T) % k16s), mt = (int)(st / ((long long)T * k16s));
    const int m = mt * kTM + row, k0 = k16 * kTK + khalf * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long o = transpose ? k0 + j : m, ch = transpose ? m : k0 + j;
      v[j] = (m < M && k0 + j < K) ? w[(o * C + ch) * T + (transpose ? T - 1 - t : t)] : 0.0f;
    }
    bf16x8 hi, lo;
    if (f16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= kF16WeightScale;
      split8_t<true>(v, hi, lo);
    } else {
      split8(v, hi, lo);
    }
    unsigned char *dst = img + st * kStage + khalf * (kTM * 16) + row * 16;
    *reinterpret_cast<bf16x8 *>(dst) = hi;
    *reinterpret_cast<bf16x8 *>(dst + kPart) = lo;
  }
}

// Both operand images of MANY weights in one launch (training re-packs every weight every step: 60 launches of a few
// microseconds each).  desc[i] = {w, img, img_t, (O << 32) | C, (T << 32) | first block, scale}; block b works on descriptor
// i with first_block[i] <= b < first_block[i + 1], one 256-item slice of each image.  scale (or 0): float [O], the images
// are those of w[o] * scale[o] -- a frozen-statistics BatchNorm behind the convolution folded into its weight
// (kgdet_amd/backbone.py _ConvBNActFold).
constexpr int kPackDescWords = 6;
__global__ __launch_bounds__(256) void conv1x1_pack_multi(const long long *__restrict__ desc, int n) {
  int lo = 0, hi = n - 1;   // uniform binary search
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)(desc[mid * kPackDescWords + 4] & 0xffffffffLL) <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long *d = desc + lo * kPackDescWords;
  const float *w = reinterpret_cast<const float *>(d[0]);
  const float *scale = reinterpret_cast<const float *>(d[5]);
  const int O = (int)(d[3] >> 32), C = (int)(d[3] & 0xffffffffLL), T = (int)((d[4] >> 32) & 0xffff);
  const bool f16_forward = (d[4] >> 62) & 1;     // forward image in fp16 parts
  f16_saturate_on();
  const long long i = (long long)((int)blockIdx.x - (int)(d[4] & 0xffffffffLL)) * 256 + threadIdx.x;
#pragma unroll
  for (int transpose = 0; transpose < 2; ++transpose) {
    unsigned char *img = reinterpret_cast<unsigned char *>(d[1 + transpose]);
    const int M = transpose ? C : O, K = transpose ? O : C;
    const int k16s = (K + kTK - 1) / kTK;
    const long long total = (long long)((M + kTM - 1) / kTM) * k16s * T * 2 * kTM;
    if (!transpose && T == 9) {
      // forward image of a 3x3 weight: a lane's row is an output channel, whose 8 channels x 9 taps are 72 CONTIGUOUS floats
      // (16-byte aligned: C % 16 == 0) -- one thread reads them once (18 float4) and writes the nine taps' items, instead of
      // nine threads each picking 8 floats 36 bytes apart out of the same 288 bytes
      if (i >= total / 9) continue;
      const int row = (int)(i % kTM), khalf = (int)((i / kTM) & 1);
      const long long s2 = i / (2 * kTM);             // (mt, k16)
      const int k16 = (int)(s2 % k16s), mt = (int)(s2 / k16s);
      const int m = mt * kTM + row, k0 = k16 * kTK + khalf * 8;
      float r[72];
      const f32x4 *src = reinterpret_cast<const f32x4 *>(w + ((long long)min(m, M - 1) * C + k0) * 9);
#pragma unroll
      for (int q = 0; q < 18; ++q) {
        const f32x4 u = src[q];
        r[4 * q] = u[0]; r[4 * q + 1] = u[1]; r[4 * q + 2] = u[2]; r[4 * q + 3] = u[3];
      }
      const float sc9 = (scale && m < M) ? scale[m] : 1.0f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = m < M ? r[j * 9 + t] * sc9 : 0.0f;
        bf16x8 hi8, lo8;
        if (f16_forward) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= kF16WeightScale;
          split8_t<true>(v, hi8, lo8);
        } else {
          split8(v, hi8, lo8);
        }
        unsigned char *dst = img + (s2 * 9 + t) * kStage + khalf * (kTM * 16) + row * 16;
        *reinterpret_cast<bf16x8 *>(dst) = hi8;
        *reinterpret_cast<bf16x8 *>(dst + kPart) = lo8;
      }
      continue;
    }
    if (i >= total) continue;
    const int row = (int)(i % kTM);
    const int khalf = (int)((i / kTM) & 1);
    const long long st = i / (2 * kTM);
    const int t = (int)(st % T);
    const int k16 = (int)((st / T) % k16s), mt = (int)(st / ((long long)T * k16s));
    const int m = mt * kTM + row, k0 = k16 * kTK + khalf * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long o = transpose ? k0 + j : m, ch = transpose ? m : k0 + j;
      v[j] = (m < M && k0 + j < K) ? w[(o * C + ch) * T + (transpose ? T - 1 - t : t)] : 0.0f;
      if (scale && m < M && k0 + j < K) v[j] *= scale[o];
    }
    bf16x8 hi8, lo8;
    if (f16_forward && !transpose) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= kF16WeightScale;
      split8_t<true>(v, hi8, lo8);
    } else {
      split8(v, hi8, lo8);
    }
    unsigned char *dst = img + st * kStage + khalf * (kTM * 16) + row * 16;
    *reinterpret_cast<bf16x8 *>(dst) = hi8;
    *reinterpret_cast<bf16x8 *>(dst + kPart) = lo8;
  }
}

// y[b][m][p] = sum_{k, t} A[m][(k, t)] * x[b][k][p + shift(t)]  (zero outside the image): a TAPS = 1 (1x1) or 9 (3x3,
// stride 1, padding 1) convolution as an implicit GEMM.  A as packed image, x [B, K, H*W], y [B, M, H*W].
// 512 threads: 8 waves as 2 (M) x 4 (N), 64 x 32 outputs each -- two waves per SIMD, so one wave's MFMAs cover the
// other's loads / conversions even when a problem has only ~1 tile per CU.  B stage: thread (pixel, k quarter) loads
// 4 channels of its pixel (dword loads, coalesced along pixels; the tap's shift is an address offset, its validity a
// bit of a per-thread mask computed once), splits, writes 8 bytes per part.  A stage: one 16-byte load per thread.
// ksplit > 1: workgroup (tile, part) reduces stages [part * per, ...) and writes y-shaped partial `part` of `y`
// (= a [ksplit][B, M, N] buffer); conv1x1_sum adds the parts.  Used when a problem has too few tiles for 256 CUs.
constexpr int kPF = 4;          // stages of global loads in flight per thread
constexpr int kNNThreads = 512;

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// NW = 4 | 5 waves along the pixels: tiles of 128 or 160 pixels (640 threads).  A CU finishes a tile at a fixed rate whatever
// shares it (see conv3x3_patch4), so what counts is the number of tiles the fullest CU draws: [2, 128, 100 x 168] is 264 tiles of
// 128 pixels (sixteen CUs draw two) but 210 of 160.
template <int TAPS, int NW = 4, bool F16 = false>
__global__ __launch_bounds__(128 * NW) void conv_nn(const unsigned char *__restrict__ img,
                                                      const float *__restrict__ x, float *__restrict__ y, int M, int K,
                                                      int H, int W, int n_mt, int n_nt, int tiles, int ksplit,
                                                      long long part_stride, int Hin, int Win, int stride,
                                                      const float *__restrict__ bias,
                                                      const float *__restrict__ residual, int relu,
    const float *__restrict__ gate = nullptr) {
  // bias [M] / residual [B, M, H, W] / relu: inference epilogue y = [relu](acc + bias[m] [+ residual]) (ksplit == 1)
  // H x W: the OUTPUT map; Hin x Win: the input map; stride 1 (Hin = H, Win = W) or 2 (H = ceil(Hin / 2), ...)
  constexpr int TN = 32 * NW, kPartB = 2 * TN * 16, kBuf = kStage + 2 * kPartB;   // B part: [khalf][TN][8 bf16]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kBuf];   // [buf][A (kStage) | B (hi, lo)]
  if constexpr (F16) f16_saturate_on();
  const int unit = xcd_tile(blockIdx.x, tiles * ksplit);
  if (unit >= tiles * ksplit) return;
  // unit order: the K parts and the m tiles of one (image, pixel tile) adjacent -> they share it through one L2
  const int part = unit % ksplit, tile = unit / ksplit;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int n_local = tid % TN, kq = tid / TN;   // pixel column, quarter of the stage's 16 channels
  const int N = H * W;
  const int S = TAPS * ((K + kTK - 1) / kTK), per = (S + ksplit - 1) / ksplit;
  const int s_begin = part * per, s_end = max(s_begin, min(S, s_begin + per));
  const int stages = s_end - s_begin;
  struct Regs {
    f32x4 a;
    float v[4];
    unsigned live;
  };
  const int mt = tile % n_mt, nt = (tile / n_mt) % n_nt, b = tile / (n_mt * n_nt);
  const int n0 = nt * TN;
  const int p = min(n0 + n_local, N - 1);   // columns past the end re-read the last one: never stored
  unsigned ok = 1u;                          // bit t: tap t of this pixel lies inside the image
  if (TAPS == 9) {
    const int h = (p / W) * stride, w = (p - (p / W) * W) * stride;
    ok = 0u;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
      ok |= (hh >= 0 && hh < Hin && ww >= 0 && ww < Win) ? (1u << t) : 0u;
    }
  }
  const int Nin = Hin * Win;
  const int pin = stride == 1 ? p : (p / W) * stride * Win + (p - (p / W) * W) * stride;   // input pixel of tap (0, 0)
  const float *xb = x + (long long)b * K * Nin + pin;
  const unsigned char *ai = img + (long long)mt * S * kStage + (tid & 511) * 16;   // (NW = 5: waves 8, 9 duplicate 0, 1 -- loads stay unconditional)

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  // gate (see the store below).  -DKGDET_GATE_EARLY (experiment, measured SLOWER: 11.35 against 11.19 ms per training step):
  // its 32 values per lane requested BEFORE the reduction loop, where their round trip hides behind the first stages -- but the
  // 32 live registers (96 -> 138) cost the gradient variants their second workgroup per CU
  float gv[2][16];
#ifdef KGDET_GATE_EARLY
  if (!F16 && gate) {     // (gradient kernels only: the forward variants are never gated and keep their register count)
    const int ng = min(n0 + wn * 32 + (lane & 31), N - 1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), M - 1);
        gv[mi][r] = gate[((long long)b * M + m) * N + ng];
      }
  }
#endif

  auto issue = [&](int s, Regs &R) {   // clamped: unconditional loads keep hipcc's vmcnt counting exact
    const int sc = s_begin + min(s, stages - 1);
    R.a = *reinterpret_cast<const f32x4 *>(ai + (long long)sc * kStage);
    const int c16 = sc / TAPS, t = sc - c16 * TAPS;
    int shift = 0;
    R.live = 1u;
    if (TAPS == 9) {
      R.live = (ok >> t) & 1u;
      shift = R.live ? (t / 3 - 1) * Win + (t % 3 - 1) : 0;
    }
    const float *xp = xb + (long long)(c16 * kTK + kq * 4) * Nin + shift;
    if (TAPS == 1 && (K & (kTK - 1)) && c16 == K / kTK) {
      // the last chunk of a reduction that is not a multiple of 16 (the head's 588-channel key-point maps): channels past the end
      // re-read the last one -- their weights are the image's zero padding (wave-uniform branch)
#pragma unroll
      for (int j = 0; j < 4; ++j) R.v[j] = xb[(long long)min(c16 * kTK + kq * 4 + j, K - 1) * Nin];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) R.v[j] = xp[(long long)j * Nin];
    }
  };
  auto commit = [&](int buf, const Regs &R) {
    unsigned char *As = smem + buf * kBuf, *Bs = As + kStage;
    *reinterpret_cast<f32x4 *>(As + (tid & 511) * 16) = R.a;
    uint2 hi, lo;
#ifdef KGDET_CONV_ABL_NOSPLIT   // ablation (timing only, wrong results): no conversion VALU
    hi.x = __float_as_uint(R.v[0]); hi.y = __float_as_uint(R.v[1]); lo.x = __float_as_uint(R.v[2]); lo.y = __float_as_uint(R.v[3]);
#else
    split_pair_t<F16>(R.live ? R.v[0] : 0.0f, R.live ? R.v[1] : 0.0f, hi.x, lo.x);
    split_pair_t<F16>(R.live ? R.v[2] : 0.0f, R.live ? R.v[3] : 0.0f, hi.y, lo.y);
#endif
    unsigned char *dst = Bs + (kq >> 1) * (TN * 16) + n_local * 16 + (kq & 1) * 8;
    *reinterpret_cast<uint2 *>(dst) = hi;
    *reinterpret_cast<uint2 *>(dst + kPartB) = lo;
  };
  auto multiply = [&](int buf) {   // wave (wm, wn): rows wm*64 .. +63, columns wn*32 .. +31
    const unsigned char *A = smem + buf * kBuf + (lane >> 5) * (kTM * 16) + (wm * 64 + (lane & 31)) * 16;
    const unsigned char *Bp = smem + buf * kBuf + kStage + (lane >> 5) * (TN * 16) + (wn * 32 + (lane & 31)) * 16;
    bf16x8 a[2][2], bb[2];
#ifdef KGDET_CONV_ABL_NOLDSREAD   // ablation: fragments from registers (no LDS reads)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      a[pt][0] = __builtin_bit_cast(bf16x8, acc[0].lo.lo); a[pt][1] = __builtin_bit_cast(bf16x8, acc[1].lo.lo);
      bb[pt] = __builtin_bit_cast(bf16x8, acc[1].hi.lo);
    }
#else
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      a[pt][0] = *reinterpret_cast<const bf16x8 *>(A + pt * kPart);
      a[pt][1] = *reinterpret_cast<const bf16x8 *>(A + pt * kPart + 32 * 16);
      bb[pt] = *reinterpret_cast<const bf16x8 *>(Bp + pt * kPartB);
    }
#endif
#ifdef KGDET_CONV_ABL_NOMFMA
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const f32x4 q = __builtin_bit_cast(f32x4, a[0][mi]) + __builtin_bit_cast(f32x4, a[1][mi]) + __builtin_bit_cast(f32x4, bb[0]) + __builtin_bit_cast(f32x4, bb[1]);
      acc[mi][0] += q[0]; acc[mi][1] += q[1]; acc[mi][2] += q[2]; acc[mi][3] += q[3];
    }
    if (false)
#endif
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {   // small terms first
      acc[mi] = mfma_t<F16>(a[1][mi], bb[0], acc[mi]);
      acc[mi] = mfma_t<F16>(a[0][mi], bb[1], acc[mi]);
      acc[mi] = mfma_t<F16>(a[0][mi], bb[0], acc[mi]);
    }
  };
  {
    Regs R[kPF];
#pragma unroll
    for (int i = 0; i < kPF; ++i) issue(i, R[i]);
#ifdef KGDET_GATE_EARLY
    if (!F16 && gate) {   // (pins the gate values HERE: hipcc otherwise sinks the loads to their use behind the loop.  They are older than
                  //  the first stage's loads, which the commit below waits for anyway)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(gv[mi][r]));
    }
#endif
    commit(0, R[0]);
    // stage s: set s % kPF was committed one body ago and is free -> loads of stage s + kPF; set (s+1) % kPF is
    // converted into the other LDS buffer while stage s is multiplied.  The main loop runs whole groups of kPF
    // bodies with NO condition around the loads (clamped addresses instead): only then does hipcc keep counted
    // s_waitcnt vmcnt(N) across the back edge -- with guarded bodies it drained the queue (vmcnt(0)) every trip.
    const int full = stages / kPF * kPF;
    for (int s0 = 0; s0 < full; s0 += kPF) {
#pragma unroll
      for (int u = 0; u < kPF; ++u) {
        const int s = s0 + u;
        __syncthreads();
        issue(s + kPF, R[u]);
        multiply(s & 1);
        commit((s + 1) & 1, R[(u + 1) % kPF]);   // past the last stage: a clamped duplicate nobody reads
      }
    }
#pragma unroll
    for (int u = 0; u < kPF - 1; ++u) {   // tail: stages full .. stages-1 are already in R[u]; no loads
      const int s = full + u;
      if (s < stages) {
        __syncthreads();
        multiply(s & 1);
        if (s + 1 < stages) commit((s + 1) & 1, R[u + 1]);
      }
    }
  }

  // store: lane holds column (lane & 31) of 16 rows per 32 x 32 block -> 128-byte row segments per half wave
  float *yb = y + (long long)part * part_stride + (long long)b * M * N;
  const int n = n0 + wn * 32 + (lane & 31);
  if constexpr (F16) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] *= 1.0f / kF16WeightScale;
  }
  if (bias || residual) {   // epilogue operands first, all loads in flight at once (clamped addresses, no branches)
    const int nc = min(n, N - 1);
    float add[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), M - 1);
        float v = bias ? bias[m] : 0.0f;
        if (residual) v += residual[((long long)b * M + m) * N + nc];
        add[mi][r] = v;
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] += add[mi][r];
  }
  if (gate) {   // y *= [gate > 0]: the backward of the ReLU that produced this convolution's input (gate = that input), see
                // kgdet_conv_apply_gated_fmt -- the consumer of y no longer takes a masking pass over it
#ifdef KGDET_GATE_EARLY
    if constexpr (F16)
#endif
    {
    const int nc = min(n, N - 1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), M - 1);
        gv[mi][r] = gate[((long long)b * M + m) * N + nc];
      }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] = gv[mi][r] > 0.0f ? acc[mi][r] : 0.0f;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M && n < N) yb[(long long)m * N + n] = relu ? fmaxf(acc[mi][r], 0.0f) : acc[mi][r];
    }
}

// 3x3, stride 1, padding 1 with the input PATCH of a tile staged once per 16-channel chunk (conv_nn<9> loads, splits and
// writes the shifted tile once per (chunk, tap): nine times the loads, conversions and address arithmetic -- and with two
// waves per SIMD the instruction count, not MFMA / LDS / HBM, is what bounds it: ablations of each changed its time by < 5 %).
// Tile = TY x TX output pixels (TY * TX <= 128, chosen by the host to fit the map: 3 x 42 for the 25 x 42 / 50 x 84 / 100 x 168
// levels), patch = (TY + 2) x (TX + 2) <= 256 pixels, zero outside the image, split once, stored [part][khalf][256][8 bf16];
// a tap is an LDS address offset: three row-base registers + immediates.  A stages stream as in conv_nn (one 16-byte load
// per thread and stage, three register sets, three LDS buffers: static indices in the nine-tap body); one barrier per stage.
#ifdef KGDET_CONV_TRACE   // experiment build (make VARIANT=ctrace EXTRA=-DKGDET_CONV_TRACE, tools/conv_trace.py): cycle sums per phase
static __device__ unsigned long long g_conv_trace[1024 * 8];   // [block][cat]: 0 barrier, 1 frag reads, 2 mfma, 3 commit, 4 lds drain, 5 prologue, 6 total, 7 stages
#define CV_NOW() __builtin_amdgcn_s_memtime()
#define CV_ADD(cat) do { const unsigned long long n__ = CV_NOW(); tr[cat] += n__ - tr_t; tr_t = n__; } while (0)
#else
#define CV_ADD(cat) do { } while (0)
#endif
constexpr int kPatchMax = 256;
constexpr int kPatchPart = 2 * kPatchMax * 16;   // bytes of one part of a B patch: [khalf][256][8 bf16]
constexpr int kPatchBuf = 2 * kPatchPart;        // hi + lo
constexpr int kPatchLds = 6 * kStage + 2 * kPatchBuf;

__global__ __launch_bounds__(kNNThreads) void conv3x3_patch(const unsigned char *__restrict__ img,
                                                            const float *__restrict__ x, float *__restrict__ y, int M,
                                                            int K, int H, int W, int n_mt, int tiles_x, int n_nt, int tiles,
                                                            int ksplit, long long part_stride, int TX, int TY,
                                                            const float *__restrict__ bias,
                                                            const float *__restrict__ residual, int relu,
    const float *__restrict__ gate = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // kPatchLds: A x 6 (two sets of three stages) | B x 2
  const int unit = xcd_tile(blockIdx.x, tiles * ksplit);
  if (unit >= tiles * ksplit) return;
  const int part = unit % ksplit, tile = unit / ksplit;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int N = H * W;
  const int chunks_all = K / kTK, per = (chunks_all + ksplit - 1) / ksplit;
  const int c_begin = part * per, c_end = max(c_begin, min(chunks_all, c_begin + per));
  const int chunks = c_end - c_begin, stages = chunks * 9;
  const int mt = tile % n_mt, nt = (tile / n_mt) % n_nt, b = tile / (n_mt * n_nt);
  const int y0 = (nt / tiles_x) * TY, x0 = (nt % tiles_x) * TX;
  const int PW = TX + 2, PP = PW * (TY + 2);

  // B patch: thread -> (khalf, patch pixel); 8 channels of that pixel per chunk
  const int khalf_l = tid >> 8, pp = tid & 255;
  const int ppy = pp / PW, ppx = pp - ppy * PW;
  const int iy = y0 - 1 + ppy, ix = x0 - 1 + ppx;
  const bool p_live = pp < PP && iy >= 0 && iy < H && ix >= 0 && ix < W;
  const float *xb = x + ((long long)b * K + khalf_l * 8) * N + (p_live ? iy * W + ix : 0);
  unsigned char *b_dst = smem + 6 * kStage + (khalf_l * kPatchMax + pp) * 16;

  // this lane's output pixel and its patch address (tap (-1, -1))
  const int q = min(wn * 32 + (lane & 31), TX * TY - 1);
  const int py = q / TX, px = q - py * TX;
  const unsigned char *b_row0 = smem + 6 * kStage + ((lane >> 5) * kPatchMax + py * PW + px) * 16;
  const int row_pitch = PW * 16;
  const unsigned char *a_rd = smem + (lane >> 5) * (kTM * 16) + (wm * 64 + (lane & 31)) * 16;
  const unsigned char *ai = img + (long long)mt * (chunks_all * 9) * kStage + (long long)min(c_begin, chunks_all - 1) * 9 * kStage + tid * 16;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  float bv[8];
  auto issue_b = [&](int ci) {   // chunk ci of this part (clamped: unconditional loads)
    const float *xp = xb + (long long)min(c_begin + min(ci, chunks - 1), chunks_all - 1) * kTK * N;
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = xp[(long long)j * N];
  };
  auto commit_b = [&](int buf) {
    float v[8];
    // (pure arithmetic floats freely through hipcc's instruction selection: without this anchor the conversion -- and the
    // wait for the loads it consumes -- lands right behind the loads, at the head of the chunk)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(bv[j]));
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p_live ? bv[j] : 0.0f;
    bf16x8 hi, lo;
    split8(v, hi, lo);
    *reinterpret_cast<bf16x8 *>(b_dst + buf * kPatchBuf) = hi;
    *reinterpret_cast<bf16x8 *>(b_dst + buf * kPatchBuf + kPatchPart) = lo;
  };
  f32x4 R[3];
#ifdef KGDET_PATCH_ABL_NOAGLOBAL
  auto issue_a = [&](int s, f32x4 &r) { r[0] = (float)s; };
#else
  auto issue_a = [&](int s, f32x4 &r) { r = *reinterpret_cast<const f32x4 *>(ai + (long long)max(min(s, stages - 1), 0) * kStage); };
#endif

  // One barrier per GROUP of three stages (a tap row).  With a barrier per stage the two waves of a SIMD run in lockstep --
  // both read fragments, both want the matrix pipe, both commit -- and a stage costs the sum of the phases (trace: 946 ticks
  // per stage, 244 of them MFMA issue, 319 waiting at the barrier for the wave that got the pipe second); inside a group the
  // waves drift apart and one multiplies while the other reads and commits.
  // A: two sets of three stage buffers; group g multiplies set g & 1 while stages of group g + 1 (in registers since group
  // g - 1) are committed into the other set and the loads of group g + 2 are issued.  Fragments of the next stage of the
  // group are read while the current one multiplies; a group's first stage reads its own (its data is only visible after
  // the barrier).
  struct Frags {
    bf16x8 a[2][2], b[2];
  };
  auto read_frags = [&](Frags &F, const unsigned char *A, const unsigned char *Bp) {
#ifdef KGDET_PATCH_ABL_NOLDSREAD   // ablations: timing only, wrong results
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      F.a[pt][0] = __builtin_bit_cast(bf16x8, acc[0].lo.lo); F.a[pt][1] = __builtin_bit_cast(bf16x8, acc[1].lo.lo);
      F.b[pt] = __builtin_bit_cast(bf16x8, acc[1].hi.lo);
    }
    return;
#endif
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      F.a[pt][0] = *reinterpret_cast<const bf16x8 *>(A + pt * kPart);
      F.a[pt][1] = *reinterpret_cast<const bf16x8 *>(A + pt * kPart + 32 * 16);
      F.b[pt] = *reinterpret_cast<const bf16x8 *>(Bp + pt * kPatchPart);
    }
  };
  auto mma = [&](const Frags &F) {
#ifdef KGDET_PATCH_ABL_NOMFMA
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const f32x4 qv = __builtin_bit_cast(f32x4, F.a[0][mi]) + __builtin_bit_cast(f32x4, F.a[1][mi]) + __builtin_bit_cast(f32x4, F.b[0]) + __builtin_bit_cast(f32x4, F.b[1]);
      acc[mi][0] += qv[0]; acc[mi][1] += qv[1]; acc[mi][2] += qv[2]; acc[mi][3] += qv[3];
    }
    return;
#endif
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {   // small terms first
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[1][mi], F.b[0], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0][mi], F.b[1], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[0][mi], F.b[0], acc[mi], 0, 0, 0);
    }
  };
#ifdef KGDET_CONV_TRACE
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tr_t = CV_NOW();
  const unsigned long long tr_start = tr_t;
#endif
  issue_b(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) issue_a(j, R[j]);
  commit_b(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    *reinterpret_cast<f32x4 *>(smem + j * kStage + tid * 16) = R[j];
    issue_a(3 + j, R[j]);
  }
  Frags F[2];
  CV_ADD(5);
  for (int ci = 0; ci < chunks; ++ci) {
    issue_b(ci + 1);
    const unsigned char *brow = b_row0 + (ci & 1) * kPatchBuf;
    const int set0 = (ci & 1) * 3 * kStage;            // A set of this chunk's rows 0 and 2; row 1 uses the other
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int g = ci * 3 + r;
      const int cur = (r & 1) ? 3 * kStage - set0 : set0, nxt = 3 * kStage - cur;
#ifndef KGDET_PATCH_ABL_NOBARRIER
      __syncthreads();
#endif
      CV_ADD(0);
      read_frags(F[0], a_rd + cur, brow + r * row_pitch);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j < 2) read_frags(F[(j + 1) & 1], a_rd + cur + (j + 1) * kStage, brow + r * row_pitch + (j + 1) * 16);
        __builtin_amdgcn_sched_barrier(0);   // (hipcc would sink the prefetch reads to their uses)
        CV_ADD(1);
        mma(F[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
        CV_ADD(2);
        *reinterpret_cast<f32x4 *>(smem + nxt + j * kStage + tid * 16) = R[j];   // A of stage 3 (g + 1) + j
        issue_a(3 * (g + 2) + j, R[j]);
        if (r == 2 && j == 1) commit_b((ci + 1) & 1);   // the next chunk's patch (its last readers finished a chunk ago)
        __builtin_amdgcn_sched_barrier(0);
        CV_ADD(3);
      }
    }
  }

#ifdef KGDET_CONV_TRACE
  tr[6] = CV_NOW() - tr_start;
  tr[7] = stages;
  tr[4] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
          __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID | HW_ID
  tr[5] = tr_start;
  if (tid == 0 && blockIdx.x < 1024)
    for (int c = 0; c < 8; ++c) g_conv_trace[blockIdx.x * 8 + c] = tr[c];
#endif
  // store (as conv_nn): lane holds pixel (lane & 31) of its wave's 32, 16 rows per 32 x 32 block
  float *yb = y + (long long)part * part_stride + (long long)b * M * N;
  const int qq = wn * 32 + (lane & 31);
  const int oy = y0 + py, ox = x0 + px;
  const bool o_live = qq < TX * TY && oy < H && ox < W;
  const int n = o_live ? oy * W + ox : 0;
  if (bias || residual) {
    float add[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), M - 1);
        float v = bias ? bias[m] : 0.0f;
        if (residual) v += residual[((long long)b * M + m) * N + n];
        add[mi][r] = v;
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] += add[mi][r];
  }
  if (gate) {   // (as conv_nn)
    float gv[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), M - 1);
        gv[mi][r] = gate[((long long)b * M + m) * N + n];
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] = gv[mi][r] > 0.0f ? acc[mi][r] : 0.0f;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M && o_live) yb[(long long)m * N + n] = relu ? fmaxf(acc[mi][r], 0.0f) : acc[mi][r];
    }
}

// conv3x3_patch with FOUR waves per workgroup: wave w owns rows 32 w .. 32 w + 31 of the 128-row tile and ALL 128 pixels.
// Its A fragments (32 rows x 16 channels, hi + lo = 2 KB per stage) come straight from the packed image in L2 into
// registers -- the image is stored in fragment order -- three stages ahead; only the input patch lives in LDS.  No A
// staging, no A commit, 8 instead of 6 LDS fragment reads for 12 instead of 6 MFMAs, and ONE barrier per chunk of nine
// stages (the patch swap).  32 KB of LDS and 4 waves per workgroup: two to three workgroups share a CU, so a launch with
// slightly more tiles than CUs (272 for [2, 128, 100, 168]) no longer waits for a CU that drew two 8-wave workgroups.
// WAVES = 4: the whole 128-row tile; WAVES = 2: a 64-row half of it (twice the workgroups, each with its own copy of the
// patch) -- for launches whose tile count is just above the CU count, where workgroups this small spread evenly.
// NB = 4 | 5 blocks of 32 pixels per tile: a tile of up to 160 pixels (e.g. 4 x 34) lets [2, *, 100, 168] take 250 tiles --
// one per CU -- where 128-pixel tiles need 272: a CU finishes a tile in ~28 us whatever shares it, so the 16 CUs that drew
// two set the time (51 us).
template <int WAVES, int NB, bool F16 = false>
__global__ __launch_bounds__(64 * WAVES) void conv3x3_patch4(const unsigned char *__restrict__ img,
                                                             const float *__restrict__ x, float *__restrict__ y, int M,
                                                             int K, int H, int W, int n_mt, int tiles_x, int n_nt,
                                                             int tiles, int ksplit, long long part_stride, int TX, int TY,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ residual, int relu,
    const float *__restrict__ gate = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kPatchBuf];
  if constexpr (F16) f16_saturate_on();
  constexpr int HALVES = 4 / WAVES, THREADS = 64 * WAVES, PXT = kPatchMax / THREADS;   // patch pixels per thread
  const int unit0 = xcd_tile(blockIdx.x, tiles * ksplit * HALVES);
  if (unit0 >= tiles * ksplit * HALVES) return;
  const int half = unit0 % HALVES, unit = unit0 / HALVES;
  const int part = unit % ksplit, tile = unit / ksplit;
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) + half * WAVES;
  const int N = H * W;
  const int chunks_all = K / kTK, per = (chunks_all + ksplit - 1) / ksplit;
  const int c_begin = part * per, c_end = max(c_begin, min(chunks_all, c_begin + per));
  const int chunks = c_end - c_begin, stages = chunks * 9;
  const int mt = tile % n_mt, nt = (tile / n_mt) % n_nt, b = tile / (n_mt * n_nt);
  const int y0 = (nt / tiles_x) * TY, x0 = (nt % tiles_x) * TX;
  const int PW = TX + 2, PP = PW * (TY + 2);

  if (mt * kTM + half * 64 >= M) return;   // (a 64-row half beyond the last output channel)
  // B patch: thread -> PXT patch pixels, all 16 channels of the chunk
  bool p_live[PXT];
  int p_off[PXT];
#pragma unroll
  for (int i = 0; i < PXT; ++i) {
    const int pp = tid + i * THREADS;
    const int ppy = pp / PW, ppx = pp - ppy * PW;
    const int iy = y0 - 1 + ppy, ix = x0 - 1 + ppx;
    p_live[i] = pp < PP && iy >= 0 && iy < H && ix >= 0 && ix < W;
    p_off[i] = p_live[i] ? iy * W + ix : 0;
  }
  const float *xb = x + (long long)b * K * N;
  unsigned char *b_dst = smem + tid * 16;

  // this lane's four output pixels (one per 32-pixel block) and their patch addresses (tap (-1, -1))
  int b_rd[NB], o_n[NB];
  bool o_live[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int qq = nb * 32 + (lane & 31), q = min(qq, TX * TY - 1);
    const int py = q / TX, px = q - py * TX;
    b_rd[nb] = ((lane >> 5) * kPatchMax + py * PW + px) * 16;
    o_live[nb] = qq < TX * TY && y0 + py < H && x0 + px < W;
    o_n[nb] = o_live[nb] ? (y0 + py) * W + x0 + px : 0;
  }
  const int row_pitch = PW * 16;
  const unsigned char *ag = img + ((long long)mt * chunks_all + min(c_begin, chunks_all - 1)) * 9 * kStage +
                            (lane >> 5) * (kTM * 16) + (wave * 32 + (lane & 31)) * 16;

  f32x16 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  float bv[PXT][16];
  auto issue_b = [&](int ci) {   // chunk ci of this part (clamped: unconditional loads)
    const float *xp = xb + (long long)min(c_begin + min(ci, chunks - 1), chunks_all - 1) * kTK * N;
#pragma unroll
    for (int i = 0; i < PXT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) bv[i][j] = xp[(long long)j * N + p_off[i]];
  };
  auto commit_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PXT; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(bv[i][j]));   // (anchor: see conv3x3_patch)
#pragma unroll
    for (int i = 0; i < PXT; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p_live[i] ? bv[i][kh * 8 + j] : 0.0f;
        bf16x8 hi, lo;
        split8_t<F16>(v, hi, lo);
        unsigned char *d = b_dst + i * THREADS * 16 + buf * kPatchBuf + kh * (kPatchMax * 16);
        *reinterpret_cast<bf16x8 *>(d) = hi;
        *reinterpret_cast<bf16x8 *>(d + kPatchPart) = lo;
      }
  };
  bf16x8 AR[3][2];
  auto issue_a = [&](int s, bf16x8 (&r)[2]) {
    const unsigned char *p = ag + (long long)max(min(s, stages - 1), 0) * kStage;
    r[0] = *reinterpret_cast<const bf16x8 *>(p);
    r[1] = *reinterpret_cast<const bf16x8 *>(p + kPart);
  };

  issue_b(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) issue_a(j, AR[j]);
  commit_b(0);
  for (int ci = 0; ci < chunks; ++ci) {
    issue_b(ci + 1);
    __syncthreads();   // patch ci is complete; every wave is done with patch ci - 1
    const unsigned char *bbuf = smem + (ci & 1) * kPatchBuf;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int s = ci * 9 + t;
      const unsigned char *bt = bbuf + (t / 3) * row_pitch + (t % 3) * 16;
      bf16x8 bf[NB][2];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bf[nb][0] = *reinterpret_cast<const bf16x8 *>(bt + b_rd[nb]);
        bf[nb][1] = *reinterpret_cast<const bf16x8 *>(bt + b_rd[nb] + kPatchPart);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_t<F16>(AR[t % 3][1], bf[nb][0], acc[nb]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_t<F16>(AR[t % 3][0], bf[nb][1], acc[nb]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_t<F16>(AR[t % 3][0], bf[nb][0], acc[nb]);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(s + 3, AR[t % 3]);
      if (t == 6) commit_b((ci + 1) & 1);   // the next chunk's patch (the barrier above freed that buffer)
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // store: lane holds pixel (lane & 31) of each 32-pixel block, rows 32 w + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float *yb = y + (long long)part * part_stride + (long long)b * M * N;
  const int m0 = mt * kTM + wave * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    if constexpr (F16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] *= 1.0f / kF16WeightScale;
    }
    if (bias || residual) {
      float add[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(m0 + (r & 3) + 8 * (r >> 2), M - 1);
        float v = bias ? bias[m] : 0.0f;
        if (residual) v += residual[((long long)b * M + m) * N + o_n[nb]];
        add[r] = v;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] += add[r];
    }
    if (gate) {   // (as conv_nn)
      float gv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) gv[r] = gate[((long long)b * M + min(m0 + (r & 3) + 8 * (r >> 2), M - 1)) * N + o_n[nb]];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = gv[r] > 0.0f ? acc[nb][r] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2);
      if (m < M && o_live[nb]) yb[(long long)m * N + o_n[nb]] = relu ? fmaxf(acc[nb][r], 0.0f) : acc[nb][r];
    }
  }
}

// grad_input of the 3x3 STRIDE-2 (padding 1) convolution on the patch machinery.  Output pixel (2i + pa, 2j + qa) receives only the
// taps with ky = 1 (pa = 0) or ky in {0, 2} (pa = 1) -- likewise in x -- each from gy(i + dy, j + dx), dy, dx in {0, 1}:
//   gx[c, 2i+pa, 2j+qa] = sum_o sum_(ky, kx of that parity class) w[o, c, ky, kx] * gy[o, i + (pa + 1 - ky) / 2, j + (qa + 1 - kx) / 2]
// So a tile of gy pixels owns four accumulator sets (the four parity classes of its 2 x 2 output blocks) and the nine taps
// are the nine stages of a chunk as in conv3x3_patch4, each adding into its class: one patch per chunk serves all classes,
// the weight fragments come from the ordinary transposed image (block 8 - (3 ky + kx): its taps are mirrored), and a lane
// stores 2 x 2 adjacent outputs.  64 gy pixels per tile (2 blocks of 32), 4 waves = 128 rows.  (MIOpen's fp32 implicit GEMM
// for this gradient runs at ~100 TFLOP/s plus two layout transposes.)
__global__ __launch_bounds__(256) void conv3x3_s2_grad_input(const unsigned char *__restrict__ img_t,
                                                             const float *__restrict__ gy, float *__restrict__ gx, int M,
                                                             int K, int H, int W, int Hin, int Win, int n_mt, int tiles_x,
                                                             int n_nt, int tiles, int TX, int TY) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kPatchBuf];
  constexpr int NB = 2;
  const int tile = xcd_tile(blockIdx.x, tiles);
  if (tile >= tiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = H * W;
  const int chunks = K / kTK, stages = chunks * 9;
  const int mt = tile % n_mt, nt = (tile / n_mt) % n_nt, b = tile / (n_mt * n_nt);
  const int y0 = (nt / tiles_x) * TY, x0 = (nt % tiles_x) * TX;
  const int PW = TX + 2, PP = PW * (TY + 2);

  const int pp = tid;
  const int ppy = pp / PW, ppx = pp - ppy * PW;
  const int iy = y0 - 1 + ppy, ix = x0 - 1 + ppx;
  const bool p_live = pp < PP && iy >= 0 && iy < H && ix >= 0 && ix < W;
  const float *xb = gy + (long long)b * K * N + (p_live ? iy * W + ix : 0);
  unsigned char *b_dst = smem + pp * 16;

  int b_rd[NB], o_y[NB], o_x[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int qq = nb * 32 + (lane & 31), q = min(qq, TX * TY - 1);
    const int py = q / TX, px = q - py * TX;
    b_rd[nb] = ((lane >> 5) * kPatchMax + py * PW + px) * 16;
    const bool live = qq < TX * TY && y0 + py < H && x0 + px < W;
    o_y[nb] = live ? 2 * (y0 + py) : Hin;     // (dead lanes: every output row fails the bound test)
    o_x[nb] = 2 * (x0 + px);
  }
  const int row_pitch = PW * 16;
  const unsigned char *ag = img_t + (long long)mt * stages * kStage + (lane >> 5) * (kTM * 16) + (wave * 32 + (lane & 31)) * 16;

  f32x16 acc[4][NB];   // [2 pa + qa][pixel block]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.0f;

  float bv[16];
  auto issue_b = [&](int ci) {
    const float *xp = xb + (long long)min(ci, chunks - 1) * kTK * N;
#pragma unroll
    for (int j = 0; j < 16; ++j) bv[j] = xp[(long long)j * N];
  };
  auto commit_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(bv[j]));
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p_live ? bv[kh * 8 + j] : 0.0f;
      bf16x8 hi, lo;
      split8(v, hi, lo);
      *reinterpret_cast<bf16x8 *>(b_dst + buf * kPatchBuf + kh * (kPatchMax * 16)) = hi;
      *reinterpret_cast<bf16x8 *>(b_dst + buf * kPatchBuf + kh * (kPatchMax * 16) + kPatchPart) = lo;
    }
  };
  // stage t of a chunk = tap (ky, kx) = (t / 3, t % 3): image block 8 - t
  bf16x8 AR[3][2];
  auto issue_a = [&](int s, bf16x8 (&r)[2]) {
    const int sc = min(s, stages - 1), ci = sc / 9, t = sc - ci * 9;
    const unsigned char *p = ag + (long long)(ci * 9 + 8 - t) * kStage;
    r[0] = *reinterpret_cast<const bf16x8 *>(p);
    r[1] = *reinterpret_cast<const bf16x8 *>(p + kPart);
  };

  issue_b(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) issue_a(j, AR[j]);
  commit_b(0);
  for (int ci = 0; ci < chunks; ++ci) {
    issue_b(ci + 1);
    __syncthreads();
    const unsigned char *bbuf = smem + (ci & 1) * kPatchBuf;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      constexpr int kDummy = 0;
      const int ky = t / 3, kx = t % 3;
      const int cls = 2 * (ky != 1) + (kx != 1), dy = ky == 0, dx = kx == 0;
      const unsigned char *bt = bbuf + (1 + dy) * row_pitch + (1 + dx) * 16;
      bf16x8 bf[NB][2];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bf[nb][0] = *reinterpret_cast<const bf16x8 *>(bt + b_rd[nb]);
        bf[nb][1] = *reinterpret_cast<const bf16x8 *>(bt + b_rd[nb] + kPatchPart);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[cls][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AR[t % 3][1], bf[nb][0], acc[cls][nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[cls][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AR[t % 3][0], bf[nb][1], acc[cls][nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[cls][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AR[t % 3][0], bf[nb][0], acc[cls][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(ci * 9 + t + 3, AR[t % 3]);
      if (t == 6) commit_b((ci + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      (void)kDummy;
    }
  }

  // store: rows 32 w + (r & 3) + 8 (r >> 2) + 4 (lane >> 5); a lane owns the 2 x 2 outputs of its gy pixel
  const long long Nin = (long long)Hin * Win;
  float *xo = gx + (long long)b * M * Nin;
  const int m0 = mt * kTM + wave * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2);
      if (m >= M) continue;
      float *row = xo + (long long)m * Nin;
#pragma unroll
      for (int pa = 0; pa < 2; ++pa) {
        const int oy = o_y[nb] + pa;
        if (oy >= Hin) continue;
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
          if (o_x[nb] + qa < Win) row[(long long)oy * Win + o_x[nb] + qa] = acc[2 * pa + qa][nb][r];
      }
    }
}

// The stem: 7x7, stride 2, padding 3, 3 -> 64 channels (mmdet/models/backbones/resnet.py:487-488) as an implicit GEMM with
// K = 147 (c, ky, kx) padded to 160 = ten stages of 16.  A tile of 8 x 16 output pixels needs a 21 x 37 x 3 input patch
// (9.3 KB, fp32, in LDS, zero outside the image).  Each lane builds ITS OWN B fragment -- pixel lane & 31, k half lane >> 5 --
// by gathering 8 patch values through a 160-entry offset table and splitting them, so the activations are never staged as
// an operand image; A fragments (64 rows: two 32-row blocks) come straight from the packed weight image in L2.  4 waves,
// wave w = pixels 32 w .. 32 w + 31 of the tile, all 64 rows.  (MIOpen's fp32 Winograd-type kernel for this layer: 220 us.)
constexpr int kStemTY = 8, kStemTX = 16, kStemPH = 2 * kStemTY + 5, kStemPW = 2 * kStemTX + 5, kStemK = 160;

template <bool F16>
__global__ __launch_bounds__(256) void stem_conv7x7_s2(const unsigned char *__restrict__ img, const float *__restrict__ x,
                                                       float *__restrict__ y, int H, int W, int Ho, int Wo, int tiles_x,
                                                       int tiles_per_image) {
  __shared__ float patch[3 * kStemPH * kStemPW];
  __shared__ int koff[kStemK];
  if constexpr (F16) f16_saturate_on();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / tiles_per_image, nt = blockIdx.x - b * tiles_per_image;
  const int oy0 = (nt / tiles_x) * kStemTY, ox0 = (nt % tiles_x) * kStemTX;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
  const float *xb = x + (long long)b * 3 * H * W;
  for (int i = tid; i < 3 * kStemPH * kStemPW; i += 256) {
    const int c = i / (kStemPH * kStemPW), r = i - c * (kStemPH * kStemPW), py = r / kStemPW, px = r - py * kStemPW;
    const int iy = iy0 + py, ix = ix0 + px;
    patch[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[((long long)c * H + iy) * W + ix] : 0.0f;
  }
  if (tid < kStemK) {
    const int c = tid / 49, r = tid - c * 49, ky = r / 7, kx = r - ky * 7;
    koff[tid] = tid < 147 ? c * (kStemPH * kStemPW) + ky * kStemPW + kx : 0;   // (k >= 147: zero weights)
  }
  __syncthreads();
  const int q = wave * 32 + (lane & 31), py = q / kStemTX, px = q - py * kStemTX;
  const float *pbase = patch + (2 * py) * kStemPW + 2 * px;
  const int kh = (lane >> 5) * 8;
  const unsigned char *ag = img + (lane >> 5) * (kTM * 16) + (lane & 31) * 16;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
  for (int s = 0; s < kStemK / kTK; ++s) {
    bf16x8 a[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      a[0][mi] = *reinterpret_cast<const bf16x8 *>(ag + (long long)s * kStage + mi * 32 * 16);
      a[1][mi] = *reinterpret_cast<const bf16x8 *>(ag + (long long)s * kStage + kPart + mi * 32 * 16);
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = pbase[koff[s * kTK + kh + j]];
    bf16x8 bhi, blo;
    split8_t<F16>(v, bhi, blo);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {   // small terms first
      acc[mi] = mfma_t<F16>(a[1][mi], bhi, acc[mi]);
      acc[mi] = mfma_t<F16>(a[0][mi], blo, acc[mi]);
      acc[mi] = mfma_t<F16>(a[0][mi], bhi, acc[mi]);
    }
  }
  const int oy = oy0 + py, ox = ox0 + px;
  if (oy < Ho && ox < Wo) {
    float *yb = y + (long long)b * 64 * Ho * Wo + (long long)oy * Wo + ox;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        yb[(long long)m * Ho * Wo] = F16 ? acc[mi][r] * (1.0f / kF16WeightScale) : acc[mi][r];
      }
  }
}

// out[i] = sum_s parts[s][i], s ascending (deterministic); n a multiple of 2
__global__ __launch_bounds__(256) void conv1x1_sum(const float *__restrict__ parts, float *__restrict__ out,
                                                   long long n, long long stride, int count) {
  for (long long i = (blockIdx.x * 256LL + threadIdx.x) * 2; i < n; i += gridDim.x * 512LL) {
    f32x2 s = {0.0f, 0.0f};
    int k = 0;
    for (; k + 8 <= count; k += 8) {      // eight loads in flight, added in slot order
      f32x2 v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const f32x2 *>(parts + (long long)(k + e) * stride + i);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; k + 2 <= count; k += 2) {
      const f32x2 v0 = *reinterpret_cast<const f32x2 *>(parts + (long long)k * stride + i);
      const f32x2 v1 = *reinterpret_cast<const f32x2 *>(parts + (long long)(k + 1) * stride + i);
      s = (s + v0) + v1;
    }
    for (; k < count; ++k) s += *reinterpret_cast<const f32x2 *>(parts + (long long)k * stride + i);
    *reinterpret_cast<f32x2 *>(out + i) = s;
  }
}

// the same sum with the convolution's epilogue in its store: out = [relu](sum + bias[channel] [+ residual]), channel =
// (i / HW) % M (a K-split problem is small; the separate epilogue pass it used to take cost a launch, ~5 us of a step each)
__global__ __launch_bounds__(256) void conv1x1_sum_epilogue(const float *__restrict__ parts, float *__restrict__ out,
                                                            long long n, long long stride, int count,
                                                            const float *__restrict__ bias, const float *__restrict__ residual,
                                                            int relu, int M, long long HW,
                                                            const float *__restrict__ gate = nullptr) {
  for (long long i = (blockIdx.x * 256LL + threadIdx.x) * 2; i < n; i += gridDim.x * 512LL) {   // (HW is even: both in one plane)
    f32x2 s = {0.0f, 0.0f};
    int k = 0;
    for (; k + 4 <= count; k += 4) {      // (a K split has at most eight parts) four loads in flight, added in slot order
      f32x2 v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const f32x2 *>(parts + (long long)(k + e) * stride + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += v[e];
    }
    for (; k < count; ++k) s += *reinterpret_cast<const f32x2 *>(parts + (long long)k * stride + i);
    if (bias) {
      const float b = bias[(i / HW) % M];
      s[0] += b; s[1] += b;
    }
    if (residual) s += *reinterpret_cast<const f32x2 *>(residual + i);
    if (relu) { s[0] = fmaxf(s[0], 0.0f); s[1] = fmaxf(s[1], 0.0f); }
    if (gate) {
      const f32x2 gv = *reinterpret_cast<const f32x2 *>(gate + i);
      s[0] = gv[0] > 0.0f ? s[0] : 0.0f; s[1] = gv[1] > 0.0f ? s[1] : 0.0f;
    }
    *reinterpret_cast<f32x2 *>(out + i) = s;
  }
}

// Weight gradient of a convolution whose BatchNorm is folded into it (kgdet_amd/backbone.py _ConvBNActFold), the split sum and
// the BatchNorm parameter gradients in ONE pass: workgroup o adds the row's partials G[o][.] (slot order), forms
// dot = <w[o], G[o]>, stores grad_w[o] = s[o] * G[o], and adds the row's BatchNorm partials: grad_beta[o] = sum g,
// grad_gamma[o] = (dot - mean[o] * grad_beta[o]) / sqrt(var[o] + eps)   (csrc/bn_act.hip bn_fold_finish_kernel as the sum's
// epilogue: a launch less per convolution and step).  T9: the partials' columns are (tap, channel), grad_w's (channel, tap).
struct ConvFoldArgs {
  const float *w, *s, *mean, *var, *bn_partial;
  float *grad_beta, *grad_gamma;
  float eps;
  int P;
};
// Launch shape (round 5): ONE workgroup per output channel with as many threads as the row has columns (up to 1024), every
// thread one or two columns, a column's partials requested in batches of 16 -- a row's 16-38 MB / O of partials arrive in one or
// two memory round trips instead of the ~9 dependent ones of the 256-thread form (7.6 / 15.6 us per launch for the 1x1 / 3x3
// problems of the backbone, i.e. 1-2 TB/s on data that sits in the Infinity Cache).  T9: the (tap, channel) -> (channel, tap)
// transposition goes through LDS (the row, <= 18 KB), so that grad_w is stored and w is read in whole lines instead of 4-byte
// pieces 36 bytes apart.
template <bool T9>
__global__ __launch_bounds__(1024) void conv_wsum_fold(const float *__restrict__ parts, float *__restrict__ out, int C,
                                                       long long stride, int count, const ConvFoldArgs f) {
  extern __shared__ float row_s[];       // T9: the summed row in grad_w's column order
  __shared__ float red[2][16];
  const int o = blockIdx.x, CK = T9 ? 9 * C : C;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const float so = f.s[o];
  const float *wr = f.w + (long long)o * CK;
  const float *pr = parts + (long long)o * CK;
  float *gr = out + (long long)o * CK;
  float dot = 0.0f, sb = 0.0f;
  for (int q = tid; q < CK; q += nthr) {                  // the partials' columns (coalesced reads); every column in slot order
    float g0 = 0.0f;
    int k = 0;
    for (; k + 16 <= count; k += 16) {
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = pr[(long long)(k + e) * stride + q];
#pragma unroll
      for (int e = 0; e < 16; ++e) g0 += v[e];
    }
    if (k + 8 <= count) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = pr[(long long)(k + e) * stride + q];
#pragma unroll
      for (int e = 0; e < 8; ++e) g0 += v[e];
      k += 8;
    }
    for (; k < count; ++k) g0 += pr[(long long)k * stride + q];
    if constexpr (T9) {
      row_s[(q % C) * 9 + q / C] = g0;                    // partial column (tap, channel) -> grad_w column (channel, tap)
    } else {
      dot += wr[q] * g0;
      gr[q] = g0 * so;
    }
  }
  if constexpr (T9) {
    __syncthreads();
    for (int j = tid; j < CK; j += nthr) {
      const float g0 = row_s[j];
      dot += wr[j] * g0;
      gr[j] = g0 * so;
    }
  }
  for (int k = tid; k < f.P; k += nthr) sb += f.bn_partial[(long long)o * f.P + k];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { dot += __shfl_xor(dot, d); sb += __shfl_xor(sb, d); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = sb; red[1][tid >> 6] = dot; }
  __syncthreads();
  if (tid == 0) {
    float b = 0.0f, d = 0.0f;
    for (int w = 0; w < (nthr >> 6); ++w) { b += red[0][w]; d += red[1][w]; }     // wave order: fixed
    if (f.grad_beta) f.grad_beta[o] = b;
    if (f.grad_gamma) f.grad_gamma[o] = (d - f.mean[o] * b) / sqrtf(f.var[o] + f.eps);
  }
}

// 3x3 grad_weight: out[o][c][t] = sum_s parts[s][o][t * C + c]  (the NT kernel's columns are (tap, channel))
__global__ __launch_bounds__(256) void conv3x3_wsum(const float *__restrict__ parts, float *__restrict__ out, int O, int C,
                                                    int count) {
  const long long n = (long long)O * C * 9;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    const int o = (int)(i / (9 * C)), rem = (int)(i - (long long)o * 9 * C), t = rem / C, ch = rem - t * C;
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= count; k += 8) {   // eight loads in flight, added in slot order
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = parts[(long long)(k + e) * n + i];
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; k < count; ++k) s += parts[(long long)k * n + i];
    out[((long long)o * C + ch) * 9 + t] = s;
  }
}

// grad_weight kernels write partial[split][m][n] (natural [M, N] layout) = sum over the split's pixels (and images) of
// a[b][m][px] * bm[b][n][px]; a [B, M, L], bm [B, N, L], L contiguous.  The B * ceil(L / 16) stages are cut into `splits`
// runs of `per` stages; a stage never straddles two images (the tail of an image is zero-filled); conv1x1_sum /
// conv3x3_wsum add the partials in fixed order.
// conv_nt8: grad_weight with 512 threads (8 waves as 2 x 4, 64 x 32 outputs each, two per SIMD) for maps with
// H*W % 4 == 0.  A thread owns (row, 4-pixel quarter of the stage): one 16-byte load per operand (TAPS == 9 with a
// column shift: two aligned loads + a static selection), 8-byte LDS writes.  
template <int TAPS, int DX>
__device__ __forceinline__ void conv_nt8_body(const float *__restrict__ a, const float *__restrict__ bm,
                                              float *__restrict__ partial, int M, int N, int L, int B, int n_mt, int n_nt,
                                              int stages_per_image, int per, int H, int W, int Cin, int unit,
                                              unsigned char *smem, float *__restrict__ row_sums, int rs_stride) {
  const int tile = unit % (n_mt * n_nt), split = unit / (n_mt * n_nt);
  const int mt = tile % n_mt, nt = tile / n_mt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int row = tid >> 2, q = tid & 3;
  // row_sums[m * rs_stride + split] = sum over the split's pixels of a[., m, .] (the BatchNorm beta gradient of a folded
  // convolution whose output gradient arrived already masked: kgdet_conv*_grad_weight_fold with bn_partial == NULL) -- written by
  // the first column tile of every (row tile, split); the values pass through this thread's registers anyway
  const bool rs_on = row_sums != nullptr && nt == 0;
  float rs = 0.0f;
  const int total = B * stages_per_image;
  const int s_begin = split * per, s_end = min(total, s_begin + per);
  const int am = min(mt * kTM + row, M - 1);
  const bool a_real = mt * kTM + row < M;
  const int tap = TAPS == 9 ? (nt * kTN) / Cin : 0;
  const int dy = TAPS == 9 ? tap / 3 - 1 : 0;
  constexpr int dx = DX, off = dx < 0 ? -4 : 0, sh = dx - off;
  constexpr int NB = (TAPS == 9 && dx != 0) ? 8 : 4;                   // floats of bm a thread loads per stage
  const int bcols = TAPS == 9 ? Cin : N;
  const int bn_raw = TAPS == 9 ? nt * kTN - tap * Cin + row : nt * kTN + row;
  const int bn = min(bn_raw, bcols - 1);
  const bool b_real = bn_raw < bcols;
  const float inv_w = 1.0f / (float)W;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  struct Regs {
    f32x4 va;
    float vb[NB];
    int p0;
  };
  auto issue = [&](int s, Regs &R) {
    const int sc = min(s, s_end - 1);
    const int img = sc / stages_per_image, st = sc - img * stages_per_image;
    const int p0 = st * kTK + q * 4;
    R.p0 = p0;
    const float *ap = a + ((long long)img * M + am) * L, *bp = bm + ((long long)img * bcols + bn) * L;
    R.va = *reinterpret_cast<const f32x4 *>(ap + min(p0, L - 4));
    const int base = p0 + dy * W + off;
#pragma unroll
    for (int k = 0; k < NB / 4; ++k) {   // clamped chunks hold wrong pixels only where the tap is outside the image
      const f32x4 w = *reinterpret_cast<const f32x4 *>(bp + min(max(base + 4 * k, 0), L - 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) R.vb[4 * k + e] = w[e];
    }
  };
  auto commit = [&](int buf, const Regs &R, bool real = true) {   // real: not the clamped duplicate past the last stage
    unsigned char *As = smem + buf * 2 * kStage, *Bs = As + kStage;
    const bool in_img = R.p0 < L;       // L % 4 == 0: a chunk is entirely inside or outside the image
    bool row_ok = in_img && b_real;
    int w0 = 0;
    if (TAPS == 9) {
      const int h = (int)(((float)R.p0 + 0.5f) * inv_w);
      w0 = R.p0 - h * W;
      row_ok = row_ok && h + dy >= 0 && h + dy < H;
    }
    float fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = (in_img && a_real) ? R.va[i] : 0.0f;
      const int col = w0 + i + dx;
      const bool ok = TAPS == 9 ? (row_ok && col >= 0 && col < W) : row_ok;
      fb[i] = ok ? R.vb[i + (TAPS == 9 ? sh : 0)] : 0.0f;
    }
    if (rs_on && real) rs += (fa[0] + fa[1]) + (fa[2] + fa[3]);
    uint2 ahi, alo, bhi, blo;
    split_pair(fa[0], fa[1], ahi.x, alo.x);
    split_pair(fa[2], fa[3], ahi.y, alo.y);
    split_pair(fb[0], fb[1], bhi.x, blo.x);
    split_pair(fb[2], fb[3], bhi.y, blo.y);
    const int o = (q >> 1) * (kTM * 16) + row * 16 + (q & 1) * 8;
    *reinterpret_cast<uint2 *>(As + o) = ahi;
    *reinterpret_cast<uint2 *>(As + kPart + o) = alo;
    *reinterpret_cast<uint2 *>(Bs + o) = bhi;
    *reinterpret_cast<uint2 *>(Bs + kPart + o) = blo;
  };
  auto multiply = [&](int buf) {   // wave (wm, wn): rows wm*64 .. +63, columns wn*32 .. +31
    const unsigned char *A = smem + buf * 2 * kStage + (lane >> 5) * (kTM * 16) + (wm * 64 + (lane & 31)) * 16;
    const unsigned char *Bp = smem + buf * 2 * kStage + kStage + (lane >> 5) * (kTN * 16) + (wn * 32 + (lane & 31)) * 16;
    bf16x8 fa[2][2], fb[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      fa[pt][0] = *reinterpret_cast<const bf16x8 *>(A + pt * kPart);
      fa[pt][1] = *reinterpret_cast<const bf16x8 *>(A + pt * kPart + 32 * 16);
      fb[pt] = *reinterpret_cast<const bf16x8 *>(Bp + pt * kPart);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][mi], fb[0], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][mi], fb[1], acc[mi], 0, 0, 0);
      acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][mi], fb[0], acc[mi], 0, 0, 0);
    }
  };
  constexpr int PF = 4;
  const int n = s_end - s_begin;
  if (n > 0) {
    Regs R[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) issue(s_begin + i, R[i]);
    commit(0, R[0]);
    const int full = n / PF * PF;   // unguarded bodies in the main loop, load-free tail (see conv_nn)
    for (int j0 = 0; j0 < full; j0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int j = j0 + u;
        __syncthreads();
        issue(s_begin + j + PF, R[u]);
        multiply(j & 1);
        commit((j + 1) & 1, R[(u + 1) % PF], j + 1 < n);
      }
    }
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
      const int j = full + u;
      if (j < n) {
        __syncthreads();
        multiply(j & 1);
        if (j + 1 < n) commit((j + 1) & 1, R[u + 1]);
      }
    }
  }
  float *out = partial + (long long)split * M * N;
  const int nn = nt * kTN + wn * 32 + (lane & 31);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M && nn < N) out[(long long)m * N + nn] = acc[mi][r];
    }
  if (rs_on) {     // the row's four quarters sit in adjacent lanes: fixed order
    rs += __shfl_xor(rs, 1);
    rs += __shfl_xor(rs, 2);
    if (q == 0 && a_real) row_sums[(long long)(mt * kTM + row) * rs_stride + split] = rs;
  }
}

// units > 0: the launch holds ceil(units / 8) * 8 workgroups and workgroup b takes unit xcd_tile(b, units) -- the tiles of one
// split (same pixels of both operands; for TAPS == 9 the nine taps re-read the same x rows) run on ONE XCD and share its L2
// instead of fetching the rows once per XCD; units == 0: workgroup b takes unit b (the order up to round 3, A/B).
template <int TAPS>
__global__ __launch_bounds__(kNNThreads) void conv_nt8(const float *__restrict__ a, const float *__restrict__ bm,
                                                       float *__restrict__ partial, int M, int N, int L, int B, int n_mt,
                                                       int n_nt, int stages_per_image, int per, int H, int W, int Cin,
                                                       int units, float *__restrict__ row_sums = nullptr, int rs_stride = 0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * kStage];
  const int unit = units > 0 ? xcd_tile(blockIdx.x, units) : (int)blockIdx.x;
  if (units > 0 && unit >= units) return;
  if (TAPS == 9) {
    const int tile = unit % (n_mt * n_nt), nt = tile / n_mt;
    const int dx = ((nt * kTN) / Cin) % 3 - 1;   // uniform: one tap per tile
    if (dx < 0) conv_nt8_body<TAPS, -1>(a, bm, partial, M, N, L, B, n_mt, n_nt, stages_per_image, per, H, W, Cin, unit, smem, row_sums, rs_stride);
    else if (dx == 0) conv_nt8_body<TAPS, 0>(a, bm, partial, M, N, L, B, n_mt, n_nt, stages_per_image, per, H, W, Cin, unit, smem, row_sums, rs_stride);
    else conv_nt8_body<TAPS, 1>(a, bm, partial, M, N, L, B, n_mt, n_nt, stages_per_image, per, H, W, Cin, unit, smem, row_sums, rs_stride);
  } else {
    conv_nt8_body<TAPS, 0>(a, bm, partial, M, N, L, B, n_mt, n_nt, stages_per_image, per, H, W, Cin, unit, smem, row_sums, rs_stride);
  }
}

// conv_ntp (round 4): the same grad_weight tile (128 x 128 outputs, partial[split][m][n]) with the work of a stage divided
// between PRODUCER and CONSUMER waves instead of done by every wave in turn.  Counters of conv_nt8 on a 3x3, 256 -> 256,
// 50 x 84 problem (gpurun, rocprofv3 --pmc): the MFMA pipe busy 25 % of the kernel, ~76 vector instructions per wave and
// 16-pixel stage for 6 MFMAs (index arithmetic, boundary selects, hi / lo split), vector ALU busy 46 %, the loads served by L2
// in ~185 cycles, 33 MB fetched for 66 MB of operands -- not a bandwidth problem: eight waves that all convert, then all
// multiply, between two barriers per 16 pixels, with both waves of a SIMD in the same phase at the same time.
//   * stage = 32 pixels (a full 128-byte line of every operand row), two LDS stages, ONE barrier per stage;
//   * 4 producer waves: thread = (row of a 32-row pass, 16-byte piece of the 128-byte row segment) -- 8 lanes read one
//     contiguous row segment --, four passes for the 128 rows of each operand, loads two stages ahead in registers;
//     boundary masks of the 3x3 taps once per stage and thread (the four passes share the pixels), none for the grad_y
//     operand (rows beyond M / N are clamped duplicates whose outputs are not stored; pixels beyond the image multiply a
//     zeroed x), image / stage counters advanced incrementally instead of divided out;
//   * 4 consumer waves: 64 x 64 outputs each (four accumulator blocks), per 16 pixels 8 fragment reads for 12 MFMAs
//     (conv_nt8: 6 for 6), products ordered so that consecutive MFMAs never share an accumulator;
//   * LDS: the four (16-pixel step, k half) blocks of a stage start 16 banks apart (kPKH = 2048 + 64 bytes): the producers'
//     8-byte stores of one instruction (8 pieces x 4 rows per half wave) fall on 64 different banks.
constexpr int kPK = 32;                         // pixels per stage
constexpr int kPKH = kTM * 16 + 64;             // [128 rows][8 bf16] + the bank offset
constexpr int kPKS = 2 * kPKH;                  // one 16-pixel MFMA step: two k halves
constexpr int kPPart = 2 * kPKS;                // one part (hi / lo) of one operand's stage
constexpr int kPOperand = 2 * kPPart;
constexpr int kPStage = 2 * kPOperand;          // A + B = 33792 bytes
#ifndef KGDET_NTP_CONS
#define KGDET_NTP_CONS 4
#endif
constexpr int kPCons = KGDET_NTP_CONS;               // consumer waves: 4 (64 x 64 outputs each) or 8 (64 x 32: two per SIMD)
constexpr int kPNI = kPCons == 8 ? 1 : 2;            // 32-column blocks per consumer wave
constexpr int kPThreads = kPCons * 64 + 256;
constexpr int kPLds = 2 * kPStage;       // (a request padded beyond 80 KB -- never two workgroups on a CU -- changed nothing)

template <int TAPS, bool PRODUCER, bool ALIGNED>
__device__ __forceinline__ void conv_ntp_role(const float *__restrict__ a, const float *__restrict__ bm,
                                              float *__restrict__ partial, int M, int N, int L, int B, int n_mt, int n_nt,
                                              int stages_per_image, int per, int H, int W, int Cin, int unit,
                                              unsigned char *smem, float *__restrict__ row_sums, int rs_stride) {
  const int tile = unit % (n_mt * n_nt), split = unit / (n_mt * n_nt);
  const int mt = tile % n_mt, nt = tile / n_mt;
  const int total = B * stages_per_image;
  const int s_begin = split * per, s_end = min(total, s_begin + per);
  const int n = s_end - s_begin;
  const int wtid = threadIdx.x, tid = PRODUCER ? wtid - kPCons * 64 : wtid;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  if constexpr (PRODUCER) {
    // 16-byte global loads need 4-byte alignment only on this part (tools/microbench/unaligned_x4.hip: same rate at every
    // shift), so a tap is a shift of the load ADDRESS -- no aligned pair + selection, any map width, any pixel count: the
    // zero-padded copies of rounds 2-3 (pad_rows2 for maps with W % 4 != 0) are gone for this kernel.
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int rp = tid >> 3, q = tid & 7;
    const int tap = TAPS == 9 ? (nt * kTN) / Cin : 0;
    const int dy = TAPS == 9 ? tap / 3 - 1 : 0, dx = TAPS == 9 ? tap % 3 - 1 : 0;
    const int shift = dy * W + dx;                                      // of the flat pixel index
    const int bcols = TAPS == 9 ? Cin : N;
    const int bn0 = TAPS == 9 ? nt * kTN - tap * Cin : nt * kTN;
    int a_off[4], b_off[4];                                             // element offsets of the four rows inside one image
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a_off[k] = min(mt * kTM + rp + 32 * k, M - 1) * L;
      b_off[k] = min(bn0 + rp + 32 * k, bcols - 1) * L;
    }
    const float inv_w = 1.0f / (float)W;
    // LDS byte offset of this thread's piece inside an operand part: pixels 4q .. 4q + 3 = step q >> 2, k half (q >> 1) & 1
    const int lds_o = (q >> 2) * kPKS + ((q >> 1) & 1) * kPKH + rp * 16 + (q & 1) * 8;

    struct Regs {
      f32x4 va[4], vb[4];
      int p0;
    };
    int img = s_begin / stages_per_image, st = s_begin - img * stages_per_image;   // of the NEXT stage to be issued
    int issued = s_begin;
    const bool rs_on = row_sums != nullptr && nt == 0;      // (as conv_nt8_body: per-row sums of the grad_y operand)
    float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto issue = [&](Regs &R) __attribute__((always_inline)) {
      // (past the end of the range: the last stage again -- the loads stay unconditional, nothing is committed from them)
      const int p0 = st * kPK + q * 4;
      R.p0 = p0;
      const float *ai = a + (long long)img * M * L, *bi = bm + (long long)img * bcols * L;
      // a piece that lies inside the image is ONE (4-byte aligned) 16-byte load; the few pieces that straddle the image's first /
      // last pixel under a tap, and the ragged last piece of an image with L % 4 != 0, load their in-range pixels one by one
      const int base = p0 + shift;
      if (p0 <= L - 4 && base >= 0 && base <= L - 4) {
        // (scalar image base + zero-extended 32-bit byte offset: the load's own addressing mode, no 64-bit vector arithmetic)
        const char *ab = reinterpret_cast<const char *>(ai), *bb = reinterpret_cast<const char *>(bi);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4u ua = *reinterpret_cast<const f32x4u *>(ab + (size_t)((unsigned)(a_off[k] + p0) * 4u));
#ifdef KGDET_NTP_ABL_NOBLOAD   // ablation: no global loads for the x operand (wrong results; how much of the time is the L2 -> CU traffic?)
          const f32x4u ub = ua;
#else
          const f32x4u ub = *reinterpret_cast<const f32x4u *>(bb + (size_t)((unsigned)(b_off[k] + base) * 4u));
#endif
          R.va[k] = f32x4{ua[0], ua[1], ua[2], ua[3]};
          R.vb[k] = f32x4{ub[0], ub[1], ub[2], ub[3]};
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            R.va[k][i] = p0 + i < L ? ai[a_off[k] + p0 + i] : 0.0f;
            R.vb[k][i] = (base + i >= 0 && base + i < L) ? bi[b_off[k] + base + i] : 0.0f;
          }
      }
      if (issued + 1 < s_end) {
        ++issued;
        if (++st == stages_per_image) { st = 0; ++img; }
      }
    };
    auto commit = [&](int buf, const Regs &R) __attribute__((always_inline)) {
      unsigned char *As = smem + buf * kPStage + lds_o, *Bs = As + kPOperand;
      const int p0 = R.p0;
      bool ok[4];
      if constexpr (ALIGNED) {
        // W % 4 == 0 (and so L % 4 == 0): a piece lies inside one row and entirely inside or outside the image -- one row test
        // for the four pixels, a column test for the one pixel a +-1 tap can push out (each vector instruction beside the MFMA
        // wave of its SIMD costs that wave ~10 cycles: the per-pixel form below is ~40 instructions per stage)
        bool row_ok = p0 < L;
        int w0 = 0;
        if (TAPS == 9) {
          const int h0 = (int)(((float)p0 + 0.5f) * inv_w);
          w0 = p0 - h0 * W;
          row_ok = row_ok && (unsigned)(h0 + dy) < (unsigned)H;
        }
        ok[0] = row_ok && (TAPS == 1 || dx >= 0 || w0 > 0);
        ok[1] = ok[2] = row_ok;
        ok[3] = row_ok && (TAPS == 1 || dx <= 0 || w0 + 4 < W);
      } else {
        const int h0 = TAPS == 9 ? (int)(((float)p0 + 0.5f) * inv_w) : 0;
        const int w0 = p0 - h0 * W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bool v = p0 + i < L;
          if (TAPS == 9) {
            const bool wrap = w0 + i >= W;                              // (W >= 4: a piece touches at most two rows)
            const int h = h0 + (wrap ? 1 : 0) + dy, w = w0 + i - (wrap ? W : 0) + dx;
            v = v && h >= 0 && h < H && w >= 0 && w < W;
          }
          ok[i] = v;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 fa = R.va[k], fb = R.vb[k];
        if (rs_on) rs[k] += (p0 < L) ? (fa[0] + fa[1]) + (fa[2] + fa[3]) : 0.0f;   // (beyond L - 4 the loads zero-filled the rest)
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = ok[i] ? fb[i] : 0.0f;       // pixels outside the image multiply a zero
        uint2 ahi, alo, bhi, blo;
#ifdef KGDET_NTP_ABL_NOSPLIT   // ablation: the operands' bits as they are (wrong results; what would operands split beforehand buy?)
        ahi = uint2{__float_as_uint(fa[0]), __float_as_uint(fa[1])}; alo = uint2{__float_as_uint(fa[2]), __float_as_uint(fa[3])};
        bhi = uint2{__float_as_uint(fb[0]), __float_as_uint(fb[1])}; blo = uint2{__float_as_uint(fb[2]), __float_as_uint(fb[3])};
#else
        split_pair(fa[0], fa[1], ahi.x, alo.x);
        split_pair(fa[2], fa[3], ahi.y, alo.y);
        split_pair(fb[0], fb[1], bhi.x, blo.x);
        split_pair(fb[2], fb[3], bhi.y, blo.y);
#endif
        *reinterpret_cast<uint2 *>(As + k * 32 * 16) = ahi;
        *reinterpret_cast<uint2 *>(As + kPPart + k * 32 * 16) = alo;
        *reinterpret_cast<uint2 *>(Bs + k * 32 * 16) = bhi;
        *reinterpret_cast<uint2 *>(Bs + kPPart + k * 32 * 16) = blo;
      }
    };
    if (n > 0) {
      Regs R0, R1;
      issue(R0);
      issue(R1);
      commit(0, R0);
      issue(R0);
      __syncthreads();
      for (int j = 0; j < n; j += 2) {
        if (j + 1 < n) commit(1, R1);       // stage j + 1
        issue(R1);                          // stage j + 3
        __syncthreads();
        if (j + 1 < n) {
          if (j + 2 < n) commit(0, R0);     // stage j + 2
          issue(R0);                        // stage j + 4
          __syncthreads();
        }
      }
    }
    if (rs_on) {    // the eight pieces of a row segment sit in adjacent lanes: fixed order
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = rs[k];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const int m = mt * kTM + rp + 32 * k;
        if (q == 0 && m < M) row_sums[(long long)m * rs_stride + split] = v;
      }
    }
  } else {
    const int wm = wave & 1, wn = wave >> 1;           // rows wm * 64 .. + 63, columns wn * 32 * kPNI .. + 32 * kPNI - 1
    f32x16 acc[2][kPNI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < kPNI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    const int fo = (lane >> 5) * kPKH + (lane & 31) * 16;
    auto multiply = [&](int buf) __attribute__((always_inline)) {
      const unsigned char *A = smem + buf * kPStage + fo + wm * 64 * 16;
      const unsigned char *Bp = smem + buf * kPStage + kPOperand + fo + wn * (32 * kPNI) * 16;
      bf16x8 fa[2][2][2], fb[2][2][kPNI];     // [step][part][block]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[ks][pt][i] = *reinterpret_cast<const bf16x8 *>(A + ks * kPKS + pt * kPPart + i * 32 * 16);
#pragma unroll
          for (int i = 0; i < kPNI; ++i) fb[ks][pt][i] = *reinterpret_cast<const bf16x8 *>(Bp + ks * kPKS + pt * kPPart + i * 32 * 16);
        }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // small terms first; the other accumulators' MFMAs between two on the same one
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < kPNI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][1][mi], fb[ks][0][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < kPNI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0][mi], fb[ks][1][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < kPNI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0][mi], fb[ks][0][ni], acc[mi][ni], 0, 0, 0);
      }
    };
    if (n > 0) {
      __syncthreads();
      for (int j = 0; j < n; j += 2) {
        multiply(0);
        __syncthreads();
        if (j + 1 < n) {
          multiply(1);
          __syncthreads();
        }
      }
    }
    float *out = partial + (long long)split * M * N;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < kPNI; ++ni) {
        const int nn = nt * kTN + wn * (32 * kPNI) + ni * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M && nn < N) out[(long long)m * N + nn] = acc[mi][ni][r];
        }
      }
  }
}

template <int TAPS, bool ALIGNED>
__global__ __launch_bounds__(kPThreads, 1) void conv_ntp(const float *__restrict__ a, const float *__restrict__ bm,
                                                         float *__restrict__ partial, int M, int N, int L, int B, int n_mt,
                                                         int n_nt, int stages_per_image, int per, int H, int W, int Cin,
                                                         int units, float *__restrict__ row_sums = nullptr, int rs_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int unit = units > 0 ? xcd_tile(blockIdx.x, units) : (int)blockIdx.x;      // (as conv_nt8)
  if (units > 0 && unit >= units) return;
  if (threadIdx.x >= kPCons * 64)
    conv_ntp_role<TAPS, true, ALIGNED>(a, bm, partial, M, N, L, B, n_mt, n_nt, stages_per_image, per, H, W, Cin, unit, smem, row_sums, rs_stride);
  else
    conv_ntp_role<TAPS, false, ALIGNED>(a, bm, partial, M, N, L, B, n_mt, n_nt, stages_per_image, per, H, W, Cin, unit, smem, row_sums, rs_stride);
}

namespace {

// measured on MI355X (tools/bench_conv1x1_wgrad.py): one workgroup per CU with >= 32 stages each beats finer cuts --
// every extra split is another [M, N] partial written and re-read
int nt_splits(int tiles, int total_stages) {
  // Rounds over the 256 CUs decide: a CU works through a tile's stages at a fixed rate whatever shares it (the chip sits at
  // its 1.4 kW limit in these kernels), so 261 workgroups take two rounds where 252 slightly longer ones take one (3x3,
  // 128 channels: 107 -> 86 us).  Time ~ ceil(tiles * s / 256) / s; ties go to the finer split.
  static const int round_up = [] { const char *e = getenv("KGDET_NT_SPLITS_UP"); return e ? atoi(e) : 0; }();   // 1: the old rule (A/B)
  const int most = (total_stages + 31) / 32;
  int splits = (256 + tiles - 1) / tiles;
  if (!round_up) {
    double best = 1e30;
    const int s_max = 2 * ((256 + tiles - 1) / tiles) < 128 ? 2 * ((256 + tiles - 1) / tiles) : 128;
    for (int s = 1; s <= s_max; ++s) {
      const double t = (double)((tiles * s + 255) / 256) / s;
      if (t < best) best = t;
    }
    // among the best: the coarsest split that still fills 7/8 of the CUs (fewer partials to add), else the finest
    splits = 0;
    for (int s = 1; s <= s_max && !splits; ++s)
      if ((double)((tiles * s + 255) / 256) / s <= best * (1.0 + 1e-9) && tiles * s >= 224 && tiles * s <= 256) splits = s;
    for (int s = s_max; s >= 1 && !splits; --s)
      if ((double)((tiles * s + 255) / 256) / s <= best * (1.0 + 1e-9)) splits = s;
  }
  if (splits > most) splits = most;
  if (splits > 128) splits = 128;
  return splits < 1 ? 1 : splits;
}

// launch shape of conv_nt8: XCD-contiguous unit order (KGDET_NT_XCD=0: workgroup b = unit b, the order up to round 3)
bool nt_xcd() {
  static const bool on = [] { const char *e = getenv("KGDET_NT_XCD"); return !e || atoi(e) != 0; }();   // A/B switch
  return on;
}
int nt_units(int units) { return nt_xcd() ? units : 0; }
int nt_grid(int units) { return nt_xcd() ? (units + 7) / 8 * 8 : units; }

// conv_ntp (producer / consumer waves) instead of conv_nt8; KGDET_NT_PC=0: the kernel up to round 3 (A/B)
// Measured (tools/bench_conv3x3_wgrad.py / bench_conv1x1_wgrad.py, with the sum pass): 3x3 76.2 / 81.7 / 99.9 / 43.2 us ->
// 66.0 / 70.6 / 89.4 / 41.4 us; 1x1 3-5 % SLOWER (31.7 -> 33.4 us ...: ~17 long stages per workgroup, fill and drain weigh
// more than the leaner stage) -- so the 3x3 problems take conv_ntp and the 1x1 problems stay on conv_nt8.  KGDET_NT_PC=0 / 2:
// neither / both.
// Maps whose rows / pixel count are not a multiple of 4 (25 x 42, 13 x 21, 7 x 11) always take conv_ntp when it is on at all: it
// reads them in place (4-byte aligned 16-byte loads), conv_nt8 needs zero-padded copies of both operands (pad_rows2).
bool ntp_on(int taps, bool odd = false) {
  static const int mode = [] { const char *e = getenv("KGDET_NT_PC"); return e ? atoi(e) : 1; }();
  return mode == 2 || (mode == 1 && (taps == 9 || odd));
}
int ntp_lds() { return kPLds; }
int ntp_attr() {
  static thread_local bool set = false;
  if (!set) {
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)conv_ntp<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ntp_lds()));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)conv_ntp<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ntp_lds()));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)conv_ntp<9, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ntp_lds()));
    KGDET_HIP_TRY(hipFuncSetAttribute((const void *)conv_ntp<9, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ntp_lds()));
    set = true;
  }
  return KGDET_OK;
}

// K parts of the NN kernel: only when the tiles alone leave most CUs idle
int nn_ksplit(long long tiles, int stages) {
  if (tiles >= 200) return 1;
  int k = (int)((384 + tiles - 1) / tiles);
  const int most = stages / 8;                        // at least 8 stages per part
  if (k > most) k = most;
  if (k > 8) k = 8;
  return k < 1 ? 1 : k;
}

// conv_nn or conv3x3_patch, the pixel tiling and the K split of one convolution (workspace query and launch agree on it)
struct NNPlan {
  bool patch;
  int TX, TY, NB, NW, tiles_x, n_nt, ks;
  long long tiles;
};
int conv_patch_mode() {   // 2: conv3x3_patch4 (default), 1: conv3x3_patch, 0: conv_nn<9>  (A/B)
  static const int on = [] { const char *e = getenv("KGDET_CONV3X3_PATCH"); return e ? atoi(e) : 2; }();
  return on;
}
bool conv_patch_enabled() { return conv_patch_mode() != 0; }
NNPlan plan_nn(long long B, int M, int K, int H, int W, int taps, int stride) {
  NNPlan p;
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const int n_mt = (M + kTM - 1) / kTM;
  p.patch = false;
  p.TX = p.TY = p.tiles_x = 0;
  p.NB = p.NW = 4;
  p.n_nt = (int)(((long long)Ho * Wo + kTN - 1) / kTN);
  if (taps == 9 && stride == 1 && conv_patch_enabled()) {
    // the tile shape with the fewest tiles (ties: the smaller patch); TX >= 8 keeps the stores in >= 32-byte runs
    long long best = -1;
    p.NB = 4;
    for (int cap = 128; cap <= (conv_patch_mode() >= 2 && M > 64 ? 160 : 128); cap += 32)   // (M <= 64: the 64-row variant, 4 blocks)
      for (int tx = 8; tx <= 126 && tx <= (W + 7) / 8 * 8; ++tx) {
        const int ty = cap / tx;
        if (ty < 1 || (tx + 2) * (ty + 2) > kPatchMax) continue;
        const long long n = (long long)((H + ty - 1) / ty) * ((W + tx - 1) / tx);
        // time ~ rounds over the 256 CUs x blocks per tile (one CU, one tile at a time: see conv3x3_patch4); below one round
        // and in the many-round regime the tile count itself decides
        const long long units = n * n_mt * B;
        const long long rounds = units <= 256 ? 256 : units <= 512 ? 512 : units;
        const long long cost = (rounds * (cap / 32)) * 4096 + n * 8 + (cap == 160) * 4 + ((tx + 2) * (ty + 2) > 230);
        if (best < 0 || cost < best) { best = cost; p.TX = tx; p.TY = ty; p.NB = cap / 32; }
      }
    static const int force_tx = [] { const char *e = getenv("KGDET_CONV_TX"); return e ? atoi(e) : 0; }();   // experiments
    if (force_tx > 0 && (force_tx + 2) * (128 / force_tx + 2) <= kPatchMax) { p.TX = force_tx; p.TY = 128 / force_tx; p.NB = 4; }
    if (best >= 0) {
      p.patch = true;
      p.tiles_x = (W + p.TX - 1) / p.TX;
      p.n_nt = p.tiles_x * ((H + p.TY - 1) / p.TY);
    }
  }
  if (!p.patch) {
    // conv_nn: 128- or 160-pixel tiles by the number of tiles the fullest of the 256 CUs draws (x the tile's width); only where
    // no K split is needed anyway, and only for a clear win
    static const int force_nw = [] { const char *e = getenv("KGDET_CONV_NW"); return e ? atoi(e) : 0; }();   // 4 | 5 (A/B)
    const long long hw = (long long)Ho * Wo;
    const long long u4 = (long long)n_mt * ((hw + 127) / 128) * B, u5 = (long long)n_mt * ((hw + 159) / 160) * B;
    const long long c4 = ((u4 + 255) / 256) * 4, c5 = ((u5 + 255) / 256) * 5;
    if (force_nw == 5 || (force_nw == 0 && u4 >= 200 && u5 >= 200 && c5 * 10 <= c4 * 9)) {
      p.NW = 5;
      p.n_nt = (int)((hw + 159) / 160);
    }
  }
  p.tiles = (long long)n_mt * p.n_nt * B;
  p.ks = nn_ksplit(p.tiles, taps * ((K + kTK - 1) / kTK));
  if (p.patch && p.tiles < 200) {
    // patch kernels: parts of whole chunks (>= 2 each), chosen by rounds over the CUs x chunks per part (+ the partials to add)
    static const int old_rule = [] { const char *e = getenv("KGDET_CONV_KS_OLD"); return e ? atoi(e) : 0; }();   // A/B
    const int chunks = K / kTK;
    double best = 1e30;
    p.ks = 1;
    for (int ks = 1; ks <= 8 && ks * 2 <= chunks; ++ks) {
      // (a chunk of a tile is ~3.5 us of a CU; a partial is written and read once: ~0.114 chunk times per MB of output)
      const double out_mb = (double)B * M * Ho * Wo * 4e-6;
      const double cost = (double)((p.tiles * ks + 255) / 256) * ((chunks + ks - 1) / ks) + 0.114 * out_mb * ks;
      if (cost < best) { best = cost; p.ks = ks; }
    }
    if (old_rule) p.ks = nn_ksplit(p.tiles, chunks * 2);
  }
  static const int force_ks = [] { const char *e = getenv("KGDET_CONV_KS"); return e ? atoi(e) : 0; }();   // experiments
  if (force_ks > 0) p.ks = force_ks;
  return p;
}

}  // namespace

}  // namespace kgdet

using namespace kgdet;

#ifdef KGDET_CONV_TRACE
extern "C" int kgdet_debug_read_conv_trace(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kgdet::g_conv_trace), sizeof(unsigned long long) * 1024 * 8);
}
#endif

extern "C" size_t kgdet_conv_packed_bytes(int32_t M, int32_t K, int32_t taps) {
  if (M <= 0 || K <= 0 || (taps == 9 && K % kTK) || (taps != 1 && taps != 9)) return 0;
  return (size_t)((M + kTM - 1) / kTM) * ((K + kTK - 1) / kTK) * taps * kStage;
}

extern "C" int kgdet_conv_pack_fmt(const float *w, int32_t O, int32_t C, int32_t taps, int32_t transpose, void *packed,
                                   int32_t operand_format, void *stream) {
  // weight [O, C, taps]; transpose = 0: rows O, reduction C (forward); 1: rows C, reduction O, taps mirrored (grad_input)
  const int M = transpose ? C : O, K = transpose ? O : C;
  KGDET_CHECK_SHAPE(taps == 1 || taps == 9, "taps must be 1 (1x1) or 9 (3x3)");
  KGDET_CHECK_SHAPE(O > 0 && C > 0 && (taps == 1 || K % kTK == 0), "reduction length %d is not a multiple of 16", K);
  KGDET_CHECK_SHAPE(w && packed, "null pointer");
  const long long total = (long long)((M + kTM - 1) / kTM) * ((K + kTK - 1) / kTK) * taps * 2 * kTM;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(conv1x1_pack, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream, w,
                     O, C, taps, transpose, (unsigned char *)packed, (unsigned char *)nullptr, operand_format == 1 ? 1 : 0);
  KGDET_CHECK_LAUNCH("conv_pack");
  return KGDET_OK;
}

extern "C" int kgdet_conv_pack(const float *w, int32_t O, int32_t C, int32_t taps, int32_t transpose, void *packed,
                               void *stream) {
  return kgdet_conv_pack_fmt(w, O, C, taps, transpose, packed, 0, stream);
}

extern "C" int kgdet_conv_pack_both_fmt(const float *w, int32_t O, int32_t C, int32_t taps, void *packed, void *packed_t,
                                        int32_t forward_format, void *stream) {
  KGDET_CHECK_SHAPE(taps == 1 || taps == 9, "taps must be 1 (1x1) or 9 (3x3)");
  KGDET_CHECK_SHAPE(O > 0 && C > 0 && (taps == 1 || (O % kTK == 0 && C % kTK == 0)), "O and C must be multiples of 16");
  KGDET_CHECK_SHAPE(w && packed && packed_t, "null pointer");
  const long long t0 = (long long)((O + kTM - 1) / kTM) * ((C + kTK - 1) / kTK) * taps * 2 * kTM;
  const long long t1 = (long long)((C + kTM - 1) / kTM) * ((O + kTK - 1) / kTK) * taps * 2 * kTM;
  const long long blocks = ((t0 > t1 ? t0 : t1) + 255) / 256;
  hipLaunchKernelGGL(conv1x1_pack, dim3((unsigned)(blocks > 32768 ? 32768 : blocks), 2), dim3(256), 0,
                     (hipStream_t)stream, w, O, C, taps, 0, (unsigned char *)packed, (unsigned char *)packed_t,
                     forward_format == 1 ? 1 : 0);
  KGDET_CHECK_LAUNCH("conv_pack_both");
  return KGDET_OK;
}

extern "C" int kgdet_conv_pack_both(const float *w, int32_t O, int32_t C, int32_t taps, void *packed, void *packed_t,
                                    void *stream) {
  return kgdet_conv_pack_both_fmt(w, O, C, taps, packed, packed_t, 0, stream);
}

extern "C" int64_t kgdet_conv_pack_blocks(int32_t O, int32_t C, int32_t taps) {
  if (O <= 0 || C <= 0 || (taps == 9 && (O % kTK || C % kTK)) || (taps != 1 && taps != 9)) return 0;
  const long long t0 = (long long)((O + kTM - 1) / kTM) * ((C + kTK - 1) / kTK) * taps * 2 * kTM;
  const long long t1 = (long long)((C + kTM - 1) / kTM) * ((O + kTK - 1) / kTK) * taps * 2 * kTM;
  return ((t0 > t1 ? t0 : t1) + 255) / 256;
}

extern "C" int kgdet_conv_pack_multi(const int64_t *desc_dev, int32_t n, int64_t total_blocks, void *stream) {
  KGDET_CHECK_SHAPE(desc_dev && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "bad descriptor table");
  hipLaunchKernelGGL(conv1x1_pack_multi, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const long long *)desc_dev, n);
  KGDET_CHECK_LAUNCH("conv_pack_multi");
  return KGDET_OK;
}

// H, W below are the INPUT map; the output map is ceil(H / stride) x ceil(W / stride) (1x1: padding 0, 3x3: padding 1)
extern "C" size_t kgdet_conv_apply_workspace_bytes(int64_t B, int32_t M, int32_t K, int32_t H, int32_t W,
                                                    int32_t taps, int32_t stride) {
  if (B <= 0 || M <= 0 || K <= 0 || H <= 0 || W <= 0 || (taps == 9 && K % kTK) || stride < 1 || stride > 2) return 0;
  const long long HW = (long long)((H + stride - 1) / stride) * ((W + stride - 1) / stride);
  const int ks = plan_nn(B, M, K, H, W, taps, stride).ks;
  return ks > 1 ? (size_t)ks * B * M * HW * sizeof(float) : 0;
}

extern "C" int kgdet_bias_act(void *x, const float *bias, const void *residual, int64_t N, int32_t C, int64_t HW,
                              int32_t dtype, int32_t relu, int32_t channels_last, void *stream);

extern "C" int kgdet_conv_apply_gated_fmt(const void *packed, const float *x, float *y, const float *bias,
                                          const float *residual, int32_t relu, const float *gate, int64_t B, int32_t M,
                                          int32_t K, int32_t H, int32_t W, int32_t taps, int32_t stride,
                                          int32_t operand_format, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int kgdet_conv_apply_epilogue_fmt(const void *packed, const float *x, float *y, const float *bias,
                                             const float *residual, int32_t relu, int64_t B, int32_t M, int32_t K,
                                             int32_t H, int32_t W, int32_t taps, int32_t stride, int32_t operand_format,
                                             void *workspace, size_t workspace_bytes, void *stream) {
  return kgdet_conv_apply_gated_fmt(packed, x, y, bias, residual, relu, nullptr, B, M, K, H, W, taps, stride, operand_format,
                                    workspace, workspace_bytes, stream);
}

// ... and `gate` [B, M, Ho, Wo] (or NULL): y = [gate > 0] * ([relu](conv + bias [+ residual])).  In the backward of
// `z = relu(...); u = conv(z)` the gradient of z is conv_grad_input(grad_u) [+ the identity branch's gradient = `residual`], and
// the node that produced z masks it with [z > 0] first thing: with gate = z (this convolution's own forward input) the mask rides
// on this kernel's store and that node's pass over the activation (read gradient, read z, write masked gradient) is not taken.
extern "C" int kgdet_conv_apply_gated_fmt(const void *packed, const float *x, float *y, const float *bias,
                                          const float *residual, int32_t relu, const float *gate, int64_t B, int32_t M,
                                          int32_t K, int32_t H, int32_t W, int32_t taps, int32_t stride,
                                          int32_t operand_format, void *workspace, size_t workspace_bytes, void *stream) {
  const bool f16 = operand_format == 1;      // the image and the on-the-fly split of x in fp16 parts (forward operands)
  KGDET_CHECK_SHAPE(B >= 0 && M > 0 && K > 0 && H >= 0 && W >= 0 && (long long)H * W < (1LL << 30), "bad sizes");
  KGDET_CHECK_SHAPE(taps == 1 || taps == 9, "taps must be 1 (1x1) or 9 (3x3)");
  KGDET_CHECK_SHAPE(stride == 1 || stride == 2, "stride must be 1 or 2");
  KGDET_CHECK_SHAPE(taps == 1 || K % kTK == 0, "reduction length %d is not a multiple of 16", K);
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const long long HW = (long long)Ho * Wo;
  if (B * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(packed && x && y, "null pointer");
  const NNPlan plan = plan_nn(B, M, K, H, W, taps, stride);
  const int n_mt = (M + kTM - 1) / kTM, n_nt = plan.n_nt;
  const long long tiles = plan.tiles;
  KGDET_CHECK_SHAPE(tiles < (1LL << 28), "too many tiles");
  const int ks = plan.ks;
  const long long part_stride = B * M * HW;
  if (ks > 1) {
    KGDET_CHECK_SHAPE(workspace && workspace_bytes >= (size_t)ks * part_stride * sizeof(float), "workspace too small");
    KGDET_CHECK_SHAPE(part_stride % 2 == 0, "B*M*Ho*Wo must be even");
    KGDET_CHECK_SHAPE(!gate || HW % 2 == 0, "a gated K-split convolution needs an even pixel count");
  }
  const int per = (int)((tiles * ks + 7) / 8);
  float *dst = ks > 1 ? (float *)workspace : y;
  if (plan.patch && (conv_patch_mode() >= 2 || f16)) {
    // fewer than ~1.6 whole-tile workgroups per CU: 64-row halves (twice the workgroups) spread evenly over the CUs
    static const int force_halves = [] { const char *e = getenv("KGDET_CONV_HALVES"); return e ? atoi(e) : -1; }();
    // ... and M <= 64 (layer 1): the second half has no rows and leaves at once instead of multiplying zeros
    const bool halves = force_halves >= 0 ? force_halves != 0 : ((ks == 1 && tiles > 256 && tiles < 400) || M <= 64);
#define KGDET_P4_ARGS (const unsigned char *)packed, x, dst, M, K, H, W, n_mt, plan.tiles_x, n_nt, (int)tiles, ks, part_stride, \
                      plan.TX, plan.TY, ks > 1 ? nullptr : bias, ks > 1 ? nullptr : residual, ks > 1 ? 0 : relu, \
                      ks > 1 ? nullptr : gate
#define KGDET_P4_LAUNCH(WV, NBK, GRID, THR)                                                                                  \
    do {                                                                                                                     \
      if (f16) hipLaunchKernelGGL((conv3x3_patch4<WV, NBK, true>), dim3(GRID), dim3(THR), 0, (hipStream_t)stream, KGDET_P4_ARGS); \
      else hipLaunchKernelGGL((conv3x3_patch4<WV, NBK, false>), dim3(GRID), dim3(THR), 0, (hipStream_t)stream, KGDET_P4_ARGS);    \
    } while (0)
    if (plan.NB == 5)
      KGDET_P4_LAUNCH(4, 5, per * 8, 256);
    else if (halves)
      KGDET_P4_LAUNCH(2, 4, (int)((tiles * ks * 2 + 7) / 8) * 8, 128);
    else
      KGDET_P4_LAUNCH(4, 4, per * 8, 256);
#undef KGDET_P4_LAUNCH
#undef KGDET_P4_ARGS
  } else if (plan.patch) {
    static thread_local bool attr_set = false;
    if (!attr_set) {
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)conv3x3_patch, hipFuncAttributeMaxDynamicSharedMemorySize, kPatchLds));
      attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_patch, dim3(per * 8), dim3(kNNThreads), kPatchLds, (hipStream_t)stream,
                       (const unsigned char *)packed, x, dst, M, K, H, W, n_mt, plan.tiles_x, n_nt, (int)tiles, ks,
                       part_stride, plan.TX, plan.TY, ks > 1 ? nullptr : bias, ks > 1 ? nullptr : residual,
                       ks > 1 ? 0 : relu, ks > 1 ? nullptr : gate);
  } else {
#define KGDET_NN_ARGS (const unsigned char *)packed, x, dst, M, K, Ho, Wo, n_mt, n_nt, (int)tiles, ks, part_stride, H, W, stride, \
                      ks > 1 ? nullptr : bias, ks > 1 ? nullptr : residual, ks > 1 ? 0 : relu, ks > 1 ? nullptr : gate
#define KGDET_NN_LAUNCH(TP, NWK, THR)                                                                                        \
    do {                                                                                                                     \
      if (f16) hipLaunchKernelGGL((conv_nn<TP, NWK, true>), dim3(per * 8), dim3(THR), 0, (hipStream_t)stream, KGDET_NN_ARGS);  \
      else hipLaunchKernelGGL((conv_nn<TP, NWK, false>), dim3(per * 8), dim3(THR), 0, (hipStream_t)stream, KGDET_NN_ARGS);     \
    } while (0)
    if (taps == 1 && plan.NW == 5)
      KGDET_NN_LAUNCH(1, 5, 640);
    else if (taps == 1)
      KGDET_NN_LAUNCH(1, 4, 512);
    else if (plan.NW == 5)
      KGDET_NN_LAUNCH(9, 5, 640);
    else
      KGDET_NN_LAUNCH(9, 4, 512);
#undef KGDET_NN_LAUNCH
#undef KGDET_NN_ARGS
  }
  KGDET_CHECK_LAUNCH("conv_nn");
  if (ks > 1) {
    const long long blocks = (part_stride / 2 + 255) / 256;
    if ((bias || residual || relu || gate) && HW % 2 == 0)   // the epilogue in the sum's store
      hipLaunchKernelGGL(conv1x1_sum_epilogue, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0,
                         (hipStream_t)stream, (const float *)workspace, y, part_stride, part_stride, ks, bias, residual, relu,
                         M, (long long)HW, gate);
    else
      hipLaunchKernelGGL(conv1x1_sum, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream,
                         (const float *)workspace, y, part_stride, part_stride, ks);
    KGDET_CHECK_LAUNCH("conv1x1_sum");
    if ((bias || residual || relu) && HW % 2 != 0) return kgdet_bias_act(y, bias, residual, B, M, HW, 0, relu, 0, stream);
  }
  return KGDET_OK;
}

extern "C" int kgdet_conv_apply_epilogue(const void *packed, const float *x, float *y, const float *bias,
                                         const float *residual, int32_t relu, int64_t B, int32_t M, int32_t K,
                                         int32_t H, int32_t W, int32_t taps, int32_t stride, void *workspace,
                                         size_t workspace_bytes, void *stream) {
  return kgdet_conv_apply_epilogue_fmt(packed, x, y, bias, residual, relu, B, M, K, H, W, taps, stride, 0, workspace,
                                       workspace_bytes, stream);
}

extern "C" int kgdet_conv3x3_s2_grad_input(const void *packed_t, const float *grad_y, float *grad_x, int64_t B, int32_t C,
                                           int32_t O, int32_t Hin, int32_t Win, void *stream) {
  // packed_t: kgdet_conv_pack(w, O, C, 9, transpose = 1) of the forward weight [O, C, 3, 3]; grad_y [B, O, ceil(Hin/2), ceil(Win/2)]
  KGDET_CHECK_SHAPE(B >= 0 && C > 0 && O > 0 && Hin > 0 && Win > 0 && O % kTK == 0, "bad sizes (O must be a multiple of 16)");
  if (B == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(packed_t && grad_y && grad_x, "null pointer");
  const int H = (Hin + 1) / 2, W = (Win + 1) / 2;
  // 64-pixel tiles of the grad_y map: the shape with the fewest tiles whose patch fits
  int TX = 0, TY = 0;
  long long best = -1;
  for (int tx = 8; tx <= 64; ++tx) {
    const int ty = 64 / tx;
    if (ty < 1 || (tx + 2) * (ty + 2) > kPatchMax) continue;
    const long long n = (long long)((H + ty - 1) / ty) * ((W + tx - 1) / tx);
    const long long cost = n * 1024 + (tx + 2) * (ty + 2);
    if (best < 0 || cost < best) { best = cost; TX = tx; TY = ty; }
  }
  const int n_mt = (C + kTM - 1) / kTM, tiles_x = (W + TX - 1) / TX, n_nt = tiles_x * ((H + TY - 1) / TY);
  const long long tiles = (long long)n_mt * n_nt * B;
  KGDET_CHECK_SHAPE(tiles < (1LL << 28), "too many tiles");
  hipLaunchKernelGGL(conv3x3_s2_grad_input, dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char *)packed_t, grad_y, grad_x, C, O, H, W, Hin, Win, n_mt, tiles_x, n_nt, (int)tiles, TX, TY);
  KGDET_CHECK_LAUNCH("conv3x3_s2_grad_input");
  return KGDET_OK;
}

extern "C" int kgdet_stem_conv7x7_s2_fmt(const void *packed, const float *x, float *y, int64_t B, int32_t H, int32_t W,
                                         int32_t operand_format, void *stream) {
  // packed: kgdet_conv_pack of the [64, 3, 7, 7] weight flattened to [64, 147] and zero-padded to [64, 160] (taps = 1,
  // transpose = 0); x [B, 3, H, W] -> y [B, 64, (H - 1) / 2 + 1, (W - 1) / 2 + 1]
  KGDET_CHECK_SHAPE(B >= 0 && H > 0 && W > 0, "bad sizes");
  if (B == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(packed && x && y, "null pointer");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int tiles_x = (Wo + kStemTX - 1) / kStemTX, tiles_per_image = tiles_x * ((Ho + kStemTY - 1) / kStemTY);
  KGDET_CHECK_SHAPE((long long)B * tiles_per_image < (1LL << 31), "too many tiles");
  if (operand_format == 1)
    hipLaunchKernelGGL(stem_conv7x7_s2<true>, dim3((unsigned)(B * tiles_per_image)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char *)packed, x, y, H, W, Ho, Wo, tiles_x, tiles_per_image);
  else
    hipLaunchKernelGGL(stem_conv7x7_s2<false>, dim3((unsigned)(B * tiles_per_image)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char *)packed, x, y, H, W, Ho, Wo, tiles_x, tiles_per_image);
  KGDET_CHECK_LAUNCH("stem_conv7x7_s2");
  return KGDET_OK;
}

extern "C" int kgdet_stem_conv7x7_s2(const void *packed, const float *x, float *y, int64_t B, int32_t H, int32_t W,
                                     void *stream) {
  return kgdet_stem_conv7x7_s2_fmt(packed, x, y, B, H, W, 0, stream);
}

extern "C" int kgdet_conv_apply(const void *packed, const float *x, float *y, int64_t B, int32_t M, int32_t K, int32_t H,
                                int32_t W, int32_t taps, int32_t stride, void *workspace, size_t workspace_bytes,
                                void *stream) {
  return kgdet_conv_apply_epilogue(packed, x, y, nullptr, nullptr, 0, B, M, K, H, W, taps, stride, workspace,
                                   workspace_bytes, stream);
}

// rows of W floats -> rows of Wp floats, zero tail, for two tensors in one launch (blockIdx.y: 0 = a, 1 = b)
__global__ __launch_bounds__(256) void pad_rows2(const float *__restrict__ a, float *__restrict__ ap, long long rows_a,
                                                 const float *__restrict__ b, float *__restrict__ bp, long long rows_b, int W,
                                                 int Wp) {
  const float *src = blockIdx.y ? b : a;
  float *dst = blockIdx.y ? bp : ap;
  const long long n = (blockIdx.y ? rows_b : rows_a) * Wp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / Wp;
    const int c = (int)(i - r * Wp);
    dst[i] = c < W ? src[r * W + c] : 0.0f;
  }
}

// threads of conv_wsum_fold for a row of `cols` columns: one column per thread up to 1024 (whole waves)
static int fold_threads(int cols) {
  const int t = (cols + 63) / 64 * 64;
  return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

// workspace = [split partials | padded operand copies (maps with H*W % 4 != 0 on the conv_nt8 route) | per-row sums of grad_y,
// [O][splits], for the folded variant called without bn_partial]; *rows_at = byte offset of the last region
static size_t conv1x1_gw_workspace(int64_t B, int32_t O, int32_t C, int64_t HW, size_t *rows_at) {
  const long long HWp = (HW + 3) & ~3LL;
  const int tiles = ((O + kTM - 1) / kTM) * ((C + kTN - 1) / kTN);
  const int stages = (int)(B * ((HWp + kTK - 1) / kTK));
  const int splits = nt_splits(tiles, stages);
  size_t bytes = (size_t)splits * O * C * sizeof(float);
  if (HWp != HW) bytes = ((bytes + 255) & ~(size_t)255) + (size_t)B * (O + C) * HWp * sizeof(float);   // padded copies of grad_y and x
  bytes = (bytes + 255) & ~(size_t)255;
  if (rows_at) *rows_at = bytes;
  return bytes + (size_t)splits * O * sizeof(float);
}

extern "C" size_t kgdet_conv1x1_grad_weight_workspace_bytes(int64_t B, int32_t O, int32_t C, int64_t HW) {
  if (B <= 0 || O <= 0 || C <= 0 || HW <= 0) return 0;
  return conv1x1_gw_workspace(B, O, C, HW, nullptr);
}

static int conv1x1_grad_weight_impl(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C,
                                    int64_t HW, void *workspace, size_t workspace_bytes, void *stream,
                                    const ConvFoldArgs *fold, int wsum3x3_C = 0) {
  // wsum3x3_C > 0: the columns are (tap, channel) of a 3x3 problem with wsum3x3_C channels (C = 9 * wsum3x3_C): the final sum
  // writes grad_w [O, wsum3x3_C, 3, 3] (kgdet_conv3x3_s2_grad_weight)
  KGDET_CHECK_SHAPE(B > 0 && O > 0 && C > 0 && HW > 0 && HW < (1LL << 30), "bad sizes");
  KGDET_CHECK_SHAPE(grad_y && x && grad_w && workspace, "null pointer");
  KGDET_CHECK_SHAPE(workspace_bytes >= kgdet_conv1x1_grad_weight_workspace_bytes(B, O, C, HW), "workspace too small");
  const long long HW_true = HW;
  size_t rows_at = 0;
  conv1x1_gw_workspace(B, O, C, HW, &rows_at);
  // the folded variant without bn_partial: the kernels below also write the per-row sums of grad_y (one slot per split)
  float *row_sums = (fold && !fold->bn_partial) ? reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + rows_at) : nullptr;
  const bool use_ntp = ntp_on(1, HW % 4 != 0) && HW >= 4 && (long long)(O > C ? O : C) * HW < (1ll << 30);
  if (HW % 4 && use_ntp) {
    HW = (HW + 3) & ~3LL;      // (the split count below is the one the workspace query computed for the padded size)
  } else if (HW % 4) {
    // conv_nt8's piece logic needs H*W % 4 == 0 (25 x 42 = 1050): both operands go into the workspace with zero pixels up to a
    // multiple of 4 (one launch; zero grad_y pixels contribute nothing).  The 8-byte-load kernel this replaces lost to
    // MIOpen's GEMM (50-54 against 41-47 us).
    const long long HWp = (HW + 3) & ~3LL;
    const int tiles_ = ((O + kTM - 1) / kTM) * ((C + kTN - 1) / kTN);
    const int stages_ = (int)(B * ((HWp + kTK - 1) / kTK));
    const size_t part = ((size_t)nt_splits(tiles_, stages_) * O * C * sizeof(float) + 255) & ~(size_t)255;
    float *gyp = reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + part);
    float *xp = gyp + (size_t)B * O * HWp;
    const long long rows_a = (long long)B * O, rows_b = (long long)B * C;
    const long long most = (rows_a > rows_b ? rows_a : rows_b) * HWp;
    const long long blocks = (most + 255) / 256;
    hipLaunchKernelGGL(pad_rows2, dim3((unsigned)(blocks > 4096 ? 4096 : blocks), 2), dim3(256), 0, (hipStream_t)stream,
                       grad_y, gyp, rows_a, x, xp, rows_b, (int)HW, (int)HWp);
    KGDET_CHECK_LAUNCH("pad_rows2");
    grad_y = gyp;
    x = xp;
    HW = HWp;
  }
  const int n_mt = (O + kTM - 1) / kTM, n_nt = (C + kTN - 1) / kTN, tiles = n_mt * n_nt;
  const int spi = (int)((HW + kTK - 1) / kTK), total = (int)(B * spi);
  const int splits = nt_splits(tiles, total);
  const int per = (total + splits - 1) / splits;
  KGDET_CHECK_SHAPE(((long long)O * C) % 2 == 0, "O*C must be even");
  if (use_ntp) {
    if (int rc = ntp_attr()) return rc;
    const int spi32 = (int)((HW_true + kPK - 1) / kPK), total32 = (int)(B * spi32), per32 = (total32 + splits - 1) / splits;
    if (HW_true % 4 == 0)
      hipLaunchKernelGGL((conv_ntp<1, true>), dim3(nt_grid(tiles * splits)), dim3(kPThreads), ntp_lds(), (hipStream_t)stream, grad_y, x,
                         (float *)workspace, O, C, (int)HW_true, (int)B, n_mt, n_nt, spi32, per32, 1, (int)HW_true, 0,
                         nt_units(tiles * splits), row_sums, splits);
    else
      hipLaunchKernelGGL((conv_ntp<1, false>), dim3(nt_grid(tiles * splits)), dim3(kPThreads), ntp_lds(), (hipStream_t)stream, grad_y, x,
                         (float *)workspace, O, C, (int)HW_true, (int)B, n_mt, n_nt, spi32, per32, 1, (int)HW_true, 0,
                         nt_units(tiles * splits), row_sums, splits);
    KGDET_CHECK_LAUNCH("conv_ntp<1>");
  } else {
    hipLaunchKernelGGL(conv_nt8<1>, dim3(nt_grid(tiles * splits)), dim3(kNNThreads), 0, (hipStream_t)stream, grad_y, x,
                       (float *)workspace, O, C, (int)HW, (int)B, n_mt, n_nt, spi, per, 1, (int)HW, 0, nt_units(tiles * splits),
                       row_sums, splits);
    KGDET_CHECK_LAUNCH("conv_nt8<1>");
  }
  const long long n = (long long)O * C;
  if (fold) {
    ConvFoldArgs f = *fold;
    if (row_sums) { f.bn_partial = row_sums; f.P = splits; }
    hipLaunchKernelGGL(conv_wsum_fold<false>, dim3(O), dim3(fold_threads(C)), 0, (hipStream_t)stream,
                       (const float *)workspace, grad_w, C, n, splits, f);
    KGDET_CHECK_LAUNCH("conv_wsum_fold");
    return KGDET_OK;
  }
  if (wsum3x3_C > 0) {
    const long long blocks9 = (n + 255) / 256;
    hipLaunchKernelGGL(conv3x3_wsum, dim3((unsigned)(blocks9 > 4096 ? 4096 : blocks9)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, grad_w, O, wsum3x3_C, splits);
    KGDET_CHECK_LAUNCH("conv3x3_wsum");
    return KGDET_OK;
  }
  const long long blocks = (n / 2 + 255) / 256;
  hipLaunchKernelGGL(conv1x1_sum, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const float *)workspace, grad_w, n, n, splits);
  KGDET_CHECK_LAUNCH("conv1x1_sum");
  return KGDET_OK;
}

// ---- 3x3 stride-2 padding-1 weight gradient (the bottleneck's conv2 at the head of layers 2-4; the FPN's extra levels) -------------
// grad_w[o][c][ky][kx] = sum_{b, oy, ox} grad_y[b][o][oy][ox] * x[b][c][2 oy + ky - 1][2 ox + kx - 1].  The nine strided views of x are
// gathered once into col [B][(tap, channel)][Ho * Wo] (zero outside the image: 9/4 of x's bytes), and the product over the pixels is the
// 1x1 weight-gradient GEMM with 9 C columns -- conv_nt8<1> / conv_ntp<1>, K split, partials added in slot order by conv3x3_wsum, which
// also turns the (tap, channel) columns into grad_w's (channel, tap) order.  Replaces MIOpen's fp32 `igemm_wrw` + its layout transposes
// (366 us per KGDet step for three convolutions), the last vendor kernels of the training step.
__global__ __launch_bounds__(256) void conv_s2_gather9(const float *__restrict__ x, float *__restrict__ col, int C, int H, int W,
                                                       int Ho, int Wo, long long rows) {
  const int HWo = Ho * Wo;
  const long long total = rows * HWo;           // rows = B * C
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / HWo;                // b * C + c
    const int p = (int)(i - r * HWo), oy = p / Wo, ox = p - oy * Wo;
    const long long b = r / C;
    const int c = (int)(r - b * C);
    const float *xp = x + r * (long long)H * W;
    float *cp = col + (b * 9 * C + c) * (long long)HWo + p;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = 2 * oy + t / 3 - 1, xx = 2 * ox + t % 3 - 1;
      cp[(long long)t * C * HWo] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xp[(long long)yy * W + xx] : 0.0f;
    }
  }
}

static size_t conv3x3_s2_col_bytes(int64_t B, int32_t C, int32_t H, int32_t W) {
  const long long HWo = (long long)((H + 1) / 2) * ((W + 1) / 2);
  return ((size_t)B * 9 * C * HWo * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" size_t kgdet_conv3x3_s2_grad_weight_workspace_bytes(int64_t B, int32_t O, int32_t C, int32_t H, int32_t W) {
  if (B <= 0 || O <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  const long long HWo = (long long)((H + 1) / 2) * ((W + 1) / 2);
  return conv3x3_s2_col_bytes(B, C, H, W) + conv1x1_gw_workspace(B, O, 9 * C, HWo, nullptr);
}

extern "C" int kgdet_conv3x3_s2_grad_weight(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C,
                                            int32_t H, int32_t W, void *workspace, size_t workspace_bytes, void *stream) {
  // x [B, C, H, W], grad_y [B, O, ceil(H/2), ceil(W/2)], grad_w [O, C, 3, 3]
  KGDET_CHECK_SHAPE(B > 0 && O > 0 && C > 0 && H > 0 && W > 0 && (long long)9 * C * ((H + 1) / 2) * ((W + 1) / 2) < (1LL << 30), "bad sizes");
  KGDET_CHECK_SHAPE(grad_y && x && grad_w && workspace, "null pointer");
  KGDET_CHECK_SHAPE(workspace_bytes >= kgdet_conv3x3_s2_grad_weight_workspace_bytes(B, O, C, H, W), "workspace too small");
  KGDET_CHECK_SHAPE(((long long)O * C) % 2 == 0, "O*C must be even");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  float *col = static_cast<float *>(workspace);
  const long long total = (long long)B * C * Ho * Wo;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(conv_s2_gather9, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, x, col, C,
                     H, W, Ho, Wo, (long long)B * C);
  KGDET_CHECK_LAUNCH("conv_s2_gather9");
  const size_t cb = conv3x3_s2_col_bytes(B, C, H, W);
  return conv1x1_grad_weight_impl(grad_y, col, grad_w, B, O, 9 * C, (long long)Ho * Wo, static_cast<unsigned char *>(workspace) + cb,
                                  workspace_bytes - cb, stream, nullptr, C);
}

extern "C" int kgdet_conv1x1_grad_weight(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O,
                                         int32_t C, int64_t HW, void *workspace, size_t workspace_bytes,
                                         void *stream) {
  return conv1x1_grad_weight_impl(grad_y, x, grad_w, B, O, C, HW, workspace, workspace_bytes, stream, nullptr);
}

static int fold_args(ConvFoldArgs &f, const float *w, const float *s, const float *mean, const float *var, float eps,
                     const float *bn_partial, int32_t P, float *grad_beta, float *grad_gamma) {
  KGDET_CHECK_SHAPE(w && s && mean && var && ((bn_partial && P > 0) || (!bn_partial && P == 0)),
                    "null pointer (folded BatchNorm arguments; bn_partial == NULL goes with P == 0)");
  f.w = w; f.s = s; f.mean = mean; f.var = var; f.bn_partial = bn_partial; f.grad_beta = grad_beta; f.grad_gamma = grad_gamma;
  f.eps = eps; f.P = P;
  return KGDET_OK;
}

extern "C" int kgdet_conv1x1_grad_weight_fold(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O,
                                              int32_t C, int64_t HW, void *workspace, size_t workspace_bytes, const float *w,
                                              const float *s, const float *mean, const float *var, float eps,
                                              const float *bn_partial, int32_t P, float *grad_beta, float *grad_gamma,
                                              void *stream) {
  ConvFoldArgs f;
  if (int rc = fold_args(f, w, s, mean, var, eps, bn_partial, P, grad_beta, grad_gamma)) return rc;
  return conv1x1_grad_weight_impl(grad_y, x, grad_w, B, O, C, HW, workspace, workspace_bytes, stream, &f);
}

static size_t conv3x3_gw_workspace(int64_t B, int32_t O, int32_t C, int32_t H, int32_t W, size_t *rows_at) {   // (as conv1x1_gw_workspace)
  const int Wp = (W + 3) & ~3;
  const int tiles = ((O + kTM - 1) / kTM) * (9 * C / kTN);
  const int stages = (int)(B * (((long long)H * Wp + kTK - 1) / kTK));
  const int splits = nt_splits(tiles, stages);
  size_t bytes = (size_t)splits * O * C * 9 * sizeof(float);
  if (Wp != W) bytes = ((bytes + 255) & ~(size_t)255) + (size_t)B * (O + C) * H * Wp * sizeof(float);   // padded copies of grad_y and x
  bytes = (bytes + 255) & ~(size_t)255;
  if (rows_at) *rows_at = bytes;
  return bytes + (size_t)splits * O * sizeof(float);
}

extern "C" size_t kgdet_conv3x3_grad_weight_workspace_bytes(int64_t B, int32_t O, int32_t C, int32_t H, int32_t W) {
  if (B <= 0 || O <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return conv3x3_gw_workspace(B, O, C, H, W, nullptr);
}

static int conv3x3_grad_weight_impl(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C,
                                    int32_t H, int32_t W, void *workspace, size_t workspace_bytes, void *stream,
                                    const ConvFoldArgs *fold) {
  KGDET_CHECK_SHAPE(B > 0 && O > 0 && C > 0 && H > 0 && W > 0 && (long long)H * (W + 3) < (1LL << 23), "bad sizes");
  if (C % kTN != 0) {
    set_error("conv3x3_grad_weight needs C %% 128 == 0 (C=%d)", C);
    return KGDET_E_UNSUPPORTED;
  }
  KGDET_CHECK_SHAPE(grad_y && x && grad_w && workspace, "null pointer");
  KGDET_CHECK_SHAPE(workspace_bytes >= kgdet_conv3x3_grad_weight_workspace_bytes(B, O, C, H, W), "workspace too small");
  const int W_true = W;
  size_t rows_at = 0;
  conv3x3_gw_workspace(B, O, C, H, W, &rows_at);
  float *row_sums = (fold && !fold->bn_partial) ? reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + rows_at) : nullptr;
  // (H * W <= 2^21: conv_ntp derives a stage's row as (int)((p0 + 0.5f) * (1.0f / W)) -- exact while p0 < 2^24 and the product's
  //  rounding error stays below half a row; larger maps take conv_nt8, which counts rows)
  const bool use_ntp = ntp_on(9) && W >= 4 && (long long)H * W >= 4 && (long long)H * W <= (1ll << 21) &&
                       (long long)(O > C ? O : C) * H * ((W + 3) & ~3) < (1ll << 30);
  if (W % 4 && use_ntp) {
    W = (W + 3) & ~3;          // (the split count below is the one the workspace query computed for the padded size)
  } else if (W % 4) {
    // conv_nt8's 16-byte row pieces need W % 4 == 0 (25 x 42 head / FPN maps): zero columns on the right of BOTH operands
    // change nothing -- grad_y is 0 there, and x's zeros are what the out-of-range taps read anyway.  One launch pads both
    // into the tail of the workspace.
    const int Wp = (W + 3) & ~3;
    const int tiles_ = ((O + kTM - 1) / kTM) * (9 * C / kTN);
    const int stages_ = (int)(B * (((long long)H * Wp + kTK - 1) / kTK));
    const size_t part = ((size_t)nt_splits(tiles_, stages_) * O * C * 9 * sizeof(float) + 255) & ~(size_t)255;
    float *gyp = reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + part);
    float *xp = gyp + (size_t)B * O * H * Wp;
    const long long rows_a = (long long)B * O * H, rows_b = (long long)B * C * H;
    const long long most = (rows_a > rows_b ? rows_a : rows_b) * Wp;
    const long long blocks = (most + 255) / 256;
    hipLaunchKernelGGL(pad_rows2, dim3((unsigned)(blocks > 4096 ? 4096 : blocks), 2), dim3(256), 0, (hipStream_t)stream,
                       grad_y, gyp, rows_a, x, xp, rows_b, W, Wp);
    KGDET_CHECK_LAUNCH("pad_rows2");
    grad_y = gyp;
    x = xp;
    W = Wp;
  }
  const int HW = H * W;
  const int n_mt = (O + kTM - 1) / kTM, n_nt = 9 * C / kTN, tiles = n_mt * n_nt;
  const int spi = (HW + kTK - 1) / kTK, total = (int)(B * spi);
  const int splits = nt_splits(tiles, total);
  const int per = (total + splits - 1) / splits;
  if (use_ntp) {
    if (int rc = ntp_attr()) return rc;
    const int L = H * W_true;
    const int spi32 = (L + kPK - 1) / kPK, total32 = (int)(B * spi32), per32 = (total32 + splits - 1) / splits;
    if (W_true % 4 == 0)
      hipLaunchKernelGGL((conv_ntp<9, true>), dim3(nt_grid(tiles * splits)), dim3(kPThreads), ntp_lds(), (hipStream_t)stream, grad_y, x,
                         (float *)workspace, O, 9 * C, L, (int)B, n_mt, n_nt, spi32, per32, H, W_true, C, nt_units(tiles * splits),
                         row_sums, splits);
    else
      hipLaunchKernelGGL((conv_ntp<9, false>), dim3(nt_grid(tiles * splits)), dim3(kPThreads), ntp_lds(), (hipStream_t)stream, grad_y, x,
                         (float *)workspace, O, 9 * C, L, (int)B, n_mt, n_nt, spi32, per32, H, W_true, C, nt_units(tiles * splits),
                         row_sums, splits);
    KGDET_CHECK_LAUNCH("conv_ntp<9>");
  } else {
    hipLaunchKernelGGL(conv_nt8<9>, dim3(nt_grid(tiles * splits)), dim3(kNNThreads), 0, (hipStream_t)stream, grad_y, x,
                       (float *)workspace, O, 9 * C, HW, (int)B, n_mt, n_nt, spi, per, H, W, C, nt_units(tiles * splits), row_sums,
                       splits);
    KGDET_CHECK_LAUNCH("conv_nt8<9>");
  }
  const long long n = (long long)O * C * 9;
  if (fold) {
    ConvFoldArgs f = *fold;
    if (row_sums) { f.bn_partial = row_sums; f.P = splits; }
    hipLaunchKernelGGL(conv_wsum_fold<true>, dim3(O), dim3(fold_threads(9 * C)), (size_t)9 * C * sizeof(float),
                       (hipStream_t)stream, (const float *)workspace, grad_w, C, n, splits, f);
    KGDET_CHECK_LAUNCH("conv_wsum_fold");
    return KGDET_OK;
  }
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(conv3x3_wsum, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const float *)workspace, grad_w, O, C, splits);
  KGDET_CHECK_LAUNCH("conv3x3_wsum");
  return KGDET_OK;
}

extern "C" int kgdet_conv3x3_grad_weight(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O,
                                         int32_t C, int32_t H, int32_t W, void *workspace, size_t workspace_bytes,
                                         void *stream) {
  return conv3x3_grad_weight_impl(grad_y, x, grad_w, B, O, C, H, W, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int kgdet_conv3x3_grad_weight_fold(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O,
                                              int32_t C, int32_t H, int32_t W, void *workspace, size_t workspace_bytes,
                                              const float *w, const float *s, const float *mean, const float *var, float eps,
                                              const float *bn_partial, int32_t P, float *grad_beta, float *grad_gamma,
                                              void *stream) {
  ConvFoldArgs f;
  if (int rc = fold_args(f, w, s, mean, var, eps, bn_partial, P, grad_beta, grad_gamma)) return rc;
  return conv3x3_grad_weight_impl(grad_y, x, grad_w, B, O, C, H, W, workspace, workspace_bytes, stream, &f);
}
