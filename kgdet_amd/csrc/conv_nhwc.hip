// 1x1 convolution on channels-last bf16 activations with the whole bottleneck epilogue in its store (gfx950):
//     D[m][n] = [relu]( bf16(sum_k A[m][k] W[n][k]) + bias[n] + R[m][n] )
// A [M][K] (M = B*H*W pixels, K input channels), W [N][K] (the BatchNorm-folded weight of a 1x1 convolution in
// channels-last storage), R / D [M][N] bf16, bias fp32.  This is conv3 + bn3 + the identity add + ReLU of a ResNet
// bottleneck at inference (mmdet/models/backbones/resnet.py:240-262) as ONE kernel.  MIOpen / hipBLASLt produce the
// convolution without the residual epilogue, and the separate pass over the widest activations of the network
// (bias_act_nhwc with a residual: three 275 MB tensors per layer-1 block at batch 8) was the largest kernel of the
// inference batch.  The product is memory-bound on the early layers (K = 64 / 128): what matters is that A, R and D
// each cross the fabric once.
//
// Orientation: the MFMA's rows are OUTPUT CHANNELS (A operand = the weight, staged once per workgroup in LDS in fragment
// order), its columns pixels (B operand = 16-byte loads straight from the activation rows).  A lane then holds, for ONE
// pixel, runs of four consecutive channels.  The residual comes in and the result goes out as whole 256-byte row segments
// (16 bytes per lane, four rows per instruction) through a wave-private LDS tile that turns them into that layout and back
// (loading the 8-byte runs straight from memory asked every cache line eight times: 2.8 TB/s).
// Workgroup = 4 waves = 128 pixels x 128 channels per step, persistent over pixel tiles.
#include "common.h"

namespace kgdet {

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
}  // namespace

// LDS: [W tile: K / 8 x 128 n x 8 k bf16][per wave: staging of a 32-pixel x 64-channel bf16 half tile, rows padded to 144 bytes]
// [128 bias values]
constexpr int kStageRow = 144;                    // 128 bytes of channels + 16: 8-byte column accesses of 32 rows spread over the banks
constexpr int kStageBytes = 32 * kStageRow;       // 4608 per wave

// IN_EPI: x is the RAW output of the convolution in front (conv2 without its epilogue): relu(x + in_bias[k]) is applied to
// the activation fragments as they are loaded -- the pass over conv2's output that did it is gone.
// R == NULL: no residual (conv1 + bn1 + ReLU of a bottleneck: MIOpen's convolution + the bias / ReLU pass over its output as ONE
// kernel); N == 64 (layer 1): the tile's upper 64 channels do not exist -- their weight rows are zero-filled, their two
// accumulator blocks are multiplied (the kernel is memory-bound there) and not stored.
template <bool RELU, int WAVES, bool IN_EPI>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : 2) void conv1x1_nhwc_res(
    const __bf16 *__restrict__ A, const __bf16 *__restrict__ Wt, const float *__restrict__ bias, const __bf16 *__restrict__ R,
    __bf16 *__restrict__ D, long long M, int K, int N, int n_ctiles, long long n_rtiles, const float *__restrict__ in_bias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int THREADS = 64 * WAVES, ROWS = 32 * WAVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = blockIdx.x % n_ctiles;                   // column tiles of one pixel tile sit in neighbouring workgroups
  const long long first = blockIdx.x / n_ctiles, step = gridDim.x / n_ctiles;
  const int n0 = ct * 128;
  const int k8s = K >> 3;
  for (int i = tid; i < k8s * 128; i += THREADS) {        // weight tile -> LDS, 16-byte pieces
    const int n = i & 127, k8 = i >> 7;
    bf16x8 wv = {};
    if (n0 + n < N) wv = *reinterpret_cast<const bf16x8 *>(Wt + (long long)(n0 + n) * K + k8 * 8);
    *reinterpret_cast<bf16x8 *>(smem + (size_t)i * 16) = wv;
  }
  float *lbias = reinterpret_cast<float *>(smem + (size_t)K * 128 * 2 + WAVES * kStageBytes);   // the tile's 128 bias values
  if (tid < 128) lbias[tid] = n0 + tid < N ? bias[n0 + tid] : 0.0f;
  const int halves = N - n0 >= 128 ? 2 : 1;                 // 64-channel halves of this tile that exist
  float *libias = lbias + 128;                                                                   // IN_EPI: the K input biases
  if constexpr (IN_EPI)
    for (int i = tid; i < K; i += THREADS) libias[i] = in_bias[i];
  __syncthreads();
  unsigned char *stage = smem + (size_t)K * 128 * 2 + wave * kStageBytes;
  const int px = lane & 31, kh = lane >> 5;
  const int ksteps = K >> 4;
  // row-major view of a half tile (32 pixels x 64 channels) for the global side: instruction j moves rows 8 j + (lane >> 3),
  // 16 bytes at column piece (lane & 7): eight whole 128-byte row segments per instruction
  const int rrow = lane >> 3, rcol = (lane & 7) * 16;
  for (long long rt = first; rt < n_rtiles; rt += step) {
    const long long m0 = rt * ROWS + wave * 32;
    // residual tile -> registers (coalesced), issued ahead of the products
    bf16x8 rv[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long m = m0 + 8 * j + rrow;
        rv[h][j] = bf16x8{};
        if (R && h < halves)
          rv[h][j] = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const unsigned char *>(R + (m < M ? m : M - 1) * N + n0 + h * 64) + rcol);
      }
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    const long long mp = m0 + px < M ? m0 + px : M - 1;
    const __bf16 *arow = A + mp * K + kh * 8;
    auto kstep = [&](int ks, bf16x8 b) {
      if constexpr (IN_EPI) {
        const f32x4 i0 = *reinterpret_cast<const f32x4 *>(libias + ks * 16 + kh * 8);
        const f32x4 i1 = *reinterpret_cast<const f32x4 *>(libias + ks * 16 + kh * 8 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)fmaxf((float)b[j] + (j < 4 ? i0[j & 3] : i1[j & 3]), 0.0f);
      }
      const unsigned char *wl = smem + ((size_t)(ks * 2 + kh) * 128 + px) * 16;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wl + nb * 32 * 16);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
      }
    };
    int ks = 0;
    for (; ks + 4 <= ksteps; ks += 4) {          // four activation fragments in flight
      bf16x8 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const bf16x8 *>(arow + (ks + u) * 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) kstep(ks + u, b[u]);
    }
    for (; ks < ksteps; ++ks) kstep(ks, *reinterpret_cast<const bf16x8 *>(arow + ks * 16));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h >= halves) break;
      // residual half through the staging tile into the accumulator layout (a lane: ONE pixel, runs of four channels)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<bf16x8 *>(stage + (8 * j + rrow) * kStageRow + rcol) = rv[h][j];
      // (a wave's LDS accesses are ordered: no barrier inside the wave-private tile)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = nh * 32 + 8 * g + 4 * kh;           // channel inside the half
          unsigned char *cell = stage + px * kStageRow + c * 2;
          const bf16x4 res = *reinterpret_cast<const bf16x4 *>(cell);
          const f32x4 bv = *reinterpret_cast<const f32x4 *>(lbias + h * 64 + c);   // (from LDS: global loads of all pieces get hoisted and spill)
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (the convolution's result rounded to bf16 first, as the two-kernel route stores it)
            float f = (float)(__bf16)acc[2 * h + nh][4 * g + e] + bv[e] + (float)res[e];
            if (RELU) f = fmaxf(f, 0.0f);
            o[e] = (__bf16)f;
          }
          *reinterpret_cast<bf16x4 *>(cell) = o;
        }
      // results back out, row-major, coalesced
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long m = m0 + 8 * j + rrow;
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(stage + (8 * j + rrow) * kStageRow + rcol);
        if (m < M) *reinterpret_cast<bf16x8 *>(reinterpret_cast<unsigned char *>(D + m * N + n0 + h * 64) + rcol) = v;
      }
    }
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" int kgdet_conv1x1_nhwc_residual_in(const void *x, const float *in_bias, const void *weight, const float *bias,
                                              const void *residual, void *out, int64_t M, int32_t K, int32_t N, int32_t relu,
                                              void *stream) {
  KGDET_CHECK_SHAPE(M >= 0 && K > 0 && N > 0 && K % 16 == 0 && (N % 128 == 0 || N == 64) && K <= 512,
                    "bad sizes (K %% 16, N %% 128 or N = 64, K <= 512)");
  if (M == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x && weight && bias && out, "null pointer");       // (residual: nullable)
  const int n_ctiles = (N + 127) / 128;
  // a 64 KB+ weight tile is shared by eight waves; beyond K = 384 tile + eight staging tiles no longer fit the 160 KB: four
  const int waves = (K >= 256 && K <= 384) ? 8 : 4;
  const long long n_rtiles = (M + 32 * waves - 1) / (32 * waves);
  const size_t lds = (size_t)K * 128 * 2 + (size_t)waves * kStageBytes + 512 + (size_t)K * 4;
  long long groups = n_rtiles < 768 ? n_rtiles : 768;      // persistent: a few workgroups per CU
  if (K > 384) {   // one workgroup per CU, 128 KB of weights each: as many row tiles per workgroup as the grid allows
    const long long fit = 256 / n_ctiles > 0 ? 256 / n_ctiles : 1;
    if (groups > fit) groups = fit;
  }
  const dim3 grid((unsigned)(groups * n_ctiles));
#define KGDET_NHWC_LAUNCH(RELU_, W_, IN_)                                                                                   \
  do {                                                                                                                      \
    static thread_local bool attr_set = false;                                                                              \
    if (!attr_set) {                                                                                                        \
      KGDET_HIP_TRY(hipFuncSetAttribute((const void *)conv1x1_nhwc_res<RELU_, W_, IN_>,                                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                           \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL((conv1x1_nhwc_res<RELU_, W_, IN_>), grid, dim3(64 * W_), lds, (hipStream_t)stream,                    \
                       (const __bf16 *)x, (const __bf16 *)weight, bias, (const __bf16 *)residual, (__bf16 *)out,            \
                       (long long)M, K, N, n_ctiles, n_rtiles, in_bias);                                                    \
  } while (0)
#define KGDET_NHWC_PICK(RELU_, W_) do { if (in_bias) KGDET_NHWC_LAUNCH(RELU_, W_, true); else KGDET_NHWC_LAUNCH(RELU_, W_, false); } while (0)
  if (waves == 8) { if (relu) KGDET_NHWC_PICK(true, 8); else KGDET_NHWC_PICK(false, 8); }
  else { if (relu) KGDET_NHWC_PICK(true, 4); else KGDET_NHWC_PICK(false, 4); }
#undef KGDET_NHWC_PICK
#undef KGDET_NHWC_LAUNCH
  KGDET_CHECK_LAUNCH("conv1x1_nhwc_residual");
  return KGDET_OK;
}

extern "C" int kgdet_conv1x1_nhwc_residual(const void *x, const void *weight, const float *bias, const void *residual,
                                           void *out, int64_t M, int32_t K, int32_t N, int32_t relu, void *stream) {
  return kgdet_conv1x1_nhwc_residual_in(x, nullptr, weight, bias, residual, out, M, K, N, relu, stream);
}
