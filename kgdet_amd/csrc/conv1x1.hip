// 1x1 convolution (stride 1, NCHW, fp32 in / fp32 out) as bf16 hi/lo-split MFMA GEMMs for gfx950.
//
// The ResNet-50 bottleneck (mmdet/models/backbones/resnet.py:142-186, 231-262) is two 1x1 convolutions around a
// 3x3; with the reference's fp32 arithmetic MIOpen runs them as fp32 GEMMs at 60-110 TFLOP/s (MI355X fp32 MFMA
// peak 157).  A 1x1 convolution at these shapes has ~50 flop per byte of fp32 activation traffic, i.e. it sits at
// ~400 TFLOP/s on the HBM roofline, so the matrix pipe, not memory, is what the fp32 GEMM leaves on the table.
// Same arithmetic as the deformable kernels (dcn_forward_plane.hip): every fp32 operand v = hi + lo with
// hi = bf16(v), lo = bf16(v - hi); a product is a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation -- fp32-level accuracy (~1e-6 of the output scale) at a third of the bf16 rate.
//
//   forward      y[b] (Cout x HW) = W (Cout x Cin)     . x[b]  (Cin x HW)    conv1x1_nn, A = packed W
//   grad_input   gx[b] (Cin x HW) = W^T (Cin x Cout)   . gy[b] (Cout x HW)   conv1x1_nn, A = packed W^T
//   grad_weight  gW (Cout x Cin)  = sum_b gy[b] (Cout x HW) . x[b]^T         conv1x1_nt + conv1x1_sum
//
// conv1x1_nn: 128 x 128 output tile per workgroup (4 waves x 64 x 64), reduction in stages of 16 channels.
//   A stage = 8 KB of the pre-split weight image [part][khalf][128 rows][8 bf16] (one 16-byte load per thread and
//   part); B stage = 16 activation rows of 128 pixels: a thread owns (pixel, khalf), issues 8 dword loads (each a
//   coalesced 256-byte row segment per wave), splits, writes one 16-byte LDS entry per part.  Loads of stage s+4 are
//   in flight while stage s+1 is converted and stage s multiplied; one barrier per stage.  Tiles are dealt to the
//   XCDs in contiguous runs so the M tiles sharing a pixel tile share an L2.
// conv1x1_nt: both operands are activations with the reduction (pixels) contiguous: a thread loads 8 consecutive
//   pixels of one row of each operand = one LDS entry each.  The pixel range is cut into `splits` chunks, one
//   workgroup per (tile, chunk) writes a partial; conv1x1_sum adds them in fixed order (deterministic).
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
constexpr int kGemmThreads = 256;

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (__bf16)v[i];
    lo[i] = (__bf16)(v[i] - (float)hi[i]);
  }
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

// image[mt][k16][part][khalf][128][8] of the logical A (M x K): A[m][k] = transpose ? w[k * ld + m] : w[m * ld + k]
__global__ __launch_bounds__(256) void conv1x1_pack(const float *__restrict__ w, int M, int K, int ld, int transpose,
                                                    unsigned char *__restrict__ img) {
  const int k16s = K / kTK;
  const long long total = (long long)((M + kTM - 1) / kTM) * k16s * 2 * kTM;   // (mt, k16, khalf, row)
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
    const int row = (int)(i % kTM);
    const int khalf = (int)((i / kTM) & 1);
    const long long st = i / (2 * kTM);
    const int k16 = (int)(st % k16s), mt = (int)(st / k16s);
    const int m = mt * kTM + row, k0 = k16 * kTK + khalf * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = m < M ? (transpose ? w[(long long)(k0 + j) * ld + m] : w[(long long)m * ld + k0 + j]) : 0.0f;
    bf16x8 hi, lo;
    split8(v, hi, lo);
    unsigned char *dst = img + st * kStage + khalf * (kTM * 16) + row * 16;
    *reinterpret_cast<bf16x8 *>(dst) = hi;
    *reinterpret_cast<bf16x8 *>(dst + kPart) = lo;
  }
}

// y[b][m][n] = sum_k A[m][k] * x[b][k][n];  A as packed image, x [B, K, N], y [B, M, N], N contiguous.
// ksplit > 1: workgroup (tile, part) reduces stages [part * per, ...) and writes y-shaped partial `part` of `y`
// (= a [ksplit][B, M, N] buffer); conv1x1_sum adds the parts.  Used when a problem has too few tiles for 256 CUs.
constexpr int kPF = 4;   // stages of global loads in flight per thread

__global__ __launch_bounds__(kGemmThreads) void conv1x1_nn(const unsigned char *__restrict__ img,
                                                           const float *__restrict__ x, float *__restrict__ y, int M,
                                                           int K, int N, int n_mt, int n_nt, int tiles, int ksplit,
                                                           long long part_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * kStage];   // [buf][A | B][kStage]
  const int unit = xcd_tile(blockIdx.x, tiles * ksplit);
  if (unit >= tiles * ksplit) return;
  // unit order: the K parts and the m tiles of one (image, pixel tile) adjacent -> they share it through one L2
  const int part = unit % ksplit, tile = unit / ksplit;
  const int mt = tile % n_mt, nt = (tile / n_mt) % n_nt, b = tile / (n_mt * n_nt);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int n_local = tid & (kTN - 1), khalf = tid >> 7;
  const int n0 = nt * kTN;
  const int n_ld = min(n0 + n_local, N - 1);   // columns past the end re-read the last one: never stored
  const float *xb = x + (long long)b * K * N + n_ld;
  const unsigned char *ai = img + (long long)mt * (K / kTK) * kStage + tid * 16;
  const int all = K / kTK, per = (all + ksplit - 1) / ksplit;
  const int s_begin = part * per, s_end = min(all, s_begin + per);
  const int stages = max(s_end - s_begin, 0);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  struct Regs {
    f32x4 a[2];
    float v[8];
  };
  auto issue = [&](int s, Regs &R) {   // clamped: unconditional loads keep hipcc's vmcnt counting exact
    const int sc = s_begin + min(s, max(stages - 1, 0));
    const unsigned char *ap = ai + (long long)sc * kStage;
    R.a[0] = *reinterpret_cast<const f32x4 *>(ap);
    R.a[1] = *reinterpret_cast<const f32x4 *>(ap + kPart);
    const float *xp = xb + (long long)(sc * kTK + khalf * 8) * N;
#pragma unroll
    for (int j = 0; j < 8; ++j) R.v[j] = xp[(long long)j * N];
  };
  auto commit = [&](int buf, const Regs &R) {
    unsigned char *As = smem + buf * 2 * kStage, *Bs = As + kStage;
    *reinterpret_cast<f32x4 *>(As + tid * 16) = R.a[0];
    *reinterpret_cast<f32x4 *>(As + kPart + tid * 16) = R.a[1];
    bf16x8 hi, lo;
    split8(R.v, hi, lo);
    unsigned char *dst = Bs + khalf * (kTN * 16) + n_local * 16;
    *reinterpret_cast<bf16x8 *>(dst) = hi;
    *reinterpret_cast<bf16x8 *>(dst + kPart) = lo;
  };
  if (stages > 0) {
    Regs R[kPF];
#pragma unroll
    for (int i = 0; i < kPF; ++i) issue(i, R[i]);
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
        const unsigned char *As = smem + (s & 1) * 2 * kStage;
        mma_stage(As, As + kStage, lane, wm, wn, acc);
        commit((s + 1) & 1, R[(u + 1) % kPF]);   // past the last stage: a clamped duplicate nobody reads
      }
    }
#pragma unroll
    for (int u = 0; u < kPF - 1; ++u) {   // tail: stages full .. stages-1 are already in R[u]; no loads
      const int s = full + u;
      if (s < stages) {
        __syncthreads();
        const unsigned char *As = smem + (s & 1) * 2 * kStage;
        mma_stage(As, As + kStage, lane, wm, wn, acc);
        if (s + 1 < stages) commit((s + 1) & 1, R[u + 1]);
      }
    }
  }

  // store: lane holds column (lane & 31) of 16 rows per 32 x 32 block -> 128-byte row segments per half wave
  float *yb = y + (long long)part * part_stride + (long long)b * M * N;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n < N) yb[(long long)m * N + n] = acc[mi][ni][r];
      }
    }
}

// out[i] = sum_s parts[s][i], s ascending (deterministic); n a multiple of 2
__global__ __launch_bounds__(256) void conv1x1_sum(const float *__restrict__ parts, float *__restrict__ out,
                                                   long long n, long long stride, int count) {
  for (long long i = (blockIdx.x * 256LL + threadIdx.x) * 2; i < n; i += gridDim.x * 512LL) {
    f32x2 s = {0.0f, 0.0f};
    int k = 0;
    for (; k + 4 <= count; k += 4) {
      const f32x2 v0 = *reinterpret_cast<const f32x2 *>(parts + (long long)k * stride + i);
      const f32x2 v1 = *reinterpret_cast<const f32x2 *>(parts + (long long)(k + 1) * stride + i);
      const f32x2 v2 = *reinterpret_cast<const f32x2 *>(parts + (long long)(k + 2) * stride + i);
      const f32x2 v3 = *reinterpret_cast<const f32x2 *>(parts + (long long)(k + 3) * stride + i);
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < count; ++k) s += *reinterpret_cast<const f32x2 *>(parts + (long long)k * stride + i);
    *reinterpret_cast<f32x2 *>(out + i) = s;
  }
}

// partial[split][m][n] (natural [M, N] layout) = sum over this split's pixels (and images) of a[b][m][px] * bm[b][n][px]
// a [B, M, L], bm [B, N, L], L contiguous.  The B * ceil(L / 16) stages are cut into `splits` runs of `per` stages; a
// stage never straddles two images (the tail of an image is zero-filled).  VEC = floats per load (4 when L % 4 == 0).
template <int VEC>
__global__ __launch_bounds__(kGemmThreads) void conv1x1_nt(const float *__restrict__ a, const float *__restrict__ bm,
                                                           float *__restrict__ partial, int M, int N, int L, int B,
                                                           int n_mt, int n_nt, int stages_per_image, int per) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * kStage];
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int tile = blockIdx.x % (n_mt * n_nt), split = blockIdx.x / (n_mt * n_nt);
  const int mt = tile % n_mt, nt = tile / n_mt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int row = tid >> 1, khalf = tid & 1;
  const int total = B * stages_per_image;
  const int s_begin = split * per, s_end = min(total, s_begin + per);
  const int am = min(mt * kTM + row, M - 1), bn = min(nt * kTN + row, N - 1);
  const bool a_real = mt * kTM + row < M, b_real = nt * kTN + row < N;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  struct Regs {
    float va[8], vb[8];
    int valid;   // pixels of this thread's 8 that exist (tail of an image)
  };
  auto issue = [&](int s, Regs &R) {
    const int sc = min(s, s_end - 1);
    const int img = sc / stages_per_image, st = sc - img * stages_per_image;
    const int p0 = st * kTK + khalf * 8;
    R.valid = min(8, L - p0);
    const float *ap = a + ((long long)img * M + am) * L, *bp = bm + ((long long)img * N + bn) * L;
#pragma unroll
    for (int j = 0; j < 8; j += VEC) {   // L % VEC == 0: the loads stay aligned; clamped at the image's end
      const int p = min(p0 + j, L - VEC);
      const vec_t u = *reinterpret_cast<const vec_t *>(ap + p), w = *reinterpret_cast<const vec_t *>(bp + p);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        R.va[j + e] = u[e];
        R.vb[j + e] = w[e];
      }
    }
  };
  auto commit = [&](int buf, Regs &R) {
    unsigned char *As = smem + buf * 2 * kStage, *Bs = As + kStage;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = j < R.valid;
      R.va[j] = (ok && a_real) ? R.va[j] : 0.0f;
      R.vb[j] = (ok && b_real) ? R.vb[j] : 0.0f;
    }
    bf16x8 hi, lo;
    split8(R.va, hi, lo);
    unsigned char *dst = As + khalf * (kTM * 16) + row * 16;
    *reinterpret_cast<bf16x8 *>(dst) = hi;
    *reinterpret_cast<bf16x8 *>(dst + kPart) = lo;
    split8(R.vb, hi, lo);
    dst = Bs + khalf * (kTN * 16) + row * 16;
    *reinterpret_cast<bf16x8 *>(dst) = hi;
    *reinterpret_cast<bf16x8 *>(dst + kPart) = lo;
  };
  constexpr int PF = 3;
  const int n = s_end - s_begin;
  if (n > 0) {
    Regs R[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) issue(s_begin + i, R[i]);
    commit(0, R[0]);
    const int full = n / PF * PF;   // as in conv1x1_nn: unguarded bodies in the main loop, load-free tail
    for (int j0 = 0; j0 < full; j0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int j = j0 + u;
        __syncthreads();
        issue(s_begin + j + PF, R[u]);
        const unsigned char *As = smem + (j & 1) * 2 * kStage;
        mma_stage(As, As + kStage, lane, wm, wn, acc);
        commit((j + 1) & 1, R[(u + 1) % PF]);
      }
    }
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
      const int j = full + u;
      if (j < n) {
        __syncthreads();
        const unsigned char *As = smem + (j & 1) * 2 * kStage;
        mma_stage(As, As + kStage, lane, wm, wn, acc);
        if (j + 1 < n) commit((j + 1) & 1, R[u + 1]);
      }
    }
  }
  float *out = partial + (long long)split * M * N;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int nn = nt * kTN + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * kTM + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && nn < N) out[(long long)m * N + nn] = acc[mi][ni][r];
      }
    }
}

namespace {

// measured on MI355X (tools/bench_conv1x1_wgrad.py): one workgroup per CU with >= 32 stages each beats finer cuts --
// every extra split is another [M, N] partial written and re-read
int nt_splits(int tiles, int total_stages) {
  int splits = (256 + tiles - 1) / tiles;
  const int most = (total_stages + 31) / 32;
  if (splits > most) splits = most;
  if (splits > 128) splits = 128;
  return splits < 1 ? 1 : splits;
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

}  // namespace

}  // namespace kgdet

using namespace kgdet;

extern "C" size_t kgdet_conv1x1_packed_bytes(int32_t M, int32_t K) {
  if (M <= 0 || K <= 0 || K % kTK) return 0;
  return (size_t)((M + kTM - 1) / kTM) * (K / kTK) * kStage;
}

extern "C" int kgdet_conv1x1_pack(const float *w, int32_t O, int32_t C, int32_t transpose, void *packed,
                                  void *stream) {
  // weight [O, C]; transpose = 0: A = W (rows O, reduction C; forward); 1: A = W^T (rows C, reduction O; grad_input)
  const int M = transpose ? C : O, K = transpose ? O : C;
  KGDET_CHECK_SHAPE(O > 0 && C > 0 && K % kTK == 0, "reduction length %d is not a multiple of 16", K);
  KGDET_CHECK_SHAPE(w && packed, "null pointer");
  const long long total = (long long)((M + kTM - 1) / kTM) * (K / kTK) * 2 * kTM;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(conv1x1_pack, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, M, K, C, transpose,
                     (unsigned char *)packed);
  KGDET_CHECK_LAUNCH("conv1x1_pack");
  return KGDET_OK;
}

extern "C" size_t kgdet_conv1x1_apply_workspace_bytes(int64_t B, int32_t M, int32_t K, int64_t HW) {
  if (B <= 0 || M <= 0 || K <= 0 || HW <= 0 || K % kTK) return 0;
  const long long tiles = (long long)((M + kTM - 1) / kTM) * ((HW + kTN - 1) / kTN) * B;
  const int ks = nn_ksplit(tiles, K / kTK);
  return ks > 1 ? (size_t)ks * B * M * HW * sizeof(float) : 0;
}

extern "C" int kgdet_conv1x1_apply(const void *packed, const float *x, float *y, int64_t B, int32_t M, int32_t K,
                                   int64_t HW, void *workspace, size_t workspace_bytes, void *stream) {
  KGDET_CHECK_SHAPE(B >= 0 && M > 0 && K > 0 && HW >= 0 && HW < (1LL << 30), "bad sizes");
  KGDET_CHECK_SHAPE(K % kTK == 0, "reduction length %d is not a multiple of 16", K);
  if (B * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(packed && x && y, "null pointer");
  const int n_mt = (M + kTM - 1) / kTM, n_nt = (int)((HW + kTN - 1) / kTN);
  const long long tiles = (long long)n_mt * n_nt * B;
  KGDET_CHECK_SHAPE(tiles < (1LL << 30), "too many tiles");
  const int ks = nn_ksplit(tiles, K / kTK);
  const long long part_stride = B * M * HW;
  if (ks > 1) {
    KGDET_CHECK_SHAPE(workspace && workspace_bytes >= (size_t)ks * part_stride * sizeof(float), "workspace too small");
    KGDET_CHECK_SHAPE(part_stride % 2 == 0, "B*M*H*W must be even");
  }
  const int per = (int)((tiles * ks + 7) / 8);
  hipLaunchKernelGGL(conv1x1_nn, dim3(per * 8), dim3(kGemmThreads), 0, (hipStream_t)stream,
                     (const unsigned char *)packed, x, ks > 1 ? (float *)workspace : y, M, K, (int)HW, n_mt, n_nt,
                     (int)tiles, ks, part_stride);
  KGDET_CHECK_LAUNCH("conv1x1_nn");
  if (ks > 1) {
    const long long blocks = (part_stride / 2 + 255) / 256;
    hipLaunchKernelGGL(conv1x1_sum, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, y, part_stride, part_stride, ks);
    KGDET_CHECK_LAUNCH("conv1x1_sum");
  }
  return KGDET_OK;
}

extern "C" size_t kgdet_conv1x1_grad_weight_workspace_bytes(int64_t B, int32_t O, int32_t C, int64_t HW) {
  if (B <= 0 || O <= 0 || C <= 0 || HW <= 0) return 0;
  const int tiles = ((O + kTM - 1) / kTM) * ((C + kTN - 1) / kTN);
  const int stages = (int)(B * ((HW + kTK - 1) / kTK));
  return (size_t)nt_splits(tiles, stages) * O * C * sizeof(float);
}

extern "C" int kgdet_conv1x1_grad_weight(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O,
                                         int32_t C, int64_t HW, void *workspace, size_t workspace_bytes,
                                         void *stream) {
  KGDET_CHECK_SHAPE(B > 0 && O > 0 && C > 0 && HW > 0 && HW < (1LL << 30), "bad sizes");
  KGDET_CHECK_SHAPE(HW % 2 == 0, "H*W = %lld must be even (8-byte loads)", (long long)HW);
  KGDET_CHECK_SHAPE(grad_y && x && grad_w && workspace, "null pointer");
  KGDET_CHECK_SHAPE(workspace_bytes >= kgdet_conv1x1_grad_weight_workspace_bytes(B, O, C, HW), "workspace too small");
  const int n_mt = (O + kTM - 1) / kTM, n_nt = (C + kTN - 1) / kTN, tiles = n_mt * n_nt;
  const int spi = (int)((HW + kTK - 1) / kTK), total = (int)(B * spi);
  const int splits = nt_splits(tiles, total);
  const int per = (total + splits - 1) / splits;
  KGDET_CHECK_SHAPE(((long long)O * C) % 2 == 0, "O*C must be even");
  if (HW % 4 == 0)
    hipLaunchKernelGGL(conv1x1_nt<4>, dim3(tiles * splits), dim3(kGemmThreads), 0, (hipStream_t)stream, grad_y, x,
                       (float *)workspace, O, C, (int)HW, (int)B, n_mt, n_nt, spi, per);
  else
    hipLaunchKernelGGL(conv1x1_nt<2>, dim3(tiles * splits), dim3(kGemmThreads), 0, (hipStream_t)stream, grad_y, x,
                       (float *)workspace, O, C, (int)HW, (int)B, n_mt, n_nt, spi, per);
  KGDET_CHECK_LAUNCH("conv1x1_nt");
  const long long n = (long long)O * C;
  const long long blocks = (n / 2 + 255) / 256;
  hipLaunchKernelGGL(conv1x1_sum, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const float *)workspace, grad_w, n, n, splits);
  KGDET_CHECK_LAUNCH("conv1x1_sum");
  return KGDET_OK;
}
