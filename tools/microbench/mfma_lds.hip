// Do ds_read_b128 fragment reads overlap with v_mfma_f32_32x32x16_bf16 on one SIMD?  (gfx950)  The conv / plane kernels'
// consumer loop: per stage a wave reads 6 fragments (6 KB) for the NEXT stage and issues 6 MFMAs on the fragments read one
// stage ago.  WAVES per workgroup (8: two per SIMD), one or two workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_cycles[2];

template <int MFMA_ON, int LDS_ON>
__global__ __launch_bounds__(512) void loop(int iters, float *sink, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 12288; i += 512) reinterpret_cast<float *>(lds)[i] = (i & 255) * 0.01f + seed;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char *base = lds + (wave & 3) * 4096 + lane * 16;       // conflict-free 16-byte reads
  bf16x8 F[2][6];
  for (int s = 0; s < 2; ++s) for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) F[s][i][e] = (__bf16)(seed + lane * 0.01f + i);
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (LDS_ON) {
#pragma unroll
        for (int i = 0; i < 6; ++i) F[h ^ 1][i] = *reinterpret_cast<const bf16x8 *>(base + i * 1024 + ((it + h) & 7) * 6144);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MFMA_ON) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[h][2 + mi], F[h][4], acc[mi], 0, 0, 0);
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[h][mi], F[h][5], acc[mi], 0, 0, 0);
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[h][mi], F[h][4], acc[mi], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i & 1][i] += (float)F[h][i][0];      // consume the fragments
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) g_cycles[0] = t1 - t0;
  float s = 0.f;
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 123.456f) sink[0] = s;
}

template <int MFMA_ON, int LDS_ON>
void run(float *sink, int wgs_per_cu) {
  const int iters = 4000;
  const int lds_bytes = wgs_per_cu == 1 ? 100 * 1024 : 60 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(loop<MFMA_ON, LDS_ON>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop<MFMA_ON, LDS_ON><<<256 * wgs_per_cu, 512, lds_bytes>>>(iters, sink, 1.5f);
  hipEventRecord(e0);
  loop<MFMA_ON, LDS_ON><<<256 * wgs_per_cu, 512, lds_bytes>>>(iters, sink, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2];
  hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cycles), sizeof(c));
  printf("%d workgroup(s) of 8 waves per CU, mfma %s, lds reads %s: %.0f cycles per stage and wave (6 MFMAs = 192 pipe cycles, 6 KB of LDS reads); kernel %.3f ms\n",
         wgs_per_cu, MFMA_ON ? "ON " : "off", LDS_ON ? "ON " : "off", (double)c[0] / iters, ms);
}

int main() {
  float *sink; hipMalloc(&sink, 64);
  for (int w = 1; w <= 2; ++w) {
    run<1, 0>(sink, w);
    run<0, 1>(sink, w);
    run<1, 1>(sink, w);
  }
  return 0;
}
