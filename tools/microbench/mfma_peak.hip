// MFMA issue-rate microbenchmark for gfx950: v_mfma_f32_32x32x16_bf16 from W waves per SIMD, ACC independent
// accumulators per wave, no memory traffic.  Prints TFLOP/s per configuration -- the ceiling the plane kernels' consumer
// waves can reach on the box at hand (clock under load included).   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_clk[4];
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// RANDOM = 0: one constant operand pair (little switching in the datapath); 1: 6 operand sets of random bf16 values
// (|v| in [0.5, 2)), cycled through -- the toggle rate of real data
template <int ACC, int RANDOM>
__global__ void spin(float *out, int iters, float seed) {
  bf16x8 a[3], b[ACC];
  for (int r = 0; r < 3; ++r)
    for (int i = 0; i < 8; ++i) {
      const unsigned h = hash(threadIdx.x * 977u + r * 131u + i * 7u + blockIdx.x);
      a[r][i] = RANDOM ? (__bf16)__uint_as_float(0x3f000000u | (h & 0x80ff0000u) | ((h & 1u) << 23)) : (__bf16)(seed + threadIdx.x * 0.001f);
    }
  for (int r = 0; r < ACC; ++r)
    for (int i = 0; i < 8; ++i) {
      const unsigned h = hash(threadIdx.x * 577u + r * 331u + i * 17u + blockIdx.x * 3u);
      b[r][i] = RANDOM ? (__bf16)__uint_as_float(0x3f000000u | (h & 0x80ff0000u) | ((h & 1u) << 23)) : (__bf16)(seed - i * 0.01f);
    }
  f32x16 acc[ACC];
  for (int k = 0; k < ACC; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < ACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b[k], acc[k], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = c1 - c0; g_clk[1] = r1 - r0; }
  float s = 0.f;
  for (int k = 0; k < ACC; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 123.456f) out[0] = s;
}

template <int ACC, int RANDOM>
void run(int waves_per_simd, float *out) {
  const int threads = waves_per_simd * 4 * 64, blocks = 256, iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<ACC, RANDOM><<<blocks, threads>>>(out, 100, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    spin<ACC, RANDOM><<<blocks, threads>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * waves_per_simd * 4 * iters * 3 * ACC * 32768.0;
  unsigned long long clk[4];
  hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
  printf("%s data  waves/SIMD %d  acc %d : %.3f ms  %.1f TFLOP/s   s_memtime %.0f MHz (vs the 100 MHz s_memrealtime)\n",
         RANDOM ? "random  " : "constant", waves_per_simd, ACC, best, flops / best / 1e9, 100.0 * clk[0] / (double)clk[1]);
}

int main() {
  float *out; hipMalloc(&out, 64);
  for (int w = 1; w <= 3; ++w) { run<4, 0>(w, out); }
  for (int w = 1; w <= 3; ++w) { run<4, 1>(w, out); }
  for (int r = 0; r < 3; ++r) { run<4, 0>(2, out); run<4, 1>(2, out); }
  return 0;
}
