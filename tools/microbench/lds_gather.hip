// LDS gather microbenchmark for gfx950: cycles per batch of 8 ds_read_b128 (the plane kernels' half-stage: 4 bilinear
// corners x 2 channel quads of a 64-byte [pixel][16 channel] row), for W waves per CU and three address patterns.
//   hipcc --offload-arch=gfx950 -O3 lds_gather.hip -o lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_out[64];
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// PATTERN 0: lane l samples pixel base + l (regular grid, the conflict-free case)
//         1: pixel base + l + jitter in [-2, 2] rows / columns (the benchmark's offsets: randn * 2)
//         2: uniformly random pixel
template <int PATTERN, int PITCH>
__global__ void gather(int iters, int HW, int Wd, float *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < HW * PITCH / 4; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = i * 0.5f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned long long total = 0;
  unsigned seed = threadIdx.x * 7919u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    seed = hash(seed + it);
    int q;
    if (PATTERN == 0) q = (seed >> 8) % 7 * 128 + (threadIdx.x & 127);
    else if (PATTERN == 1) { const int b = ((it * 131) % 7) * 128 + (threadIdx.x & 127); q = b + (int)(seed % 5) - 2 + ((int)((seed >> 4) % 5) - 2) * Wd; }
    else q = seed % (unsigned)HW;
    q = min(max(q, 0), HW - Wd - 2);
    const int qs[4] = {q, q + 1, q + Wd, q + Wd + 1};
    unsigned off[4];
    for (int e = 0; e < 4; ++e) off[e] = PITCH == 64 ? (unsigned)(qs[e] * 64 + (((qs[e] >> 2) & 3) << 4)) : (unsigned)(qs[e] * PITCH);
    const unsigned long long t0 = __builtin_readcyclecounter();
    f32x4 v[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        v[c][e] = *reinterpret_cast<const f32x4 *>(lds + (PITCH == 64 ? (off[e] ^ (unsigned)(c << 4)) : off[e] + (unsigned)(c << 4)));
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const unsigned long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += v[0][e] + v[1][e];
  }
  if (lane == 0 && blockIdx.x == 0) g_out[threadIdx.x >> 6] = total;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
}

template <int PATTERN, int PITCH>
void run(int waves, float *sink) {
  const int HW = 1050, Wd = 42, iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(gather<PATTERN, PITCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  gather<PATTERN, PITCH><<<256, waves * 64, HW * PITCH>>>(iters, HW, Wd, sink);
  hipDeviceSynchronize();
  unsigned long long out[64];
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_out), sizeof(out));
  double mean = 0;
  for (int w = 0; w < waves; ++w) mean += (double)out[w] / iters / waves;
  printf("pitch %d pattern %d  waves/CU %2d : %.0f cycles per batch of 8 ds_read_b128 (timer overhead included)\n", PITCH, PATTERN, waves, mean);
}

int main() {
  float *sink; hipMalloc(&sink, 64);
  for (int w : {1, 4, 8, 12}) { run<0, 64>(w, sink); run<1, 64>(w, sink); run<2, 64>(w, sink); }
  for (int w : {4}) { run<0, 80>(w, sink); run<1, 80>(w, sink); run<2, 80>(w, sink); }
  return 0;
}
