#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int iters, float* out, int lds_ops) {
  __shared__ float buf[4096];
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (lds_ops) { buf[threadIdx.x] = acc + i; }
    __syncthreads();
    if (lds_ops) acc += buf[(threadIdx.x + 64) & 1023];
  }
  if (acc == 12345.f) out[0] = acc;
}
int main() {
  float* d; hipMalloc(&d, 4);
  for (int lds = 0; lds < 2; ++lds)
  for (int T : {256, 512, 768, 1024}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<256, T>>>(1000, d, lds); hipDeviceSynchronize();
    hipEventRecord(a); k<<<256, T>>>(10000, d, lds); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("lds=%d T=%d: %.1f ns per barrier iteration\n", lds, T, ms * 1e6 / 10000);
  }
  return 0;
}
