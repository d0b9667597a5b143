// Do 16-byte global loads work at 4-byte alignment on gfx950 (SH_MEM_CONFIG alignment mode), and what do they cost?
//   hipcc -O3 --offload-arch=gfx950 unaligned_x4.hip -o unaligned_x4 && ./unaligned_x4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void k(const float *x, float *y, int shift, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i * 4 + shift + 4 > n) return;
  const f32x4u v = *reinterpret_cast<const f32x4u *>(x + i * 4 + shift);
  *reinterpret_cast<f32x4 *>(y + i * 4) = f32x4{v[0], v[1], v[2], v[3]};
}
int main() {
  const int n = 64 << 20;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)(i & 0xffff);
  float *x, *y;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(y, 0, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<n / 1024, 256>>>(x, y, shift, n);
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) k<<<n / 1024, 256>>>(x, y, shift, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<float> o(1024);
    hipMemcpy(o.data(), y + 4096, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += o[i] != h[4096 + i + shift];
    printf("shift %d: %s, %.1f us per pass (%.2f TB/s)\n", shift, bad ? "WRONG" : "ok", ms * 100, 2.0 * n * 4 / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
