// How much do MFMA waves slow a VALU wave on the same SIMD?  (gfx950)  Workgroup of 12 waves as in the plane kernels:
// waves 0-7 issue v_mfma_f32_32x32x16_bf16 back to back (accumulators in VGPRs or AGPRs), waves 8-11 run the producers'
// instruction mix (packed fp32 FMAs, bf16 conversions, optional ds_read_b128 gathers) and time themselves.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ unsigned long long g_out[8];

__device__ unsigned g_simd[12];
// PLACE 0: waves 0-7 MFMA, 8-11 VALU (one VALU wave beside two MFMA waves on every SIMD)
//       1: waves with (w & 3) == 3 VALU (3 waves, all on one SIMD if waves are dealt to SIMDs cyclically), w = 10 idle,
//          the other 8 MFMA (3 + 3 + 2 per SIMD)
template <int MFMA_ON, int AGPR, int GATHER, int PLACE, int PACE = 0>
__global__ __launch_bounds__(768, 1) void mix(int iters, float *sink, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 16384; i += 768) reinterpret_cast<float *>(lds)[i] = i * 0.25f;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_simd[wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
  const bool valu_wave = PLACE == 0 ? wave >= 8 : (wave & 3) == 3;
  if (PLACE == 1 && wave == 10) return;
  const int vslot = PLACE == 0 ? wave - 8 : wave >> 2;
  if (!valu_wave) {
    if (!MFMA_ON) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.37f + i); b[i] = (__bf16)(seed * 3 - i * 0.11f + threadIdx.x); }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
          else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
          // PACE: the MFMA wave idles between its MFMAs (s_nop does not hold the vector issue port) instead of queueing the
          // next MFMA behind the busy pipe
          if (PACE == 1) asm volatile("s_nop 7");
          if (PACE == 2) asm volatile("s_nop 15");
          if (PACE == 3) asm volatile("s_nop 15\n s_nop 7");
          if (PACE == 4) asm volatile("s_nop 15\n s_nop 15");
          if (PACE == 5) asm volatile("s_nop 15\n s_nop 15\n s_nop 7");
          if (PACE == 6) asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
          if (PACE == 7) asm volatile("s_sleep 1");
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 123.456f) sink[0] = s;
    return;
  }
  __builtin_amdgcn_s_setprio(2);
  // producer-like half-stage: [8 gathers] 16 packed FMAs (4 chains of 4), 8 -> bf16 hi/lo split, 2 ds_write_b128
  f32x2 w[4];
  for (int e = 0; e < 4; ++e) w[e] = f32x2{seed + e, seed + e};
  f32x4 v[2][4];
  for (int c = 0; c < 2; ++c) for (int e = 0; e < 4; ++e) v[c][e] = f32x4{seed * c, seed + e, seed - c, seed * e};
  unsigned off = ((threadIdx.x * 2654435761u) >> 12) % 1000u * 64u;
  float keep = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (GATHER) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          v[c][e] = *reinterpret_cast<const f32x4 *>(lds + ((off + e * 64 * (e & 1 ? 1 : 42)) & 0xffc0u) + c * 16);
      off = (off * 5u + 64u * 17u) % 64000u;
    }
    f32x2 sv[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x2 ve = {v[c][e][2 * h], v[c][e][2 * h + 1]};
          sv[c][h] = e == 0 ? w[e] * ve : __builtin_elementwise_fma(w[e], ve, sv[c][h]);
        }
    bf16x8 hi, lo;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int q = c * 4 + h * 2;
        hi[q] = (__bf16)sv[c][h][0]; hi[q + 1] = (__bf16)sv[c][h][1];
        const f32x2 lf = sv[c][h] - f32x2{(float)hi[q], (float)hi[q + 1]};
        lo[q] = (__bf16)lf[0]; lo[q + 1] = (__bf16)lf[1];
      }
    unsigned char *dst = lds + 65536 + (vslot * 64 + (threadIdx.x & 63)) * 16 + (it & 3) * 8192;
    *reinterpret_cast<bf16x8 *>(dst) = hi;
    *reinterpret_cast<bf16x8 *>(dst + 4096) = lo;
    if (!GATHER) { v[0][0][0] += (float)lo[0]; w[1][0] += 1e-9f; }
    keep += (float)hi[3];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_out[vslot] = t1 - t0;
  if (keep == 1.2345f) sink[1] = keep;
}

template <int MFMA_ON, int AGPR, int GATHER, int PLACE, int PACE = 0>
void run(float *sink) {
  const int iters = 3000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(mix<MFMA_ON, AGPR, GATHER, PLACE, PACE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mix<MFMA_ON, AGPR, GATHER, PLACE, PACE><<<256, 768, 100 * 1024>>>(iters, sink, 1.5f);
  hipEventRecord(e0);
  mix<MFMA_ON, AGPR, GATHER, PLACE, PACE><<<256, 768, 100 * 1024>>>(iters, sink, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long out[8];
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_out), sizeof(out));
  unsigned simd[12];
  hipMemcpyFromSymbol(simd, HIP_SYMBOL(g_simd), sizeof(simd));
  printf("pace %d place %d mfma %s acc in %s  gathers %s : VALU wave %.0f cycles per half-stage;  kernel %.3f ms   SIMD of waves 0..11:", PACE, PLACE, MFMA_ON ? "ON " : "off",
         AGPR ? "AGPR" : "VGPR", GATHER ? "yes" : "no ", (double)out[0] / iters, ms);
  for (int w = 0; w < 12; ++w) printf(" %u", (simd[w] >> 4) & 3);
  printf("\n");
  if (MFMA_ON) printf("      (MFMA waves: %.1f cycles per MFMA and SIMD if they ran the whole %.3f ms at 2.4 GHz)\n", ms * 1e-3 * 2.4e9 / (iters * 4.0 * 12 * 2), ms);
}


// Phased: 8 waves (2 per SIMD), every wave alternates a sampling phase (UNITS producer half-stages with gathers) and an
// MFMA phase (STAGES x 12 MFMAs), a workgroup barrier after each: the two instruction classes never meet on a SIMD.
template <int UNITS, int STAGES>
__global__ __launch_bounds__(512, 1) void phased(int groups, float *sink, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float *>(lds)[i] = i * 0.25f;
  __syncthreads();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.37f + i); b[i] = (__bf16)(seed * 3 - i * 0.11f + threadIdx.x); }
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  f32x2 w[4];
  for (int e = 0; e < 4; ++e) w[e] = f32x2{seed + e, seed + e};
  unsigned off = ((threadIdx.x * 2654435761u) >> 12) % 1000u * 64u;
  float keep = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int g = 0; g < groups; ++g) {
    f32x4 v[UNITS][2][4];
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          v[u][c][e] = *reinterpret_cast<const f32x4 *>(lds + ((off + e * 64 * (e & 1 ? 1 : 42)) & 0xffc0u) + c * 16);
      off = (off * 5u + 64u * 17u) % 64000u;
    }
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      f32x2 sv[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 ve = {v[u][c][e][2 * h], v[u][c][e][2 * h + 1]};
            sv[c][h] = e == 0 ? w[e] * ve : __builtin_elementwise_fma(w[e], ve, sv[c][h]);
          }
      bf16x8 hi, lo;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = c * 4 + h * 2;
          hi[q] = (__bf16)sv[c][h][0]; hi[q + 1] = (__bf16)sv[c][h][1];
          const f32x2 lf = sv[c][h] - f32x2{(float)hi[q], (float)hi[q + 1]};
          lo[q] = (__bf16)lf[0]; lo[q + 1] = (__bf16)lf[1];
        }
      unsigned char *dst = lds + 65536 + threadIdx.x * 16 + (u & 1) * 16384;
      *reinterpret_cast<bf16x8 *>(dst) = hi;
      *reinterpret_cast<bf16x8 *>(dst + 8192) = lo;
      keep += (float)hi[3];
    }
    __syncthreads();
    for (int st = 0; st < STAGES; ++st) {
      bf16x8 bb[2][4];
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bb[part][ni] = *reinterpret_cast<const bf16x8 *>(lds + 65536 + part * 8192 + ni * 512 + (threadIdx.x & 63) * 16 + (st & 1) * 16384);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb[1][k], acc[k], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, bb[0][k], acc[k], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb[0][k], acc[k], 0, 0, 0);
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_out[threadIdx.x >> 6] = t1 - t0;
  float s = keep;
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 123.456f) sink[0] = s;
}

template <int UNITS, int STAGES>
void run_phased(float *sink) {
  const int groups = 400;
  hipFuncSetAttribute(reinterpret_cast<const void *>(phased<UNITS, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  phased<UNITS, STAGES><<<256, 512, 100 * 1024>>>(groups, sink, 1.5f);
  phased<UNITS, STAGES><<<256, 512, 100 * 1024>>>(groups, sink, 1.5f);
  hipDeviceSynchronize();
  unsigned long long out[8];
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_out), sizeof(out));
  printf("phased: %d stages per group (%d sampling units per wave): %.0f cycles per stage  (MFMA alone: %d)\n", STAGES, UNITS,
         (double)out[0] / groups / STAGES, 24 * 32);
}

int main() {
  float *sink; hipMalloc(&sink, 64);
  run<0, 0, 0, 0>(sink); run<1, 0, 0, 0>(sink); run<1, 1, 0, 0>(sink);
  run<0, 0, 1, 0>(sink); run<1, 0, 1, 0>(sink);
  run<0, 0, 1, 1>(sink); run<1, 0, 1, 1>(sink); run<1, 0, 0, 1>(sink);
  run<1, 0, 1, 0, 1>(sink); run<1, 0, 1, 0, 2>(sink); run<1, 0, 1, 0, 3>(sink); run<1, 0, 1, 0, 4>(sink); run<1, 0, 1, 0, 5>(sink);
  run<1, 0, 1, 0, 6>(sink); run<1, 0, 1, 0, 7>(sink);
  run_phased<1, 2>(sink); run_phased<2, 4>(sink); run_phased<3, 6>(sink); run_phased<4, 8>(sink);
  return 0;
}
