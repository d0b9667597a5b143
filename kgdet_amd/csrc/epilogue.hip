// Fused convolution epilogue for inference: x = [relu](x + bias[c] [+ residual]) in place, NCHW, fp32 or bf16.
// With BatchNorm folded into the convolution (kgdet_amd/backbone.py conv_bn) a ResNet bottleneck would otherwise
// run a bias-add, a residual add and a clamp as three full passes over the activation; MIOpen's convolution has no
// fused epilogue on this path.  One 16-byte load / store per thread per tensor, grid-stride over (n, c) planes.
#include "common.h"

namespace kgdet {

template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int n = 4;
};
template <>
struct Vec16<__bf16> {
  static constexpr int n = 8;
};

template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_kernel(T *__restrict__ x, const float *__restrict__ bias,
                                                       const T *__restrict__ res, int C, long long HW, long long planes) {
  constexpr int V = Vec16<T>::n;
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const bool vec_ok = (HW % V) == 0;
  for (long long pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const float b = bias ? bias[pl % C] : 0.0f;
    T *xp = x + pl * HW;
    const T *rp = RES ? res + pl * HW : nullptr;
    if (vec_ok) {
      const long long nv = HW / V;
      for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += gridDim.x * 256LL) {
        vec_t v = reinterpret_cast<vec_t *>(xp)[i];
        vec_t r;
        if (RES) r = reinterpret_cast<const vec_t *>(rp)[i];
#pragma unroll
        for (int k = 0; k < V; ++k) {
          float f = (float)v[k] + b;
          if (RES) f += (float)r[k];
          if (RELU) f = fmaxf(f, 0.0f);
          v[k] = (T)f;
        }
        reinterpret_cast<vec_t *>(xp)[i] = v;
      }
    } else {
      for (long long i = blockIdx.x * 256LL + threadIdx.x; i < HW; i += gridDim.x * 256LL) {
        float f = (float)xp[i] + b;
        if (RES) f += (float)rp[i];
        if (RELU) f = fmaxf(f, 0.0f);
        xp[i] = (T)f;
      }
    }
  }
}

// channels-last storage ([N, HW, C], C a multiple of the vector width): a flat pass, bias index = element % C
template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(T *__restrict__ x, const float *__restrict__ bias,
                                                            const T *__restrict__ res, int C, long long nvec) {
  constexpr int V = Vec16<T>::n;
  typedef T vec_t __attribute__((ext_vector_type(V)));
  // the channel of a vector only depends on i mod (C / V): track it incrementally instead of dividing per element
  const unsigned cv = (unsigned)(C / V);
  const long long step = gridDim.x * 256LL;
  const unsigned cstep = (unsigned)(step % cv);
  long long i = blockIdx.x * 256LL + threadIdx.x;
  unsigned ci = (unsigned)(i % cv);
  for (; i < nvec; i += step) {
    vec_t v = reinterpret_cast<vec_t *>(x)[i];
    vec_t r;
    if (RES) r = reinterpret_cast<const vec_t *>(res)[i];
    float bv[V];
    if (bias) {
#pragma unroll
      for (int k = 0; k < V; k += 4) {
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + ci * V + k);
        bv[k] = b4.x; bv[k + 1] = b4.y; bv[k + 2] = b4.z; bv[k + 3] = b4.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) bv[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float f = (float)v[k] + bv[k];
      if (RES) f += (float)r[k];
      if (RELU) f = fmaxf(f, 0.0f);
      v[k] = (T)f;
    }
    reinterpret_cast<vec_t *>(x)[i] = v;
    ci += cstep;
    if (ci >= cv) ci -= cv;
  }
}

// The stem at inference (folded BatchNorm): y = maxpool3x3/s2/p1(relu(x + bias[c])) on channels-last tensors in one pass --
// instead of the in-place epilogue over the full-resolution map and torch's pooling kernel (bf16, batch 8: 275 MB read +
// written, then read again).  Each element is rounded to T after bias + ReLU, as the two-pass route does, so the results
// are identical.  x [N, H, W, C] -> y [N, Ho, Wo, C]; a thread owns one output pixel's vector of 16 bytes.
template <typename T>
__global__ __launch_bounds__(256) void bias_relu_maxpool_nhwc_kernel(const T *__restrict__ x, const float *__restrict__ bias,
                                                                     T *__restrict__ y, int C, int H, int W, int Ho, int Wo,
                                                                     long long nvec) {
  constexpr int V = Vec16<T>::n;
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int cv = C / V;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nvec; i += gridDim.x * 256LL) {
    const int c = (int)(i % cv);
    long long p = i / cv;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const long long n = p / Ho;
    float bv[V], m[V];
#pragma unroll
    for (int k = 0; k < V; k += 4) {
      const float4 b4 = bias ? *reinterpret_cast<const float4 *>(bias + c * V + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[k] = b4.x; bv[k + 1] = b4.y; bv[k + 2] = b4.z; bv[k + 3] = b4.w;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) m[k] = 0.0f;   // relu(.) >= 0 and every window holds a valid element
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int iy = 2 * oy + dy;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int ix = 2 * ox + dx;
        if (ix < 0 || ix >= W) continue;
        const vec_t v = reinterpret_cast<const vec_t *>(x)[((n * H + iy) * W + ix) * cv + c];
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k], (float)(T)fmaxf((float)v[k] + bv[k], 0.0f));
      }
    }
    vec_t o;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = (T)m[k];
    reinterpret_cast<vec_t *>(y)[i] = o;
  }
}

}  // namespace kgdet

using namespace kgdet;

extern "C" int kgdet_bias_relu_maxpool_nhwc(const void *x, const float *bias, void *y, int64_t N, int32_t C, int32_t H,
                                            int32_t W, int32_t dtype, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && H > 0 && W > 0 && (dtype == 0 || dtype == 1), "bad arguments");
  if (N == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x && y, "null pointer");
  const int V = dtype == 0 ? 4 : 8;
  if (C % V != 0) {
    set_error("bias_relu_maxpool_nhwc: C = %d is not a multiple of %d", C, V);
    return KGDET_E_UNSUPPORTED;
  }
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long nvec = (long long)N * Ho * Wo * (C / V);
  const long long want = (nvec + 255) / 256;
  dim3 grid((unsigned)(want > 65536 ? 65536 : want));
  if (dtype == 0)
    hipLaunchKernelGGL(bias_relu_maxpool_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)x, bias,
                       (float *)y, C, H, W, Ho, Wo, nvec);
  else
    hipLaunchKernelGGL(bias_relu_maxpool_nhwc_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, bias,
                       (__bf16 *)y, C, H, W, Ho, Wo, nvec);
  KGDET_CHECK_LAUNCH("bias_relu_maxpool_nhwc");
  return KGDET_OK;
}

extern "C" int kgdet_bias_act(void *x, const float *bias, const void *residual, int64_t N, int32_t C, int64_t HW,
                              int32_t dtype, int32_t relu, int32_t channels_last, void *stream) {
  KGDET_CHECK_SHAPE(N >= 0 && C > 0 && HW >= 0, "bad sizes");
  KGDET_CHECK_SHAPE(dtype == 0 || dtype == 1, "dtype must be 0 (float32) or 1 (bfloat16)");
  if (N * HW == 0) return KGDET_OK;
  KGDET_CHECK_SHAPE(x != nullptr, "null pointer");
  const long long planes = N * C;
  const int per = dtype == 0 ? 4 : 8;
  if (channels_last) {
    if (C % per != 0) {
      set_error("channels-last epilogue needs C %% %d == 0 (got C=%d)", per, C);
      return KGDET_E_UNSUPPORTED;
    }
    const long long nvec = N * HW * C / per;
    const long long want = (nvec + 255) / 256;
    dim3 grid((unsigned)(want > 16384 ? 16384 : want));
#define LAUNCH_CL(T, RES, RELU)                                                                              \
  hipLaunchKernelGGL((bias_act_nhwc_kernel<T, RES, RELU>), grid, dim3(256), 0, (hipStream_t)stream, (T *)x, \
                     bias, (const T *)residual, C, nvec)
    if (dtype == 0) {
      if (residual) { if (relu) LAUNCH_CL(float, true, true); else LAUNCH_CL(float, true, false); }
      else { if (relu) LAUNCH_CL(float, false, true); else LAUNCH_CL(float, false, false); }
    } else {
      if (residual) { if (relu) LAUNCH_CL(__bf16, true, true); else LAUNCH_CL(__bf16, true, false); }
      else { if (relu) LAUNCH_CL(__bf16, false, true); else LAUNCH_CL(__bf16, false, false); }
    }
#undef LAUNCH_CL
    KGDET_CHECK_LAUNCH("bias_act_nhwc");
    return KGDET_OK;
  }
  const int gx = (int)((HW / per + 255) / 256) > 0 ? (int)((HW / per + 255) / 256) : 1;
  dim3 grid(gx > 64 ? 64 : gx, planes > 4096 ? 4096 : (int)planes);
#define LAUNCH(T, RES, RELU)                                                                                      \
  hipLaunchKernelGGL((bias_act_kernel<T, RES, RELU>), grid, dim3(256), 0, (hipStream_t)stream, (T *)x, bias, \
                     (const T *)residual, C, (long long)HW, planes)
  if (dtype == 0) {
    if (residual) { if (relu) LAUNCH(float, true, true); else LAUNCH(float, true, false); }
    else { if (relu) LAUNCH(float, false, true); else LAUNCH(float, false, false); }
  } else {
    if (residual) { if (relu) LAUNCH(__bf16, true, true); else LAUNCH(__bf16, true, false); }
    else { if (relu) LAUNCH(__bf16, false, true); else LAUNCH(__bf16, false, false); }
  }
#undef LAUNCH
  KGDET_CHECK_LAUNCH("bias_act");
  return KGDET_OK;
}
