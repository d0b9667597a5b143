// Shared host-side helpers for libkgdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/kgdet_hip.h"

namespace kgdet {

void set_error(const char *fmt, ...);

#define KGDET_CHECK_SHAPE(cond, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      ::kgdet::set_error(__VA_ARGS__); \
      return KGDET_E_SHAPE;            \
    }                                  \
  } while (0)

// after a kernel launch: report (not printf, as the reference does) launch failures
#define KGDET_CHECK_LAUNCH(what)                                                      \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      ::kgdet::set_error("%s: HIP error %d (%s)", what, (int)e__, hipGetErrorString(e__)); \
      return KGDET_E_HIP;                                                             \
    }                                                                                 \
  } while (0)

#define KGDET_HIP_TRY(expr)                                                            \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      ::kgdet::set_error("%s: HIP error %d (%s)", #expr, (int)e__, hipGetErrorString(e__)); \
      return KGDET_E_HIP;                                                              \
    }                                                                                  \
  } while (0)

extern int g_options[];  // kgdet_set_option (include/kgdet_hip.h KGDET_OPT_*)
int cu_count();  // cached multiProcessorCount of the current device

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace kgdet
