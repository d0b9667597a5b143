// Error reporting and device queries for libkgdet_hip.so.
#include "common.h"

namespace kgdet {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int g_options[KGDET_OPT_COUNT] = {0};

int cu_count() {
  static thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != cached_dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    cached = prop.multiProcessorCount;
    cached_dev = dev;
  }
  return cached;
}

}  // namespace kgdet

extern "C" {
const char *kgdet_last_error(void) { return kgdet::g_err; }
int kgdet_version(void) { return 1; }
int kgdet_device_cu_count(void) { return kgdet::cu_count(); }
int kgdet_set_option(int32_t option, int32_t value) {
  if (option < 0 || option >= KGDET_OPT_COUNT) {
    kgdet::set_error("unknown option %d", option);
    return KGDET_E_SHAPE;
  }
  kgdet::g_options[option] = value;
  return KGDET_OK;
}
}
