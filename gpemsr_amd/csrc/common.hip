// Error reporting + device info for libgpemsr_hip.so
#include "common.h"
#include <string.h>

namespace gpemsr {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}
}  // namespace gpemsr

extern "C" int gpemsr_abi_version(void) { return GPEMSR_ABI_VERSION; }
extern "C" const char* gpemsr_last_error(void) { return gpemsr::err_buf(); }
extern "C" int gpemsr_device_info(char* name, int name_len, int* cu_count, int64_t* hbm_bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return gpemsr::fail(GPEMSR_ELAUNCH, "device_info: no HIP device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return gpemsr::fail(GPEMSR_ELAUNCH, "device_info: query failed");
  if (name && name_len > 0) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return GPEMSR_OK;
}
