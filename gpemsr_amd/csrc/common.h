// Shared helpers for libgpemsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "gpemsr_hip.h"

namespace gpemsr {

// thread-local error string, reported through gpemsr_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GPEMSR_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return GPEMSR_OK;
}

#define GP_REQUIRE(cond, ...) \
  do { if (!(cond)) return ::gpemsr::fail(GPEMSR_EINVAL, __VA_ARGS__); } while (0)

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case GPEMSR_ACT_RELU: return v > 0.f ? v : 0.f;
    case GPEMSR_ACT_LRELU: return v > 0.f ? v : 0.1f * v;
    case GPEMSR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case GPEMSR_ACT_LRELU_SIGMOID: { float t = v > 0.f ? v : 0.1f * v; return 1.f / (1.f + expf(-t)); }
    default: return v;
  }
}

// Per-DEVICE "done once" flags for per-device settings (hipFuncSetAttribute of the dynamic LDS limit): bit d = done on device d.
// A plain static bool would leave a second device of the same process at the default limit; two threads racing here both set the
// attribute, which is harmless.  Devices >= 64 simply redo the call every time.
typedef std::atomic<unsigned long long> dev_once_t;
inline bool dev_once_begin(dev_once_t& m) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  return (m.load(std::memory_order_acquire) & (1ull << dev)) == 0ull;
}
inline void dev_once_done(dev_once_t& m) {
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) m.fetch_or(1ull << dev, std::memory_order_release);
}

// CUs of the CURRENT device, cached per device id (common.hip): grids of the persistent kernels
int device_cus();

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// gpemsr_conv_desc.transposed == 3: the Winograd F(2x2, 3x3) form (conv_wino.hip); name_buf != NULL: write the kernel's name, launch nothing
int conv2d_winograd(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap, int* parts_only = nullptr);
// gpemsr_conv_desc.transposed == 4: the 1-D Winograd F(2, 7) form of a 7x7 stride-1 convolution (conv7_wino.hip)
int conv2d_winograd7(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap);
int conv2d_winograd4(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap, int* parts_only = nullptr);
// gpemsr_conv_desc.transposed == 6: the 2-D Winograd F(2x2, 7x7) form of a 7x7 stride-1 convolution (conv7_wino2d.hip)
int conv2d_winograd77(const gpemsr_conv_desc* d, void* stream, char* name_buf, int name_cap);

}  // namespace gpemsr
