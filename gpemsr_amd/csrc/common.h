// Shared helpers for libgpemsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
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

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace gpemsr
