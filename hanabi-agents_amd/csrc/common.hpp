// common.hpp — error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace hb {
extern thread_local std::string g_error;
int fail(int code, const char* fmt, ...);
}  // namespace hb

// HB_ERR_HIP = -3 (include/hanabi_hip.h)
#define HB_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t hb_err_ = (expr);                                                                  \
    if (hb_err_ != hipSuccess) return hb::fail(-3, "%s failed: %s", #expr, hipGetErrorString(hb_err_)); \
  } while (0)

#define HB_HIP_OR(expr, cleanup)                                                                  \
  do {                                                                                            \
    hipError_t hb_err_ = (expr);                                                                  \
    if (hb_err_ != hipSuccess) {                                                                  \
      cleanup;                                                                                    \
      return hb::fail(-3, "%s failed: %s", #expr, hipGetErrorString(hb_err_));                    \
    }                                                                                             \
  } while (0)
