// Shared host-side helpers of libmw_cdna4.so (error reporting; no CPU fallbacks live here).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <cstdio>

namespace mw {

void set_error(const std::string &msg);           // defined in mw_host.cpp

#define MW_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess) {                                                                           \
      char b__[512];                                                                                   \
      snprintf(b__, sizeof(b__), "%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      mw::set_error(b__);                                                                              \
      return 1;                                                                                        \
    }                                                                                                  \
  } while (0)

#define MW_FAIL(msg)                                                                                   \
  do { mw::set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); return 1; } while (0)

#define MW_LAUNCH_CHECK() MW_HIP(hipGetLastError())

} // namespace mw
