// phf_common.h — error reporting shared by the C-ABI translation units (host side only).
#ifndef PHF_COMMON_H
#define PHF_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/pyhillfit_amd.h"

// thread-local message buffer behind phf_last_error(); defined in phf_capi.hip
char* phf_error_buffer();
constexpr int kPhfErrorBufferSize = 512;

inline int phf_fail(int code, const char* msg) {
  std::snprintf(phf_error_buffer(), kPhfErrorBufferSize, "%s", msg);
  return code;
}

inline int phf_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return PHF_OK;
  std::snprintf(phf_error_buffer(), kPhfErrorBufferSize, "%s: %s", what, hipGetErrorString(e));
  return PHF_ERR_HIP;
}

// phf_simd_count() (public header): SIMDs of the current device, looked up once per device; defined in phf_capi.hip

#endif  // PHF_COMMON_H
