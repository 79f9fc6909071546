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

// The kernels advance moments with hardware fp64 atomics (global_atomic_add_f64 without return) and hand blocks of a queued launch
// over with agent-scope release / acquire: both are only guaranteed on ordinary (coarse-grained) device memory.  A buffer that is
// host-pinned, managed or fine-grained (hipExtMallocWithFlags(hipDeviceMallocFinegrained)) is refused here instead of giving
// silently wrong sums (include/pyhillfit_amd.h, "MEMORY KIND").  NULL passes: optional buffers are checked for NULL by their
// callers.  What the runtime cannot classify at all (hipPointerGetAttributes fails: e.g. a range mapped through the
// virtual-memory API) passes too — nothing is known against it, and the header says so.
// The verdict of the last few distinct (address, device) pairs is remembered (a steady-state launch loop passes the same three buffers
// every time: one runtime query per buffer, not per launch).  An entry is forgotten when it falls out of that window of 8, when
// phf_forget_device_memory_verdicts() is called (the *_init entry points call it: a new sampler starts from a clean slate, so a
// buffer freed and re-allocated as another kind at the same address between two samplers is always re-examined), and the cache is
// per thread.  What remains: a caller that frees a state / moments buffer and re-allocates it as managed or fine-grained memory at
// the same address WITHIN one sampler's launch loop, inside the next 8 checks — such a caller must call an *_init in between
// (include/pyhillfit_amd.h, "MEMORY KIND").
struct phf_memory_verdicts {
  static constexpr int kRemembered = 8;
  const void* ptr[kRemembered] = {};
  int dev[kRemembered] = {};
  int next = 0;
};
inline phf_memory_verdicts& phf_memory_verdict_cache() {
  static thread_local phf_memory_verdicts c;
  return c;
}
inline void phf_forget_device_memory_verdicts() { phf_memory_verdict_cache() = phf_memory_verdicts(); }

inline int phf_require_device_memory(const void* p, const char* what) {
  if (!p) return PHF_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  phf_memory_verdicts& c = phf_memory_verdict_cache();
  for (int i = 0; i < phf_memory_verdicts::kRemembered; ++i)
    if (c.ptr[i] == p && c.dev[i] == dev) return PHF_OK;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return PHF_OK;
  }
  const bool fine_grained = (attr.allocationFlags & hipDeviceMallocFinegrained) != 0;
  if (attr.type != hipMemoryTypeDevice || attr.isManaged || fine_grained) {
    std::snprintf(phf_error_buffer(), kPhfErrorBufferSize,
                  "%s: must be ordinary coarse-grained device memory (hipMalloc), not host-pinned, managed or fine-grained memory: "
                  "fp64 atomics and release/acquire hand-overs are not guaranteed there", what);
    return PHF_ERR_INVALID_ARGUMENT;
  }
  c.ptr[c.next] = p;
  c.dev[c.next] = dev;
  c.next = (c.next + 1) % phf_memory_verdicts::kRemembered;
  return PHF_OK;
}

// phf_simd_count() (public header): SIMDs of the current device, looked up once per device; defined in phf_capi.hip

#endif  // PHF_COMMON_H
