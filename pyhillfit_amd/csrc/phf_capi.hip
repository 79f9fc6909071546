// phf_capi.hip — library-wide pieces of the C ABI (include/pyhillfit_amd.h).
#include "phf_common.h"

char* phf_error_buffer() {
  static thread_local char buf[kPhfErrorBufferSize] = "";
  return buf;
}

extern "C" int phf_simd_count(void) {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 1024; }
  if (dev < 0 || dev >= 64) return 1024;
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    cached[dev] = 4 * cus;
  }
  return cached[dev];
}

extern "C" const char* phf_last_error(void) { return phf_error_buffer(); }
