// phf_capi.hip — library-wide pieces of the C ABI (include/pyhillfit_amd.h).
#include "phf_common.h"

char* phf_error_buffer() {
  static thread_local char buf[kPhfErrorBufferSize] = "";
  return buf;
}

extern "C" const char* phf_last_error(void) { return phf_error_buffer(); }
