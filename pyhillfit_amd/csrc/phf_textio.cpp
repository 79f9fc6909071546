// phf_textio.cpp — the chain files' text, host side (no GPU code): rows of doubles in exactly the bytes numpy.savetxt writes with
// its default format '%.18e', ' ' between columns, '\n' after each row (python/PyHillFit.py:865-867,514-515; PyHillTemp.py:169).
//
// Why native: once sampling takes a second, formatting 75 001 x 4 numbers per pair through Python's `fmt % tuple(row)` (~0.4 us per
// number and core) is most of a run's wall time.  std::to_chars(double, scientific, 18) is libstdc++'s Ryu-printf:
// the correctly rounded 19 significant digits that printf("%.18e") and Python produce (ties to even on the exact binary value), at
// ~0.1 us per number.  Checked byte for byte against numpy.savetxt in tests/test_host.py (random bit patterns, subnormals, exact
// ties such as 2^-28, signed zeros, infinities, NaN).
#include "../../include/pyhillfit_textio.h"

#include <charconv>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

// one number, Python's spelling of the non-finite ones ('%.18e' % x): inf, -inf, nan
inline char* put(char* p, double v) {
  if (__builtin_expect(!std::isfinite(v), 0)) {
    if (std::isnan(v)) { std::memcpy(p, "nan", 3); return p + 3; }
    if (v < 0) { std::memcpy(p, "-inf", 4); return p + 4; }
    std::memcpy(p, "inf", 3); return p + 3;
  }
  return std::to_chars(p, p + 32, v, std::chars_format::scientific, 18).ptr;
}

constexpr int kMaxPerNumber = 28;   // '-' d '.' 18 digits 'e' sign 3 digits + separator, with slack

}  // namespace

extern "C" {

// rows x cols doubles (row r at data + r * row_stride) -> text in out; returns the number of bytes, or -(bytes needed) if out_capacity is too small
int64_t phf_format_rows(const double* data, int64_t rows, int64_t cols, int64_t row_stride, char* out, int64_t out_capacity) {
  const int64_t need = rows * cols * kMaxPerNumber;
  if (out_capacity < need) return -need;
  char* p = out;
  for (int64_t r = 0; r < rows; ++r) {
    const double* row = data + r * row_stride;
    for (int64_t c = 0; c < cols; ++c) {
      p = put(p, row[c]);
      *p++ = (c + 1 < cols) ? ' ' : '\n';
    }
    if (cols == 0) *p++ = '\n';
  }
  return p - out;
}

// write (append != 0: append) `header` followed by the rows to `path`; returns 0 or an errno value
int phf_savetxt(const char* path, int append, const char* header, int64_t header_len, const double* data, int64_t rows, int64_t cols,
                int64_t row_stride) {
  std::FILE* f = std::fopen(path, append ? "ab" : "wb");
  if (!f) return errno ? errno : EIO;
  int rc = 0;
  if (header_len > 0 && std::fwrite(header, 1, (size_t)header_len, f) != (size_t)header_len) rc = errno ? errno : EIO;
  const int64_t chunk = 8192;
  std::vector<char> buf((size_t)(chunk * (cols > 0 ? cols : 1) * kMaxPerNumber));
  for (int64_t r0 = 0; r0 < rows && rc == 0; r0 += chunk) {
    const int64_t n = (rows - r0 < chunk) ? rows - r0 : chunk;
    const int64_t bytes = phf_format_rows(data + r0 * row_stride, n, cols, row_stride, buf.data(), (int64_t)buf.size());
    if (bytes < 0 || std::fwrite(buf.data(), 1, (size_t)bytes, f) != (size_t)bytes) rc = errno ? errno : EIO;
  }
  if (std::fclose(f) != 0 && rc == 0) rc = errno ? errno : EIO;
  return rc;
}

}  // extern "C"
