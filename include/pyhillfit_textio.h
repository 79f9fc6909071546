/* pyhillfit_textio.h — C ABI of pyhillfit_amd/lib/libphf_textio.so: the chain files' text (host only, no GPU code).
 *
 * Replaces the reference's np.savetxt calls on the output side of the sampling step —
 *   python/PyHillFit.py:865-867 (single-level chain), :514-515,522-525 (hierarchical chain, alpha/mu samples),
 *   python/PyHillTemp.py:169 (tempered chains), python/construct_hierarchical_cdfs.py:130-131,144-149 (tables) —
 * with the same bytes: default format '%.18e', ' ' between columns, '\n' after each row, 'inf' / '-inf' / 'nan' as Python
 * spells them.  Source: pyhillfit_amd/csrc/phf_textio.cpp; bound with ctypes in pyhillfit_amd/chainio.py (write_text).      */
#ifndef PYHILLFIT_TEXTIO_H
#define PYHILLFIT_TEXTIO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rows x cols doubles, row r at data + r * row_stride  ->  text in out.
 * Returns the number of bytes written, or -(capacity needed) when out_capacity is too small (28 bytes per number suffice). */
int64_t phf_format_rows(const double* data, int64_t rows, int64_t cols, int64_t row_stride, char* out, int64_t out_capacity);

/* Create (append == 0: truncate, then write the header_len bytes of header) or append to `path`, then write the rows.
 * Returns 0 or an errno value. */
int phf_savetxt(const char* path, int append, const char* header, int64_t header_len, const double* data, int64_t rows,
                int64_t cols, int64_t row_stride);

#ifdef __cplusplus
}
#endif
#endif /* PYHILLFIT_TEXTIO_H */
