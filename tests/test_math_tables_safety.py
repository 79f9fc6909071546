"""The elementary functions of pyhillfit_amd/csrc/phf_math.h index small tables (LDS on the device) with bits of their ARGUMENT.
A kernel evaluates them on whatever a proposal produces — negative scales, NaN, infinities — before a select discards the result,
so no bit pattern may index outside a table.  Checked here on the host build under AddressSanitizer + UBSan (the device cannot
run sanitizers on this pool): every table function over random 64-bit patterns, special values and the interval joints."""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "phf_math.h"
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t next(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double from(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
int main(void) {
  const double special[] = {0.0, -0.0, 1.0, -1.0, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, -1.7976931348623157e308,
                            __builtin_inf(), -__builtin_inf(), __builtin_nan(""), 6.0, 5.999999999999999, 131071.0, 131072.0, 1e300, -1e300,
                            709.782712893384, -745.2, 0.125, 0.375, 0.7071067811865476, 1.4142135623730951};
  volatile double sink = 0.0;
  PHF_KFETCH_V(ke, phf_k_exp, PHF_K_EXP_N);
  PHF_KFETCH_V(kl, phf_k_log, PHF_K_LOG_N);
  long n = 0;
  for (long i = 0; i < 3000000 + (long)(sizeof special / sizeof special[0]); ++i) {
    const uint64_t u = next();
    const double x = i < 3000000 ? from(u) : special[i - 3000000];
    sink += phf_exp_fast_k(x, ke) == 0.5;          /* clamps its argument */
    sink += phf_log_pos_k(x, kl) == 0.5;           /* any bit pattern */
    sink += phf_log_fast_k(x, kl) == 0.5;
    sink += phf_normal_u32((uint32_t)u) == 0.5;
    sink += phf_log_ndtr_tab(x, -x * PHF_INV_SQRT2) == 0.5;
    sink += phf_log_ndtr_tab(x, x) == 0.5;         /* y of any sign and size */
    sink += phf_erfc_tab(x) == 0.5;
    ++n;
  }
  printf("evaluated %ld patterns\n", n);
  return 0;
}
'''


def test_no_bit_pattern_indexes_outside_a_table(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "tables_safety.c"
    src.write_text(SRC)
    exe = tmp_path / "tables_safety"
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-mfma", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(REPO, "pyhillfit_amd", "csrc"), str(src), "-o", str(exe), "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this gcc has no sanitizer runtime: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "evaluated 3000024 patterns" in run.stdout
