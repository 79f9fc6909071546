/* phf_math.h — bit-reproducible fp64 elementary functions for the PyHillFit MH kernels.
 *
 * Everything here is built from IEEE-754 +, -, *, /, sqrt, fma, rint and integer bit moves only,
 * in a FIXED evaluation order, so the same source gives bit-identical results on the gfx950
 * device (hipcc, -ffp-contract=off) and on the host (gcc, -ffp-contract=off -mfma).  That is
 * what makes "same Philox stream => same accept sequence, same chain, bit for bit" testable:
 * ROCm's OCML exp/log/erfc and glibc's differ in the last ulp and would not be.
 *
 * Coefficients are Chebyshev-interpolant (near-minimax) fits produced by tools/gen_math_coeffs.py
 * (mpmath, 60 digits); approximation errors are quoted per function.  No table lookups: every
 * lane runs the same instruction stream whatever its argument (no divergence inside a wave).
 *
 * Replaces, on the hot path, the third-party numerics the reference calls:
 *   numpy ** / 10**x           (python/doseresponse.py:84-88)      -> phf_exp of a log-domain argument
 *   scipy.stats.norm.logcdf/sf (python/doseresponse.py:218-219,244-245) -> phf_log_ndtr
 *   scipy.stats.norm.cdf       (python/PyHillFit.py:124)           -> phf_ndtr
 *   np.log / np.exp            (doseresponse.py:220,308; PyHillFit.py:135,146) -> phf_log / phf_exp
 */
#ifndef PHF_MATH_H
#define PHF_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PHF_HD static __host__ __device__ __forceinline__
#else
#define PHF_HD static inline
#endif

#define PHF_INF (__builtin_inf())
#define PHF_NAN (__builtin_nan(""))
#define PHF_LN2_HI 0x1.62e42fee00000p-1   /* ln2 with 21 trailing zero bits */
#define PHF_LN2_LO 0x1.a39ef35793c76p-33  /* ln2 - PHF_LN2_HI */
#define PHF_LOG2E 0x1.71547652b82fep+0
#define PHF_LN10 0x1.26bb1bbb55516p+1
#define PHF_INV_SQRT2 0x1.6a09e667f3bcdp-1
#define PHF_TWO_OVER_SQRTPI 0x1.20dd750429b6dp+0
#define PHF_INV_SQRTPI 0x1.20dd750429b6dp-1
#define PHF_2PI_2M32 0x1.921fb54442d18p-30 /* 2*pi / 2^32 */

PHF_HD uint64_t phf_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
PHF_HD double phf_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
PHF_HD double phf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
PHF_HD double phf_sqrt(double x) { return __builtin_sqrt(x); } /* correctly rounded on both sides (checked in tests) */
PHF_HD double phf_pow2i(int k) { return phf_from_bits((uint64_t)(k + 1023) << 52); } /* -1022 <= k <= 1023 */

/* exp(x).  k = rint(x/ln2), r = x - k ln2 (two fma), exp(r) = 1 + r + r^2 q(r), q degree 9
 * (approximation error 1.6e-17 relative on |r| <= ln2/2), result scaled by 2^k in two exact-or-
 * single-rounding multiplications so subnormal results round once, identically everywhere.      */
PHF_HD double phf_exp(double x) {
  if (!(x <= 709.782712893384)) return (x > 0.0) ? PHF_INF : x; /* overflow; NaN falls through as NaN */
  if (x < -745.2) return 0.0;
  const double kd = __builtin_rint(x * PHF_LOG2E);
  double r = phf_fma(kd, -PHF_LN2_HI, x);
  r = phf_fma(kd, -PHF_LN2_LO, r);
  double q = 0x1.af38a9b0ec855p-26;
  q = phf_fma(q, r, 0x1.289185613a3d6p-22);
  q = phf_fma(q, r, 0x1.71de0dae63bb3p-19);
  q = phf_fma(q, r, 0x1.a019b90d2ae7ap-16);
  q = phf_fma(q, r, 0x1.a01a01a7c41d5p-13);
  q = phf_fma(q, r, 0x1.6c16c1788bd90p-10);
  q = phf_fma(q, r, 0x1.11111111109b3p-7);
  q = phf_fma(q, r, 0x1.5555555553d63p-5);
  q = phf_fma(q, r, 0x1.5555555555556p-3);
  q = phf_fma(q, r, 0x1.0000000000001p-1);
  const double p = phf_fma(r * r, q, r) + 1.0;
  const int k = (int)kd;
  const int k1 = k >> 1;
  return (p * phf_pow2i(k1)) * phf_pow2i(k - k1);
}

/* log(x).  x = 2^k m, m in [sqrt(1/2), sqrt 2), f = m-1, s = f/(2+f), z = s^2,
 * log(1+f) = f - (f^2/2 - s (f^2/2 + z G(z))), G degree 6 (approximation error 4.6e-18).       */
PHF_HD double phf_log(double x) {
  uint64_t u = phf_bits(x);
  int k = 0;
  if (u < 0x0010000000000000ull || (u >> 63)) { /* +0, +subnormal, or sign bit set */
    if (x == 0.0) return -PHF_INF;
    if (u >> 63) return (x != x) ? x : PHF_NAN;
    x *= 0x1p54; u = phf_bits(x); k = -54;
  } else if (u >= 0x7ff0000000000000ull) {
    return x; /* +inf or NaN */
  }
  u += 0x3ff0000000000000ull - 0x3fe6a09e667f3bcdull;
  k += (int)(u >> 52) - 1023;
  u = (u & 0x000fffffffffffffull) + 0x3fe6a09e667f3bcdull;
  const double f = phf_from_bits(u) - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double g = 0x1.2b59b713616c9p-3;
  g = phf_fma(g, z, 0x1.39fe42e9740a7p-3);
  g = phf_fma(g, z, 0x1.7462b58e46ebep-3);
  g = phf_fma(g, z, 0x1.c71c62e26208bp-3);
  g = phf_fma(g, z, 0x1.2492492df3ba9p-2);
  g = phf_fma(g, z, 0x1.99999999952ccp-2);
  g = phf_fma(g, z, 0x1.5555555555558p-1);
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  const double t = phf_fma(dk, PHF_LN2_LO, s * phf_fma(z, g, hfsq));
  return phf_fma(dk, PHF_LN2_HI, f - (hfsq - t));
}

/* erfcx(y) = exp(y^2) erfc(y) for y >= 0:  (1+2y) erfcx(y) = P(t), t = (y-4)/(y+4) in [-1,1],
 * P degree 22 (approximation error 3.1e-16); one division serves both t and the final scaling. */
PHF_HD double phf_erfcx_nonneg(double y) {
  if (y > 1e100) return PHF_INV_SQRTPI / y;
  const double a = y + 4.0, b = phf_fma(2.0, y, 1.0);
  const double r = 1.0 / (a * b);
  const double t = ((y - 4.0) * b) * r;
  double p = 0x1.c2e324cb33784p-30;
  p = phf_fma(p, t, -0x1.dff032d300316p-28);
  p = phf_fma(p, t, -0x1.1f8f10ba20f78p-26);
  p = phf_fma(p, t, 0x1.8f0920c7d5e28p-24);
  p = phf_fma(p, t, 0x1.355884b1ca9fcp-24);
  p = phf_fma(p, t, -0x1.ae86b29807edbp-21);
  p = phf_fma(p, t, 0x1.3ebb0291516c9p-22);
  p = phf_fma(p, t, 0x1.7dcf4dcc6199dp-18);
  p = phf_fma(p, t, -0x1.789e79eb906c5p-17);
  p = phf_fma(p, t, -0x1.9928561ea5afcp-16);
  p = phf_fma(p, t, 0x1.3be0e09412ec0p-13);
  p = phf_fma(p, t, -0x1.a1e16f900a258p-13);
  p = phf_fma(p, t, -0x1.8d4aa41628fedp-11);
  p = phf_fma(p, t, 0x1.49c676f414b52p-8);
  p = phf_fma(p, t, -0x1.09623878c700ep-6);
  p = phf_fma(p, t, 0x1.3079ede17a234p-5);
  p = phf_fma(p, t, -0x1.0fb06dfe8afa8p-4);
  p = phf_fma(p, t, 0x1.7fee004ef1101p-4);
  p = phf_fma(p, t, -0x1.9ddb23c3e6861p-4);
  p = phf_fma(p, t, 0x1.16ecefcf9cb1ep-4);
  p = phf_fma(p, t, 0x1.f7f5df66fd40dp-7);
  p = phf_fma(p, t, -0x1.1df1ad154a1c8p-3);
  p = phf_fma(p, t, 0x1.3ba5916e9fd7fp+0);
  return (p * a) * r;
}

/* exp(-x^2/2) with the rounding error of x*x compensated (keeps the Gaussian tail to ~1 ulp). */
PHF_HD double phf_exp_mhalf_sq(double x) {
  const double hi = x * x;
  const double lo = phf_fma(x, x, -hi);
  const double e = phf_exp(-0.5 * hi);
  return phf_fma(e, -0.5 * lo, e);
}

/* log Phi(x), standard normal log-CDF.  x <= 0 (the only case the censored likelihood produces,
 * because predictions lie in [0,100]):  log(erfcx(-x/sqrt2)/2) - x^2/2, no cancellation anywhere.
 * x > 0:  log(1 - q), q = erfcx(x/sqrt2) exp(-x^2/2)/2, with the log1p correction term.        */
PHF_HD double phf_log_ndtr(double x) {
  if (x <= 0.0) {
    const double e = phf_erfcx_nonneg(-x * PHF_INV_SQRT2);
    return phf_fma(-0.5 * x, x, phf_log(0.5 * e));
  }
  if (!(x == x)) return x;
  const double q = 0.5 * phf_erfcx_nonneg(x * PHF_INV_SQRT2) * phf_exp_mhalf_sq(x);
  const double w = 1.0 - q;
  return phf_log(w) - ((w - 1.0) + q) / w;
}

/* Phi(x). */
PHF_HD double phf_ndtr(double x) {
  if (!(x == x)) return x;
  const double ax = (x < 0.0) ? -x : x;
  const double q = 0.5 * phf_erfcx_nonneg(ax * PHF_INV_SQRT2) * phf_exp_mhalf_sq(ax);
  return (x < 0.0) ? q : 1.0 - q;
}

/* sin and cos of 2*pi*w/2^32 for a 32-bit integer w: exact octant reduction in the integer
 * domain, |x| <= pi/8 kernels of degree 11/12 (approximation error 5e-18), exact rotations.    */
PHF_HD void phf_sincos_2pi_u32(uint32_t w, double* sn, double* cs) {
  const uint32_t q = ((w >> 28) + 1u) >> 1;                 /* nearest multiple of pi/4: 0..8 */
  const int64_t rem = (int64_t)w - ((int64_t)q << 29);       /* [-2^28, 2^28) */
  const double x = (double)rem * PHF_2PI_2M32;
  const double z = x * x;
  double ps = -0x1.ad54503fdffb8p-26;
  ps = phf_fma(ps, z, 0x1.71ddf0ef66ef1p-19);
  ps = phf_fma(ps, z, -0x1.a01a018fee5fbp-13);
  ps = phf_fma(ps, z, 0x1.111111110fd1dp-7);
  ps = phf_fma(ps, z, -0x1.5555555555554p-3);
  double pc = 0x1.1e5217c71f176p-29;
  pc = phf_fma(pc, z, -0x1.27e4d184456c9p-22);
  pc = phf_fma(pc, z, 0x1.a01a0196dbfc7p-16);
  pc = phf_fma(pc, z, -0x1.6c16c16c160afp-10);
  pc = phf_fma(pc, z, 0x1.5555555555555p-5);
  const double s = phf_fma(x * z, ps, x);
  const double c = phf_fma(z * z, pc, phf_fma(-0.5, z, 1.0));
  const double a = (q & 1u) ? PHF_INV_SQRT2 * (s + c) : s;  /* sin(x + (q&1) pi/4) */
  const double b = (q & 1u) ? PHF_INV_SQRT2 * (c - s) : c;  /* cos(x + (q&1) pi/4) */
  const uint32_t h = (q >> 1) & 3u;                          /* quarter turns */
  *sn = (h == 0u) ? a : (h == 1u) ? b : (h == 2u) ? -a : -b;
  *cs = (h == 0u) ? b : (h == 1u) ? -a : (h == 2u) ? -b : a;
}

/* Box-Muller pair from two 32-bit words: u1 = (w1+0.5)/2^32 in (0,1), angle = 2 pi w2/2^32.
 * |z| <= 6.66; the proposal stays symmetric, which is all Metropolis needs.                    */
PHF_HD void phf_box_muller(uint32_t w1, uint32_t w2, double* z0, double* z1) {
  const double u1 = ((double)w1 + 0.5) * 0x1p-32;
  const double rad = phf_sqrt(-2.0 * phf_log(u1));
  double sn, cs;
  phf_sincos_2pi_u32(w2, &sn, &cs);
  *z0 = rad * cs;
  *z1 = rad * sn;
}

/* 53-bit uniform on [0,1) from two words — numpy's random_sample() construction
 * (the reference's npr.rand(), python/PyHillFit.py:834).                                       */
PHF_HD double phf_uniform53(uint32_t w1, uint32_t w2) {
  return ((double)(w1 >> 5) * 67108864.0 + (double)(w2 >> 6)) * 0x1p-53;
}

#endif /* PHF_MATH_H */
