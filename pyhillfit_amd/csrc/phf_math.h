/* phf_math.h — bit-reproducible fp64 elementary functions for the PyHillFit MH kernels.
 *
 * Everything here is built from IEEE-754 +, -, *, /, sqrt, fma, rint, min/max and integer bit moves only, in
 * a FIXED evaluation order, so the same source gives bit-identical results on the gfx950 device (hipcc,
 * -ffp-contract=off) and on the host (gcc, -ffp-contract=off -mfma).  That is what makes "same Philox
 * stream => same accept sequence, same chain, bit for bit" testable: ROCm's OCML exp/log/erfc and glibc's
 * differ in the last ulp and would not be.
 *
 * Shaped for CDNA4's fp64 VALU (measured with tools/microbench.hip on MI355X: v_fma_f64 2.25 ns per
 * wave-instruction per SIMD, v_mov_b64 2.1 ns, IEEE division ~26 ns, v_rcp/v_sqrt_f64 7 ns):
 *   - polynomial steps are literal 3-operand v_fma_f64 with the coefficient in an SGPR pair fetched through the
 *     scalar cache or held in VGPRs (PHF_KFETCH / PHF_KFETCH_V below);
 *   - polynomials are split into even/odd halves (two independent dependency chains) so a lone wave on a
 *     SIMD is not latency-bound;
 *   - the *_fast / *_core entry points used inside the kernels are branch-free: range problems are handled by
 *     clamping and IEEE overflow/underflow, never by divergent control flow;
 *   - every division is exposed (reduce / finish pairs) so callers can share one IEEE division among several
 *     evaluations (batched reciprocal, see phf_model.h).
 * No table lookups: every lane runs the same instruction stream whatever its argument.
 *
 * Coefficients: Chebyshev-interpolant (near-minimax) fits from tools/gen_math_coeffs.py (mpmath, 60 digits);
 * approximation errors are quoted per function.
 *
 * Replaces, on the hot path, the third-party numerics the reference calls:
 *   numpy ** / 10**x           (python/doseresponse.py:84-88)            -> phf_exp of a log-domain argument
 *   scipy.stats.norm.logcdf/sf (python/doseresponse.py:218-219,244-245)  -> phf_log_ndtr
 *   scipy.stats.norm.cdf       (python/PyHillFit.py:124)                 -> phf_ndtr
 *   np.log / np.exp            (doseresponse.py:220,308; PyHillFit.py:135,146) -> phf_log / phf_exp
 */
#ifndef PHF_MATH_H
#define PHF_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PHF_HD static __host__ __device__ __forceinline__
#else
#define PHF_HD static inline __attribute__((always_inline))
#endif

/* Polynomial coefficients and how they reach the VALU (hipcc on its own emits "v_mov_b64 acc, coef ; v_fmac_f64 acc, p, t"
 * for a constant addend — two fp64-rate instructions per term; and would hoist all ~60 coefficients out of the MH
 * loop, pinning 120+ registers):
 *   PHF_FMA_K / PHF_FMA_KV   p*t + c as a literal 3-operand v_fma_f64 with the coefficient in an SGPR / a VGPR pair;
 *   PHF_KFETCH(k, table, n)  the n coefficients loaded through the scalar cache into SGPRs where the macro stands;
 *   PHF_KFETCH_V(k, table, n) the n coefficients placed in VGPRs (done once per kernel for the exp and log tables, which
 *                            every polynomial of an iteration uses: measured, a lone wave per SIMD cannot hide the
 *                            ~100-cycle scalar-load latency of refetching them).
 * On the host all of these are plain C.                                                                          */
#if defined(__HIPCC__)
#define PHF_UNROLL _Pragma("unroll")
#else
#define PHF_UNROLL
#endif
typedef const double* phf_ktab;   /* coefficients of one polynomial, already in registers (device) / the table itself (host) */
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(4))) * phf_ktab4;
/* SGPR-resident: loaded through the scalar cache where the macro stands (erfcx, sin/cos: big, used in one phase) */
#define PHF_KFETCH(name, table, n)                                                          \
  double name##_buf[n];                                                                     \
  {                                                                                         \
    phf_ktab4 phf_p_ = (phf_ktab4)(table);                                                  \
    asm volatile("" : "+s"(phf_p_));                                                        \
    PHF_UNROLL                                                                              \
    for (int phf_i_ = 0; phf_i_ < (n); ++phf_i_) name##_buf[phf_i_] = phf_p_[phf_i_];       \
  }                                                                                         \
  const phf_ktab name = name##_buf
/* VGPR-resident: materialised once (kernels do it before the MH loop) and kept in vector registers (exp, log: 17
 * coefficients used by every polynomial of the iteration — no scalar-load latency, no SGPR pressure) */
#define PHF_KFETCH_V(name, table, n)                                                        \
  double name##_buf[n];                                                                     \
  PHF_UNROLL                                                                                \
  for (int phf_i_ = 0; phf_i_ < (n); ++phf_i_) {                                            \
    name##_buf[phf_i_] = (table)[phf_i_];                                                   \
    asm volatile("" : "+v"(name##_buf[phf_i_]));                                            \
  }                                                                                         \
  const phf_ktab name = name##_buf
/* `name` = `resident` when `have` (a compile-time constant) is non-zero, else fetched like PHF_KFETCH */
#define PHF_KFETCH_UNLESS(name, have, resident, table, n)                                   \
  double name##_buf[n];                                                                     \
  if (!(have)) {                                                                            \
    phf_ktab4 phf_p_ = (phf_ktab4)(table);                                                  \
    asm volatile("" : "+s"(phf_p_));                                                        \
    PHF_UNROLL                                                                              \
    for (int phf_i_ = 0; phf_i_ < (n); ++phf_i_) name##_buf[phf_i_] = phf_p_[phf_i_];       \
  }                                                                                         \
  const phf_ktab name = (have) ? (resident) : name##_buf
/* SGPR-resident coefficient.  Inline asm pins the 3-operand form with the SGPR pair as the addend (left to itself hipcc copies
 * some coefficients to VGPRs and uses v_fmac: more VALU instructions, which is what a SIMD shared by two wavefronts pays for),
 * at the price of an `s_nop` between dependent steps (see PHF_FMA_KV).  A translation unit whose wavefronts run one per SIMD,
 * where an s_nop costs as much as an fma, defines PHF_FMA_K_AS_BUILTIN before including this header (phf_hierarchical.hip:
 * 486 -> 248 scalar instructions per iteration of the Ne = 3 kernel, VALU count unchanged). */
#if defined(PHF_FMA_K_AS_BUILTIN)
#define PHF_FMA_K(p, t, c) __builtin_fma((p), (t), (c))
#else
#define PHF_FMA_K(p, t, c) __extension__({ double phf_r_; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(phf_r_) : "v"(p), "v"(t), "s"(c)); phf_r_; })
#endif
/* VGPR-resident coefficient: the compiler itself emits the 3-operand v_fma_f64 here (the coefficient register stays live, so
 * v_fmac cannot overwrite it), and — unlike around an inline-asm statement, whose instruction class the hazard recogniser cannot
 * see — it inserts no `s_nop` between two dependent steps: 540 s_nop per iteration of the Ne = 3 hierarchical kernel, ~70 of the
 * single-level one, each a full issue slot of a wavefront that has its SIMD to itself. */
#define PHF_FMA_KV(p, t, c) __builtin_fma((p), (t), (c))
#else
#define PHF_KFETCH_UNLESS(name, have, resident, table, n) const phf_ktab name = (have) ? (resident) : (table)
#define PHF_KFETCH(name, table, n) const phf_ktab name = (table)
#define PHF_KFETCH_V(name, table, n) const phf_ktab name = (table)
#define PHF_FMA_K(p, t, c) __builtin_fma((p), (t), (c))
#define PHF_FMA_KV(p, t, c) __builtin_fma((p), (t), (c))
#endif
#define PHF_KTABLE static const double   /* internal linkage: addressed pc-relatively (a __constant__ symbol goes through the GOT: one more dependent load per fetch) */

#define PHF_INF (__builtin_inf())
#define PHF_NAN (__builtin_nan(""))
#define PHF_DBL_MIN 0x1p-1022
#define PHF_LN2_HI 0x1.62e42fee00000p-1   /* ln2 with 21 trailing zero bits */
#define PHF_LN2_LO 0x1.a39ef35793c76p-33  /* ln2 - PHF_LN2_HI */
#define PHF_LOG2E 0x1.71547652b82fep+0
#define PHF_LN10 0x1.26bb1bbb55516p+1
#define PHF_INV_SQRT2 0x1.6a09e667f3bcdp-1
#define PHF_INV_SQRTPI 0x1.20dd750429b6dp-1
#define PHF_2PI_2M32 0x1.921fb54442d18p-30 /* 2*pi / 2^32 */

PHF_HD uint64_t phf_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
PHF_HD double phf_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
PHF_HD double phf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
PHF_HD double phf_sqrt(double x) { return __builtin_sqrt(x); } /* correctly rounded on both sides (checked in tests) */

/* Division and square root of the MH loops.  hipcc expands an fp64 division into v_div_scale x2, v_rcp_f64, two Newton steps,
 * q = n y, r = n - d q, v_div_fmas, v_div_fixup (11 instructions) and sqrt into a scaled v_rsq_f64 iteration (18): the scaling
 * and fix-up only serve operands within 2^+-~250 of the ends of the exponent range, zeros, infinities and NaNs.  The loops divide
 * by products of O(1)..1e70 numbers and take roots of O(1e-30)..O(1e30) numbers, so the device versions below are the SAME
 * iterations without the range handling — 7 / 8 / 10 instructions — and return the same correctly rounded result (which is what
 * the host computes with / and sqrt): bit-identity with the twin is kept and checked (tests/test_gpu_parity.py, 400 k arguments
 * per function across 2^-600..2^600).  Outside that range (never reached by a live chain: such operands only arise where the
 * target is -inf anyway and a select discards them) device and host may differ in NaN-versus-infinity.                        */
#if defined(__HIP_DEVICE_COMPILE__)
PHF_HD double phf_rcp_refined_(double d) {                  /* 1/d to ~0.5 ulp: hardware estimate + two Newton steps */
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  return __builtin_fma(y, e, y);
}
PHF_HD double phf_rcp(double d) {                           /* 1.0 / d */
  const double y = phf_rcp_refined_(d);
  const double r = __builtin_fma(-d, y, 1.0);               /* q = 1.0 * y = y exactly */
  return __builtin_fma(r, y, y);
}
PHF_HD double phf_div(double n, double d) {                 /* n / d */
  const double y = phf_rcp_refined_(d);
  const double q = n * y;
  const double r = __builtin_fma(-d, q, n);
  return __builtin_fma(r, y, q);
}
PHF_HD double phf_sqrt_pos(double x) {                      /* sqrt(x), x > 0 */
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
#else
PHF_HD double phf_rcp(double d) { return 1.0 / d; }
PHF_HD double phf_div(double n, double d) { return n / d; }
PHF_HD double phf_sqrt_pos(double x) { return __builtin_sqrt(x); }
#endif
/* sqrt(x) for x >= 0 with sqrt(0) = 0 (pivots of a degenerate factor); anything else non-positive also gives 0 */
PHF_HD double phf_sqrt_nonneg(double x) {
  const double r = phf_sqrt_pos(x);
  return (x > 0.0) ? r : 0.0;
}
PHF_HD double phf_pow2i(int k) { return phf_from_bits((uint64_t)(k + 1023) << 52); } /* -1022 <= k <= 1023 */

/* ------------------------------------------------------------------------------------------------ exp
 * k = nearest integer to x/ln2 (magic-number add), r = x - k ln2 (two fma), exp(r) = 1 + r + r^2 q(r), q degree 9
 * (approximation error 1.6e-17 relative on |r| <= ln2/2), scaled by 2^k with ldexp (subnormal results round once).  The argument is clamped to [-746, 710]: the scaling then overflows to +inf /
 * underflows to 0 by itself, no branches.  phf_exp_fast(NaN) = 0 (min/max drop the NaN); phf_exp keeps NaN. */
/* [10] is not a coefficient: the constant 100 of the percent scale, riding along in the table every target holds in registers — a
 * double that is not an inline constant costs two v_mov_b32 at EVERY use otherwise (hipcc rematerialises it: 10 per single-level
 * iteration, 24 per hierarchical one) */
#define PHF_K_EXP_N 11
#define PHF_K100(k_exp) ((k_exp)[10])
PHF_KTABLE phf_k_exp[PHF_K_EXP_N] = {   /* (exp(r)-1-r)/r^2, coefficient of r^i */
    0x1.0000000000001p-1, 0x1.5555555555556p-3, 0x1.5555555553d63p-5, 0x1.11111111109b3p-7, 0x1.6c16c1788bd90p-10,
    0x1.a01a01a7c41d5p-13, 0x1.a019b90d2ae7ap-16, 0x1.71de0dae63bb3p-19, 0x1.289185613a3d6p-22, 0x1.af38a9b0ec855p-26,
    100.0};

#define PHF_EXP_MAGIC 0x1.8p52   /* adding it rounds to an integer and leaves that integer in the low mantissa bits */

/* core: x already within [-746, 710].  SPLIT = 1: even/odd halves (two dependency chains, for lone evaluations);
 * SPLIT = 0: plain Horner (one instruction fewer; used where several evaluations already interleave).           */
PHF_HD double phf_exp_core_k(double xc, phf_ktab k, int split) {
  const double t = phf_fma(xc, PHF_LOG2E, PHF_EXP_MAGIC);
  const double kd = t - PHF_EXP_MAGIC;
  const int ki = (int)(int32_t)(uint32_t)phf_bits(t);
  double r = phf_fma(kd, -PHF_LN2_HI, xc);
  r = phf_fma(kd, -PHF_LN2_LO, r);
  const double r2 = r * r;
  double q;
  if (split) {
    double qe = k[8];                             /* even coefficients in r^2 */
    qe = PHF_FMA_KV(qe, r2, k[6]);
    qe = PHF_FMA_KV(qe, r2, k[4]);
    qe = PHF_FMA_KV(qe, r2, k[2]);
    qe = PHF_FMA_KV(qe, r2, k[0]);
    double qo = k[9];                             /* odd coefficients */
    qo = PHF_FMA_KV(qo, r2, k[7]);
    qo = PHF_FMA_KV(qo, r2, k[5]);
    qo = PHF_FMA_KV(qo, r2, k[3]);
    qo = PHF_FMA_KV(qo, r2, k[1]);
    q = phf_fma(qo, r, qe);
  } else {
    q = k[9];
    q = PHF_FMA_KV(q, r, k[8]);
    q = PHF_FMA_KV(q, r, k[7]);
    q = PHF_FMA_KV(q, r, k[6]);
    q = PHF_FMA_KV(q, r, k[5]);
    q = PHF_FMA_KV(q, r, k[4]);
    q = PHF_FMA_KV(q, r, k[3]);
    q = PHF_FMA_KV(q, r, k[2]);
    q = PHF_FMA_KV(q, r, k[1]);
    q = PHF_FMA_KV(q, r, k[0]);
  }
  const double p = phf_fma(r2, q, r) + 1.0;
  return __builtin_ldexp(p, ki);                  /* v_ldexp_f64: exact scaling, one rounding if the result is subnormal */
}

PHF_HD double phf_exp_fast_k(double x, phf_ktab k) {
  return phf_exp_core_k(__builtin_fmin(__builtin_fmax(x, -746.0), 710.0), k, 1);
}

/* x known to be <= 709 (callers that have already capped their argument): lower clamp only */
PHF_HD double phf_exp_capped_k(double x, phf_ktab k, int split) {
  return phf_exp_core_k(__builtin_fmax(x, -746.0), k, split);
}

PHF_HD double phf_exp_fast(double x) { PHF_KFETCH_V(k, phf_k_exp, PHF_K_EXP_N); return phf_exp_fast_k(x, k); }

PHF_HD double phf_exp(double x) {
  const double r = phf_exp_fast(x);
  return (x != x) ? x : r;
}

/* ------------------------------------------------------------------------------------------------ log
 * x = 2^k m, m in [sqrt(1/2), sqrt 2), f = m-1, s = f/(2+f), z = s^2,
 * log(1+f) = f - (f^2/2 - s (f^2/2 + z G(z))), G degree 6 as even + z*odd (approximation error 4.6e-18).
 * reduce/finish are split so the caller may obtain s = f/(2+f) from a shared (batched) reciprocal.            */
typedef struct { double f; double dk; } phf_logred;

PHF_HD phf_logred phf_log_reduce(double x) { /* exact for positive normal finite x; harmless bit-twiddling otherwise */
  uint64_t u = phf_bits(x);
  u += 0x3ff0000000000000ull - 0x3fe6a09e667f3bcdull;
  const int k = (int)(u >> 52) - 1023;
  u = (u & 0x000fffffffffffffull) + 0x3fe6a09e667f3bcdull;
  phf_logred lr;
  lr.f = phf_from_bits(u) - 1.0;
  lr.dk = (double)k;
  return lr;
}

PHF_KTABLE phf_k_log[7] = {    /* G(z), coefficient of z^i */
    0x1.5555555555558p-1, 0x1.99999999952ccp-2, 0x1.2492492df3ba9p-2, 0x1.c71c62e26208bp-3, 0x1.7462b58e46ebep-3,
    0x1.39fe42e9740a7p-3, 0x1.2b59b713616c9p-3};

PHF_HD double phf_log_finish_k(phf_logred lr, double s, phf_ktab k) {
  const double f = lr.f, dk = lr.dk;
  const double z = s * s;
  const double z2 = z * z;
  double ge = k[6];
  ge = PHF_FMA_KV(ge, z2, k[4]);
  ge = PHF_FMA_KV(ge, z2, k[2]);
  ge = PHF_FMA_KV(ge, z2, k[0]);
  double go = k[5];
  go = PHF_FMA_KV(go, z2, k[3]);
  go = PHF_FMA_KV(go, z2, k[1]);
  const double g = phf_fma(go, z, ge);
  const double hfsq = 0.5 * f * f;
  const double t = phf_fma(dk, PHF_LN2_LO, s * phf_fma(z, g, hfsq));
  return phf_fma(dk, PHF_LN2_HI, f - (hfsq - t));
}

PHF_HD double phf_log_finish(phf_logred lr, double s) { PHF_KFETCH_V(k, phf_k_log, 7); return phf_log_finish_k(lr, s, k); }

/* positive normal finite x only (no checks) */
PHF_HD double phf_log_core(double x) {
  const phf_logred lr = phf_log_reduce(x);
  return phf_log_finish(lr, phf_div(lr.f, 2.0 + lr.f));
}

/* kernels: x >= 2^-1022 exact; anything below (0, negatives, subnormals) gives -inf; branch-free.
 * (+inf and NaN are not handled: the kernels never produce them here.)                              */
PHF_HD double phf_log_fast(double x) {
  const double r = phf_log_core(x);
  return (x < PHF_DBL_MIN) ? -PHF_INF : r;
}

/* full IEEE behaviour (subnormals, 0, negatives, inf, NaN) */
PHF_HD double phf_log(double x) {
  const uint64_t u = phf_bits(x);
  if (u < 0x0010000000000000ull || (u >> 63)) { /* +0, +subnormal, or sign bit set */
    if (x == 0.0) return -PHF_INF;
    if (u >> 63) return (x != x) ? x : PHF_NAN;
    const phf_logred lr = phf_log_reduce(x * 0x1p54);
    phf_logred l2; l2.f = lr.f; l2.dk = lr.dk - 54.0;
    return phf_log_finish(l2, lr.f / (2.0 + lr.f));
  }
  if (u >= 0x7ff0000000000000ull) return x; /* +inf or NaN */
  return phf_log_core(x);
}

/* ------------------------------------------------------------------------------------------------ erfcx
 * erfcx(y) = exp(y^2) erfc(y) for y >= 0:  (1+2y) erfcx(y) = P(t), t = (y-4)/(y+4) in [-1,1], P degree 22
 * as even + t*odd (approximation error 3.1e-16).  den/finish are split: r = 1/((y+4)(1+2y)) serves both t and
 * the final scaling and may come from a batched reciprocal.                                                    */
PHF_HD double phf_erfcx_den(double y) { return (y + 4.0) * phf_fma(2.0, y, 1.0); }

PHF_KTABLE phf_k_erfcx[24] = { /* (1+2y) erfcx(y) in t = (y-4)/(y+4), coefficient of t^i (24th entry pads the burst) */
    0x1.3ba5916e9fd7fp+0, -0x1.1df1ad154a1c8p-3, 0x1.f7f5df66fd40dp-7, 0x1.16ecefcf9cb1ep-4, -0x1.9ddb23c3e6861p-4,
    0x1.7fee004ef1101p-4, -0x1.0fb06dfe8afa8p-4, 0x1.3079ede17a234p-5, -0x1.09623878c700ep-6, 0x1.49c676f414b52p-8,
    -0x1.8d4aa41628fedp-11, -0x1.a1e16f900a258p-13, 0x1.3be0e09412ec0p-13, -0x1.9928561ea5afcp-16,
    -0x1.789e79eb906c5p-17, 0x1.7dcf4dcc6199dp-18, 0x1.3ebb0291516c9p-22, -0x1.ae86b29807edbp-21,
    0x1.355884b1ca9fcp-24, 0x1.8f0920c7d5e28p-24, -0x1.1f8f10ba20f78p-26, -0x1.dff032d300316p-28,
    0x1.c2e324cb33784p-30, 0.0};

/* the coefficient operand of a Horner step: from an SGPR pair (table fetched through the scalar cache) or a VGPR pair */
#define PHF_FMA_KX(p, t, c, in_vgpr) ((in_vgpr) ? PHF_FMA_KV(p, t, c) : PHF_FMA_K(p, t, c))

PHF_HD double phf_erfcx_finish_kx(double y, double r, phf_ktab k, int in_vgpr) {
  const double a = y + 4.0, b = phf_fma(2.0, y, 1.0);
  const double t = ((y - 4.0) * b) * r;
  const double t2 = t * t;
  double pe = k[22], po = k[21];
  PHF_UNROLL
  for (int i = 20; i >= 0; i -= 2) {               /* even and odd chains alternate: neighbours are independent */
    pe = PHF_FMA_KX(pe, t2, k[i], in_vgpr);
    if (i >= 2) po = PHF_FMA_KX(po, t2, k[i - 1], in_vgpr);
  }
  const double p = phf_fma(po, t, pe);
  return (p * a) * r;
}

PHF_HD double phf_erfcx_finish_k(double y, double r, phf_ktab k) { return phf_erfcx_finish_kx(y, r, k, 0); }

/* two arguments at once: the four Horner chains advance in turn */
PHF_HD void phf_erfcx_finish_x2_kx(double y0, double r0, double y1, double r1, phf_ktab k, int in_vgpr, double* e0, double* e1) {
  const double a0 = y0 + 4.0, b0 = phf_fma(2.0, y0, 1.0), a1 = y1 + 4.0, b1 = phf_fma(2.0, y1, 1.0);
  const double t0 = ((y0 - 4.0) * b0) * r0, t1 = ((y1 - 4.0) * b1) * r1;
  const double s0 = t0 * t0, s1 = t1 * t1;
  double pe0 = k[22], pe1 = k[22], po0 = k[21], po1 = k[21];
  PHF_UNROLL
  for (int i = 20; i >= 0; i -= 2) {
    pe0 = PHF_FMA_KX(pe0, s0, k[i], in_vgpr);
    pe1 = PHF_FMA_KX(pe1, s1, k[i], in_vgpr);
    if (i >= 2) {
      po0 = PHF_FMA_KX(po0, s0, k[i - 1], in_vgpr);
      po1 = PHF_FMA_KX(po1, s1, k[i - 1], in_vgpr);
    }
  }
  *e0 = (phf_fma(po0, t0, pe0) * a0) * r0;
  *e1 = (phf_fma(po1, t1, pe1) * a1) * r1;
}

PHF_HD double phf_erfcx_finish(double y, double r) { PHF_KFETCH(k, phf_k_erfcx, 24); return phf_erfcx_finish_k(y, r, k); }

/* 0 <= y < ~1e150 (no checks) */
PHF_HD double phf_erfcx_core(double y) { return phf_erfcx_finish(y, phf_rcp(phf_erfcx_den(y))); }

/* any y >= 0 */
PHF_HD double phf_erfcx_nonneg(double y) {
  if (y > 1e100) return PHF_INV_SQRTPI / y;
  return phf_erfcx_core(y);
}

/* exp(-x^2/2) with the rounding error of x*x compensated (keeps the Gaussian tail to ~1 ulp). */
PHF_HD double phf_exp_mhalf_sq(double x) {
  const double hi = x * x;
  const double lo = phf_fma(x, x, -hi);
  const double e = phf_exp_fast(-0.5 * hi);
  return phf_fma(e, -0.5 * lo, e);
}

/* ------------------------------------------------------------------------------------------------ normal CDF
 * log Phi(x) for x <= 0 — the only case the censored likelihood produces, because predictions lie in
 * [0,100]:  log(erfcx(-x/sqrt2)/2) - x^2/2, no cancellation anywhere, no branches.                             */
PHF_HD double phf_log_ndtr_nonpos(double x) {
  const double e = phf_erfcx_core(-x * PHF_INV_SQRT2);
  return phf_fma(-0.5 * x, x, phf_log_core(0.5 * e));
}

/* two at once, sharing one division for the two erfcx and one for the two logs */
PHF_HD void phf_log_ndtr_nonpos_x2_kx(double x0, double x1, double* r0, double* r1, phf_ktab ke, int ke_in_vgpr, phf_ktab kl) {
  const double y0 = -x0 * PHF_INV_SQRT2, y1 = -x1 * PHF_INV_SQRT2;
  const double q0 = phf_erfcx_den(y0), q1 = phf_erfcx_den(y1);
  const double iq = phf_rcp(q0 * q1);
  double e0, e1;
  phf_erfcx_finish_x2_kx(y0, iq * q1, y1, iq * q0, ke, ke_in_vgpr, &e0, &e1);
  const phf_logred l0 = phf_log_reduce(0.5 * e0), l1 = phf_log_reduce(0.5 * e1);
  const double d0 = 2.0 + l0.f, d1 = 2.0 + l1.f;
  const double id = phf_rcp(d0 * d1);
  *r0 = phf_fma(-0.5 * x0, x0, phf_log_finish_k(l0, l0.f * (id * d1), kl));
  *r1 = phf_fma(-0.5 * x1, x1, phf_log_finish_k(l1, l1.f * (id * d0), kl));
}

PHF_HD void phf_log_ndtr_nonpos_x2_k(double x0, double x1, double* r0, double* r1, phf_ktab ke, phf_ktab kl) {
  phf_log_ndtr_nonpos_x2_kx(x0, x1, r0, r1, ke, 0, kl);
}

PHF_HD void phf_log_ndtr_nonpos_x2(double x0, double x1, double* r0, double* r1) {
  PHF_KFETCH(ke, phf_k_erfcx, 24);
  PHF_KFETCH_V(kl, phf_k_log, 7);
  phf_log_ndtr_nonpos_x2_k(x0, x1, r0, r1, ke, kl);
}

/* one, with the log table from the caller */
PHF_HD double phf_log_ndtr_nonpos_kx(double x, phf_ktab ke, int ke_in_vgpr, phf_ktab kl) {
  const double yv = -x * PHF_INV_SQRT2;
  const double e = phf_erfcx_finish_kx(yv, phf_rcp(phf_erfcx_den(yv)), ke, ke_in_vgpr);
  const phf_logred lr = phf_log_reduce(0.5 * e);
  return phf_fma(-0.5 * x, x, phf_log_finish_k(lr, phf_div(lr.f, 2.0 + lr.f), kl));
}

PHF_HD double phf_log_ndtr_nonpos_k(double x, phf_ktab ke, phf_ktab kl) { return phf_log_ndtr_nonpos_kx(x, ke, 0, kl); }

/* log Phi(x), any x.  x > 0: log(1 - q), q = erfcx(x/sqrt2) exp(-x^2/2)/2, with the log1p correction term. */
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

/* ------------------------------------------------------------------------------------------------ sin/cos
 * sin and cos of 2*pi*w/2^32 for a 32-bit integer w: exact quadrant reduction in the integer domain,
 * |x| <= pi/4 kernels of degree 13/14 (approximation error 2e-17), rotation by swap + sign-bit flips (no branches). */
PHF_KTABLE phf_k_sincos[12] = { /* sin: S(z) coefficients of z^0..z^5, then cos: C(z) of z^0..z^5, |x| <= pi/4 */
    -0x1.5555555555555p-3, 0x1.1111111110bb1p-7, -0x1.a01a019e8357dp-13, 0x1.71de37961e4c6p-19, -0x1.ae600a926c89ap-26,
    0x1.5e0af186af739p-33,
    0x1.5555555555555p-5, -0x1.6c16c16c16966p-10, 0x1.a01a019f4e867p-16, -0x1.27e4fa17a41b4p-22, 0x1.1eeb68b109173p-29,
    -0x1.907d7aebd5e3dp-37};

PHF_HD void phf_sincos_2pi_u32_k(uint32_t w, double* sn, double* cs, phf_ktab k) {
  const uint32_t q = ((w >> 29) + 1u) >> 1;                 /* nearest multiple of pi/2: 0..4 */
  const int32_t rem = (int32_t)(w - (q << 30));              /* [-2^29, 2^29); q = 4 wraps to the same value */
  const double x = (double)rem * PHF_2PI_2M32;
  const double z = x * x;
  double ps = k[5];
  ps = PHF_FMA_KV(ps, z, k[4]);
  ps = PHF_FMA_KV(ps, z, k[3]);
  ps = PHF_FMA_KV(ps, z, k[2]);
  ps = PHF_FMA_KV(ps, z, k[1]);
  ps = PHF_FMA_KV(ps, z, k[0]);
  double pc = k[11];
  pc = PHF_FMA_KV(pc, z, k[10]);
  pc = PHF_FMA_KV(pc, z, k[9]);
  pc = PHF_FMA_KV(pc, z, k[8]);
  pc = PHF_FMA_KV(pc, z, k[7]);
  pc = PHF_FMA_KV(pc, z, k[6]);
  const double s = phf_fma(x * z, ps, x);
  const double c = phf_fma(z * z, pc, phf_fma(-0.5, z, 1.0));
  /* quarter turns h = q mod 4: (sin, cos) = (s,c), (c,-s), (-s,-c), (-c,s): a swap and two sign flips (sign-bit xor) */
  const uint32_t h = q & 3u;
  const double a = (h & 1u) ? c : s;
  const double b = (h & 1u) ? s : c;
  *sn = phf_from_bits(phf_bits(a) ^ ((uint64_t)(h >> 1) << 63));
  *cs = phf_from_bits(phf_bits(b) ^ ((uint64_t)(((h + 1u) >> 1) & 1u) << 63));
}

PHF_HD void phf_sincos_2pi_u32(uint32_t w, double* sn, double* cs) { PHF_KFETCH_V(k, phf_k_sincos, 12); phf_sincos_2pi_u32_k(w, sn, cs, k); }

/* 53-bit uniform on [0,1) from two words — numpy's random_sample() construction
 * (the reference's npr.rand(), python/PyHillFit.py:834).                                       */
PHF_HD double phf_uniform53(uint32_t w1, uint32_t w2) {
  return ((double)(w1 >> 5) * 67108864.0 + (double)(w2 >> 6)) * 0x1p-53;
}

/* Box-Muller radius argument: u1 = (w+0.5)/2^32 in (0,1), so -2 log u1 is finite; |z| <= 6.66.
 * The proposal stays symmetric, which is all Metropolis needs.                                  */
PHF_HD double phf_unit_open32(uint32_t w) { return ((double)w + 0.5) * 0x1p-32; }

/* the same from the top 24 bits of a field: u1 = (v+0.5)/2^24, |z| <= 5.89 */
PHF_HD double phf_unit_open24(uint32_t v24) { return ((double)v24 + 0.5) * 0x1p-24; }

/* Box-Muller pair from two 32-bit words (own division; the samplers use phf_mh_draws in phf_model.h) */
PHF_HD void phf_box_muller(uint32_t w1, uint32_t w2, double* z0, double* z1) {
  const double rad = phf_sqrt_pos(-2.0 * phf_log_core(phf_unit_open32(w1)));
  double sn, cs;
  phf_sincos_2pi_u32(w2, &sn, &cs);
  *z0 = rad * cs;
  *z1 = rad * sn;
}

#endif /* PHF_MATH_H */
