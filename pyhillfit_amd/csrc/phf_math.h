/* phf_math.h — bit-reproducible fp64 elementary functions for the PyHillFit MH kernels.
 *
 * Everything here is built from IEEE-754 +, -, *, /, sqrt, fma, rint, min/max and integer bit moves only, in
 * a FIXED evaluation order, so the same source gives bit-identical results on the gfx950 device (hipcc,
 * -ffp-contract=off) and on the host (gcc, -ffp-contract=off -mfma).  That is what makes "same Philox
 * stream => same accept sequence, same chain, bit for bit" testable: ROCm's OCML exp/log/erfc and glibc's
 * differ in the last ulp and would not be.
 *
 * Shaped for CDNA4's fp64 VALU (measured with tools/microbench.hip on MI355X: v_fma_f64 2.25 ns per
 * wave-instruction per SIMD, v_mov_b64 2.1 ns, IEEE division ~26 ns, v_rcp/v_sqrt_f64 7 ns) — the kernels are bound by
 * vector-ALU issue, so what counts is the NUMBER of vector instructions:
 *   - exp and log reduce their argument through a small table in LDS (a gather is a ds_read, not a vector-ALU
 *     instruction): 11 and 12 fp64 operations instead of 17 and 27, no division in the logarithm;
 *   - polynomial steps are literal 3-operand v_fma_f64 with the coefficient in an SGPR pair fetched through the
 *     scalar cache or held in VGPRs (PHF_KFETCH / PHF_KFETCH_V below);
 *   - the long polynomials (erfcx, sin/cos) are split into even/odd halves (two independent dependency chains) so a
 *     lone wave on a SIMD is not latency-bound;
 *   - the *_fast / *_core entry points used inside the kernels are branch-free: range problems are handled by
 *     clamping and IEEE overflow/underflow, never by divergent control flow: every lane runs the same instruction
 *     stream whatever its argument (a table lookup differs in the address only);
 *   - the remaining divisions are exposed (den / finish pairs of erfcx, the Hill curve's 1/(1+x)) so callers can share
 *     one IEEE division among several evaluations (batched reciprocal, see phf_model.h).
 *
 * Tables and coefficients: tools/gen_math_coeffs.py (mpmath, 60 digits; Chebyshev-interpolant = near-minimax fits);
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
/* VGPR-resident: materialised once (kernels do it before the MH loop) and kept in vector registers (exp, log: 8
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
/* a wave-uniform "does any lane ...": rare, expensive alternatives run under a scalar branch (and every lane then selects by its
 * OWN condition, so a chain's value never depends on the other chains of its wavefront); on the host: the one value itself */
#if defined(__HIP_DEVICE_COMPILE__)
#define PHF_ANY_LANE(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#else
#define PHF_ANY_LANE(c) (c)
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
/* g = sqrt(x) AND 1/g for x > 0 — a Cholesky pivot and its reciprocal (phf_single_level.hip: chol_packed) — from ONE hardware
 * estimate: the square-root iteration above carries h ~ 1/(2 sqrt x) along, so 2 h is already 1/g to a few ulp, and what
 * phf_rcp does after ITS estimate and first Newton step — one more Newton step, then the correcting step — finishes it.  Same two
 * correctly rounded results as phf_sqrt_pos(x) and phf_rcp(g) (checked bit for bit against sqrt and 1.0 / sqrt on 1.2 M arguments,
 * tests/test_gpu_parity.py), 15 instructions and one quarter-rate one instead of 17 and two (round 4: C3 -1.x %). */
PHF_HD double phf_sqrt_rcp_pos(double x, double* inv) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);                               /* sqrt(x), correctly rounded (as phf_sqrt_pos) */
  double yi = h + h;                                        /* 1/g to a few ulp (h was refined once: relative error ~2^-51) */
  const double e = __builtin_fma(-g, yi, 1.0);
  yi = __builtin_fma(yi, e, yi);                            /* one Newton step: as phf_rcp_refined_ leaves its y (~0.5 ulp) */
  const double c = __builtin_fma(-g, yi, 1.0);
  *inv = __builtin_fma(c, yi, yi);                          /* 1.0 / g, correctly rounded (as phf_rcp) */
  return g;
}
#else
PHF_HD double phf_rcp(double d) { return 1.0 / d; }
PHF_HD double phf_div(double n, double d) { return n / d; }
PHF_HD double phf_sqrt_pos(double x) { return __builtin_sqrt(x); }
PHF_HD double phf_sqrt_rcp_pos(double x, double* inv) { const double g = __builtin_sqrt(x); *inv = 1.0 / g; return g; }
#endif
/* sqrt(x) for x >= 0 with sqrt(0) = 0 (pivots of a degenerate factor); anything else non-positive also gives 0 */
PHF_HD double phf_sqrt_nonneg(double x) {
  const double r = phf_sqrt_pos(x);
  return (x > 0.0) ? r : 0.0;
}
PHF_HD double phf_pow2i(int k) { return phf_from_bits((uint64_t)(k + 1023) << 52); } /* -1022 <= k <= 1023 */

/* ------------------------------------------------------------------------------------------------ lookup tables
 * exp and log reduce their argument with a small table: 2^(j/64) (64 doubles) and {1/c_j, log c_j} on a grid of 129 points c_j of
 * [sqrt(1/2), sqrt 2] (tools/gen_math_coeffs.py).  On the device the tables live in LDS — 2 576 bytes per workgroup, filled by
 * PHF_MATH_TABLES_TO_LDS(), which EVERY kernel that evaluates anything of this header calls first, with all its threads — and a
 * lookup is one ds_read (b64 / b128) at a per-lane address: a gather costs no vector-ALU slot, which is what these kernels are
 * short of (measured, C3: polynomial-only exp and log -> tables: 151.4 -> 138.4 ms per 8 000 iterations).
 * On the host the tables are the arrays themselves: same values, same operations, same results.                              */
typedef struct { double invc, logc; } phf_logtab;
PHF_KTABLE phf_t_exp2[64] = {   /* 2^(j/64), correctly rounded */
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
#define PHF_LOG_TAB_N 129
#define PHF_LOG_TAB_BASE 0x1ff35   /* (bits(sqrt(1/2) rounded up) + 2^44) >> 45: entry j belongs to the double with bits (BASE + j) << 45 */
static const phf_logtab phf_t_log[PHF_LOG_TAB_N] = {   /* {1/c rounded, -log of that rounded reciprocal}; entry 53 is c = 1: {1, 0} */
    {0x1.6a13cd1537290p+0, -0x1.630030b3aac48p-2},   /* c = 0.70703125 */
    {0x1.6816816816817p+0, -0x1.5d5bddf595f31p-2},
    {0x1.661ec6a5122f9p+0, -0x1.57bf753c8d1fbp-2},
    {0x1.642c8590b2164p+0, -0x1.522ae0738a3d7p-2},
    {0x1.623fa77016240p+0, -0x1.4c9e09e172c3dp-2},
    {0x1.6058160581606p+0, -0x1.4718dc271c41cp-2},
    {0x1.5e75bb8d015e7p+0, -0x1.419b423d5e8c6p-2},
    {0x1.5c9882b931057p+0, -0x1.3c25277333183p-2},
    {0x1.5ac056b015ac0p+0, -0x1.36b6776be1116p-2},
    {0x1.58ed2308158edp+0, -0x1.314f1e1d35ce3p-2},
    {0x1.571ed3c506b3ap+0, -0x1.2bef07cdc9355p-2},
    {0x1.5555555555555p+0, -0x1.269621134db91p-2},
    {0x1.5390948f40febp+0, -0x1.214456d0eb8d5p-2},
    {0x1.51d07eae2f815p+0, -0x1.1bf99635a6b95p-2},
    {0x1.5015015015015p+0, -0x1.16b5ccbacfb73p-2},
    {0x1.4e5e0a72f0539p+0, -0x1.1178e8227e47ap-2},
    {0x1.4cab88725af6ep+0, -0x1.0c42d676162e2p-2},   /* c = 0.76953125 */
    {0x1.4afd6a052bf5bp+0, -0x1.07138604d5864p-2},
    {0x1.49539e3b2d067p+0, -0x1.01eae5626c691p-2},
    {0x1.47ae147ae147bp+0, -0x1.f991c6cb3b37ap-3},
    {0x1.460cbc7f5cf9ap+0, -0x1.ef5ade4dcffe5p-3},
    {0x1.446f86562d9fbp+0, -0x1.e530effe71013p-3},
    {0x1.42d6625d51f87p+0, -0x1.db13db0d48941p-3},
    {0x1.4141414141414p+0, -0x1.d1037f2655e7bp-3},
    {0x1.3fb013fb013fbp+0, -0x1.c6ffbc6f00f71p-3},
    {0x1.3e22cbce4a902p+0, -0x1.bd087383bd8aap-3},
    {0x1.3c995a47babe7p+0, -0x1.b31d8575bce3bp-3},
    {0x1.3b13b13b13b14p+0, -0x1.a93ed3c8ad9e5p-3},
    {0x1.3991c2c187f63p+0, -0x1.9f6c407089663p-3},
    {0x1.3813813813814p+0, -0x1.95a5adcf70182p-3},
    {0x1.3698df3de0748p+0, -0x1.8beafeb38fe8fp-3},
    {0x1.3521cfb2b78c1p+0, -0x1.823c16551a3c0p-3},
    {0x1.33ae45b57bcb2p+0, -0x1.7898d85444c74p-3},   /* c = 0.83203125 */
    {0x1.323e34a2b10bfp+0, -0x1.6f0128b756ab9p-3},
    {0x1.30d190130d190p+0, -0x1.6574ebe8c1339p-3},
    {0x1.2f684bda12f68p+0, -0x1.5bf406b543db0p-3},
    {0x1.2e025c04b8097p+0, -0x1.527e5e4a1b58dp-3},
    {0x1.2c9fb4d812ca0p+0, -0x1.4913d8333b563p-3},
    {0x1.2b404ad012b40p+0, -0x1.3fb45a59928cap-3},
    {0x1.29e4129e4129ep+0, -0x1.365fcb0159014p-3},
    {0x1.288b01288b013p+0, -0x1.2d1610c86813dp-3},
    {0x1.27350b8812735p+0, -0x1.23d712a49c201p-3},
    {0x1.25e22708092f1p+0, -0x1.1aa2b7e23f729p-3},
    {0x1.2492492492492p+0, -0x1.1178e8227e47ap-3},
    {0x1.23456789abcdfp+0, -0x1.08598b59e3a07p-3},
    {0x1.21fb78121fb78p+0, -0x1.fe89139dbd565p-4},
    {0x1.20b470c67c0d9p+0, -0x1.ec739830a1126p-4},
    {0x1.1f7047dc11f70p+0, -0x1.da7276384469ep-4},
    {0x1.1e2ef3b3fb874p+0, -0x1.c885801bc4b20p-4},   /* c = 0.89453125 */
    {0x1.1cf06ada2811dp+0, -0x1.b6ac88dad5b1dp-4},
    {0x1.1bb4a4046ed29p+0, -0x1.a4e7640b1bc38p-4},
    {0x1.1a7b9611a7b96p+0, -0x1.9335e5d594988p-4},
    {0x1.19453808ca29cp+0, -0x1.8197e2f40e3f0p-4},
    {0x1.1811811811812p+0, -0x1.700d30aeac0e8p-4},
    {0x1.16e0689427379p+0, -0x1.5e95a4d9791cdp-4},
    {0x1.15b1e5f75270dp+0, -0x1.4d3115d207eacp-4},
    {0x1.1485f0e0acd3bp+0, -0x1.3bdf5a7d1ee5ep-4},
    {0x1.135c81135c811p+0, -0x1.2aa04a44717a1p-4},
    {0x1.12358e75d3033p+0, -0x1.1973bd1465561p-4},
    {0x1.1111111111111p+0, -0x1.08598b59e3a06p-4},
    {0x1.0fef010fef011p+0, -0x1.eea31c006b87cp-5},
    {0x1.0ecf56be69c90p+0, -0x1.ccb73cdddb2d0p-5},
    {0x1.0db20a88f4696p+0, -0x1.aaef2d0fb1108p-5},
    {0x1.0c9714fbcda3bp+0, -0x1.894aa149fb34bp-5},
    {0x1.0b7e6ec259dc8p+0, -0x1.67c94f2d4bb65p-5},   /* c = 0.95703125 */
    {0x1.0a6810a6810a7p+0, -0x1.466aed42de3f9p-5},
    {0x1.0953f39010954p+0, -0x1.252f32f8d1840p-5},
    {0x1.0842108421084p+0, -0x1.0415d89e74440p-5},
    {0x1.073260a47f7c6p+0, -0x1.c63d2ec14aad7p-6},
    {0x1.0624dd2f1a9fcp+0, -0x1.8492528c8cac5p-6},
    {0x1.05197f7d73404p+0, -0x1.432a925980cbcp-6},
    {0x1.0410410410410p+0, -0x1.0205658935837p-6},
    {0x1.03091b51f5e1ap+0, -0x1.82448a388a283p-7},
    {0x1.0204081020408p+0, -0x1.010157588de69p-7},
    {0x1.0101010101010p+0, -0x1.0080559588b25p-8},
    {0x1.0000000000000p+0, 0x0.0p+0},   /* c = 1 */
    {0x1.fc07f01fc07f0p-1, 0x1.fe02a6b106799p-8},
    {0x1.f81f81f81f820p-1, 0x1.fc0a8b0fc03c4p-7},
    {0x1.f44659e4a4271p-1, 0x1.7b91b07d5b126p-6},
    {0x1.f07c1f07c1f08p-1, 0x1.f829b0e7832f8p-6},
    {0x1.ecc07b301ecc0p-1, 0x1.39e87b9febd68p-5},   /* c = 1.0390625 */
    {0x1.e9131abf0b767p-1, 0x1.77458f632dcffp-5},
    {0x1.e573ac901e574p-1, 0x1.b42dd711971b9p-5},
    {0x1.e1e1e1e1e1e1ep-1, 0x1.f0a30c01162a8p-5},
    {0x1.de5d6e3f8868ap-1, 0x1.16536eea37ae3p-4},
    {0x1.dae6076b981dbp-1, 0x1.341d7961bd1d0p-4},
    {0x1.d77b654b82c34p-1, 0x1.51b073f06183cp-4},
    {0x1.d41d41d41d41dp-1, 0x1.6f0d28ae56b4ep-4},
    {0x1.d0cb58f6ec074p-1, 0x1.8c345d6319b23p-4},
    {0x1.cd85689039b0bp-1, 0x1.a926d3a4ad562p-4},
    {0x1.ca4b3055ee191p-1, 0x1.c5e548f5bc743p-4},
    {0x1.c71c71c71c71cp-1, 0x1.e27076e2af2eap-4},
    {0x1.c3f8f01c3f8f0p-1, 0x1.fec9131dbeabcp-4},
    {0x1.c0e070381c0e0p-1, 0x1.0d77e7cd08e5bp-3},
    {0x1.bdd2b899406f7p-1, 0x1.1b72ad52f67a2p-3},
    {0x1.bacf914c1bad0p-1, 0x1.29552f81ff521p-3},
    {0x1.b7d6c3dda338bp-1, 0x1.371fc201e8f75p-3},   /* c = 1.1640625 */
    {0x1.b4e81b4e81b4fp-1, 0x1.44d2b6ccb7d1cp-3},
    {0x1.b2036406c80d9p-1, 0x1.526e5e3a1b438p-3},
    {0x1.af286bca1af28p-1, 0x1.5ff3070a793d6p-3},
    {0x1.ac5701ac5701bp-1, 0x1.6d60fe719d21bp-3},
    {0x1.a98ef606a63bep-1, 0x1.7ab890210d907p-3},
    {0x1.a6d01a6d01a6dp-1, 0x1.87fa06520c911p-3},
    {0x1.a41a41a41a41ap-1, 0x1.9525a9cf456b6p-3},
    {0x1.a16d3f97a4b02p-1, 0x1.a23bc1fe2b561p-3},
    {0x1.9ec8e951033d9p-1, 0x1.af3c94e80bff3p-3},
    {0x1.9c2d14ee4a102p-1, 0x1.bc286742d8cd4p-3},
    {0x1.999999999999ap-1, 0x1.c8ff7c79a9a20p-3},
    {0x1.970e4f80cb872p-1, 0x1.d5c216b4fbb94p-3},
    {0x1.948b0fcd6e9e0p-1, 0x1.e27076e2af2e8p-3},
    {0x1.920fb49d0e229p-1, 0x1.ef0adcbdc5935p-3},
    {0x1.8f9c18f9c18fap-1, 0x1.fb9186d5e3e29p-3},
    {0x1.8d3018d3018d3p-1, 0x1.0402594b4d041p-2},   /* c = 1.2890625 */
    {0x1.8acb90f6bf3aap-1, 0x1.0a324e27390e2p-2},
    {0x1.886e5f0abb04ap-1, 0x1.1058bf9ae4ad4p-2},
    {0x1.8618618618618p-1, 0x1.1675cababa60fp-2},
    {0x1.83c977ab2beddp-1, 0x1.1c898c16999fbp-2},
    {0x1.8181818181818p-1, 0x1.22941fbcf7966p-2},
    {0x1.7f405fd017f40p-1, 0x1.2895a13de86a4p-2},
    {0x1.7d05f417d05f4p-1, 0x1.2e8e2bae11d31p-2},
    {0x1.7ad2208e0ecc3p-1, 0x1.347dd9a987d56p-2},
    {0x1.78a4c8178a4c8p-1, 0x1.3a64c556945eap-2},
    {0x1.767dce434a9b1p-1, 0x1.404308686a7e4p-2},
    {0x1.745d1745d1746p-1, 0x1.4618bc21c5ec2p-2},
    {0x1.724287f46debcp-1, 0x1.4be5f957778a1p-2},
    {0x1.702e05c0b8170p-1, 0x1.51aad872df82ep-2},
    {0x1.6e1f76b4337c7p-1, 0x1.5767717455a6cp-2},
    {0x1.6c16c16c16c17p-1, 0x1.5d1bdbf5809cap-2},
    {0x1.6a13cd1537290p-1, 0x1.62c82f2b9c796p-2},   /* c = 1.4140625 */
};
#if defined(__HIP_DEVICE_COMPILE__)
static __shared__ double phf_lds_exp2[64];
static __shared__ __attribute__((aligned(16))) phf_logtab phf_lds_log[PHF_LOG_TAB_N];
#define PHF_T_EXP2(j) phf_lds_exp2[j]
#define PHF_T_LOG(j) phf_lds_log[j]
#define PHF_MATH_TABLES_TO_LDS()                                                                                          \
  do {                                                                                                                    \
    for (int phf_i_ = threadIdx.x; phf_i_ < 64; phf_i_ += blockDim.x) phf_lds_exp2[phf_i_] = phf_t_exp2[phf_i_];          \
    for (int phf_i_ = threadIdx.x; phf_i_ < PHF_LOG_TAB_N; phf_i_ += blockDim.x) phf_lds_log[phf_i_] = phf_t_log[phf_i_]; \
    __syncthreads();                                                                                                      \
  } while (0)
#else
#define PHF_T_EXP2(j) phf_t_exp2[j]
#define PHF_T_LOG(j) phf_t_log[j]
#define PHF_MATH_TABLES_TO_LDS() do { } while (0)
#endif
#define PHF_MATH_LDS_BYTES (64 * 8 + PHF_LOG_TAB_N * 16)

/* ------------------------------------------------------------------------------------------------ exp
 * n = nearest integer to 64 x / ln2 (magic-number add) = 64 k + j, r = x - n ln2/64 (two fma, |r| <= ln2/128),
 * exp(x) = 2^k T[j] (1 + p(r)), p(r) = r + r^2 (1/2 + r q(r)), q degree 2 (approximation error 8.9e-18 relative), T[j] = 2^(j/64)
 * from the table, scaled by 2^k with ldexp (subnormal results round once); ~1 ulp.  The argument is clamped to [-746, 710]: the
 * scaling then overflows to +inf / underflows to 0 by itself, no branches.  phf_exp_fast(NaN) = 0 (min/max drop the NaN); phf_exp keeps NaN. */
/* [3] is not a coefficient: the constant 100 of the percent scale, riding along in the table every target holds in registers — a
 * double that is not an inline constant costs two v_mov_b32 at EVERY use otherwise (hipcc rematerialises it: 10 per single-level
 * iteration, 24 per hierarchical one) */
#define PHF_K_EXP_N 4
#define PHF_K100(k_exp) ((k_exp)[3])
PHF_KTABLE phf_k_exp[PHF_K_EXP_N] = {   /* (expm1(r) - r - r^2/2)/r^3, coefficient of r^i */
    0x1.5555555555555p-3, 0x1.555565c3ff8a9p-5, 0x1.11111a74dffd2p-7,
    100.0};

#define PHF_EXP_MAGIC 0x1.8p52        /* adding it rounds to an integer and leaves that integer in the low mantissa bits */
#define PHF_64_LOG2E 0x1.71547652b82fep+6
#define PHF_LN2_64_HI 0x1.62e42fee00000p-7   /* PHF_LN2_HI / 64: 21 trailing zero bits, n * this is exact for |n| < 2^21 */
#define PHF_LN2_64_LO 0x1.a39ef35793c76p-39  /* PHF_LN2_LO / 64 */

/* core: x already within [-746, 710] */
PHF_HD double phf_exp_core_k(double xc, phf_ktab k) {
  const double t = phf_fma(xc, PHF_64_LOG2E, PHF_EXP_MAGIC);
  const double nd = t - PHF_EXP_MAGIC;
  const int n = (int)(int32_t)(uint32_t)phf_bits(t);
  const double tj = PHF_T_EXP2(n & 63);
  double r = phf_fma(nd, -PHF_LN2_64_HI, xc);
  r = phf_fma(nd, -PHF_LN2_64_LO, r);
  const double r2 = r * r;
  double q = PHF_FMA_KV(k[2], r, k[1]);
  q = PHF_FMA_KV(q, r, k[0]);
  q = phf_fma(q, r, 0.5);
  const double p = phf_fma(r2, q, r);
  return __builtin_ldexp(phf_fma(tj, p, tj), n >> 6);   /* v_ldexp_f64: exact scaling, one rounding if the result is subnormal */
}

PHF_HD double phf_exp_fast_k(double x, phf_ktab k) {
  return phf_exp_core_k(__builtin_fmin(__builtin_fmax(x, -746.0), 710.0), k);
}

/* x known to be <= 709 (callers that have already capped their argument): lower clamp only */
PHF_HD double phf_exp_capped_k(double x, phf_ktab k) {
  return phf_exp_core_k(__builtin_fmax(x, -746.0), k);
}

PHF_HD double phf_exp_fast(double x) { PHF_KFETCH_V(k, phf_k_exp, PHF_K_EXP_N); return phf_exp_fast_k(x, k); }

PHF_HD double phf_exp(double x) {
  const double r = phf_exp_fast(x);
  return (x != x) ? x : r;
}

/* ------------------------------------------------------------------------------------------------ log
 * x = 2^k m, m in [sqrt(1/2), sqrt 2); c_j = the grid point nearest to m (the double whose bits are (BASE + j) << 45: 128 steps per
 * binade, c = 1 among them); r = m (1/c_j) - 1 by ONE fma (|r| <= 2^-8; the table's reciprocal is rounded, its logarithm is that of
 * the rounded value, so nothing is lost); log x = k ln2 + log c_j + (r + r^2 q(r)), q degree 4 with q(0) = -1/2 exactly
 * (approximation error 3.2e-17 of log1p(r)).  No division.  <= 2 ulp; near x = 1 the table entry is {1, 0} and the result is the
 * polynomial alone, so log(1 + tiny) keeps its relative accuracy.                                                              */
#define PHF_K_LOG_N 4
PHF_KTABLE phf_k_log[PHF_K_LOG_N] = {   /* (log1p(r) - r)/r^2 + 1/2, coefficients of r^1..r^4 */
    0x1.55555555276f7p-2, -0x1.ffffffffafadap-3, 0x1.999b080ce97c7p-3, -0x1.555695fa425fap-3};

/* positive normal finite x only (no checks; harmless bit-twiddling on anything else) */
PHF_HD double phf_log_pos_k(double x, phf_ktab k) {
  uint64_t u = phf_bits(x);
  u += 0x3ff0000000000000ull - 0x3fe6a09e667f3bcdull;
  const int e = (int)(u >> 52) - 1023;
  u = (u & 0x000fffffffffffffull) + 0x3fe6a09e667f3bcdull;                     /* bits of m */
  /* nearest grid point: ((bits + 2^44) >> 45) - BASE, with the subtraction folded into the addition in front of the shift (BASE << 13 has
   * no bits below the shift: the same integer) — the index then is a small non-negative number whose table address needs no further
   * add: 3 vector-ALU instructions per logarithm instead of 4 (round 5, tools/isa_itemise.py) */
  const int j = (int)(((uint32_t)(u >> 32) + (0x1000u - ((uint32_t)PHF_LOG_TAB_BASE << 13))) >> 13);
  const phf_logtab c = PHF_T_LOG(j);
  const double r = phf_fma(phf_from_bits(u), c.invc, -1.0);
  const double dk = (double)e;
  const double r2 = r * r;
  double q = PHF_FMA_KV(k[3], r, k[2]);
  q = PHF_FMA_KV(q, r, k[1]);
  q = PHF_FMA_KV(q, r, k[0]);
  q = phf_fma(q, r, -0.5);
  const double hi = phf_fma(dk, PHF_LN2_HI, c.logc);
  const double lo = phf_fma(dk, PHF_LN2_LO, r2 * q);
  return hi + (r + lo);
}

PHF_HD double phf_log_core(double x) { PHF_KFETCH_V(k, phf_k_log, PHF_K_LOG_N); return phf_log_pos_k(x, k); }

/* kernels: x >= 2^-1022 exact; anything below (0, negatives, subnormals) gives -inf; branch-free.
 * (+inf and NaN are not handled: the kernels never produce them here.)                              */
PHF_HD double phf_log_fast_k(double x, phf_ktab k) {
  const double r = phf_log_pos_k(x, k);
  return (x < PHF_DBL_MIN) ? -PHF_INF : r;
}

PHF_HD double phf_log_fast(double x) { PHF_KFETCH_V(k, phf_k_log, PHF_K_LOG_N); return phf_log_fast_k(x, k); }

/* full IEEE behaviour (subnormals, 0, negatives, inf, NaN) */
PHF_HD double phf_log(double x) {
  const uint64_t u = phf_bits(x);
  if (u < 0x0010000000000000ull || (u >> 63)) { /* +0, +subnormal, or sign bit set */
    if (x == 0.0) return -PHF_INF;
    if (u >> 63) return (x != x) ? x : PHF_NAN;
    return phf_log_core(x * 0x1p54) - 54.0 * (PHF_LN2_HI + PHF_LN2_LO);
  }
  if (u >= 0x7ff0000000000000ull) return x; /* +inf or NaN */
  return phf_log_core(x);
}

/* ------------------------------------------------------------------------------------------------ erfcx
 * erfcx(y) = exp(y^2) erfc(y) for y >= 0:  (1+2y) erfcx(y) = P(t), t = (y-4)/(y+4) in [-1,1], P degree 22
 * as even + t*odd (approximation error 3.1e-16).  den/finish are split: r = 1/((y+4)(1+2y)) serves both t and
 * the final scaling and may come from a batched reciprocal.
 * Since round 3 the MH loops no longer call this family: the single-level censored likelihood takes log Phi from
 * phf_log_ndtr_tab and the hierarchical truncation masses take erfc from phf_erfc_tab (both below).  It stays as the full-range,
 * full-relative-accuracy form behind phf_log_ndtr / phf_ndtr (any argument) and as what the tables are tested against.   */
PHF_HD double phf_erfcx_den(double y) { return (y + 4.0) * phf_fma(2.0, y, 1.0); }

PHF_KTABLE phf_k_erfcx[24] = { /* (1+2y) erfcx(y) in t = (y-4)/(y+4), coefficient of t^i (24th entry pads the burst) */
    0x1.3ba5916e9fd7fp+0, -0x1.1df1ad154a1c8p-3, 0x1.f7f5df66fd40dp-7, 0x1.16ecefcf9cb1ep-4, -0x1.9ddb23c3e6861p-4,
    0x1.7fee004ef1101p-4, -0x1.0fb06dfe8afa8p-4, 0x1.3079ede17a234p-5, -0x1.09623878c700ep-6, 0x1.49c676f414b52p-8,
    -0x1.8d4aa41628fedp-11, -0x1.a1e16f900a258p-13, 0x1.3be0e09412ec0p-13, -0x1.9928561ea5afcp-16,
    -0x1.789e79eb906c5p-17, 0x1.7dcf4dcc6199dp-18, 0x1.3ebb0291516c9p-22, -0x1.ae86b29807edbp-21,
    0x1.355884b1ca9fcp-24, 0x1.8f0920c7d5e28p-24, -0x1.1f8f10ba20f78p-26, -0x1.dff032d300316p-28,
    0x1.c2e324cb33784p-30, 0.0};

PHF_HD double phf_erfcx_finish_k(double y, double r, phf_ktab k) {
  const double a = y + 4.0, b = phf_fma(2.0, y, 1.0);
  const double t = ((y - 4.0) * b) * r;
  const double t2 = t * t;
  double pe = k[22], po = k[21];
  PHF_UNROLL
  for (int i = 20; i >= 0; i -= 2) {               /* even and odd chains alternate: neighbours are independent */
    pe = PHF_FMA_K(pe, t2, k[i]);
    if (i >= 2) po = PHF_FMA_K(po, t2, k[i - 1]);
  }
  const double p = phf_fma(po, t, pe);
  return (p * a) * r;
}

/* two arguments at once: the four Horner chains advance in turn */
PHF_HD void phf_erfcx_finish_x2_k(double y0, double r0, double y1, double r1, phf_ktab k, double* e0, double* e1) {
  const double a0 = y0 + 4.0, b0 = phf_fma(2.0, y0, 1.0), a1 = y1 + 4.0, b1 = phf_fma(2.0, y1, 1.0);
  const double t0 = ((y0 - 4.0) * b0) * r0, t1 = ((y1 - 4.0) * b1) * r1;
  const double s0 = t0 * t0, s1 = t1 * t1;
  double pe0 = k[22], pe1 = k[22], po0 = k[21], po1 = k[21];
  PHF_UNROLL
  for (int i = 20; i >= 0; i -= 2) {
    pe0 = PHF_FMA_K(pe0, s0, k[i]);
    pe1 = PHF_FMA_K(pe1, s1, k[i]);
    if (i >= 2) {
      po0 = PHF_FMA_K(po0, s0, k[i - 1]);
      po1 = PHF_FMA_K(po1, s1, k[i - 1]);
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

/* two at once, sharing one division for the two erfcx */
PHF_HD void phf_log_ndtr_nonpos_x2_k(double x0, double x1, double* r0, double* r1, phf_ktab ke, phf_ktab kl) {
  const double y0 = -x0 * PHF_INV_SQRT2, y1 = -x1 * PHF_INV_SQRT2;
  const double q0 = phf_erfcx_den(y0), q1 = phf_erfcx_den(y1);
  const double iq = phf_rcp(q0 * q1);
  double e0, e1;
  phf_erfcx_finish_x2_k(y0, iq * q1, y1, iq * q0, ke, &e0, &e1);
  *r0 = phf_fma(-0.5 * x0, x0, phf_log_pos_k(0.5 * e0, kl));
  *r1 = phf_fma(-0.5 * x1, x1, phf_log_pos_k(0.5 * e1, kl));
}

PHF_HD void phf_log_ndtr_nonpos_x2(double x0, double x1, double* r0, double* r1) {
  PHF_KFETCH(ke, phf_k_erfcx, 24);
  PHF_KFETCH_V(kl, phf_k_log, PHF_K_LOG_N);
  phf_log_ndtr_nonpos_x2_k(x0, x1, r0, r1, ke, kl);
}

/* one, with the log table from the caller */
PHF_HD double phf_log_ndtr_nonpos_k(double x, phf_ktab ke, phf_ktab kl) {
  const double yv = -x * PHF_INV_SQRT2;
  const double e = phf_erfcx_finish_k(yv, phf_rcp(phf_erfcx_den(yv)), ke);
  return phf_fma(-0.5 * x, x, phf_log_pos_k(0.5 * e, kl));
}

/* erfc(y), y >= 0, to an ABSOLUTE accuracy of 3.6e-17 through a table — the hierarchical target's form: its truncation masses
 * Phi(b) - Phi(a) = 1 - (erfc(ya) + erfc(yb))/2 need the tails to half an ulp of 1, not to a relative accuracy.  Centres
 * c_j = j/4 (the nearest to y, magic-number rounding), s = y - c_j in [-1/8, 1/8], degree 11 (tools/gen_math_coeffs.py erfc);
 * erfc(6) = 2.2e-17 is below half an ulp of the 1 it is subtracted from: from PHF_ERFC_CUT on the tail is DEFINED as zero.
 * 15 fp64 operations and six 16-byte LDS reads where erfcx x exp(-y^2) took ~50, no division, no branch.  The table (2.4 KB)
 * goes to LDS with PHF_ERFC_TABLE_TO_LDS().  (The index is clamped: any argument is safe to evaluate.) */
#define PHF_ERFC_TAB_N 25
#define PHF_ERFC_CUT 6.0
typedef struct { double c[12]; } phf_erfctab;
static const phf_erfctab phf_t_erfc[PHF_ERFC_TAB_N] = {
    {{0x1.0000000000000p+0, -0x1.20dd750429b6dp+0, -0x1.1011ffb71c7eep-188, 0x1.812746b0379a0p-2, 0x1.f4572a9871ca6p-180, -0x1.ce2f21a00f2fcp-4, -0x1.45a0000000000p-172, 0x1.b82ce2367f627p-6, 0x1.6155555555555p-166, -0x1.565a247787ddap-8, -0x1.1155555555555p-161, 0x1.bd39fdb90fa3ap-11}},
    {{0x1.728558ee694fcp-1, -0x1.0f5d1602f7e41p+0, 0x1.0f5d1602f7d94p-2, 0x1.3c974458cbde1p-2, -0x1.040e8a6d43d1dp-3, -0x1.47e5cfee31131p-4, 0x1.4c0b24317213ep-5, 0x1.08d9468a5e011p-6, -0x1.3db4f8210378dp-7, -0x1.55613142c98afp-9, 0x1.e2631bbd80df3p-10, 0x1.64bb2a48c9969p-12}},
    {{0x1.eb02147ce245cp-2, -0x1.c1efca49a5011p-1, 0x1.c1efca49a4f9ep-2, 0x1.2bf531866e06fp-3, -0x1.76f27de7df78fp-3, -0x1.dfeeb5a62a146p-8, 0x1.99f13a73477c9p-5, -0x1.623c5f112004ep-8, -0x1.493bedf1a6959p-7, 0x1.1c13edd64a268p-9, 0x1.9907484199c5fp-10, -0x1.02aaeb6b9d521p-11}},
    {{0x1.27c6d14c5e341p-2, -0x1.492e42d78d2c5p-1, 0x1.edc5644353c65p-2, -0x1.b6e8591f66b94p-6, -0x1.349b5eaa2af7dp-3, 0x1.b42a18908d619p-5, 0x1.b8477a27cd97dp-6, -0x1.2e0afa814c26fp-6, -0x1.2db61c7e31d1cp-9, 0x1.040f51c2dcaa8p-8, -0x1.7bc355700ad98p-13, -0x1.454a7084b2f36p-11}},
    {{0x1.4226162fbddd5p-3, -0x1.a911f096fbc26p-2, 0x1.a911f096fbc9bp-2, -0x1.1b614b0f5282fp-3, -0x1.1b614b0fa8153p-4, 0x1.1b614b0f5a72fp-4, -0x1.2e459fb0c45c4p-8, -0x1.f096fdb402567p-7, 0x1.390e85fe8a0b8p-8, 0x1.ee31df8a34c33p-10, -0x1.3f03a0b1dcd9ap-10, -0x1.6a5cad6230683p-14}},
    {{0x1.3bcd133aa0ffcp-4, -0x1.e4652fadcb6b2p-3, 0x1.2ebf3dcc9f24ep-2, -0x1.571d01c5c5700p-3, 0x1.93a9a7ba1302dp-8, 0x1.8281ce0b64bf9p-5, -0x1.5d00034e6d934p-6, -0x1.db43cca1e52edp-9, 0x1.756671dff7a0ep-8, -0x1.cc1c2fce1d469p-11, -0x1.9ebaf6388a9d8p-11, 0x1.50f558b0669b1p-12}},
    {{0x1.15aaa8ec85205p-5, -0x1.e723726b824a9p-4, 0x1.6d5a95d0a1b24p-3, -0x1.1c2a02beb6acdp-3, 0x1.6d5a95d0e3fe8p-5, 0x1.e723726bc0292p-7, -0x1.3ca3d7b9ba81ep-6, 0x1.36d739dde13e0p-8, 0x1.35b0703fc2452p-9, -0x1.c0370dbb499bdp-10, 0x1.76846ae70e497p-14, 0x1.09ecd2fea9675p-12}},
    {{0x1.b4be201caa4b4p-7, -0x1.b055303221015p-5, 0x1.7a4a8a2bdcd71p-4, -0x1.7148c3d57c2e9p-4, 0x1.8a0da54340f86p-5, -0x1.b22257dd7a0f7p-8, -0x1.25b379c47520ap-7, 0x1.8d10fbec88ca9p-8, -0x1.7eba3eba4f77dp-11, -0x1.d4d0fcefe33e5p-11, 0x1.cce38c94b0662p-12, 0x1.a4e2425477377p-18}},
    {{0x1.328f5ec350e67p-8, -0x1.529b9e8cf9a1ep-6, 0x1.529b9e8cf9a05p-5, -0x1.8b0ae3a4788d7p-5, 0x1.1a2c59757f58ap-5, -0x1.ace7404c62debp-7, -0x1.e193612087723p-12, 0x1.bae0ac8b67b83p-9, -0x1.a113e810f8d98p-10, 0x1.a44fa9c5c5b49p-15, 0x1.134e399508787p-12, -0x1.b00973c1ff3acp-14}},
    {{0x1.7f713f9cc9783p-10, -0x1.d4143a9dfe965p-8, 0x1.074b60f8df41dp-6, -0x1.63ef61e824418p-6, 0x1.38a98327624afp-6, -0x1.5d3b17bbe5a9dp-7, 0x1.7cae0e90978f9p-9, 0x1.5f83140762f9fp-11, -0x1.0604228718f3fp-10, 0x1.83527fb65d54dp-12, 0x1.a448645e3fd86p-17, -0x1.10edcd64bc8ecp-14}},
    {{0x1.aab859b20ac9dp-12, -0x1.1d83170fbf6fbp-9, 0x1.64e3dcd3af6cep-8, -0x1.119da0c46cd09p-7, 0x1.1a89b97cba5c5p-7, -0x1.90e81283da98bp-8, 0x1.6ecdc0365d488p-9, -0x1.1c610cdd72ff0p-11, -0x1.11583acbda9f2p-12, 0x1.067280fda61ebp-12, -0x1.47bfe2d08e3d2p-14, -0x1.676f97d1bc118p-18}},
    {{0x1.a609f7584d32bp-14, -0x1.3360ccd23db3ap-11, 0x1.a6a519a114e18p-10, -0x1.69cf466cce02dp-9, 0x1.ab0c273ab313dp-9, -0x1.693596063fda3p-9, 0x1.b2755c1e0e460p-10, -0x1.52b626dcdd7acp-11, 0x1.755677c721618p-14, 0x1.2abd9d6d3f0e4p-14, -0x1.cceaa0d3bd40fp-15, 0x1.07b3894bcebcdp-16}},
    {{0x1.729df65034230p-16, -0x1.2408e9ba3327fp-13, 0x1.b60d5e974c4a4p-12, -0x1.9db74b1d1dd06p-11, 0x1.11c85b1ee2cfdp-10, -0x1.0a7b5546b179fp-10, 0x1.82f234ff7fdd5p-11, -0x1.998b47d5abf31p-12, 0x1.1aa737717e520p-13, -0x1.d2a25165478bap-17, -0x1.085eb63742f48p-16, 0x1.69f8d84a727a9p-17}},
    {{0x1.20c1303550f1ep-18, -0x1.e9b5e8d00ce78p-16, 0x1.8de3cd29095c7p-14, -0x1.9aa489e3caa8ep-13, 0x1.2c7d5ef12599fp-12, -0x1.490a4d234ab2cp-12, 0x1.1454640a7f08ap-12, -0x1.647f719bfadbfp-13, 0x1.56762921e98d1p-14, -0x1.b2c50e744d729p-16, 0x1.2c97ed504a2ddp-19, 0x1.83e07111513d8p-19}},
    {{0x1.8ef2a9a18d875p-21, -0x1.6a597219a93dcp-18, 0x1.3d0e43d673074p-16, -0x1.62ccea63cab12p-15, 0x1.1c07721b8d476p-14, -0x1.586bafca3e808p-14, 0x1.46153ee733221p-14, -0x1.e827f9defad7ep-15, 0x1.1f63cf8207188p-15, -0x1.0136377850d21p-16, 0x1.360c879c6f4e1p-18, -0x1.ce4e1e4173e89p-22}},
    {{0x1.e87470e4f4217p-24, -0x1.d9371e2ff7c39p-21, 0x1.bba3ac4cf9f12p-19, -0x1.0b6a7b0f1b125p-17, 0x1.d06f585f5d223p-17, -0x1.3436bca02a923p-16, 0x1.4357b5fa5e733p-16, -0x1.110de42f28ac0p-16, 0x1.75661e9389f9fp-17, -0x1.99f6970f427dfp-18, 0x1.5df833cfbf755p-19, -0x1.9ecf06081e0b7p-21}},
    {{0x1.08ddd13bd34e5p-26, -0x1.10b1488aeb230p-23, 0x1.10b1488af4309p-21, -0x1.603a5308c5b3bp-20, 0x1.4980e24bed079p-19, -0x1.da5f10db61713p-19, 0x1.105056fc200cep-18, -0x1.fd7c66700181bp-19, 0x1.88c44a0ee8b6ap-19, -0x1.f4236bf395920p-20, 0x1.05f7f3583ffd8p-20, -0x1.af339c270ec00p-22}},
    {{0x1.fcae93fb72ab6p-30, -0x1.155a09065d4d2p-26, 0x1.26afa996e4fe8p-24, -0x1.95ea6fe000658p-23, 0x1.96ba734e1156bp-22, -0x1.3b468015f8c43p-21, 0x1.8868eefc08490p-21, -0x1.916e963513e10p-21, 0x1.5668a994c857ap-21, -0x1.eab106ddf01dap-22, 0x1.2a2a33cbd26a9p-22, -0x1.29928ccdcbb0bp-23}},
    {{0x1.b05cfe2e9814bp-33, -0x1.f1e3523b41c8ap-30, 0x1.180fde41aa25dp-27, -0x1.99b8665629efcp-26, 0x1.b598cb0807b78p-25, -0x1.6b1baf38f543ap-24, 0x1.e651045ac6e5ap-24, -0x1.0d67962a69900p-23, 0x1.f5e32a19fa10fp-24, -0x1.8d27fc15c07cap-24, 0x1.0fc42053f27b4p-24, -0x1.37c69e3cfed70p-25}},
    {{0x1.4531410820bfap-36, -0x1.8a61745ec7af6p-33, 0x1.d453ba31d2f76p-31, -0x1.6a8aeba49d1a4p-29, 0x1.9b017a453afd4p-28, -0x1.6b43c936c2f9ep-27, 0x1.042f4a0abfcb2p-26, -0x1.35dc9229118ffp-26, 0x1.3825a6988c396p-26, -0x1.0d45b29454bb1p-26, 0x1.976d4877bedb0p-27, -0x1.051d7ee7af912p-27}},
    {{0x1.b0c1a759f0883p-40, -0x1.13af4f04f95dcp-36, 0x1.589b22c82a29cp-34, -0x1.196da0aaacb06p-32, 0x1.516d3c01da1efp-31, -0x1.3c51d079a2b09p-30, 0x1.e235e7a9a4964p-30, -0x1.32c74326ea416p-29, 0x1.4bb756cf02be1p-29, -0x1.34f9678364e4cp-29, 0x1.fe675890886b9p-30, -0x1.670ef93c9a518p-30}},
    {{0x1.fd5f08ad19cb9p-44, -0x1.5422ef5d88a3cp-40, 0x1.be6dda2fa6900p-38, -0x1.7f8a0f3ede434p-36, 0x1.e4cb4922bd157p-35, -0x1.e044b362f0c07p-34, 0x1.83eac510d9ee3p-33, -0x1.06597cf8e4042p-32, 0x1.2ea81305ac120p-32, -0x1.2e26e583a68e2p-32, 0x1.0dfb64c216932p-32, -0x1.9be597e51954cp-33}},
    {{0x1.09182b3258f7ap-47, -0x1.7258610b39a88p-44, 0x1.fd398579a93c7p-42, -0x1.cb12e2f795f80p-40, 0x1.31011cba1778bp-38, -0x1.3e4a1eee24db1p-37, 0x1.0f6f08ccac2e9p-36, -0x1.84a533bac1111p-36, 0x1.dbfb9bc353330p-36, -0x1.fa52c68d06182p-36, 0x1.e60013716d50bp-36, -0x1.8e6a4b07d9332p-36}},
    {{0x1.e7eea02e0eceep-52, -0x1.63daf8b4af110p-48, 0x1.ff8ac595c1f3bp-46, -0x1.e2d06d72fda7bp-44, 0x1.505d91eda3dffp-42, -0x1.70b6ffef37c60p-41, 0x1.4aee47ed7f06cp-40, -0x1.f3c62fd9046dcp-40, 0x1.438b4d928ebb0p-39, -0x1.6cf31221e9dcbp-39, 0x1.7638142d73f1cp-39, -0x1.476976e6fedf5p-39}},
    {{0x1.8cf8155772405p-56, -0x1.2dc1190952aa1p-52, 0x1.c4a1a5a8f3195p-50, -0x1.be584a62e0a25p-48, 0x1.45542a15a4be2p-46, -0x1.75a81a2865782p-45, 0x1.5ff9234088ef0p-44, -0x1.177287afcb477p-43, 0x1.7d2495b5025fep-43, -0x1.c60964539d7f9p-43, 0x1.ef1d0d92952aap-43, -0x1.cbc4bc38a2e47p-43}},
};
#if defined(__HIP_DEVICE_COMPILE__)
static __shared__ __attribute__((aligned(16))) phf_erfctab phf_lds_erfc[PHF_ERFC_TAB_N];
#define PHF_T_ERFC(j) phf_lds_erfc[j]
#define PHF_ERFC_TABLE_TO_LDS()                                                                                     \
  do {                                                                                                              \
    for (int phf_i_ = threadIdx.x; phf_i_ < PHF_ERFC_TAB_N * 12; phf_i_ += blockDim.x)                              \
      (&phf_lds_erfc[0].c[0])[phf_i_] = (&phf_t_erfc[0].c[0])[phf_i_];                                              \
    __syncthreads();                                                                                                \
  } while (0)
#else
#define PHF_T_ERFC(j) phf_t_erfc[j]
#define PHF_ERFC_TABLE_TO_LDS() do { } while (0)
#endif

PHF_HD double phf_erfc_tab(double y) {
  const double t = phf_fma(y, 4.0, PHF_EXP_MAGIC);                      /* nearest integer to 4 y in the low mantissa bits */
  const double s = phf_fma(t - PHF_EXP_MAGIC, -0.25, y);                /* y - c_j, exact */
  const uint32_t jr = (uint32_t)phf_bits(t);
  const uint32_t j = jr < (uint32_t)PHF_ERFC_TAB_N ? jr : (uint32_t)(PHF_ERFC_TAB_N - 1);
  const phf_erfctab e = PHF_T_ERFC(j);
  double p = phf_fma(e.c[11], s, e.c[10]);
  p = phf_fma(p, s, e.c[9]);
  p = phf_fma(p, s, e.c[8]);
  p = phf_fma(p, s, e.c[7]);
  p = phf_fma(p, s, e.c[6]);
  p = phf_fma(p, s, e.c[5]);
  p = phf_fma(p, s, e.c[4]);
  p = phf_fma(p, s, e.c[3]);
  p = phf_fma(p, s, e.c[2]);
  p = phf_fma(p, s, e.c[1]);
  p = phf_fma(p, s, e.c[0]);
  return (y < PHF_ERFC_CUT) ? p : 0.0;
}

/* log Phi(x) for x <= 0 through a table — the single-level censored likelihood's form: log Phi(x) = -x^2/2 + g(y), y = -x/sqrt2,
 * g(y) = log(erfcx(y)/2), smooth and slowly varying, so a degree-9 polynomial per interval does it: v = y + 1 = 2^E m, interval
 * 8 E + floor(8 (m - 1)), E = 0..16, i.e. 0 <= y < 131071, g = P(m - 1) to 9.4e-16 absolute (tools/gen_math_coeffs.py logphi):
 * 12 fp64 operations and five 16-byte LDS reads where erfcx + log took ~50, and no branch.  The range holds every argument
 * the likelihood can produce while it is not -inf anyway: |pred - y| <= 100 and sigma above its floor of 1e-3 give y <= 70 711.
 * (The index is clamped, so any other argument is safe to evaluate; its value is meaningless.)  The table (10.6 KB) goes to
 * LDS with PHF_LOGPHI_TABLE_TO_LDS(). */
#define PHF_LOGPHI_TAB_N 136
#define PHF_LOGPHI_Y_MAX 131071.0
typedef struct { double c[10]; } phf_logphitab;
static const phf_logphitab phf_t_logphi[PHF_LOGPHI_TAB_N] = {
    {{-0x1.62e42fefa39efp-1, -0x1.20dd750429b6dp+0, 0x1.7419f246c6d8cp-2, -0x1.a4f4e2a190066p-4, 0x1.3966857717addp-6, 0x1.b6b8432d6421cp-13, -0x1.bcaa6f2438639p-10, 0x1.35bf16dc0c7f7p-11, -0x1.ff599d624e3e7p-16, -0x1.88e331ca07a8ep-15}},
    {{-0x1.62e42fefa5645p-1, -0x1.20dd7503f75c2p+0, 0x1.7419f232c1a25p-2, -0x1.a4f4ddf04ee85p-4, 0x1.3965cd982361fp-6, 0x1.b91eb89e87a43p-13, -0x1.be0c36d42cde1p-10, 0x1.3e5072c5af549p-11, -0x1.035c87f6930d4p-14, -0x1.308cf66baa915p-16}},
    {{-0x1.62e42ff087942p-1, -0x1.20dd74f56e9fap+0, 0x1.7419eed90a817p-2, -0x1.a4f46962ddcbap-4, 0x1.395b7f78acabbp-6, 0x1.ccd332a4f6608p-13, -0x1.c46c54b5c72e8p-10, 0x1.53e16a9767452p-11, -0x1.b088988fa343fp-14, 0x1.3db8124b73b12p-21}},
    {{-0x1.62e42ffe7b78ap-1, -0x1.20dd74542078ap+0, 0x1.7419d4c9a8926p-2, -0x1.a4f1f084775cbp-4, 0x1.3934a25d79307p-6, 0x1.000e0f732f135p-12, -0x1.cfc9983a87eecp-10, 0x1.6dfb46dad24aap-11, -0x1.1ebf3f534f905p-13, 0x1.68cd3bd227726p-17}},
    {{-0x1.62e4303cd0ae1p-1, -0x1.20dd7227890d5p+0, 0x1.74198f7a7af5fp-2, -0x1.a4ece24f01dfdp-4, 0x1.38f7b14aa23f8p-6, 0x1.1ecb289b943b0p-12, -0x1.da2ad70f2ea41p-10, 0x1.80153f472f079p-11, -0x1.43ba081ee8a61p-13, 0x1.efa5113553381p-17}},
    {{-0x1.62e4302aac17ap-1, -0x1.20dd72831fa7ep+0, 0x1.741994c5b0119p-2, -0x1.a4ecf7fb6f253p-4, 0x1.38f65b65aa28fp-6, 0x1.20307d8fa6ad7p-12, -0x1.dac55c202a788p-10, 0x1.813b4efbe6f87p-11, -0x1.461601cff1039p-13, 0x1.f7de2b2edf5b0p-17}},
    {{-0x1.62e42b2cf92ffp-1, -0x1.20dd8fc8f2095p+0, 0x1.741bf818ca4d5p-2, -0x1.a50a1a6178d73p-4, 0x1.39db36b5e53e3p-6, 0x1.aa54aa246936dp-13, -0x1.ca590b0ae5886p-10, 0x1.6eb62e0b4a0b2p-11, -0x1.2dadc3f2f856ep-13, 0x1.be959fccd8842p-17}},
    {{-0x1.62e410cb964a6p-1, -0x1.20de15ccdcb9bp+0, 0x1.74256fd719a43p-2, -0x1.a56e1ddae6590p-4, 0x1.3c834acd93edap-6, 0x1.438c7bcb87f50p-16, -0x1.a5ce1d389c867p-10, 0x1.4b119e50eecffp-11, -0x1.05102f0c68050p-13, 0x1.6c30bc6636b9cp-17}},
    {{-0x1.8af1d71f7bf99p+0, -0x1.4726c001b090cp+0, 0x1.4150fb975f3f8p-1, -0x1.5477f09f1b56cp-2, 0x1.48c9d50ae8ec5p-3, -0x1.033fa53e03d02p-4, 0x1.f1d6cb39f635cp-7, 0x1.b2b5f5a9c1652p-9, -0x1.b2978cb5dd1eap-8, 0x1.e2d7494e23338p-9}},
    {{-0x1.8af1d71f45a4fp+0, -0x1.4726c00dc7e87p+0, 0x1.4150fe002c137p-1, -0x1.5478391edb593p-2, 0x1.48cf66e565d6bp-3, -0x1.038a77035b933p-4, 0x1.fcad5efd0aef5p-7, 0x1.2b3de27d6d2a8p-9, -0x1.2f5fbe13cebd5p-8, 0x1.e08f1ba41e943p-10}},
    {{-0x1.8af1d7072c11cp+0, -0x1.4726c328ca635p+0, 0x1.415159ea9e5fbp-1, -0x1.547e7d7256f79p-2, 0x1.49169ac039d5fp-3, -0x1.05ad6ce5e0027p-4, 0x1.149046d6d3645p-6, -0x1.7e09cb03dcb82p-16, -0x1.2dd2ab32287b6p-9, 0x1.947875cef17d3p-11}},
    {{-0x1.8af1d5a031bacp+0, -0x1.4726e39848600p+0, 0x1.4153f8f9e1a8fp-1, -0x1.549e562c9c087p-2, 0x1.4a1114f52a8eep-3, -0x1.0ad873d847c63p-4, 0x1.393cc67ac9d65p-6, -0x1.542d98de19d68p-9, -0x1.27ff01b1f8aecp-11, 0x1.016856f0e2620p-12}},
    {{-0x1.8af1ce3c54ea2p+0, -0x1.472766996d692p+0, 0x1.415c1196b112ep-1, -0x1.54e95d8073a9ep-2, 0x1.4bd1f6710b1dep-3, -0x1.11de8c609751cp-4, 0x1.5eea1c26f4b16p-6, -0x1.2c948c019841dp-8, 0x1.d319b65d03660p-12, 0x1.19e4d0805e98ap-16}},
    {{-0x1.8af1bb8a0c788p+0, -0x1.472873263fd0cp+0, 0x1.41697fb2a222fp-1, -0x1.554df024b137ap-2, 0x1.4db776b068cf8p-3, -0x1.17fd4db435c22p-4, 0x1.7951f538044b7p-6, -0x1.7604ae18f0ccap-8, 0x1.d877bd0bf88d3p-11, -0x1.13e2a71f66badp-14}},
    {{-0x1.8af1a98cf0659p+0, -0x1.47294f4116e70p+0, 0x1.4172dd34b4d7ap-1, -0x1.55898868019cep-2, 0x1.4eab9a57c428ep-3, -0x1.1a99035bac12fp-4, 0x1.82d82c2273807p-6, -0x1.8c67ed75bccc0p-8, 0x1.0af8954a8f90cp-10, -0x1.5f0095d8aba0ep-14}},
    {{-0x1.8af1d65024631p+0, -0x1.47278f5dc0344p+0, 0x1.41634b14c4574p-1, -0x1.5538a47d66dcbp-2, 0x1.4d9d46396d843p-3, -0x1.183e509ebe33cp-4, 0x1.7bd7439c115e4p-6, -0x1.7f00b256cb359p-8, 0x1.f7fe58c59ecfbp-11, -0x1.413cfdfec11e5p-14}},
    {{-0x1.34ede72930114p+1, -0x1.370b35013b9b7p+0, 0x1.5d06cc9bbcb42p-1, -0x1.e5bee8cd2b547p-2, 0x1.62c31d10063c7p-2, -0x1.01a01a4fc16eep-2, 0x1.6957b207fbe27p-3, -0x1.dc4c758cdb037p-4, 0x1.16c9a96e6be47p-4, -0x1.cbdb7ef9940b3p-6}},
    {{-0x1.34ede729e8e6ep+1, -0x1.370b34aefd973p+0, 0x1.5d06bc3a28b96p-1, -0x1.e5bcfc2ad3d00p-2, 0x1.62b0347e60d67p-2, -0x1.01213e09378edp-2, 0x1.64c16b11c3e03p-3, -0x1.bfb179c736365p-4, 0x1.bf0df10c3f612p-5, -0x1.003b81957a387p-6}},
    {{-0x1.34ede77e6e38bp+1, -0x1.370b1eec87d72p+0, 0x1.5d0438f370eb1p-1, -0x1.e5912ece59e59p-2, 0x1.61b7b94d9b6ccp-2, -0x1.fad09d8cb5078p-3, 0x1.516d0e917cc8cp-3, -0x1.7e16b912b19bbp-4, 0x1.3ae500cb75b6cp-5, -0x1.0f99610fba29ep-7}},
    {{-0x1.34edecde3d9b1p+1, -0x1.370a26fe07469p+0, 0x1.5cf03e860eac9p-1, -0x1.e49f35f923ebap-2, 0x1.5e032db23d983p-2, -0x1.e74eb96ab0c30p-3, 0x1.2eeefa5acfed1p-3, -0x1.2f118731e3fd6p-4, 0x1.a0efbaf25055fp-6, -0x1.1e641be48a96ep-8}},
    {{-0x1.34ee0f59cbdbdp+1, -0x1.3705668aa5b34p+0, 0x1.5ca573aed3473p-1, -0x1.e1edb7608de7fp-2, 0x1.55ff8f8b8e561p-2, -0x1.c7606e64f5c81p-3, 0x1.045797e8a8b39p-3, -0x1.cb69b3018cbf4p-5, 0x1.0cde12e63b71ep-6, -0x1.31fb1d6e4a4b4p-9}},
    {{-0x1.34ee916e33e37p+1, -0x1.36f6eea16469fp+0, 0x1.5bedde84d9ed0p-1, -0x1.dc9b3f5f096ebp-2, 0x1.494480220b3aep-2, -0x1.9ea950292c13fp-3, 0x1.b1a1c0d2857d6p-4, -0x1.536d393f84226p-5, 0x1.58410a16f8cf0p-7, -0x1.4ddde7553e49ap-10}},
    {{-0x1.34eff099b2641p+1, -0x1.36d63bbcd4f1ep+0, 0x1.5a92cc7d34296p-1, -0x1.d4323e82408c8p-2, 0x1.38774ebb2bc13p-2, -0x1.71d227ea60339p-3, 0x1.61b237a2c5389p-4, -0x1.ef4912a413bd8p-6, 0x1.ba20846976e9ap-8, -0x1.75399b563b9d3p-11}},
    {{-0x1.34f2e58dcca11p+1, -0x1.3699aa53620f2p+0, 0x1.586a9e6fc43b5p-1, -0x1.c8b5946abc3c2p-2, 0x1.24c664686be23p-2, -0x1.44c032e428b15p-3, 0x1.1cd34dec068e2p-4, -0x1.67c8d9c5e9fc5p-6, 0x1.1e64dba39da90p-8, -0x1.abcdf22f37757p-12}},
    {{-0x1.9c561f90047a9p+1, -0x1.1ee25c5c3261ep+0, 0x1.3b7909241c425p-1, -0x1.c61be51c78652p-2, 0x1.69164984575a3p-2, -0x1.2cc8642b84059p-2, 0x1.00215fcc5e2b3p-2, -0x1.b4b7d1baf2898p-3, 0x1.5cb933c19bd71p-3, -0x1.8692f96ce3cbep-4}},
    {{-0x1.9c561f92b358dp+1, -0x1.1ee25b280c5c3p+0, 0x1.3b78cb1ec2916p-1, -0x1.c61482991147ap-2, 0x1.68cc6b3693086p-2, -0x1.2accd1bd9a4fdp-2, 0x1.ed50a66ac17a8p-3, -0x1.772c92df3c782p-3, 0x1.bfa26e98eafabp-4, -0x1.27597d6304776p-5}},
    {{-0x1.9c56206cc81c1p+1, -0x1.1ee2228b581f0p+0, 0x1.3b72318d8ff93p-1, -0x1.c5a0388c32996p-2, 0x1.6630c297a570ap-2, -0x1.20aa43d88a14cp-2, 0x1.b7ee6455b76f9p-3, -0x1.1b35e72610f6dp-3, 0x1.037d825fb4cfap-4, -0x1.e4e6a403d9f24p-7}},
    {{-0x1.9c562b23de7fep+1, -0x1.1ee031939b626p+0, 0x1.3b49eb6eccff6p-1, -0x1.c3b57c850ab26p-2, 0x1.5ea0dd9825073p-2, -0x1.0ca0b526c9f80p-2, 0x1.7097e1652efedp-3, -0x1.91de2b7fcb766p-4, 0x1.27d3e505cca35p-5, -0x1.aba4e864b6681p-8}},
    {{-0x1.9c5663f814304p+1, -0x1.1ed8565586617p+0, 0x1.3acdd8af4f157p-1, -0x1.bf39bd1b66f3cp-2, 0x1.513da3beec5a8p-2, -0x1.e3b930517de38p-3, 0x1.28edde51ed2a3p-3, -0x1.15fa5d888e70ep-4, 0x1.54b1204c37ef7p-6, -0x1.917258bd4b90bp-9}},
    {{-0x1.9c571dd58ca27p+1, -0x1.1ec39f5eddb3bp+0, 0x1.39c6782675962p-1, -0x1.b792fffd8b7b7p-2, 0x1.3ee665823c8bfp-2, -0x1.a8f0e189210ecp-3, 0x1.d3e86108869c5p-4, -0x1.7dfbbed43663bp-5, 0x1.903faace0be3dp-7, -0x1.8df3e5da2738cp-10}},
    {{-0x1.9c58e0128273ep+1, -0x1.1e99a51e9ae33p+0, 0x1.38085471e487bp-1, -0x1.acbfc675a4edfp-2, 0x1.293ec317b84cbp-2, -0x1.6f12078b35a68p-3, 0x1.6c9aed0e14227p-4, -0x1.0735b6b3861f8p-5, 0x1.e13d5d6c7a4b4p-8, -0x1.9dc6ed2148913p-11}},
    {{-0x1.9c5c5cfb27e81p+1, -0x1.1e52231552b56p+0, 0x1.357bd9308f421p-1, -0x1.9f29e9c91ad6ap-2, 0x1.11efbde9f870ap-2, -0x1.39ac50dcb9977p-3, 0x1.1af01a4a93434p-4, -0x1.6d99bc5fce16cp-6, 0x1.283eec126647ap-8, -0x1.c0b50f39ae25ep-12}},
    {{-0x1.fce61b9c38f4dp+1, -0x1.0fddc62427210p+0, 0x1.1f75329552839p-1, -0x1.9380d57467359p-2, 0x1.3d393b153296ep-2, -0x1.08dd26f91f501p-2, 0x1.ca5f46826f02cp-3, -0x1.92362160c47d7p-3, 0x1.4d2485258a67bp-3, -0x1.81d81dabce0cdp-4}},
    {{-0x1.fce61b9edf659p+1, -0x1.0fddc4f375c4ap+0, 0x1.1f74f52cb913ep-1, -0x1.9379824f4944ep-2, 0x1.3cefcad4e9fe7p-2, -0x1.06e2f7886d3bdp-2, 0x1.b766a1c67860cp-3, -0x1.5444217276ca9p-3, 0x1.9d53aa98ff8cbp-4, -0x1.14a678e08e9edp-5}},
    {{-0x1.fce61c6cbca3ep+1, -0x1.0fdd8f73993e4p+0, 0x1.1f6eb680d9ea3p-1, -0x1.930b5207b1652p-2, 0x1.3a762a6ed9aa2p-2, -0x1.fa80e03810210p-3, 0x1.848ee21c1a620p-3, -0x1.f90644fbaf1d8p-4, 0x1.d2e3d7dc322cap-5, -0x1.b76764ad6f4fap-7}},
    {{-0x1.fce62638f82bdp+1, -0x1.0fdbc8bb45763p+0, 0x1.1f49d6330e4d7p-1, -0x1.9149a6fa4c27cp-2, 0x1.3386e46ce373ep-2, -0x1.d5b9616489051p-3, 0x1.43084d17786a6p-3, -0x1.61c083d0ff1fbp-4, 0x1.0593811c96bf7p-5, -0x1.7b9424ff607c2p-8}},
    {{-0x1.fce6590d05bdbp+1, -0x1.0fd4c134912e7p+0, 0x1.1edac8570b573p-1, -0x1.8d45f12a5658bp-2, 0x1.2788e47445c97p-2, -0x1.a5c0a572bde87p-3, 0x1.02c94657efe63p-3, -0x1.e54812afa9680p-5, 0x1.29fbb5ab0857bp-6, -0x1.5fcc753214f45p-9}},
    {{-0x1.fce6fcd70c83fp+1, -0x1.0fc27f2dd1c26p+0, 0x1.1df29884a16e7p-1, -0x1.8686d05d87468p-2, 0x1.175bfb873c51bp-2, -0x1.71e637fbeb28cp-3, 0x1.96720e3b41170p-4, -0x1.4bc09d2ccf98fp-5, 0x1.5bd0d16197b43p-7, -0x1.5a17afde03a72p-10}},
    {{-0x1.fce885c8ab180p+1, -0x1.0f9ddb7022347p+0, 0x1.1c6d2586230cap-1, -0x1.7d1376bde6dd9p-2, 0x1.0473b8c9382f3p-2, -0x1.3f5da7ae71af8p-3, 0x1.3c3aef92b5b27p-4, -0x1.c807886769148p-6, 0x1.a0c24ba3271eap-8, -0x1.665bd1eeee105p-11}},
    {{-0x1.fceb8c66db335p+1, -0x1.0f5fd1b0e2d0cp+0, 0x1.1a370b03b3cb3p-1, -0x1.7149d9082e738p-2, 0x1.e0742185de7ccp-3, -0x1.11079e0e63a0dp-3, 0x1.eab79e1e8222dp-5, -0x1.3c761e1a37915p-6, 0x1.0033a4782b8b8p-8, -0x1.83e55878cbd42p-12}},
    {{-0x1.2ccd1c4217f06p+2, -0x1.07fbd9f912100p+0, 0x1.0fef440ececb3p-1, -0x1.751d18eccdb71p-2, 0x1.1faae85a6c99dp-2, -0x1.d8a4f64b005b4p-3, 0x1.93ba0f7a6ea53p-3, -0x1.5ee54483d0e07p-3, 0x1.20e0ed865197bp-3, -0x1.4d90f9e50b7cdp-4}},
    {{-0x1.2ccd1c433cc5cp+2, -0x1.07fbd8f20c571p+0, 0x1.0fef0f0c7f1b5p-1, -0x1.7516c64eb173fp-2, 0x1.1f6b848924757p-2, -0x1.d53b249100a05p-3, 0x1.835a3c36bb8c0p-3, -0x1.296e7f9bf51b4p-3, 0x1.67706155216f2p-4, -0x1.dfc8f289b2723p-6}},
    {{-0x1.2ccd1c9c4daddp+2, -0x1.07fbaaa81f502p+0, 0x1.0fe9a7f8d1d68p-1, -0x1.74b77497f2f45p-2, 0x1.1d4774a770135p-2, -0x1.c490bdb4973f7p-3, 0x1.5762d33328dcep-3, -0x1.bb1c899986b50p-4, 0x1.97d97de8b288cp-5, -0x1.7ed331b896189p-7}},
    {{-0x1.2ccd20dd8d362p+2, -0x1.07fa1fc11cad2p+0, 0x1.0fc9a263a36d1p-1, -0x1.733106561f87ap-2, 0x1.174239ffbf863p-2, -0x1.a4a381f440887p-3, 0x1.1e833185f2557p-3, -0x1.37d4426b584f6p-4, 0x1.cb608f244ed78p-6, -0x1.4c7ebaba492e5p-8}},
    {{-0x1.2ccd370a20d2bp+2, -0x1.07f3fdcb8cb20p+0, 0x1.0f68c1bb01a67p-1, -0x1.6fb09837f8ee7p-2, 0x1.0ccca8626c863p-2, -0x1.7acdad780004fp-3, 0x1.ccfaf08bc57f3p-4, -0x1.ade6b12acbb8dp-5, 0x1.07121a4360315p-6, -0x1.35e2082e985f2p-9}},
    {{-0x1.2ccd7ed8bb148p+2, -0x1.07e3fbd4c3fc4p+0, 0x1.0e9d33fb3867bp-1, -0x1.69c6947f50c7ep-2, 0x1.fd3e03448b372p-3, -0x1.4d5b2909f68a6p-3, 0x1.6b970c900fb33p-4, -0x1.275c8cd33e765p-5, 0x1.34b3a883162f0p-7, -0x1.328824b43c3d0p-10}},
    {{-0x1.2cce2bfba9931p+2, -0x1.07c3b2b447e52p+0, 0x1.0d460e6e775ddp-1, -0x1.617304e40ad2cp-2, 0x1.dbed9c54346dfp-3, -0x1.20d7062266397p-3, 0x1.1c1f682d86b8fp-4, -0x1.97fa71deeefe9p-6, 0x1.73ca0301bf51cp-8, -0x1.3f11ad0719289p-11}},
    {{-0x1.2ccf82f9b51ecp+2, -0x1.078cc2f8edfebp+0, 0x1.0b50c8d104275p-1, -0x1.5703243a50986p-2, 0x1.b81d44f51ecf3p-3, -0x1.efa28ad431cc9p-4, 0x1.bac38e43154f9p-5, -0x1.1c6df5d3eabb2p-6, 0x1.cb56326b3c13ep-9, -0x1.5b1753c561467p-12}},
    {{-0x1.5a2955556c61ep+2, -0x1.03ff7d8ee4366p+0, 0x1.07fdf3d9d0e7ap-1, -0x1.654e79fb5984ap-2, 0x1.0ff599b1d1605p-2, -0x1.b97842a7129fdp-3, 0x1.74decda57c75fp-3, -0x1.40c33460c0746p-3, 0x1.05d0099be78c3p-3, -0x1.2c84a2b7ee0d0p-4}},
    {{-0x1.5a29555673f2fp+2, -0x1.03ff7ca2301d6p+0, 0x1.07fdc42772f64p-1, -0x1.6548c9e374485p-2, 0x1.0fbc976dc319dp-2, -0x1.b666b20d7b67cp-3, 0x1.66282b2250db2p-3, -0x1.10c01ff1c8410p-3, 0x1.47b492c7bf9f7p-4, -0x1.b3cea865e58d0p-6}},
    {{-0x1.5a2955a7290bap+2, -0x1.03ff52b24b092p+0, 0x1.07f8df5c139a1p-1, -0x1.64f277d75374dp-2, 0x1.0dcc655f7def0p-2, -0x1.a7511db2874edp-3, 0x1.3e5fbab0fe622p-3, -0x1.983afd287ecc3p-4, 0x1.762649fd6a5fbp-5, -0x1.5e39cd4eecdc2p-7}},
    {{-0x1.5a29598862b58p+2, -0x1.03fdeaa8cdc92p+0, 0x1.07dbaeafe0a4cp-1, -0x1.638e9f024b844p-2, 0x1.084fecc2186f3p-2, -0x1.8a3a6c20f8e1dp-3, 0x1.0a90b59025e5ep-3, -0x1.20a9376a20f05p-4, 0x1.a7d381e3e6979p-6, -0x1.321541ab28b85p-8}},
    {{-0x1.5a296dde77569p+2, -0x1.03f84afac32c9p+0, 0x1.0782da9556d70p-1, -0x1.6058c3b33504cp-2, 0x1.fd72c6447bc05p-3, -0x1.63e1d6df1602fp-3, 0x1.ae716c9f9a030p-4, -0x1.8fc470072941fp-5, 0x1.e7e9d24364c97p-7, -0x1.1ed49cbbb870cp-9}},
    {{-0x1.5a29b011c87dep+2, -0x1.03e98947d6bc8p+0, 0x1.06c7393e742cbp-1, -0x1.5ae54d833582fp-2, 0x1.e3507612456cdp-3, -0x1.3a00259a10409p-3, 0x1.54b3969339ba7p-4, -0x1.13cdeca65361cp-5, 0x1.1f959421fb88dp-7, -0x1.1d19841972fb0p-10}},
    {{-0x1.5a2a5070ee531p+2, -0x1.03cba1faa3376p+0, 0x1.05896a31dad69p-1, -0x1.532f4220482c6p-2, 0x1.c476c6fbd6086p-3, -0x1.10c77907ac57fp-3, 0x1.0b1ed1fce7d52p-4, -0x1.7e6923ebe0c1ap-6, 0x1.5bc67fed79d86p-8, -0x1.2a08312f1549cp-11}},
    {{-0x1.5a2b8f7ef796cp+2, -0x1.039888601d06bp+0, 0x1.03b72bc11c5c3p-1, -0x1.497a34a877e4cp-2, 0x1.a327eba52c074p-3, -0x1.d541f02d295b3p-4, 0x1.a18cd51e2469bp-5, -0x1.0b85da83f8d14p-6, 0x1.af3f618c1fec5p-9, -0x1.45720f8eeee00p-12}},
    {{-0x1.8705d1d942782p+2, -0x1.01ffefd8735f4p+0, 0x1.03ffbf3a51c6ap-1, -0x1.5d547c99c36d7p-2, 0x1.07fea99fdc5d9p-2, -0x1.a9920338a98bfp-3, 0x1.650600f23c5e7p-3, -0x1.312a5caa6c79fp-3, 0x1.ef6344bb8ea87p-4, -0x1.1b2c251acaa19p-4}},
    {{-0x1.8705d1da3aae7p+2, -0x1.01ffeef98ed1dp+0, 0x1.03ff9251c1305p-1, -0x1.5d4f21d6f4f5dp-2, 0x1.07c901681e264p-2, -0x1.a6aed4a7edaddp-3, 0x1.572efedee6042p-3, -0x1.040602811f4dfp-3, 0x1.3745774a159f6p-4, -0x1.9cf691bcd7c2bp-6}},
    {{-0x1.8705d22696a74p+2, -0x1.01ffc74d4e7b6p+0, 0x1.03faf1544e237p-1, -0x1.5cfd7f0d2b777p-2, 0x1.05f3d1c1a5554p-2, -0x1.986c21f077a6fp-3, 0x1.3194a647502a9p-3, -0x1.865354d637f82p-4, 0x1.64ce1a52ac20dp-5, -0x1.4d6a98ae3a9e8p-7}},
    {{-0x1.8705d5d631f57p+2, -0x1.01fe7148a57f9p+0, 0x1.03df375bdaae0p-1, -0x1.5bab87436e294p-2, 0x1.00be0fbf91264p-2, -0x1.7ccd2d403d3b6p-3, 0x1.00644ea430369p-3, -0x1.14d0d4b901ad9p-4, 0x1.959e065507222p-6, -0x1.24899bea3c760p-8}},
    {{-0x1.8705e93a6ef8ap+2, -0x1.01f91493683c7p+0, 0x1.038a869d173e5p-1, -0x1.589c03c312bcbp-2, 0x1.ef346b95023f4p-3, -0x1.583ff66654cbdp-3, 0x1.9ee8a72f7c28ep-4, -0x1.8069346dabf20p-5, 0x1.d468212b062a4p-7, -0x1.130e6eb7b9faap-9}},
    {{-0x1.8706288f106b4p+2, -0x1.01eaf6d2891b1p+0, 0x1.02d70c23e8765p-1, -0x1.53653f9a7e513p-2, 0x1.d6356e3683824p-3, -0x1.3031aae6c27a1p-3, 0x1.491578c141d88p-4, -0x1.09dd4448fb703p-5, 0x1.14d72e161d7eap-7, -0x1.1230179cd8b0dp-10}},
    {{-0x1.8706c26b11624p+2, -0x1.01ce46a6ce90bp+0, 0x1.01a62a9e09757p-1, -0x1.4bff933219c9bp-2, 0x1.b89d93adb05b3p-3, -0x1.08a76587f2959p-3, 0x1.0281aa7dba503p-4, -0x1.71719a27caddcp-6, 0x1.4f9648aa23c58p-8, -0x1.1f5927d6d929dp-11}},
    {{-0x1.8707f54b1b36ap+2, -0x1.019d20c79b7bdp+0, 0x1.ffcb7e10f3e75p-2, -0x1.42a996cfeaea9p-2, 0x1.989526db88f4bp-3, -0x1.c7ee19485bff6p-4, 0x1.94cbc65d1cfd8p-5, -0x1.02f6ad423be29p-6, 0x1.a1049f7b58418p-9, -0x1.3a799aa3a6463p-12}},
    {{-0x1.b3a256aa39d39p+2, -0x1.00fffdfd835e1p+0, 0x1.01fff7f331bd0p-1, -0x1.59553a28e7147p-2, 0x1.03ffc73d2fd65p-2, -0x1.a19546b66b750p-3, 0x1.5d0cc0c353b09p-3, -0x1.294db3e42c29ep-3, 0x1.e121166050c02p-4, -0x1.1266d8dbf7362p-4}},
    {{-0x1.b3a256ab2a468p+2, -0x1.00fffd259a73dp+0, 0x1.01ffcc738f94ep-1, -0x1.59500a8cc3e56p-2, 0x1.03cbd0f6171efp-2, -0x1.9ec98df0234dfp-3, 0x1.4fa6df2cdaa87p-3, -0x1.fb3a35d376275p-4, 0x1.2efc8c4f54e9ep-4, -0x1.91705de0e8663p-6}},
    {{-0x1.b3a256f55485fp+2, -0x1.00ffd69de0536p+0, 0x1.01fb4dae7dca2p-1, -0x1.5900c50debd74p-2, 0x1.0204436843689p-2, -0x1.90f14eea87f5ep-3, 0x1.2b264562e916dp-3, -0x1.7d513c07e1cdep-4, 0x1.5c12d53813bb7p-5, -0x1.44f35b6acc81cp-7}},
    {{-0x1.b3a25a8bef63ap+2, -0x1.00fe89ad98f38p+0, 0x1.01e0508740e82p-1, -0x1.57b7cfc8c365cp-2, 0x1.f9e4487770321p-3, -0x1.760fb6b23dfbdp-3, 0x1.f68fe3e28823bp-4, -0x1.0edc986b7f577p-4, 0x1.8c75f29e339e6p-6, -0x1.1db9587773842p-8}},
    {{-0x1.b3a26d7676005p+2, -0x1.00f94ead0be1ap+0, 0x1.018db4f3bcd97p-1, -0x1.54bb98a2177e0p-2, 0x1.e81007000a931p-3, -0x1.5269b685f84a1p-3, 0x1.971c0faaa6f12p-4, -0x1.78b2a7dbca186p-5, 0x1.ca9b2d775f192p-7, -0x1.0d23b36a4c6a2p-9}},
    {{-0x1.b3a2ab59b2d45p+2, -0x1.00eb835a98e49p+0, 0x1.00de53c390cecp-1, -0x1.4fa3592cca60ep-2, 0x1.cfa38545ad197p-3, -0x1.2b46701481810p-3, 0x1.43410c6072b56p-4, -0x1.04dff1c758e33p-5, 0x1.0f7243a58d30bp-7, -0x1.0cb55598f95a5p-10}},
    {{-0x1.b3a341f01f78fp+2, -0x1.00cf6f7f706b8p+0, 0x1.ff67e1fb5e48dp-2, -0x1.48660e02644fcp-2, 0x1.b2ad5ca4282d9p-3, -0x1.049469d26c0e9p-3, 0x1.fc5f30599b846p-5, -0x1.6af029073151dp-6, 0x1.49788eb8a13d1p-8, -0x1.19fc864d8ea9ap-11}},
    {{-0x1.b3a46eb2a4b5bp+2, -0x1.009f4489cd922p+0, 0x1.fbf8f0cf7402bp-2, -0x1.3f3fcd4bb83f2p-2, 0x1.9348d6eb55292p-3, -0x1.c13fdda93a37fp-4, 0x1.8e66acc039fe3p-5, -0x1.fd579a43c3c24p-7, 0x1.99e182187a3c1p-9, -0x1.34f8cba87af56p-12}},
    {{-0x1.e01edc82b39e2p+2, -0x1.007fffbfd7ecep+0, 0x1.00fffefed6c9ep-1, -0x1.575551ae29af6p-2, 0x1.01ffead9900c0p-2, -0x1.9d95bb81240a6p-3, 0x1.590e8f47769e1p-3, -0x1.255d656e9f9f5p-3, 0x1.d9fba62dc1f02p-4, -0x1.0e01285ac39cdp-4}},
    {{-0x1.e01edc83a02cbp+2, -0x1.007ffeeb6f3d9p+0, 0x1.00ffd43429187p-1, -0x1.575037b46d40ep-2, 0x1.01ccce2389225p-2, -0x1.9ad5c5f8b1ca0p-3, 0x1.4be166bca4a3bp-3, -0x1.f4ce6acf38b9fp-4, 0x1.2ad5f983a0e1bp-4, -0x1.8baa1512ed433p-6}},
    {{-0x1.e01edcccb0eeap+2, -0x1.007fd8f64c62dp+0, 0x1.00fb66952f750p-1, -0x1.5702218b672d2p-2, 0x1.000c15a23c3eep-2, -0x1.8d32e047e13e7p-3, 0x1.27ee005d63bfap-3, -0x1.78ce719f420e0p-4, 0x1.57b35899f46d3p-5, -0x1.40b5d49acdb02p-7}},
    {{-0x1.e01ee056c592ep+2, -0x1.007e909266746p+0, 0x1.00e0c8044d7c4p-1, -0x1.55bdafba06f96p-2, 0x1.f617a1305e5ecp-3, -0x1.72b024da05318p-3, 0x1.f172051799c7ap-4, -0x1.0be17e7579e44p-4, 0x1.87e047779463fp-6, -0x1.1a4ff031c4354p-8}},
    {{-0x1.e01ef304583f5p+2, -0x1.0079667310f31p+0, 0x1.008f37737d849p-1, -0x1.52cb22b65e799p-2, 0x1.e47d30ba52559p-3, -0x1.4f7defcfe6f98p-3, 0x1.9334c2731e292p-4, -0x1.74d6497fa26dfp-5, 0x1.c5b33835c24b7p-7, -0x1.0a2d6658c2180p-9}},
    {{-0x1.e01f302ea2969p+2, -0x1.006bc4662f64cp+0, 0x1.ffc3c73e11fa1p-2, -0x1.4dc22af252ff9p-2, 0x1.cc5a068731327p-3, -0x1.28d0553f9e354p-3, 0x1.40562daae8ecap-4, -0x1.0260abc25dc92p-5, 0x1.0cbf1b073231ap-7, -0x1.09f736e6ec1fap-10}},
    {{-0x1.e01fc521c76f9p+2, -0x1.004ffecace041p+0, 0x1.fd75826710ee8p-2, -0x1.4699167ea2003p-2, 0x1.afb4cf4b5a494p-3, -0x1.028a8f4ddc067p-3, 0x1.f80c42470d87ep-5, -0x1.67aebe2a38897p-6, 0x1.4669013773899p-8, -0x1.174d94ed9a953p-11}},
    {{-0x1.e020eed4bdc03p+2, -0x1.0020516acc4fep+0, 0x1.fa0f86938b93bp-2, -0x1.3d8ab9e27f6ccp-2, 0x1.90a252ae55e87p-3, -0x1.bde8380653f18p-4, 0x1.8b3390113199ap-5, -0x1.f90beb4bd4e87p-7, 0x1.964f3efd90569p-9, -0x1.3237d4224cacep-12}},
    {{-0x1.0645b13dfd885p+3, -0x1.003ffff7fd520p+0, 0x1.007fffdf92895p-1, -0x1.5655549c76a12p-2, 0x1.00ffef643fae0p-2, -0x1.9b95d0b6cb325p-3, 0x1.570f4501eee1fp-3, -0x1.2364ffd5e605bp-3, 0x1.d668650fd1205p-4, -0x1.0bcdf05a05eefp-4}},
    {{-0x1.0645b13e72d66p+3, -0x1.003fff25550c1p+0, 0x1.007fd56f6e7c1p-1, -0x1.56504575c8178p-2, 0x1.00cd3f898bff1p-2, -0x1.98dbbdd415318p-3, 0x1.49fe7e01a3d34p-3, -0x1.f1982a1b12ae4p-4, 0x1.28c26d6589260p-4, -0x1.88c68c462903dp-6}},
    {{-0x1.0645b162b4ce1p+3, -0x1.003fd97988328p+0, 0x1.007b7064b9426p-1, -0x1.5602c70d63edbp-2, 0x1.fe1fe41953b58p-3, -0x1.8b5388ae5d250p-3, 0x1.2651bbb09dd31p-3, -0x1.768cd55b8351dp-4, 0x1.55835fcadcc37p-5, -0x1.3e96d4f9477c8p-7}},
    {{-0x1.0645b3249d331p+3, -0x1.003e935c18216p+0, 0x1.00610124a9157p-1, -0x1.54c0973b26567p-2, 0x1.f431364e244c5p-3, -0x1.710041636eb9dp-3, 0x1.eee2e5f7aaf66p-4, -0x1.0a63d25d195e1p-4, 0x1.85953e584d23ep-6, -0x1.189b13bada6b0p-8}},
    {{-0x1.0645bc6c28065p+3, -0x1.003971ae379eap+0, 0x1.000ff622aa73ap-1, -0x1.51d2dfc632278p-2, 0x1.e2b3b1484e7ebp-3, -0x1.4e07f7d0c355bp-3, 0x1.9140fbfde6d47p-4, -0x1.72e7f7aedf5a5p-5, 0x1.c33f0eac97272p-7, -0x1.08b22237abc08p-9}},
    {{-0x1.0645dad30c9bbp+3, -0x1.002be445e4ea7p+0, 0x1.fec7522569dddp-2, -0x1.4cd18c83c2ae7p-2, 0x1.cab53605f20fdp-3, -0x1.2795384d0aabep-3, 0x1.3ee0a973c28bep-4, -0x1.0120f5595ccd3p-5, 0x1.0b657085a43a7p-7, -0x1.0898101684faep-10}},
    {{-0x1.064624e3c550bp+3, -0x1.001045cd4ec59p+0, 0x1.fc7c4de6b65b9p-2, -0x1.45b2942c4a090p-2, 0x1.ae387a82c6c2ep-3, -0x1.0185969064bdfp-3, 0x1.f5e2b0059b883p-5, -0x1.660df2a770aa8p-6, 0x1.44e1249b8aa99p-8, -0x1.15f6085bc387bp-11}},
    {{-0x1.0646b8f95014ep+3, -0x1.ffc1ae785c610p-1, 0x1.f91acd09bb232p-2, -0x1.3cb02a63ff95cp-2, 0x1.8f4f051e98b52p-3, -0x1.bc3c545e8d33fp-4, 0x1.8999efe51c186p-5, -0x1.f6e5fa2f0c2b0p-7, 0x1.948607167d946p-9, -0x1.30d74687d4e27p-12}},
    {{-0x1.1c77f43cad04ep+3, -0x1.001ffffeffa96p+0, 0x1.003ffffb9eae1p-1, -0x1.55d554fa5666bp-2, 0x1.007ff0035d0f2p-2, -0x1.9a95d6b2b6152p-3, 0x1.560f99b8be440p-3, -0x1.2268c54c64fccp-3, 0x1.d49eb382db521p-4, -0x1.0ab4487b0b669p-4}},
    {{-0x1.1c77f43d21d64p+3, -0x1.001fff2d37a19p+0, 0x1.003fd5b8c15bfp-1, -0x1.55d04b3d69a2fp-2, 0x1.004d7698a74eap-2, -0x1.97deb543726f1p-3, 0x1.490d041cf4f6fp-3, -0x1.effcfe6ee481ep-4, 0x1.27b89f0ee560ep-4, -0x1.8754bb6afca80p-6}},
    {{-0x1.1c77f46140981p+3, -0x1.001fd9a61711ep+0, 0x1.003b74f855169p-1, -0x1.558318b808491p-2, 0x1.fd23bd5b41991p-3, -0x1.8a63d8de6ae05p-3, 0x1.2583951ba00ebp-3, -0x1.756c0062f9711p-4, 0x1.546b5c1fafe68p-5, -0x1.3d874dae5b4d5p-7}},
    {{-0x1.1c77f62197facp+3, -0x1.001e94abea84bp+0, 0x1.00211d61611a0p-1, -0x1.544209edf46a1p-2, 0x1.f33dfdf8da6adp-3, -0x1.70284c5b61bd7p-3, 0x1.ed9b507912b5fp-4, -0x1.09a4f873207bep-4, 0x1.846fb3612fa74p-6, -0x1.17c0a07ce02d0p-8}},
    {{-0x1.1c77ff61835bfp+3, -0x1.00197736df7b9p+0, 0x1.ffa0aa50ee95cp-2, -0x1.5156bd4fe70d5p-2, 0x1.e1ceef08769a1p-3, -0x1.4d4cf94005043p-3, 0x1.904714cd8579ep-4, -0x1.71f0ca783c715p-5, 0x1.c204f41363410p-7, -0x1.07f47c797e233p-9}},
    {{-0x1.1c781db147281p+3, -0x1.000bf4210ca2dp+0, 0x1.fe4916fb22c8fp-2, -0x1.4c593c6344e31p-2, 0x1.c9e2cba386b7bp-3, -0x1.26f7a7e51a863p-3, 0x1.3e25e4bff3a0ep-4, -0x1.008117bb380fbp-5, 0x1.0ab8988245ab7p-7, -0x1.07e879c35f2cdp-10}},
    {{-0x1.1c78678d920a2p+3, -0x1.ffe0d274711d0p-1, 0x1.fbffb31048a7ap-2, -0x1.453f5231e20a5p-2, 0x1.ad7a4e5ce2d64p-3, -0x1.010318c3e957dp-3, 0x1.f4cde381ebb87p-5, -0x1.653d8a26e3d7cp-6, 0x1.441d339593ec8p-8, -0x1.154a3f99c09bdp-11}},
    {{-0x1.1c78fb4123094p+3, -0x1.ff82342213dd9p-1, 0x1.f8a06fb7d12d0p-2, -0x1.3c42e1ec353c6p-2, 0x1.8ea55cea0ccb2p-3, -0x1.bb66607227f89p-4, 0x1.88cd1d96ff556p-5, -0x1.f5d2fe703bd3dp-7, 0x1.93a1685b047d9p-9, -0x1.3026fd8250ffap-12}},
    {{-0x1.32a8373b9de85p+3, -0x1.000fffffdfcefp+0, 0x1.001fffff1fa26p-1, -0x1.5595550631d34p-2, 0x1.003ff01e48115p-2, -0x1.9a15d91d404b9p-3, 0x1.558fc35002676p-3, -0x1.21eaa710ec14bp-3, 0x1.d3b9d89ed09f2p-4, -0x1.0a2773115627fp-4}},
    {{-0x1.32a8373c127b6p+3, -0x1.000fff2e87e76p+0, 0x1.001fd5d2e5eabp-1, -0x1.55904dfe2d44fp-2, 0x1.000d91ebdb81bp-2, -0x1.9760306bc3eddp-3, 0x1.4894467a4bff6p-3, -0x1.ef2f672fcf9ccp-4, 0x1.2733b6db9a134p-4, -0x1.869bd170503a9p-6}},
    {{-0x1.32a837601fa22p+3, -0x1.000fd9b9bda5bp+0, 0x1.001b7737a2f29p-1, -0x1.5543416aa3395p-2, 0x1.fca5a9976b523p-3, -0x1.89ec007669d8ep-3, 0x1.251c8149b91acp-3, -0x1.74db950c950cbp-4, 0x1.53df59626d35cp-5, -0x1.3cff891a7292cp-7}},
    {{-0x1.32a8391fae822p+3, -0x1.000e955133ec1p+0, 0x1.00012b7553634p-1, -0x1.5402c325b64aep-2, 0x1.f2c46171eadf2p-3, -0x1.6fbc516e0c6efp-3, 0x1.ecf784fc7e7cdp-4, -0x1.09458b02c9f32p-4, 0x1.83dced1953603p-6, -0x1.1753663e13506p-8}},
    {{-0x1.32a8425bca244p+3, -0x1.000979f896fc6p+0, 0x1.ff61094252ac1p-2, -0x1.5118abf50c4e2p-2, 0x1.e15c8d97e12d6p-3, -0x1.4cef79a5b61bfp-3, 0x1.8fca20b6f347dp-4, -0x1.7175335384da1p-5, 0x1.c167e60cbd684p-7, -0x1.0795a9250a331p-9}},
    {{-0x1.32a8609ffd7c1p+3, -0x1.fff7f81816335p-1, 0x1.fe09f95249fa3p-2, -0x1.4c1d1435eeaf5p-2, 0x1.c979962e3f18fp-3, -0x1.26a8df7373c62p-3, 0x1.3dc882133a1a2p-4, -0x1.0031289f2448fp-5, 0x1.0a622c287f27fp-7, -0x1.0790ae3cae140p-10}},
    {{-0x1.32a8aa621155dp+3, -0x1.ffc0f5dc47276p-1, 0x1.fbc16592522c7p-2, -0x1.4505b11a93cb2p-2, 0x1.ad1b3811dc80bp-3, -0x1.00c1d9b00e398p-3, 0x1.f4437cd3f64bdp-5, -0x1.64d5558eeeb6dp-6, 0x1.43bb3abbcf220p-8, -0x1.14f45ae9ced54p-11}},
    {{-0x1.32a93de4a5402p+3, -0x1.ff6276f1fc210p-1, 0x1.f86340fd43a05p-2, -0x1.3c0c3d9949f14p-2, 0x1.8e5088a248f4ap-3, -0x1.bafb663909e01p-4, 0x1.8866b42916f47p-5, -0x1.f549802b016a1p-7, 0x1.932f18a47e903p-9, -0x1.2fced8b8b0952p-12}},
    {{-0x1.48d77a3a96f78p+3, -0x1.0007fffffbd15p+0, 0x1.000fffff8fca9p-1, -0x1.55755507be41ep-2, 0x1.001ff0252b58cp-2, -0x1.99d5da401f24dp-3, 0x1.554fd8032ab4dp-3, -0x1.21ab97d4676a0p-3, 0x1.d3476ae93c7bdp-4, -0x1.09e1082d4e675p-4}},
    {{-0x1.48d77a3b0b6b6p+3, -0x1.0007ff2edbfa0p+0, 0x1.000fd5dea7e7bp-1, -0x1.55704f5a2eb81p-2, 0x1.ffdb3f1dd9475p-3, -0x1.9720edee07e75p-3, 0x1.4857e792f6a17p-3, -0x1.eec89b6339e2dp-4, 0x1.26f142a104ec4p-4, -0x1.863f5c416f8c7p-6}},
    {{-0x1.48d77a5f0fc49p+3, -0x1.0007d9c33ce47p+0, 0x1.000b7855fa304p-1, -0x1.552355bf9b0b0p-2, 0x1.fc669fa8eab77p-3, -0x1.89b014326e40ap-3, 0x1.24e8f74fdf5a5p-3, -0x1.74935f462a56bp-4, 0x1.539957e6e2e1ap-5, -0x1.3cbba6b2bd18dp-7}},
    {{-0x1.48d77c1e3a631p+3, -0x1.000695a384b6dp+0, 0x1.ffe264fbff421p-2, -0x1.53e31fbd63c7dp-2, 0x1.f2879322ed792p-3, -0x1.6f8653ea3c888p-3, 0x1.eca59f2694260p-4, -0x1.0915d43b3961cp-4, 0x1.839389dbe5b19p-6, -0x1.171cc90abb01ep-8}},
    {{-0x1.48d785586e24fp+3, -0x1.00017b591f3f2p+0, 0x1.ff4138b87839ep-2, -0x1.50f9a343a95fep-2, 0x1.e1235cd5845eap-3, -0x1.4cc0b9ce540ffp-3, 0x1.8f8ba69be3766p-4, -0x1.713767b002709p-5, 0x1.c1195ef23185ep-7, -0x1.07663f6c2a0f2p-9}},
    {{-0x1.48d7a396d9409p+3, -0x1.ffe800026f706p-1, 0x1.fdea6a7b677c7p-2, -0x1.4bff001ba1a2dp-2, 0x1.c944fb6b1b85ap-3, -0x1.26817b32ecf47p-3, 0x1.3d99d0b2866dep-4, -0x1.000931077d6d6p-5, 0x1.0a36f5f09cbd7p-7, -0x1.0764c86db5c67p-10}},
    {{-0x1.48d7ed4bd1926p+3, -0x1.ffb1078f8fd34p-1, 0x1.fba23ed0ff81dp-2, -0x1.44e8e08baa382p-2, 0x1.acebace558b9dp-3, -0x1.00a13a206ea71p-3, 0x1.f3fe496f7bfd8p-5, -0x1.64a13b3801b89p-6, 0x1.438a3e4417000p-8, -0x1.14c96887fb144p-11}},
    {{-0x1.48d880b5e6ec0p+3, -0x1.ff52985951f0ap-1, 0x1.f844a99dca428p-2, -0x1.3bf0eb6cf43b5p-2, 0x1.8e261e78b8cadp-3, -0x1.bac5e9141fb24p-4, 0x1.88337f694a0a0p-5, -0x1.f504c0fbada45p-7, 0x1.92f5f0be25b93p-9, -0x1.2fa2c64b073b5p-12}},
    {{-0x1.5f063d39910c1p+3, -0x1.0003ffffff518p+0, 0x1.0007ffff9dd98p-1, -0x1.55655507f864ap-2, 0x1.000ff027cad91p-2, -0x1.99b5dacf423bcp-3, 0x1.552fe259b04e3p-3, -0x1.218c10324cfbcp-3, 0x1.d30e340602c58p-4, -0x1.09bdd2b56681ep-4}},
    {{-0x1.5f063d3a05706p+3, -0x1.0003ff2efb826p+0, 0x1.0007d5e45ee20p-1, -0x1.55605007a37a5p-2, 0x1.ffbb4cbf48727p-3, -0x1.97014cacedb98p-3, 0x1.4839b81c8c6b8p-3, -0x1.ee9535774ee4ep-4, 0x1.26d0087f9d643p-4, -0x1.861121a3cf6c7p-6}},
    {{-0x1.5f063d5e0562dp+3, -0x1.0003d9c7f2037p+0, 0x1.000378e4fbde3p-1, -0x1.55135fe98c532p-2, 0x1.fc471ab017fe5p-3, -0x1.89921e0e71709p-3, 0x1.24cf3250d61d4p-3, -0x1.746f445f8ea24p-4, 0x1.5376572581615p-5, -0x1.3c99b57b2b0c3p-7}},
    {{-0x1.5f063f1cfde08p+3, -0x1.000295cca2a01p+0, 0x1.ffd26c0458502p-2, -0x1.53d34e08b42afp-2, 0x1.f2692bf9fe4d8p-3, -0x1.6f6b5526b0373p-3, 0x1.ec7cac38ab803p-4, -0x1.08fdf8d584cf0p-4, 0x1.836ed839ff8b9p-6, -0x1.17017a6e91206p-8}},
    {{-0x1.5f0648563db22p+3, -0x1.fffaf812b1e3bp-1, 0x1.ff315073397aep-2, -0x1.50ea1eea79486p-2, 0x1.e106c47313e5bp-3, -0x1.4ca959e15bed0p-3, 0x1.8f6c698c630e5p-4, -0x1.711881dc1ccabp-5, 0x1.c0f21b6204fe9p-7, -0x1.074e8a8de58f5p-9}},
    {{-0x1.5f066691c4af8p+3, -0x1.ffe003f78769bp-1, 0x1.fddaa30fa783fp-2, -0x1.4beff60e06edbp-2, 0x1.c92aae0879eb3p-3, -0x1.266dc911b343bp-3, 0x1.3d827800e1f05p-4, -0x1.ffea6a74ed27fp-6, 0x1.0a215ad34bdf2p-7, -0x1.074ed584c5e7fp-10}},
    {{-0x1.5f06b0402f3cdp+3, -0x1.ffa910691fdffp-1, 0x1.fb92ab700b476p-2, -0x1.44da7843cd2d8p-2, 0x1.acd3e74e36e1fp-3, -0x1.0090ea57e8b82p-3, 0x1.f3dbafbb8f250p-5, -0x1.64872e0b2d225p-6, 0x1.4371c006e06e8p-8, -0x1.14b3ef55d60afp-11}},
    {{-0x1.5f07439e054d5p+3, -0x1.ff4aa90ce90ffp-1, 0x1.f8355dedc7485p-2, -0x1.3be342566d698p-2, 0x1.8e10e9633b05fp-3, -0x1.baab2a809f5c1p-4, 0x1.8819e50848a60p-5, -0x1.f4e261626d2aep-7, 0x1.92d95cc996c67p-9, -0x1.2f8cbd1317948p-12}},
    {{-0x1.7534c0388b416p+3, -0x1.0001ffffffc18p+0, 0x1.0003ffff9fa0bp-1, -0x1.555d550803f4cp-2, 0x1.0007f02900570p-2, -0x1.99a5db168a43ap-3, 0x1.551fe7849153fp-3, -0x1.217c4c60c4d06p-3, 0x1.d2f1989358191p-4, -0x1.09ac37f8b6288p-4}},
    {{-0x1.7534c038ff9dep+3, -0x1.0001ff2f09f67p+0, 0x1.0003d5e7351efp-1, -0x1.5558505e4c5ddp-2, 0x1.ffab538fcbcd9p-3, -0x1.96f17c0c19242p-3, 0x1.482aa060ff690p-3, -0x1.ee7b8280a5764p-4, 0x1.26bf6b6e66115p-4, -0x1.85fa0454397bcp-6}},
    {{-0x1.7534c05cfd5cep+3, -0x1.0001d9ca4b42fp+0, 0x1.fffef258eeeecp-2, -0x1.550b64fe73a46p-2, 0x1.fc3758337c587p-3, -0x1.898322fc332e0p-3, 0x1.24c24fd10df9ep-3, -0x1.745d36ebd4085p-4, 0x1.5364d6c45d230p-5, -0x1.3c88bcdeeb293p-7}},
    {{-0x1.7534c21bdcca4p+3, -0x1.000095e130454p+0, 0x1.ffca6f887a71dp-2, -0x1.53cb652e4b925p-2, 0x1.f259f86558ac4p-3, -0x1.6f5dd5c4b5876p-3, 0x1.ec6832c158c6dp-4, -0x1.08f20b226d026p-4, 0x1.835c7f68a698fp-6, -0x1.16f3d3202c808p-8}},
    {{-0x1.7534cb54a2a3bp+3, -0x1.fff6f8c2e8fadp-1, 0x1.ff295c508feb3p-2, -0x1.50e25cbdd169cp-2, 0x1.e0f87841b36a2p-3, -0x1.4c9da9eab6fb5p-3, 0x1.8f5ccb0463cfdp-4, -0x1.71090ef1e56fcp-5, 0x1.c0de799991f0bp-7, -0x1.0742b01e88c73p-9}},
    {{-0x1.7534e98eb791ep+3, -0x1.ffdc05f210d1cp-1, 0x1.fdd2bf59bdb10p-2, -0x1.4be871072b0e4p-2, 0x1.c91d875707266p-3, -0x1.2663f000f7a4bp-3, 0x1.3d76cba7e65ffp-4, -0x1.ffd66ea7997a1p-6, 0x1.0a168d44777eap-7, -0x1.0743dc101f85cp-10}},
    {{-0x1.75353339db3cep+3, -0x1.ffa514d5e55d4p-1, 0x1.fb8ae1bf87ce0p-2, -0x1.44d3441fd1a0ep-2, 0x1.acc8048289f96p-3, -0x1.0088c2738efdap-3, 0x1.f3ca62e162c6bp-5, -0x1.647a27749717cp-6, 0x1.436580e819d93p-8, -0x1.14a932bc9c246p-11}},
    {{-0x1.7535c69191a8bp+3, -0x1.ff46b166b226bp-1, 0x1.f82db815bd025p-2, -0x1.3bdc6dcb1e827p-2, 0x1.8e064ed8656f0p-3, -0x1.ba9dcb36bdcb6p-4, 0x1.880d17d7a498bp-5, -0x1.f4d131959a1f1p-7, 0x1.92cb12cf22ff8p-9, -0x1.2f81b876fc649p-12}},
    {{-0x1.8b632337857abp+3, -0x1.0000ffffffcf9p+0, 0x1.0001ffff9fdc4p-1, -0x1.55595508078ccp-2, 0x1.0003f02997cdep-2, -0x1.999ddb3a2517ap-3, 0x1.5517ea19f59eap-3, -0x1.21746a77f15d2p-3, 0x1.d2e34ad9e10aep-4, -0x1.09a36a9a4670ap-4}},
    {{-0x1.8b632337f9d35p+3, -0x1.0000ff2f11068p+0, 0x1.0001d5e89f957p-1, -0x1.555450899e9ffp-2, 0x1.ffa356f806f41p-3, -0x1.96e993bba5ea1p-3, 0x1.482314832deb6p-3, -0x1.ee6ea9053a425p-4, 0x1.26b71ce5b9f70p-4, -0x1.85ee75ac55c8ep-6}},
    {{-0x1.8b63235bf678ap+3, -0x1.0000d9cb77b8bp+0, 0x1.fffaf2a069388p-2, -0x1.55076788e522bp-2, 0x1.fc2f76f5283cap-3, -0x1.897ba5730c11dp-3, 0x1.24bbde9121780p-3, -0x1.74543031e9242p-4, 0x1.535c1693bc94dp-5, -0x1.3c804090bc5cfp-7}},
    {{-0x1.8b63251ac95ddp+3, -0x1.ffff2bd6eddbdp-1, 0x1.ffc6714a8a360p-2, -0x1.53c770c1152cdp-2, 0x1.f2525e9b001a8p-3, -0x1.6f571613b19eep-3, 0x1.ec5df605a39ddp-4, -0x1.08ec1448d96c0p-4, 0x1.835352ffed645p-6, -0x1.16ecff78f03b2p-8}},
    {{-0x1.8b632e53523b3p+3, -0x1.fff4f91b0432fp-1, 0x1.ff25623f39dd7p-2, -0x1.50de7ba77b801p-2, 0x1.e0f15228fe24bp-3, -0x1.4c97d1ef5f66bp-3, 0x1.8f54fbc05c4f9p-4, -0x1.7101557cc131bp-5, 0x1.c0d4a8b54cd10p-7, -0x1.073cc2e6d3127p-9}},
    {{-0x1.8b634c8cae21cp+3, -0x1.ffda06ef55334p-1, 0x1.fdcecd7ec78cdp-2, -0x1.4be4ae83bb4dfp-2, 0x1.c916f3fe49853p-3, -0x1.265f037895fc9p-3, 0x1.3d70f57b636dcp-4, -0x1.ffcc70c0e6090p-6, 0x1.0a11267d07d06p-7, -0x1.073e5f55c6869p-10}},
    {{-0x1.8b6396362e5bap+3, -0x1.ffa3170c47cacp-1, 0x1.fb86fce744e5dp-2, -0x1.44cfaa0dd239bp-2, 0x1.acc2131cb005ap-3, -0x1.0084ae815f481p-3, 0x1.f3c1bc7445d99p-5, -0x1.6473a429469acp-6, 0x1.435f6158b1253p-8, -0x1.14a3d46ffa451p-11}},
    {{-0x1.8b64298ad4f52p+3, -0x1.ff44b5939662fp-1, 0x1.f829e529b6c64p-2, -0x1.3bd90385759f3p-2, 0x1.8e010192f7cd2p-3, -0x1.ba971b91c8d67p-4, 0x1.8806b13f4e26cp-5, -0x1.f4c899af2a3f8p-7, 0x1.92c3edd1e3928p-9, -0x1.2f7c3628ea613p-12}},
};
#if defined(__HIP_DEVICE_COMPILE__)
static __shared__ __attribute__((aligned(16))) phf_logphitab phf_lds_logphi[PHF_LOGPHI_TAB_N];
#define PHF_T_LOGPHI(j) phf_lds_logphi[j]
#define PHF_LOGPHI_TABLE_TO_LDS()                                                                                   \
  do {                                                                                                              \
    for (int phf_i_ = threadIdx.x; phf_i_ < PHF_LOGPHI_TAB_N * 10; phf_i_ += blockDim.x)                            \
      (&phf_lds_logphi[0].c[0])[phf_i_] = (&phf_t_logphi[0].c[0])[phf_i_];                                          \
    __syncthreads();                                                                                                \
  } while (0)
#else
#define PHF_T_LOGPHI(j) phf_t_logphi[j]
#define PHF_LOGPHI_TABLE_TO_LDS() do { } while (0)
#endif

/* y = -x/sqrt2 in [0, PHF_LOGPHI_Y_MAX): log Phi(x).  Any other y: some finite or NaN value, no out-of-table access. */
PHF_HD double phf_log_ndtr_tab(double x, double y) {
  const uint64_t vb = phf_bits(y + 1.0);
  const uint32_t jr = (uint32_t)(vb >> 49) - (0x3ffu << 3);           /* 8 E + the top three mantissa bits */
  const uint32_t j = jr < (uint32_t)PHF_LOGPHI_TAB_N ? jr : (uint32_t)(PHF_LOGPHI_TAB_N - 1);
  const double sft = phf_from_bits((vb & 0x000fffffffffffffull) | 0x3ff0000000000000ull) - 1.0;
  const phf_logphitab e = PHF_T_LOGPHI(j);
  double g = phf_fma(e.c[9], sft, e.c[8]);
  g = phf_fma(g, sft, e.c[7]);
  g = phf_fma(g, sft, e.c[6]);
  g = phf_fma(g, sft, e.c[5]);
  g = phf_fma(g, sft, e.c[4]);
  g = phf_fma(g, sft, e.c[3]);
  g = phf_fma(g, sft, e.c[2]);
  g = phf_fma(g, sft, e.c[1]);
  g = phf_fma(g, sft, e.c[0]);
  return phf_fma(-0.5 * x, x, g);
}

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

/* ------------------------------------------------------------------------------------------------ standard normal
 * One standard normal from one 32-bit word by a piecewise inverse CDF (the single-level sampler's proposals; the hierarchical
 * sampler keeps Box-Muller, its LDS is full): w = the low 31 bits, p = (w + 1/2) / 2^32 in (0, 1/2), |z| = -Phi^-1(p), the sign
 * is the top bit — so v and v ^ 2^31 give +-z: the proposal is exactly symmetric, which is all Metropolis needs, and |z| <= 6.34.
 * a = 2 w + 1 = 2^E m converts to a double exactly; its exponent and top mantissa bit select one of 64 intervals, on which
 * |z| = P(m - 1), degree 5 (max error 1.7e-7 over all intervals: tools/gen_math_coeffs.py normal — a proposal need not be more normal
 * than that) — 6 fp64 operations and three 16-byte LDS reads where Box-Muller spent a logarithm, a square root and half a
 * sine/cosine pair (~21).  The table (3 KB) goes to LDS with PHF_NORMAL_TABLE_TO_LDS() in the kernels that draw with it. */
#define PHF_NORMAL_TAB_N 64
typedef struct { double c[6]; } phf_normtab;
static const phf_normtab phf_t_normal[PHF_NORMAL_TAB_N] = {
    {{0x1.95a1194d4d6d3p+2, -0x1.3b9b4c25fb156p-3, 0x1.3368aac04ead2p-4, -0x1.867b02d30c867p-5, 0x1.d253900d225bfp-6, -0x1.4b2805f73d8e9p-7}},
    {{0x1.959dad4d74622p+2, -0x1.37a742beff387p-3, 0x1.1493d85521b8ap-4, -0x1.062ab62b937d1p-5, 0x1.6906ac6fba8e7p-7, -0x1.e5b848eb35130p-10}},
    {{0x1.8ebc94c10e998p+2, -0x1.40d048fbe0de4p-3, 0x1.383d3b76ff2d4p-4, -0x1.8c7c106522c03p-5, 0x1.d969c61a0a9cep-6, -0x1.5028a621e6d0fp-7}},
    {{0x1.8eb91b7e3baaap+2, -0x1.3cccf08a6f25cp-3, 0x1.18f1248186ef9p-4, -0x1.0a3c016600325p-5, 0x1.6e9557dce92d6p-7, -0x1.ed2a8d3954477p-10}},
    {{0x1.87ba784e37994p+2, -0x1.46487e5bcd01ep-3, 0x1.3d4cf9d533f46p-4, -0x1.92c50a0c323aap-5, 0x1.e0d3db2b42d39p-6, -0x1.55641d1c23cc4p-7}},
    {{0x1.87b6f12c19cfcp+2, -0x1.4235229320c7fp-3, 0x1.1d84232009183p-4, -0x1.0e7e7737f4138p-5, 0x1.7466a3aaeb832p-7, -0x1.f4f5b8237c502p-10}},
    {{0x1.80993f6ce224ap+2, -0x1.4c09b7fdae55fp-3, 0x1.429cb7f426e93p-4, -0x1.995ba8cf23060p-5, 0x1.e8986668e76f6p-6, -0x1.5adf02fc683b5p-7}},
    {{0x1.8095a9c2ed885p+2, -0x1.47e5967893047p-3, 0x1.225138b4939dbp-4, -0x1.12f60a49a8267p-5, 0x1.7a7fddcd72fd3p-7, -0x1.fd20d5c0df71bp-10}},
    {{0x1.7957437ee7b56p+2, -0x1.521a79f7b6ab9p-3, 0x1.4831d80a9d70cp-4, -0x1.a0464ca22ca09p-5, 0x1.f0bebcbbf07fdp-6, -0x1.609e73409fe04p-7}},
    {{0x1.79539e96f6c20p+2, -0x1.4de4c0a4decafp-3, 0x1.275d4da1cbbcep-4, -0x1.17a7217b22e11p-5, 0x1.80e6eefe3e557p-7, -0x1.02d9df6ddc42fp-9}},
    {{0x1.71f2b77fd133bp+2, -0x1.58821f24173b0p-3, 0x1.4e12629cf66dfp-4, -0x1.a78c1572ba8cep-5, 0x1.f94f0ce528a9bp-6, -0x1.66a8202010165p-7}},
    {{0x1.71ef029492a9bp+2, -0x1.5439ea7708b63p-3, 0x1.2cade28c30881p-4, -0x1.1c96a97dd50b1p-5, 0x1.87a271f624107p-7, -0x1.075b9bc1b82d0p-9}},
    {{0x1.6a69a2fb4d174p+2, -0x1.5f48fddc5821dp-3, 0x1.544520f5fd32ep-4, -0x1.af3500e87fdfap-5, 0x1.012940647d431p-5, -0x1.6d02696b7fe42p-7}},
    {{0x1.6a65dd3674a24p+2, -0x1.5aed56b7fb312p-3, 0x1.324928b12af59p-4, -0x1.21ca29c503a8ep-5, 0x1.8eb9cf032c45bp-7, -0x1.0c1a89ab18806p-9}},
    {{0x1.62b9dc22c267cp+2, -0x1.667894a80fec5p-3, 0x1.5ad1bce95e008p-4, -0x1.b74a0dee29f1ep-5, 0x1.05e9b28d9d6abp-5, -0x1.73b477c67b10cp-7}},
    {{0x1.62b6049af0753p+2, -0x1.62086df57cfc7p-3, 0x1.38361f200405ep-4, -0x1.2747dd917899bp-5, 0x1.96355cf9ec2e6p-7, -0x1.111c3066a1ce1p-9}},
    {{0x1.5ae100d053933p+2, -0x1.6e1bc0fb3357cp-3, 0x1.61c0e727cb74ap-4, -0x1.bfd5675e06793p-5, 0x1.0aeeac72b8a6cp-5, -0x1.7ac65d39ecf5fp-7}},
    {{0x1.5add1686a6023p+2, -0x1.6995f4dc46663p-3, 0x1.3e7cb604f0016p-4, -0x1.2d16d21258439p-5, 0x1.9e1e88c4b4a75p-7, -0x1.1666ce982055bp-9}},
    {{0x1.52dc6e382c531p+2, -0x1.763f02cc347dep-3, 0x1.691c85caff800p-4, -0x1.c8e2978fe987fp-5, 0x1.103ec364f3a8fp-5, -0x1.82413c6606c4fp-7}},
    {{0x1.52d87015663d4p+2, -0x1.71a24f557f6dap-3, 0x1.4525f98dfa5c9p-4, -0x1.333f0adb33cb1p-5, 0x1.a680052cd6b68p-7, -0x1.1c0179a3a4f2ep-9}},
    {{0x1.4aa936f71bba5p+2, -0x1.7ef0d0b6ce5bep-3, 0x1.70efed3d2fa63p-4, -0x1.d27ec705e6d2bp-5, 0x1.15e17537434b7p-5, -0x1.8a2f77f519d4fp-7}},
    {{0x1.4aa523c84e323p+2, -0x1.7a3bd4177bcbep-3, 0x1.4c3c46667ff56p-4, -0x1.39c9ae4c8a9c0p-5, 0x1.af6604f0dd4bdp-7, -0x1.21f443befaa2ap-9}},
    {{0x1.424417158f237p+2, -0x1.884201aa00abap-3, 0x1.794826482c345p-4, -0x1.dcb909208dbc3p-5, 0x1.1bdf529823e5ep-5, -0x1.929cec6aab3d3p-7}},
    {{0x1.423fed88271b7p+2, -0x1.837335a1b97e2p-3, 0x1.53cb8a633150bp-4, -0x1.40c13c086b76fp-5, 0x1.b8de81d8c26a0p-7, -0x1.28486a7a71080p-9}},
    {{0x1.39a96575920e7p+2, -0x1.924652f123f48p-3, 0x1.823444fd1f4e7p-4, -0x1.e7a2baa9927c2p-5, 0x1.22423319b7f9fp-5, -0x1.9b97370ca74edp-7}},
    {{0x1.39a52412a83e1p+2, -0x1.8d5c076b9442ep-3, 0x1.5be194ce2de79p-4, -0x1.4831d034ced45p-5, 0x1.c2f9944dd9e73p-7, -0x1.2f0890092c2d7p-9}},
    {{0x1.30d501f5e8f5fp+2, -0x1.9d1514029bc81p-3, 0x1.8bc5d55f042d1p-4, -0x1.f34ff73c5e791p-5, 0x1.2915759703362p-5, -0x1.a52e0d7fff132p-7}},
    {{0x1.30d0a71ca6e23p+2, -0x1.980d68a51911cp-3, 0x1.648e7ae77e67cp-4, -0x1.50297731ccf11p-5, 0x1.cdc9e01cce51cp-7, -0x1.3641024b129cep-9}},
    {{0x1.27c23f564fd6ep+2, -0x1.a8ca0548b8309p-3, 0x1.9611646a96434p-4, -0x1.ffd82c40422e6p-5, 0x1.306650857cb3dp-5, -0x1.af73aadc9cc5ep-7}},
    {{0x1.27bdc93530526p+2, -0x1.a3a2e1afd5436p-3, 0x1.6de516c30bc61p-4, -0x1.58b896a3be685p-5, 0x1.d9651c9353d87p-7, -0x1.3e001398aaf2ep-9}},
    {{0x1.1e6bc79126983p+2, -0x1.b5867cb57a86ap-3, 0x1.a12f2e71b18a6p-4, -0x1.06ab693c2115bp-4, 0x1.384436f757960p-5, -0x1.ba7d5896e351ap-7}},
    {{0x1.1e67341dc6e89p+2, -0x1.b03d870cacd03p-3, 0x1.77fba8e236a7dp-4, -0x1.61f2726497a44p-5, 0x1.e5e4c04d98d91p-7, -0x1.46568ab019d18p-9}},
    {{0x1.14cb78e0ac079p+2, -0x1.c372eb68c0c51p-3, 0x1.ad3bff1052097p-4, -0x1.0df62d338f42fp-4, 0x1.40c158c0845dfp-5, -0x1.c6641bd5dd073p-7}},
    {{0x1.14c6c5cdcbbb3p+2, -0x1.be057cebb3e65p-3, 0x1.82eca8026eea2p-4, -0x1.6bedd650dc2d4p-5, 0x1.f366dc01cc619p-7, -0x1.4f58310b403adp-9}},
    {{0x1.0ada38ed240b6p+2, -0x1.d2c0ecba8301cp-3, 0x1.ba5a53b2a9649p-4, -0x1.15dfae292626fp-4, 0x1.49f3468b03aa7p-5, -0x1.d34592e983b93p-7}},
    {{0x1.0ad5639e5200ep+2, -0x1.cd2c038b93964p-3, 0x1.8ed7cf015261cp-4, -0x1.76c5f140375f2p-5, 0x1.010799e49ed5ap-6, -0x1.591c899627b85p-9}},
    {{0x1.008fba8d1cbcbp+2, -0x1.e3ae1b89d8feep-3, 0x1.c8b3d850f06d8p-4, -0x1.1e7f11918124bp-4, 0x1.53f3c5fb98407p-5, -0x1.e14512f8dbe1fp-7}},
    {{0x1.008ac007647a3p+2, -0x1.ddee491777e89p-3, 0x1.9be37efb5f41ep-4, -0x1.829b72408f6eap-5, 0x1.0904562cea7d0p-6, -0x1.63bfbd8b21d83p-9}},
    {{0x1.ebc461b593fbdp+1, -0x1.f6880bf40dfebp-3, 0x1.d87b5f845e5a8p-4, -0x1.27f0594e53b38p-4, 0x1.5ee1e6c6d3938p-5, -0x1.f08d1c3f9e524p-7}},
    {{0x1.ebba1b62f05f7p+1, -0x1.f0995ee1c25b3p-3, 0x1.aa3e94b41810cp-4, -0x1.8f95ffff5a1afp-5, 0x1.11c3951199a6ap-6, -0x1.6f63d2a8f9fb4p-9}},
    {{0x1.d58bcf967004ep+1, -0x1.05d8ff7d1d07cp-2, 0x1.e9ef847580cb7p-4, -0x1.3255bb4a72668p-4, 0x1.6ae37062fd8a2p-5, -0x1.00a8a212ea758p-6}},
    {{0x1.d581300ef10acp+1, -0x1.02c7f3909fdb8p-2, 0x1.ba22dcde055ccp-4, -0x1.9de62c16f1019p-5, 0x1.1b65063a586d8p-6, -0x1.7c3245c2a9f83p-9}},
    {{0x1.be596c8e675b6p+1, -0x1.11d6a274a5552p-2, 0x1.fd5e375938362p-4, -0x1.3dd96ddadeb27p-4, 0x1.7826c9c3802e7p-5, -0x1.09e860a9d29bep-6}},
    {{0x1.be4e6a91a8bbbp+1, -0x1.0ea937be2affcp-2, 0x1.cbd863dbd76c0p-4, -0x1.adc811e9832acp-5, 0x1.260fc02463752p-6, -0x1.8a5e2fd549991p-9}},
    {{0x1.a60a6d9e1e9f1p+1, -0x1.1f92f6c78bb83p-2, 0x1.0994c5f5bcdd6p-3, -0x1.4ab0166edb355p-4, 0x1.86e58a6ef90fep-5, -0x1.142cec2078a0bp-6}},
    {{0x1.a5fefe481404dp+1, -0x1.1c460b55bc836p-2, 0x1.dfb9f2979fc0bp-4, -0x1.bf86e438bdd85p-5, 0x1.31f47f87253a1p-6, -0x1.9a272bb502093p-9}},
    {{0x1.8c734f45ed6aep+1, -0x1.2f8351507b92fp-2, 0x1.15e71b484472cp-3, -0x1.591c1a198becep-4, 0x1.9767f7a1bca49p-5, -0x1.1fa7404d2f614p-6}},
    {{0x1.8c6765a4ec183p+1, -0x1.2c132c42fbb31p-2, 0x1.f63b1cdcfcd56p-4, -0x1.d381c6c0f8e3dp-5, 0x1.3f50b7de8d150p-6, -0x1.abdd4867dde74p-9}},
    {{0x1.715c7b2d8fe33p+1, -0x1.424c7b62b1066p-2, 0x1.23f60ee003d30p-3, -0x1.697225e1150c0p-4, 0x1.aa09c9d4f4d8dp-5, -0x1.2c955f64e6832p-6}},
    {{0x1.715007b61aed8p+1, -0x1.3eb4a4b670395p-2, 0x1.07f81dabf0ee7p-3, -0x1.ea3270681aed5p-5, 0x1.4e72c5e20aef4p-6, -0x1.bfe670f45194bp-9}},
    {{0x1.547d16451e9cep+1, -0x1.58de40a94c3cap-2, 0x1.34286f25850a4p-3, -0x1.7c1f5b01744bep-4, 0x1.bf40b4c3ca2bap-5, -0x1.3b46abe25deadp-6}},
    {{0x1.54700616c0ebbp+1, -0x1.55194b917e87dp-2, 0x1.16cc101e8cbf8p-3, -0x1.021b172d281bdp-4, 0x1.5fbfab2290763p-6, -0x1.d6c5d23163d18p-9}},
    {{0x1.357291e029f9bp+1, -0x1.74a533bad91b3p-2, 0x1.4700d664160d5p-3, -0x1.91b194863cca1p-4, 0x1.d7a541b945a08p-5, -0x1.4c21d62b192ebp-6}},
    {{0x1.3564cdca55b52p+1, -0x1.70ac76a3e7bfep-2, 0x1.28126ec9f9c02p-3, -0x1.112cf1719f11ep-4, 0x1.73bac15dc1178p-6, -0x1.f125d943a3c36p-9}},
    {{0x1.13b2296715684p+1, -0x1.97eb398704eeep-2, 0x1.5d1c1265a4ad2p-3, -0x1.aae236fe37390p-4, 0x1.f3fe6f6d5bcc4p-5, -0x1.5facc10b34fccp-6}},
    {{0x1.13a39498da50fp+1, -0x1.93b66dabaf268p-2, 0x1.3c5b8d62aadf3p-3, -0x1.22d4c8d12ebcdp-4, 0x1.8b0f984e39747p-6, -0x1.07f29a9b40f2fp-8}},
    {{0x1.dcdbfc944eb1bp+0, -0x1.c6a4d7fea2c1ep-2, 0x1.770e9f1d8804ap-3, -0x1.c8a53e526e5dbp-4, 0x1.0aa7d3d63be5cp-4, -0x1.76966b71decdbp-6}},
    {{0x1.dcbce92e8107ep+0, -0x1.c2299a529a5c1p-2, 0x1.542b973bf5a1dp-3, -0x1.37c3673ffa78dp-4, 0x1.a69ce98b86539p-6, -0x1.1a1330397f3fbp-8}},
    {{0x1.88bc1d4550e15p+0, -0x1.042f860463847p-1, 0x1.94aaae5ff6d37p-3, -0x1.ec51565855da4p-4, 0x1.1e717cabce953p-4, -0x1.91c22ac040d24p-6}},
    {{0x1.889ac4b10ee8ap+0, -0x1.01c81ed5652bdp-1, 0x1.6f3f98ab31055p-3, -0x1.50f850198b03ep-4, 0x1.c77677342785bp-6, -0x1.2fb16e78af501p-8}},
    {{0x1.267d495d95f38p+0, -0x1.36e38e6ee5102p-1, 0x1.b1363adbbcadap-3, -0x1.0c6ff3b3cc2bcp-3, 0x1.35fce63729b18p-4, -0x1.b2691cbd1137bp-6}},
    {{0x1.26593b1e882cep+0, -0x1.344a46c85258bp-1, 0x1.88c56f267cd2ep-3, -0x1.7107c8f2e2784p-4, 0x1.ee597db6d5e13p-6, -0x1.4a5310aa0a7a7p-8}},
    {{0x1.5956b2b4fa470p-1, -0x1.92c938bdc840dp-1, 0x1.aa41293f2adcdp-3, -0x1.325999fb241f0p-3, 0x1.4d178625abeeap-4, -0x1.e3716ab3b4d4cp-6}},
    {{0x1.5910e54d44bd7p-1, -0x1.903c4310c32f9p-1, 0x1.81ede58f6400ep-3, -0x1.ba8b879322b83p-4, 0x1.1a67394d0a4a5p-5, -0x1.d28f57304e568p-8}},
};
#if defined(__HIP_DEVICE_COMPILE__)
static __shared__ __attribute__((aligned(16))) phf_normtab phf_lds_normal[PHF_NORMAL_TAB_N];
#define PHF_T_NORMAL(j) phf_lds_normal[j]
#define PHF_NORMAL_TABLE_TO_LDS()                                                                                   \
  do {                                                                                                              \
    for (int phf_i_ = threadIdx.x; phf_i_ < PHF_NORMAL_TAB_N * 6; phf_i_ += blockDim.x)                             \
      (&phf_lds_normal[0].c[0])[phf_i_] = (&phf_t_normal[0].c[0])[phf_i_];                                          \
    __syncthreads();                                                                                                \
  } while (0)
#else
#define PHF_T_NORMAL(j) phf_t_normal[j]
#define PHF_NORMAL_TABLE_TO_LDS() do { } while (0)
#endif

PHF_HD double phf_normal_u32(uint32_t v) {
  const uint64_t ab = phf_bits((double)((v << 1) | 1u));             /* a = 2 w + 1 in [1, 2^32): exact */
  const uint32_t hi = (uint32_t)(ab >> 32);
  /* 2 E + the top mantissa bit: 0..63; the bias leaves ahead of the shift (it has no bits below it), as in phf_log_pos_k */
  const int j = (int)((hi - ((uint32_t)(0x3ff << 1) << 19)) >> 19);
  const double sft = phf_from_bits((ab & 0x000fffffffffffffull) | 0x3ff0000000000000ull) - 1.0;   /* m - 1 in [0, 1) */
  const phf_normtab e = PHF_T_NORMAL(j);
  double z = phf_fma(e.c[5], sft, e.c[4]);
  z = phf_fma(z, sft, e.c[3]);
  z = phf_fma(z, sft, e.c[2]);
  z = phf_fma(z, sft, e.c[1]);
  z = phf_fma(z, sft, e.c[0]);
  /* |P| with the sign bit of v (one v_bfi_b32): exactly symmetric even where the polynomial's error crosses zero (p -> 1/2) */
  return phf_from_bits((phf_bits(z) & 0x7fffffffffffffffull) | ((uint64_t)(v & 0x80000000u) << 32));
}

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
