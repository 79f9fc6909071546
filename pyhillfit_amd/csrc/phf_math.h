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
  const int j = (int)(((uint32_t)(u >> 32) + 0x1000u) >> 13) - PHF_LOG_TAB_BASE;  /* nearest grid point: (bits + 2^44) >> 45 */
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

/* two at once, sharing one division for the two erfcx */
PHF_HD void phf_log_ndtr_nonpos_x2_kx(double x0, double x1, double* r0, double* r1, phf_ktab ke, int ke_in_vgpr, phf_ktab kl) {
  const double y0 = -x0 * PHF_INV_SQRT2, y1 = -x1 * PHF_INV_SQRT2;
  const double q0 = phf_erfcx_den(y0), q1 = phf_erfcx_den(y1);
  const double iq = phf_rcp(q0 * q1);
  double e0, e1;
  phf_erfcx_finish_x2_kx(y0, iq * q1, y1, iq * q0, ke, ke_in_vgpr, &e0, &e1);
  *r0 = phf_fma(-0.5 * x0, x0, phf_log_pos_k(0.5 * e0, kl));
  *r1 = phf_fma(-0.5 * x1, x1, phf_log_pos_k(0.5 * e1, kl));
}

PHF_HD void phf_log_ndtr_nonpos_x2_k(double x0, double x1, double* r0, double* r1, phf_ktab ke, phf_ktab kl) {
  phf_log_ndtr_nonpos_x2_kx(x0, x1, r0, r1, ke, 0, kl);
}

PHF_HD void phf_log_ndtr_nonpos_x2(double x0, double x1, double* r0, double* r1) {
  PHF_KFETCH(ke, phf_k_erfcx, 24);
  PHF_KFETCH_V(kl, phf_k_log, PHF_K_LOG_N);
  phf_log_ndtr_nonpos_x2_k(x0, x1, r0, r1, ke, kl);
}

/* one, with the log table from the caller */
PHF_HD double phf_log_ndtr_nonpos_kx(double x, phf_ktab ke, int ke_in_vgpr, phf_ktab kl) {
  const double yv = -x * PHF_INV_SQRT2;
  const double e = phf_erfcx_finish_kx(yv, phf_rcp(phf_erfcx_den(yv)), ke, ke_in_vgpr);
  return phf_fma(-0.5 * x, x, phf_log_pos_k(0.5 * e, kl));
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

/* ------------------------------------------------------------------------------------------------ standard normal
 * One standard normal from one 32-bit word by a piecewise inverse CDF (the single-level sampler's proposals; the hierarchical
 * sampler keeps Box-Muller, its LDS is full): w = the low 31 bits, p = (w + 1/2) / 2^32 in (0, 1/2), |z| = -Phi^-1(p), the sign
 * is the top bit — so v and v ^ 2^31 give +-z: the proposal is exactly symmetric, which is all Metropolis needs, and |z| <= 6.34.
 * a = 2 w + 1 = 2^E m converts to a double exactly; its exponent and two mantissa bits select one of 128 intervals, on which
 * |z| = P(m - 1), degree 5 (max error 4.4e-9 over all intervals: tools/gen_math_coeffs.py normal) — 6 fp64 operations and three
 * 16-byte LDS reads where Box-Muller spent a logarithm, a square root and half a sine/cosine pair (~21).  The table (6 KB) goes
 * to LDS with PHF_NORMAL_TABLE_TO_LDS() in the kernels that draw with it. */
#define PHF_NORMAL_TAB_N 128
typedef struct { double c[6]; } phf_normtab;
static const phf_normtab phf_t_normal[PHF_NORMAL_TAB_N] = {
    {{0x1.95a1198e14418p+2, -0x1.3b9fc1bb6c69dp-3, 0x1.34339dd030bf6p-4, -0x1.93529ba159b71p-5, 0x1.14a12e05662a6p-5, -0x1.0f87a210ea2d9p-6}},
    {{0x1.95a0ea99ed462p+2, -0x1.3b378d5c7917cp-3, 0x1.2e2ee0317a3efp-4, -0x1.64651173421bbp-5, 0x1.6692ddc48b0f5p-6, -0x1.8e2ff230dbf1cp-8}},
    {{0x1.959f3f3705db7p+2, -0x1.392b531249d00p-3, 0x1.1de00cce9c160p-4, -0x1.227a51efdc39dp-5, 0x1.bea74476fe4f6p-7, -0x1.59ceaad6c8bf8p-9}},
    {{0x1.959a43173ca3bp+2, -0x1.3507773278c54p-3, 0x1.07b9666c037dcp-4, -0x1.cda6d95bebc0cp-6, 0x1.1cf37cb5a91ffp-7, -0x1.52ad7da7e8973p-10}},
    {{0x1.8ebc9502cd7a4p+2, -0x1.40d4cfa4f89e1p-3, 0x1.390b37ac7a2e8p-4, -0x1.9984d4d0849e0p-5, 0x1.18d2af06be7ddp-5, -0x1.139d799332a2ep-6}},
    {{0x1.8ebc655a2d3e3p+2, -0x1.406b0ae695bb6p-3, 0x1.32ef5d3fb4d49p-4, -0x1.69e330dc6ee31p-5, 0x1.6c0b3c404b895p-6, -0x1.943ae460d0061p-8}},
    {{0x1.8ebab381c7c8bp+2, -0x1.3e56e4fcf3f57p-3, 0x1.22617adc92731p-4, -0x1.26f9a1c762b1ep-5, 0x1.c582add5dbf21p-7, -0x1.5f17d55f7c365p-9}},
    {{0x1.8eb5a3fa32222p+2, -0x1.3a22ebdfc4315p-3, 0x1.0be4a07da11c8p-4, -0x1.d4d543d1e74a1p-6, 0x1.2159d2d9336a5p-7, -0x1.57e3014eb2484p-10}},
    {{0x1.87ba7890f9ceep+2, -0x1.464d16df6892dp-3, 0x1.3e1e2289f4ba3p-4, -0x1.a0013696fdd19p-5, 0x1.1d35b0aa58d42p-5, -0x1.17e32c8f65e3fp-6}},
    {{0x1.87ba482ba2c5cp+2, -0x1.45e1af7b73392p-3, 0x1.37ea1d63d7223p-4, -0x1.6fa342e5866b4p-5, 0x1.71c498298afadp-6, -0x1.9a8d3ea1bb200p-8}},
    {{0x1.87b88f91bf93dp+2, -0x1.43c540c6d6f5bp-3, 0x1.271a466f12271p-4, -0x1.2baf308ca813ep-5, 0x1.ccb0143ca5837p-7, -0x1.649feb01386ecp-9}},
    {{0x1.87b36bbc993f8p+2, -0x1.3f806ba821cc5p-3, 0x1.10433bbdefad0p-4, -0x1.dc5ab67fe0a71p-6, 0x1.25f512826d1dcp-7, -0x1.5d56ea0004446p-10}},
    {{0x1.80993fb0b3d6cp+2, -0x1.4c0e633209b13p-3, 0x1.4371333d3a2ddp-4, -0x1.a6cda638eb47bp-5, 0x1.21ce142dba020p-5, -0x1.1c5c73aacb877p-6}},
    {{0x1.80990e85c4cbdp+2, -0x1.4ba1457795a46p-3, 0x1.3d23e086a6646p-4, -0x1.75aa8a0193539p-5, 0x1.77c411f7c2d2ep-6, -0x1.a12c9afcc8f5bp-8}},
    {{0x1.80974ed874a4fp+2, -0x1.497c29861e0bdp-3, 0x1.2c0ef58c2659bp-4, -0x1.309f56ed8acc4p-5, 0x1.d435fa28f1dfep-7, -0x1.6a6be4b03be3cp-9}},
    {{0x1.809215bdea211p+2, -0x1.4525aaa3db7e9p-3, 0x1.14d96e0715c4fp-4, -0x1.e43e332c9fc20p-6, 0x1.2ac975849e679p-7, -0x1.630e2e29270c6p-10}},
    {{0x1.795743c3d603bp+2, -0x1.521f38c41f661p-3, 0x1.4909cf05c29dbp-4, -0x1.adf0b4cbbc3afp-5, 0x1.26a02a3993665p-5, -0x1.210d7188f03e3p-6}},
    {{0x1.795711c9b7ef4p+2, -0x1.51b04f6f36c6dp-3, 0x1.42a1f4a9db3f9p-4, -0x1.7bfee269d75edp-5, 0x1.7e0f5e5542b72p-6, -0x1.a81f3479b0c9ap-8}},
    {{0x1.79554ab072446p+2, -0x1.4f8219b898ef4p-3, 0x1.314495dbb082fp-4, -0x1.35ceed43fd765p-5, 0x1.dc1b9f81fdf9bp-7, -0x1.70814b5e0e985p-9}},
    {{0x1.794ffb4498c7fp+2, -0x1.4b1912a344c3dp-3, 0x1.19abebd15b67dp-4, -0x1.ec878a724c0ffp-6, 0x1.2fdbb16521462p-7, -0x1.690e54da913c5p-10}},
    {{0x1.71f2b7c5ea520p+2, -0x1.5886f282ded58p-3, 0x1.4eee01c9318c5p-4, -0x1.b571b8f8217cfp-5, 0x1.2bb0c35501662p-5, -0x1.25fac250189a9p-6}},
    {{0x1.71f284f23b888p+2, -0x1.5816268e61b58p-3, 0x1.486a4bccc8fe0p-4, -0x1.82a6d93e52ff7p-5, 0x1.84acdc25f8303p-6, -0x1.af6bfee7e17f1p-8}},
    {{0x1.71f0b60d1bab6p+2, -0x1.55de618541356p-3, 0x1.36c0d198d324cp-4, -0x1.3b435ee0d8970p-5, 0x1.e4691defbf005p-7, -0x1.76e64d6950bb0p-9}},
    {{0x1.71eb4f2d9c2a3p+2, -0x1.5161e13060b5dp-3, 0x1.1ebffbcf209a8p-4, -0x1.f53f7b26938e3p-6, 0x1.353109fa113b6p-7, -0x1.6f5d8b74d199fp-10}},
    {{0x1.6a69a342a06e0p+2, -0x1.5f4de6dd1124ep-3, 0x1.552498997416bp-4, -0x1.bd58ed9258543p-5, 0x1.31054363428f8p-5, -0x1.2b298e05a7feap-6}},
    {{0x1.6a696f8a1cab4p+2, -0x1.5edb1f4da2322p-3, 0x1.4e83960c03a11p-4, -0x1.89a9c801bdd8dp-5, 0x1.8ba3aea79a607p-6, -0x1.b71ac30dea28ep-8}},
    {{0x1.6a67987101752p+2, -0x1.5c994b4cb2bbdp-3, 0x1.3c8a08919d414p-4, -0x1.4102c0fd968aep-5, 0x1.ed278a81460e0p-7, -0x1.7da1d80c731dcp-9}},
    {{0x1.6a6218e26868bp+2, -0x1.58084bd3a6053p-3, 0x1.241b8e57d657fp-4, -0x1.fe6fd7bc5f74dp-6, 0x1.3acf678b9144cp-7, -0x1.7602bf66b4f05p-10}},
    {{0x1.62b9dc6b60b9bp+2, -0x1.667d947223dfdp-3, 0x1.5bb54187003e3p-4, -0x1.c5af962285045p-5, 0x1.36a3b8da28e99p-5, -0x1.309f9e63fa1efp-6}},
    {{0x1.62b9a7c1c571dp+2, -0x1.6608b618d29aep-3, 0x1.54f560f9d5b0fp-4, -0x1.910ff56f895bdp-5, 0x1.92fbdc97b8bf4p-6, -0x1.bf34403b53865p-8}},
    {{0x1.62b7c8034dba9p+2, -0x1.63bc4824e0ed3p-3, 0x1.42a76e28d0e4dp-4, -0x1.4713ee2b6300fp-5, 0x1.f6611dd6e4454p-7, -0x1.84bbb5b238713p-9}},
    {{0x1.62b22e6dee3a3p+2, -0x1.5f15ac3583569p-3, 0x1.29c55987a5cb2p-4, -0x1.0411d97fcfc79p-5, 0x1.40bd7144d0991p-7, -0x1.7d05bce940a57p-10}},
    {{0x1.5ae1011a4f28ep+2, -0x1.6e20d8d0e2d6ap-3, 0x1.62a8b207d6364p-4, -0x1.ce802ac2a4a65p-5, 0x1.3c92f88da0f08p-5, -0x1.366378f583583p-6}},
    {{0x1.5ae0cb723b640p+2, -0x1.6da9c603f7216p-3, 0x1.5bc83d73bdc36p-4, -0x1.98e2bcef3c061p-5, 0x1.9abe758b2a328p-6, -0x1.c7c2547bdcae3p-8}},
    {{0x1.5adee29298602p+2, -0x1.6b52265a6238dp-3, 0x1.49212d916712dp-4, -0x1.4d7ea7500ac0bp-5, 0x1.0010b22b9af45p-6, -0x1.8c3cb25472870p-9}},
    {{0x1.5ad92d7eed065p+2, -0x1.6694b633a1f9fp-3, 0x1.2fc4fb3977bd2p-4, -0x1.0933caf5e31efp-5, 0x1.4702acf953a7cp-7, -0x1.846f53e5a444dp-10}},
    {{0x1.52dc6e83992dap+2, -0x1.7644340e16b73p-3, 0x1.6a08d59a26fa9p-4, -0x1.d7d68d1b0f81dp-5, 0x1.42dabf28e8f35p-5, -0x1.3c7c7e7840986p-6}},
    {{0x1.52dc37ce68086p+2, -0x1.75cacc560035fp-3, 0x1.6305ed963d83fp-4, -0x1.a12cbe42ab8cdp-5, 0x1.a2f5bef4cdd49p-6, -0x1.d0d02d0246241p-8}},
    {{0x1.52da4545fcfe4p+2, -0x1.736754bd28056p-3, 0x1.500095d51960cp-4, -0x1.544bbb8dc4ea2p-5, 0x1.053abc31562f6p-6, -0x1.942ec75a0f022p-9}},
    {{0x1.52d473186bf26p+2, -0x1.6e91baab7685bp-3, 0x1.36232255caad7p-4, -0x1.0ea4e06dfe8fcp-5, 0x1.4da7a58bb6fc9p-7, -0x1.8c49848496594p-10}},
    {{0x1.4aa937440fcf2p+2, -0x1.7ef61ce7ea822p-3, 0x1.71e106c723edfp-4, -0x1.e1c048ca0aa0bp-5, 0x1.4983d9bd6c76ep-5, -0x1.42f310d50a39dp-6}},
    {{0x1.4aa8ff71ac822p+2, -0x1.7e7a3c9dd9caep-3, 0x1.6ab99cec54b82p-4, -0x1.a9fa178405c59p-5, 0x1.abad6ac654a9bp-6, -0x1.da6a80d3b7368p-8}},
    {{0x1.4aa702ab712bdp+2, -0x1.7c0a366a74236p-3, 0x1.57504faeca5d9p-4, -0x1.5b8538d81483fp-5, 0x1.0ab624843de32p-6, -0x1.9c9d50d3ebce2p-9}},
    {{0x1.4aa1119f52088p+2, -0x1.771afaa673876p-3, 0x1.3ce9c161de3f1p-4, -0x1.146d3d7dec318p-5, 0x1.54b619a354ab6p-7, -0x1.949fb554b376cp-10}},
    {{0x1.4244176422a21p+2, -0x1.88476a74a2810p-3, 0x1.7a3e55535f0d6p-4, -0x1.ec4ce2324dcabp-5, 0x1.50985738923b7p-5, -0x1.49d0c15673eeap-6}},
    {{0x1.4243de62d4a81p+2, -0x1.87c8ea4fb1a5ap-3, 0x1.72f0259ed579ap-4, -0x1.b358ac27f933cp-5, 0x1.b4f2da2678847p-6, -0x1.e49fd85552174p-8}},
    {{0x1.4241d6ba6cdf2p+2, -0x1.854b8c0ea1103p-3, 0x1.5f1c9fe2817e4p-4, -0x1.6336a77f55d12p-5, 0x1.108b783f5f7ddp-6, -0x1.a5954e8579b8fp-9}},
    {{0x1.423bc4dc269f1p+2, -0x1.80410fdb5aee3p-3, 0x1.44244cd2fdff7p-4, -0x1.1a9639dfdfd32p-5, 0x1.5c3934de96012p-7, -0x1.9d7ef5883fb87p-10}},
    {{0x1.39a965c5dfb40p+2, -0x1.924bda2c954dap-3, 0x1.832fdd4f3af36p-4, -0x1.f78e379856222p-5, 0x1.5823c51edf1d7p-5, -0x1.5120894b3ce16p-6}},
    {{0x1.39a92b820d4b0p+2, -0x1.91ca8eb994e41p-3, 0x1.7bb8664a14ab7p-4, -0x1.bd587c8ee1a37p-5, 0x1.bed56f7a17d39p-6, -0x1.ef80e529cb880p-8}},
    {{0x1.39a71841850d8p+2, -0x1.8f3ef96987426p-3, 0x1.6773b9970b56ap-4, -0x1.6b6d53b2f83ccp-5, 0x1.16c48f501de7fp-6, -0x1.af25b3e426506p-9}},
    {{0x1.39a0e3678f039p+2, -0x1.8a177146380b4p-3, 0x1.4be008788e788p-4, -0x1.212a9dad9add3p-5, 0x1.643dd65c8b576p-7, -0x1.a6f64e8ffe48bp-10}},
    {{0x1.30d502480e701p+2, -0x1.9d1abbb9f8380p-3, 0x1.8cc733f3870e7p-4, -0x1.01cc7cdb4f1f7p-4, 0x1.60337aa96342ap-5, -0x1.58ef0fed41a22p-6}},
    {{0x1.30d4c6abf3339p+2, -0x1.9c9674b8e50a9p-3, 0x1.8523ad5660c7fp-4, -0x1.c80c12d35b425p-5, 0x1.c966f404560d9p-6, -0x1.fb20eee0c54bdp-8}},
    {{0x1.30d2a70909702p+2, -0x1.99fbb0794f297p-3, 0x1.706625607462cp-4, -0x1.7438a8f288e36p-5, 0x1.1d6cce870d044p-6, -0x1.b95fcb39acb88p-9}},
    {{0x1.30cc4ccb8ba0fp+2, -0x1.94b51d2ea7f4ap-3, 0x1.542c686b16a5cp-4, -0x1.2836ec63e498fp-5, 0x1.6cd2e8631b911p-7, -0x1.b117296475aa1p-10}},
    {{0x1.27c23faa6e3aep+2, -0x1.a8cfcfc31f088p-3, 0x1.9718f0e7b0e44p-4, -0x1.0842a153c20e2p-4, 0x1.68d6f6662f67dp-5, -0x1.614b0150fbb9cp-6}},
    {{0x1.27c2029dc2f48p+2, -0x1.a8485763dbe95p-3, 0x1.8f46405beb774p-4, -0x1.d3890b1441db1p-5, 0x1.d4bc16c899116p-6, -0x1.03cb2d38e38a2p-7}},
    {{0x1.27bfd5b6a75e1p+2, -0x1.a59d4f775eaf6p-3, 0x1.7a07434833b62p-4, -0x1.7daaa4ba56817p-5, 0x1.24917a37f8622p-6, -0x1.c457b174bd91cp-9}},
    {{0x1.27b9536563205p+2, -0x1.a03575b7fca71p-3, 0x1.5d1b8b76e8db1p-4, -0x1.2fc9c31359dacp-5, 0x1.7609ce28da071p-7, -0x1.bbf5cd3d4eb8bp-10}},
    {{0x1.1e6bc7e762f2dp+2, -0x1.b58c6c7ff7466p-3, 0x1.a23d5ceb0ecefp-4, -0x1.0f37aa3116dfbp-4, 0x1.722053e1faaa3p-5, -0x1.6a457b642f73dp-6}},
    {{0x1.1e6b894ef2e3ap+2, -0x1.b5018676059f5p-3, 0x1.9a38087005d65p-4, -0x1.dfe8bfa129739p-5, 0x1.e0ed0c47b95adp-6, -0x1.0a7daa41139f4p-7}},
    {{0x1.1e694e2652d78p+2, -0x1.b245046c38d80p-3, 0x1.846df060f3a89p-4, -0x1.87d867907a628p-5, 0x1.2c421e6f09986p-6, -0x1.d024f217dda55p-9}},
    {{0x1.1e62a0bc62921p+2, -0x1.acb962aa77f00p-3, 0x1.66c2d71cf5dc3p-4, -0x1.37f44fad1fbe9p-5, 0x1.7ff6ee86b8e03p-7, -0x1.c7a9ff709ef33p-10}},
    {{0x1.14cb7939300c4p+2, -0x1.c37903625ce85p-3, 0x1.ae51522c35df0p-4, -0x1.16bc45ab10e08p-4, 0x1.7c24e0d17a8c4p-5, -0x1.73f297c129c0ap-6}},
    {{0x1.14cb38f656cd1p+2, -0x1.c2ea6bc4c76cdp-3, 0x1.a6156f816eb8fp-4, -0x1.ed492474c25a8p-5, 0x1.ee165961c15dfp-6, -0x1.11b754a1e747ap-7}},
    {{0x1.14c8ee6e8a3b8p+2, -0x1.c01b1191d2eadp-3, 0x1.8fb55b9fe6eb6p-4, -0x1.92daee2b27aacp-5, 0x1.3491138b520e8p-6, -0x1.dce34d589cb2fp-9}},
    {{0x1.14c2128363e50p+2, -0x1.ba68d38ff04fdp-3, 0x1.713bc047c81d1p-4, -0x1.40cae9811878cp-5, 0x1.8ab264c372736p-7, -0x1.d44fcef5bdbc6p-10}},
    {{0x1.0ada39481eebap+2, -0x1.d2c73023187a8p-3, 0x1.bb775f4c4915ap-4, -0x1.1ee44b2c75cf5p-4, 0x1.86fddbe892b49p-5, -0x1.7e6a1b8c6c972p-6}},
    {{0x1.0ad9f7382874bp+2, -0x1.d2349a0a53eeap-3, 0x1.b3007f8f17c60p-4, -0x1.fbcde1a8f141fp-5, 0x1.fc59d76ba8c28p-6, -0x1.198a6c54f3a18p-7}},
    {{0x1.0ad79c0d409fcp+2, -0x1.cf50dac042ff3p-3, 0x1.9bfe1a343e81cp-4, -0x1.9ed0000c577d8p-5, 0x1.3d942893af14bp-6, -0x1.eab3b634522d2p-9}},
    {{0x1.0ad08dc236972p+2, -0x1.c974ca37ed590p-3, 0x1.7ca4d01871434p-4, -0x1.4a65d60d9c7e5p-5, 0x1.9658e4166927dp-7, -0x1.e20899e873356p-10}},
    {{0x1.008fbaeac432bp+2, -0x1.e3b48e146e8f4p-3, 0x1.c9d944bbc4c48p-4, -0x1.27c785da28c6bp-4, 0x1.92c96b64efa0cp-5, -0x1.89c85a0361b2ep-6}},
    {{0x1.008f76e626796p+2, -0x1.e31da1e33de73p-3, 0x1.c1225b218fd37p-4, -0x1.05d0e14d08c79p-4, 0x1.05f00270413b6p-5, -0x1.220cbda03677fp-7}},
    {{0x1.008d09a68b95dp+2, -0x1.e023b8a78d0c9p-3, 0x1.a96f91fb1149fp-4, -0x1.abdb661aca18bp-5, 0x1.4765802d746a5p-6, -0x1.f9bd9b7d22aa2p-9}},
    {{0x1.0085c48fc5385p+2, -0x1.da1a2a2f1c407p-3, 0x1.8922fa44898cep-4, -0x1.54e249947f029p-5, 0x1.a30cdf5beabacp-7, -0x1.f0fc60ea8c7dbp-10}},
    {{0x1.ebc46276b7335p+1, -0x1.f68eb1ddcba88p-3, 0x1.d9a9eda073442p-4, -0x1.3182be496a255p-4, 0x1.9fabdeca6f163p-5, -0x1.962f5b46c9b3bp-6}},
    {{0x1.ebc3d6297c7bep+1, -0x1.f5f30b1bad300p-3, 0x1.d0ad2fd6b96acp-4, -0x1.0e7c4c07b30e4p-4, 0x1.0e6cdeda742fdp-5, -0x1.2b5889f00debap-7}},
    {{0x1.ebbed42fd84eep+1, -0x1.f2e0efd047d73p-3, 0x1.b839d9c72899ap-4, -0x1.ba2883254f8b8p-5, 0x1.5224b2364b9d4p-6, -0x1.05184b2e30d58p-8}},
    {{0x1.ebafd23ee6e3dp+1, -0x1.eca5aad77cef0p-3, 0x1.96e36324ef7e4p-4, -0x1.6063b91a2781fp-5, 0x1.b0f80d678adb1p-7, -0x1.00adc20fa18d6p-9}},
    {{0x1.d58bd05df59dbp+1, -0x1.05dc6e93b219fp-2, 0x1.eb2812d5e616cp-4, -0x1.3c391ac5e39c8p-4, 0x1.add15725e3fb6p-5, -0x1.a3c85feb0981ep-6}},
    {{0x1.d58b3f63fa645p+1, -0x1.058c03e4f67fep-2, 0x1.e1ded10af6055p-4, -0x1.1808dc0d6e8b8p-4, 0x1.17c140a017a8fp-5, -0x1.358dbae931193p-7}},
    {{0x1.d58612198df87p+1, -0x1.03f5b057b2d15p-2, 0x1.c8983bff8006cp-4, -0x1.c9ec72d5a02f7p-5, 0x1.5df84b53a44e7p-6, -0x1.0e2351c0267a1p-8}},
    {{0x1.d5768c9b8eae4p+1, -0x1.00bcc0813d62ep-2, 0x1.a61dbfbc1ce4ap-4, -0x1.6d159c5288953p-5, 0x1.c04d6c5913cb5p-7, -0x1.09b088623aea4p-9}},
    {{0x1.be596d5cf4928p+1, -0x1.11da30851d0b6p-2, 0x1.fea1c8d5d0ec0p-4, -0x1.4815f72ea4b9fp-4, 0x1.bd6ffc6828cafp-5, -0x1.b2c5e3e39bbd1p-6}},
    {{0x1.be58d73c452c5p+1, -0x1.1186eafef51b2p-2, 0x1.f5043ab0255c3p-4, -0x1.229dac0396585p-4, 0x1.2211bdac7b657p-5, -0x1.40d37ba1ff25dp-7}},
    {{0x1.be537a2757880p+1, -0x1.0fe1f2a781574p-2, 0x1.dad4997b7524bp-4, -0x1.db68e2b43cd56p-5, 0x1.6b0fce45ab78ap-6, -0x1.18238dc68644dp-8}},
    {{0x1.be4363568f590p+1, -0x1.0c8ad9fbca41cp-2, 0x1.b7178a7352f59p-4, -0x1.7b2dcaa24191fp-5, 0x1.d14bf49d54696p-7, -0x1.13aaeb4d57952p-9}},
    {{0x1.a60a6e7475412p+1, -0x1.1f96a727f246bp-2, 0x1.0a3ca80cc14cep-3, -0x1.554f6307f57bcp-4, 0x1.cecaf672b10bap-5, -0x1.c36651bf79046p-6}},
    {{0x1.a609d29de01e2p+1, -0x1.1f40377f65db7p-2, 0x1.053f296c15994p-3, -0x1.2e6b8abc6af72p-4, 0x1.2d8bce45cbb32p-5, -0x1.4d5a5f9472a8dp-7}},
    {{0x1.a604407a72ff4p+1, -0x1.1d8afd91530afp-2, 0x1.ef4c00127b603p-4, -0x1.eeefee0bcd45dp-5, 0x1.79a669a9e5b55p-6, -0x1.23455aa7e6114p-8}},
    {{0x1.a5f3881ab4798p+1, -0x1.1a125dd3d8671p-2, 0x1.ca285b8beb7a8p-4, -0x1.8aefaffdef59ap-5, 0x1.e442406ef1ec5p-7, -0x1.1eca49a134f96p-9}},
    {{0x1.8c735024f2e0cp+1, -0x1.2f8727f3cf328p-2, 0x1.1695ca66bf3d9p-3, -0x1.642986d34cd78p-4, 0x1.e23676d7f8b8cp-5, -0x1.d5f7ab7a13c11p-6}},
    {{0x1.8c72adeec0d22p+1, -0x1.2f2d303131d9bp-2, 0x1.11642aa88f0f2p-3, -0x1.3bb012b7a504fp-4, 0x1.3a68962346e54p-5, -0x1.5b5f4f28ebdf3p-7}},
    {{0x1.8c6ce07ba99a9p+1, -0x1.2d65ca6382119p-2, 0x1.033a6df43cea0p-3, -0x1.0274b0f286de2p-4, 0x1.8a06a95781c9ap-6, -0x1.2fc07d554aa08p-8}},
    {{0x1.8c5b7339d465fp+1, -0x1.29c7a14f35e09p-2, 0x1.dfbfc5fcfc334p-4, -0x1.9cb0ade62f3a8p-5, 0x1.f993883a41e09p-7, -0x1.2b47d0b1b43afp-9}},
    {{0x1.715c7c16567bfp+1, -0x1.42507d01d3016p-2, 0x1.24ac61df41040p-3, -0x1.74fb48bedb4b9p-4, 0x1.f81d3e292a62fp-5, -0x1.eadc89d08b6bbp-6}},
    {{0x1.715bd2b553423p+1, -0x1.41f28ca6acb42p-2, 0x1.1f4026933df94p-3, -0x1.4ab9e498af632p-4, 0x1.48f0b437404d3p-5, -0x1.6b2f878fdf37cp-7}},
    {{0x1.7155c270eb5b3p+1, -0x1.4016ae9c6d158p-2, 0x1.1073909dc7d13p-3, -0x1.0eecf85edc4c3p-4, 0x1.9c8f84d8ffdbap-6, -0x1.3ddbe09cf0a4dp-8}},
    {{0x1.7143890e3fc45p+1, -0x1.3c4e29cbd938ap-2, 0x1.f86d0db7df36ep-4, -0x1.b0de18814ec53p-5, 0x1.08df3b7c200e9p-6, -0x1.396c61236beb7p-9}},
    {{0x1.547d1738f1e96p+1, -0x1.58e272f9795fdp-2, 0x1.34e769b72fc0fp-3, -0x1.88349fcd30139p-4, 0x1.0884156a4de26p-4, -0x1.01497ad08b873p-5}},
    {{0x1.547c65b724e04p+1, -0x1.588001df5165cp-2, 0x1.2f38bc0f42577p-3, -0x1.5bee68b066e29p-4, 0x1.598179d114fa5p-5, -0x1.7d2e15b61ccc8p-7}},
    {{0x1.547609857a6c7p+1, -0x1.568ce1aeb61a0p-2, 0x1.1fb3214c4e517p-3, -0x1.1d367e963a5ffp-4, 0x1.b1bb48ae89937p-6, -0x1.4df2b299f8c4ep-8}},
    {{0x1.5462e7bb67715p+1, -0x1.529423b0a0029p-2, 0x1.0a743fbc7c744p-3, -0x1.c805576b0fbc8p-5, 0x1.16b33e25d56e6p-6, -0x1.4995e23644b5fp-9}},
    {{0x1.357292e09f5b7p+1, -0x1.74a99db6a509ep-2, 0x1.47c9b60a12b51p-3, -0x1.9e670ea731380p-4, 0x1.16d43d90bfd04p-4, -0x1.0edee05239852p-5}},
    {{0x1.3571d810db0f8p+1, -0x1.7442052e981bfp-2, 0x1.41cefdaa1e025p-3, -0x1.6fd182ad17913p-4, 0x1.6c93f84dcba51p-5, -0x1.91db49bbb3b67p-7}},
    {{0x1.356b24c331db0p+1, -0x1.7234368552d26p-2, 0x1.3175282fa5f3ep-3, -0x1.2dc0d42c2787fp-4, 0x1.ca28e44306ac3p-6, -0x1.607b517a3404dp-8}},
    {{0x1.3556f7c5dcbc1p+1, -0x1.6e0407c6be3cbp-2, 0x1.1b0ddc6b1ef31p-3, -0x1.e2deb0dd65f01p-5, 0x1.26b02443c4dc1p-6, -0x1.5c3e6ee97c9cbp-9}},
    {{0x1.13b22a7622494p+1, -0x1.97efe3d04a1cdp-2, 0x1.5df05fd924be5p-3, -0x1.b850bb11a8b1ep-4, 0x1.2772a90a88a14p-4, -0x1.1e9883147697bp-5}},
    {{0x1.13b164e354efbp+1, -0x1.978255698a22ap-2, 0x1.579dbb17dd256p-3, -0x1.870f9d230af3cp-4, 0x1.82c6359e32ea4p-5, -0x1.a9de934c7a3ffp-7}},
    {{0x1.13aa4c96b6510p+1, -0x1.9555982a66973p-2, 0x1.464de258f1989p-3, -0x1.411de345a2465p-4, 0x1.e6a7e89063369p-6, -0x1.76105b236947fp-8}},
    {{0x1.1394e916c55dfp+1, -0x1.90e4fd2ea5c93p-2, 0x1.2e8e2b74161f4p-3, -0x1.012d9f870a887p-4, 0x1.3960d6860ca96p-6, -0x1.72059db2ea4e8p-9}},
    {{0x1.dcdbfed47d6d0p+0, -0x1.c6a9cd60b6c89p-2, 0x1.77f045721a1f7p-3, -0x1.d6ebd894f229dp-4, 0x1.3af610580db83p-4, -0x1.3101064838134p-5}},
    {{0x1.dcda5a82a1c7ap+0, -0x1.c63546abbb3bdp-2, 0x1.7136d53471961p-3, -0x1.a28bbb83d09d8p-4, 0x1.9ce60674696cdp-5, -0x1.c612b8a72f8bcp-7}},
    {{0x1.dccb3d131192fp+0, -0x1.c3e445c1b60b8p-2, 0x1.5ec6982dd84e9p-3, -0x1.580e06e8f305dp-4, 0x1.04231071351d8p-5, -0x1.8f7b9f9973f51p-8}},
    {{0x1.dc9d9c2dff856p+0, -0x1.bf27f460e58b8p-2, 0x1.4572214c8d3ffp-3, -0x1.13dcb501feaa7p-4, 0x1.4f7961e84a1eep-6, -0x1.8bbb72bd11b43p-9}},
    {{0x1.88bc1fadc1d1dp+0, -0x1.04322d101845fp-1, 0x1.959c18d1761e4p-3, -0x1.fb972d0d84299p-4, 0x1.521edd6cf8cfbp-4, -0x1.46ccba7400e5cp-5}},
    {{0x1.88ba5d94cc7f6p+0, -0x1.03f3ca8219704p-1, 0x1.8e6919b66d316p-3, -0x1.c3867392d1d3fp-4, 0x1.bbf7bca210740p-5, -0x1.e7939a7019970p-7}},
    {{0x1.88aa274b43f67p+0, -0x1.02b5ca1b2c46cp-1, 0x1.7aa2f198216f8p-3, -0x1.73a4e013ffe97p-4, 0x1.182a1164d0613p-5, -0x1.adc2fc15322cdp-8}},
    {{0x1.887922eb0c3fep+0, -0x1.002aa41373486p-1, 0x1.5f6d7d2c91613p-3, -0x1.2a656fcb70361p-4, 0x1.69d682c1fdde6p-6, -0x1.aa6f1c69a5fd0p-9}},
    {{0x1.267d4bf612409p+0, -0x1.36e66a695b88dp-1, 0x1.b23a75f8c9f8cp-3, -0x1.14ab2a56e3b4ap-3, 0x1.6db045c70243ep-4, -0x1.60ec374968867p-5}},
    {{0x1.267b664c4648ap+0, -0x1.36a31ba63064ap-1, 0x1.aa764e224df42p-3, -0x1.ecddbfccc6256p-4, 0x1.e0e614bc6710bp-5, -0x1.07ff8ad8775cfp-6}},
    {{0x1.2669e08420c49p+0, -0x1.354b6c3e2e00fp-1, 0x1.9517ae61898b4p-3, -0x1.968baf5849910p-4, 0x1.2fe75792349c0p-5, -0x1.d2b89dbe18dddp-8}},
    {{0x1.2634d38077555p+0, -0x1.328ab5da09f17p-1, 0x1.77a578c2fe7f6p-3, -0x1.4746fa57b3105p-4, 0x1.890223f97e316p-6, -0x1.d0c8b8cd2c01dp-9}},
    {{0x1.5956b84f073e7p-1, -0x1.92cc4e91aff80p-1, 0x1.ab59fa2fcf973p-3, -0x1.3b3bd220e7704p-3, 0x1.8937eee82009dp-4, -0x1.844145a20f41dp-5}},
    {{0x1.5952a76a2d221p-1, -0x1.92841ea5f28fbp-1, 0x1.a30389d55d5b0p-3, -0x1.1abc4ae46ba6bp-3, 0x1.0270d6ec0079fp-4, -0x1.2c74d7ec6925fp-6}},
    {{0x1.592ee34ead63fp-1, -0x1.912437023f679p-1, 0x1.8d0f1d2101bb3p-3, -0x1.dc7c0cd26dadcp-4, 0x1.4dcf9502fc6a9p-5, -0x1.27313253e40f7p-7}},
    {{0x1.58da7fade0131p-1, -0x1.8eec6703bde81p-1, 0x1.7506af3887551p-3, -0x1.9af78b95237c4p-4, 0x1.e7e5ff36ae1c7p-6, -0x1.8801800bb2926p-8}},
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
  const int j = (int)(hi >> 18) - (0x3ff << 2);                       /* 4 E + the top two mantissa bits: 0..127 */
  const double sft = phf_from_bits((ab & 0x000fffffffffffffull) | 0x3ff0000000000000ull) - 1.0;   /* m - 1 in [0, 1) */
  const phf_normtab e = PHF_T_NORMAL(j);
  double z = phf_fma(e.c[5], sft, e.c[4]);
  z = phf_fma(z, sft, e.c[3]);
  z = phf_fma(z, sft, e.c[2]);
  z = phf_fma(z, sft, e.c[1]);
  z = phf_fma(z, sft, e.c[0]);
  return phf_from_bits(phf_bits(z) | ((uint64_t)(v & 0x80000000u) << 32));   /* z > 0: the sign bit comes from v */
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
