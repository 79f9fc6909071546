/* phf_hier_model.h — the hierarchical log-target (python/PyHillFit.py:113-193) and the per-iteration draws of
 * the hierarchical sampler, shared by the gfx950 kernel and the host twin like phf_model.h.
 *
 *   theta = [alpha, beta, mu, s, pIC50_1, Hill_1, ..., pIC50_Ne, Hill_Ne, sigma]           (PyHillFit.py:178-181)
 *   log target = - sum_i [ n_i ln sigma + SSE_i/(2 sigma^2) + sum_j ln(Phi((100-p_ij)/sigma) - Phi((0-p_ij)/sigma)) ]  (:113-132, truncated Gaussian)
 *              + sum_i [ ln beta - beta ln alpha + (beta-1) ln Hill_i - 2 ln(1 + (Hill_i/alpha)^beta) ]              (:134-142, log-logistic)
 *              + sum_i [ -(pIC50_i-mu)/s - ln s - 2 ln(1 + exp(-(pIC50_i-mu)/s)) ]                                   (:144-154, logistic)
 *              + sum over (alpha, beta, mu, s, sigma) of the shifted-Gamma log-priors                                (:187, doseresponse.py:304-317)
 *   support: theta[:4] <= locs[:4], any Hill_i < 0, any pIC50_i < -2, sigma <= locs[4]  ->  -inf                       (:176,182)
 *
 * theta is read through a strided accessor (th[i*ts]) so the same code serves registers/LDS/global layouts.
 * Points are stored experiment by experiment: expt_start[i] .. expt_start[i+1]-1 are experiment i's points.    */
#ifndef PHF_HIER_MODEL_H
#define PHF_HIER_MODEL_H

#include "phf_math.h"
#include "phf_philox.h"

#define PHF_HIER_PIC50_LOWER (-2.0)  /* pic50_prior[0], PyHillFit.py:215 */

#ifndef PHF_HIER_PRIOR_DEFINED
#define PHF_HIER_PRIOR_DEFINED
typedef struct phf_hier_prior {   /* Gamma hyper-priors of (alpha, beta, mu, s, sigma): PyHillFit.py:301,363-364 */
  double shape_m1[5];             /* shapes - 1 */
  double inv_scale[5];            /* 1/scales */
  double loc[5];
} phf_hier_prior;
#endif

#if defined(__HIPCC__)
#define PHF_UNROLL _Pragma("unroll")
#else
#define PHF_UNROLL
#endif

/* capacity of the fixed-size work arrays (experiments per pair).  With a compile-time Ne every loop is unrolled and
 * only the first Ne entries exist (registers); the generic-Ne kernel and the host twin index them at run time
 * (the reference's synthetic data set has a pair with 50 experiments).                                          */
#define PHF_HIER_CAP 64
#define PHF_MAX_BATCH (2 + 9 + 2 * PHF_HIER_CAP)

/* v[i] <- 1/v[i] for i < n with ONE division (prefix products, then back-substitution): 3(n-1) multiplications.
 * n must be a compile-time constant at the call site (fully unrolled, static register indices).                 */
PHF_HD void phf_batch_recip(double* v, int n) {
  double pre[PHF_MAX_BATCH];
  if (n <= 0) return;
  pre[0] = v[0];
  PHF_UNROLL
  for (int i = 1; i < n; ++i) pre[i] = pre[i - 1] * v[i];
  double inv = phf_rcp(pre[n - 1]);
  PHF_UNROLL
  for (int i = n - 1; i > 0; --i) {
    const double vi = v[i];
    v[i] = inv * pre[i - 1];
    inv = inv * vi;
  }
  v[0] = inv;
}

/* Truncated-Gaussian term of a point (PyHillFit.py:121-125): the mass Phi(b) - Phi(a), a = (0-pred)/sigma <= 0 <= b = (100-pred)/sigma,
 * = 1 - [erfc(ya) + erfc(yb)]/2 with ya = -a/sqrt2, yb = b/sqrt2 >= 0: both tails from the erfc table of phf_math.h (absolute
 * accuracy below half an ulp of the 1 they are subtracted from; a tail with argument >= PHF_ERFC_CUT = 6 — more than 8.5 sigma
 * from its bound — is DEFINED as zero).  No division; the only branch is the one-lane kernels' skip of a pair's upper tails
 * (phf_trunc_mass_x2).  The per-Ne target multiplies the masses of a half's points and takes one logarithm
 * (phf_hier_target_half); the per-experiment form takes one per point.
 * (Rounds 1-3: erfcx(y) exp(-y^2) per tail, two tails per division, the upper tails of a pair of points skipped under a
 * wave-uniform branch when negligible on every lane — ~50 fp64 operations per tail against 15.)                              */
PHF_HD double phf_trunc_mass(double pred, double inv_s, phf_ktab kx) {
  const double a = -pred * inv_s, b = (PHF_K100(kx) - pred) * inv_s;
  const double ya = -a * PHF_INV_SQRT2, yb = b * PHF_INV_SQRT2;
  return phf_fma(-0.5, phf_erfc_tab(ya) + phf_erfc_tab(yb), 1.0);
}

/* two points: the product of their masses.  skip (a literal): the two UPPER tails — beyond the cut on every lane for two thirds of the
 * Crumb point pairs — are evaluated only if some lane needs one; phf_erfc_tab is 0 beyond the cut anyway, so skipping changes
 * nobody's value.  One-lane kernels skip (C4 14.58 -> 14.09 ms); the two-lane kernels, bound by the latency of one iteration, do
 * better without the branch (9.71 against 9.99 us per Ne = 6 iteration). */
PHF_HD double phf_trunc_mass_x2(double pred0, double pred1, double inv_s, phf_ktab kx, int skip) {
  const double ya0 = (pred0 * inv_s) * PHF_INV_SQRT2, ya1 = (pred1 * inv_s) * PHF_INV_SQRT2;
  const double yb0 = ((PHF_K100(kx) - pred0) * inv_s) * PHF_INV_SQRT2, yb1 = ((PHF_K100(kx) - pred1) * inv_s) * PHF_INV_SQRT2;
  double t0 = phf_erfc_tab(ya0), t1 = phf_erfc_tab(ya1);
  if (!skip || PHF_ANY_LANE((yb0 < PHF_ERFC_CUT) | (yb1 < PHF_ERFC_CUT))) { t0 += phf_erfc_tab(yb0); t1 += phf_erfc_tab(yb1); }
  return phf_fma(-0.5, t0, 1.0) * phf_fma(-0.5, t1, 1.0);
}

PHF_HD double phf_trunc_term(double pred, double inv_s, phf_ktab kx, phf_ktab kl) { return phf_log_fast_k(phf_trunc_mass(pred, inv_s, kx), kl); }

/* ---- the target as the sum of two HALVES ---------------------------------------------------------------------------
 * log target = P_0 + P_1 (or -inf outside the support), where each half is a fixed sequence of operations on its own share of
 * the work.  The halves exist so that TWO LANES can share one chain (hier_advance2_kernel: lane 2c computes P_0, lane 2c+1
 * computes P_1 of chain c, one cross-lane addition joins them); a kernel or the twin that gives a chain one lane evaluates
 * the two halves one after the other.  Same value either way, bit for bit — the addition commutes.
 *
 *   first batch of logarithms (9 + Ne of them), in this order:
 *       ln alpha | ln Hill_1 .. ln Hill_Ne | ln beta | ln(alpha-loc0) ln(beta-loc1) ln(mu-loc2) ln(s-loc3) ln(sigma-loc4) | ln s | ln sigma
 *     with weights
 *       -Ne beta |  beta - 1  (each)       |   Ne    |   shape_k - 1  (each: the Gamma priors, :187)                       | -Ne  | -n_pts
 *     half 0 takes the first ceil((9+Ne)/2) (always ln alpha and every ln Hill_i), half 1 the rest (a logarithm is a table
 *     lookup and a short polynomial, phf_math.h: no division); 1/sigma and 1/s share ONE division;
 *   half 0: + sum_k -(x_k - loc_k)/scale_k   (linear part of the Gamma priors)
 *           - 2 ln prod_i (1 + (Hill_i/alpha)^beta)                          (log-logistic, :134-142: the sum of Ne logarithms as one)
 *   half 1: + sum_i -(pIC50_i - mu)/s  - 2 ln prod_i (1 + exp(-(pIC50_i - mu)/s))  (logistic, :144-154; either product term by
 *           term if it is not below 2^1000)
 *   both:   - [ SSE_h / (2 sigma^2) + ln prod over the half's points of (Phi((100-p)/sigma) - Phi((0-p)/sigma)) ]   (:113-132; an upper
 *           tail beyond PHF_ERFC_CUT counts as zero; a product that underflows makes the half -inf)
 *     where experiment i's n points are split  first 2*floor((n+2)/4) -> half 0, the others -> half 1  (4 points: 2 + 2).
 *
 * `h` is a literal at the call site (twin, one-lane kernels: the selects below fold away) or the lane's parity (two-lane
 * kernel: each select is a v_cndmask pair; only ARGUMENTS and WEIGHTS are selected, never results of expensive work).
 * n_expts must be a compile-time constant at the call site when theta lives in registers.
 * fixed_n > 0: a SHAPE CODE, PHF_HIER_SHAPE(per, last) — every experiment has exactly `per` points (a multiple of 4), except that
 * the last one has `last` (1..3) if that is not 0 — all literals: the point loops unroll, an iteration of the sampler is
 * straight-line code (147 of the 210 Crumb pairs are 3 experiments x 4 points, 32 more are 4 + 4 + 4 + 1); 0: experiment i's
 * points are expt_start[i] .. expt_start[i+1]-1.  Same operations in the same order either way.                            */
#define PHF_HIER_SHAPE(per, last) ((per) | ((last) << 4))
#define PHF_PICK(h, x0, x1) ((h) ? (x1) : (x0))
/* a value the optimiser must treat as computed here: keeps a pick between two ARRAY ELEMENTS a select of two registers (left to
 * itself hipcc turns it into one load with a selected index, which puts the whole array in scratch memory)               */
#if defined(__HIP_DEVICE_COMPILE__)
#define PHF_OPAQUE(x) asm("" : "+v"(x))
#else
#define PHF_OPAQUE(x) ((void)0)
#endif

/* support (:176,182): the same for both halves */
PHF_HD int phf_hier_out_of_support(int n_expts, const double* th, int ts, const phf_hier_prior* pr) {
  const int dim = 5 + 2 * n_expts;
  const double alpha = th[0], beta = th[1 * ts], mu = th[2 * ts], s = th[3 * ts], sigma = th[(dim - 1) * ts];
  int bad = (alpha <= pr->loc[0]) | (beta <= pr->loc[1]) | (mu <= pr->loc[2]) | (s <= pr->loc[3]) | (sigma <= pr->loc[4]);
  PHF_UNROLL
  for (int i = 0; i < n_expts; ++i) bad |= (th[(5 + 2 * i) * ts] < 0.0) | (th[(4 + 2 * i) * ts] < PHF_HIER_PIC50_LOWER);
  return bad;
}

PHF_HD double phf_hier_target_half(int h, int n_expts, int fixed_n, const int* expt_start, const double* lc, const double* y,
                                   const double* th, int ts, const phf_hier_prior* pr, phf_ktab k_exp, phf_ktab k_log, int skip_tails) {
  const int dim = 5 + 2 * n_expts;
  const double alpha = th[0], beta = th[1 * ts], mu = th[2 * ts], s = th[3 * ts], sigma = th[(dim - 1) * ts];
  /* fixed_n is a SHAPE CODE (PHF_HIER_SHAPE): low 4 bits = points of every experiment (a multiple of 4), the bits above = points of the
   * LAST experiment if it differs (1..3; then h must be a literal: the halves' shares of that experiment are unequal) */
  const int fper = fixed_n & 15, flast = fixed_n >> 4;
#define PHF_NI_(i_) ((flast && (i_) == n_expts - 1) ? flast : fper)                       /* points of experiment i_ */
#define PHF_NF_(i_) ((2 * ((PHF_NI_(i_) + 2) / 4) < PHF_NI_(i_)) ? 2 * ((PHF_NI_(i_) + 2) / 4) : PHF_NI_(i_))   /* of them, half 0's */
  const int n_pts = fixed_n ? (n_expts - 1) * fper + PHF_NI_(n_expts - 1) : expt_start[n_expts];
  const int nl = 9 + n_expts, n0 = (nl + 1) / 2;
  /* fixed shapes (straight-line bodies): this half's points are read HERE, a few hundred instructions ahead of their use, so that a
   * wavefront that has its SIMD to itself does not wait out the LDS latency pair by pair inside the point loop */
  double plc[32], py[32];                                         /* this half's points, experiment after experiment */
  if (fixed_n) {
    int off = 0;
    PHF_UNROLL
    for (int i = 0; i < n_expts; ++i) {
      const int cnt = flast ? (h ? PHF_NI_(i) - PHF_NF_(i) : PHF_NF_(i)) : fper / 2;       /* equal halves when every experiment has fper points */
      const int j0 = i * fper + (h ? 1 : 0) * PHF_NF_(i);
      PHF_UNROLL
      for (int p = 0; p < cnt; ++p) { plc[off + p] = lc[j0 + p]; py[off + p] = y[j0 + p]; }
      off += cnt;
    }
  }
  /* ---- this half's share of the first batch of logarithms; 1/sigma and 1/s: one division ---- */
  double ga[10 + PHF_HIER_CAP], gw[10 + PHF_HIER_CAP];            /* arguments and weights of all 9 + Ne (cheap), then the pick */
  double xl[5];
  ga[0] = alpha; gw[0] = -(double)n_expts * beta;
  PHF_UNROLL
  for (int i = 0; i < n_expts; ++i) { ga[1 + i] = th[(5 + 2 * i) * ts]; gw[1 + i] = beta - 1.0; }
  ga[n_expts + 1] = beta; gw[n_expts + 1] = (double)n_expts;
  {
    const double hv[5] = {alpha, beta, mu, s, sigma};
    PHF_UNROLL
    for (int k = 0; k < 5; ++k) { xl[k] = hv[k] - pr->loc[k]; ga[n_expts + 2 + k] = xl[k]; gw[n_expts + 2 + k] = pr->shape_m1[k]; }
  }
  ga[n_expts + 7] = s; gw[n_expts + 7] = -(double)n_expts;
  ga[n_expts + 8] = sigma; gw[n_expts + 8] = -(double)n_pts;
  ga[nl] = 1.0; gw[nl] = 0.0;                                     /* pads half 1 when 9 + Ne is odd: ln 1 = 0 */
  double rc[2] = {sigma, s};
  phf_batch_recip(rc, 2);
  const double inv_s = rc[0], inv_sc = rc[1];
  double lg[5 + PHF_HIER_CAP / 2 + 1];
  double part = 0.0;
  PHF_UNROLL
  for (int j = 0; j < n0; ++j) {
    double a0 = ga[j], a1 = ga[n0 + j], w0 = gw[j], w1 = gw[n0 + j];
    PHF_OPAQUE(a0); PHF_OPAQUE(a1); PHF_OPAQUE(w0); PHF_OPAQUE(w1);
    lg[j] = phf_log_fast_k(PHF_PICK(h, a0, a1), k_log);
    part = phf_fma(PHF_PICK(h, w0, w1), lg[j], part);
  }
  /* ---- linear terms: Gamma priors (half 0), logistic density of the pIC50_i (half 1) ---- */
  double lin0 = 0.0, lin1 = 0.0;
  PHF_UNROLL
  for (int k = 0; k < 5; ++k) lin0 = phf_fma(-xl[k], pr->inv_scale[k], lin0);
  double la[PHF_HIER_CAP];                                         /* 1 + (Hill_i/alpha)^beta (half 0), 1 + exp(-z_i) (half 1) */
  PHF_UNROLL
  for (int i = 0; i < n_expts; ++i) {
    const double z = (th[(4 + 2 * i) * ts] - mu) * inv_sc;
    lin1 -= z;
    const double e_arg = PHF_PICK(h, beta * (lg[1 + i] - lg[0]), -z);     /* half 0 holds ln alpha in lg[0], ln Hill_i in lg[1+i] */
    la[i] = 1.0 + phf_exp_fast_k(e_arg, k_exp);
  }
  part += PHF_PICK(h, lin0, lin1);
  {   /* the Ne deferred logarithms sum_i ln la_i as ONE logarithm of the product (every la_i >= 1: it cannot underflow); only if
       * the product is not below 2^1000 — an overflowed power somewhere, hundreds of log-units from any posterior — term by term,
       * under a wave-uniform branch (a chain's value depends on its own product alone) */
    double prod = la[0];
    PHF_UNROLL
    for (int i = 1; i < n_expts; ++i) prod *= la[i];
    const int big = !(prod < 0x1p1000);
    double v = phf_log_pos_k(prod, k_log);
    if (PHF_ANY_LANE(big)) {
      double vs = 0.0;
      PHF_UNROLL
      for (int i = 0; i < n_expts; ++i) {
        const double vi = phf_log_pos_k(la[i], k_log);
        vs += (la[i] > 0x1p1000) ? PHF_INF : vi;                             /* overflowed power: log(inf) = inf */
      }
      v = big ? vs : v;
    }
    part = phf_fma(-2.0, v, part);
  }
  /* ---- this half's points (:117-125) ---- */
  double sse = 0.0, mass = 1.0;                                    /* mass: product of the truncation masses Phi(b) - Phi(a) of this half's points */
  int poff = 0;                                                    /* fixed shapes: where experiment i's points start in plc / py */
  PHF_UNROLL
  for (int i = 0; i < n_expts; ++i) {
    const double pic50 = th[(4 + 2 * i) * ts], hill = th[(5 + 2 * i) * ts];
    const double ln_ic50 = PHF_LN10 * (6.0 - pic50);
    /* this half's points of experiment i: [j, jend).  Fixed shapes: the counts are literals, the points already in plc / py */
    const int sb = fixed_n ? i * fper : expt_start[i];
    const int se = fixed_n ? sb + PHF_NI_(i) : expt_start[i + 1];
    const int nf = fixed_n ? PHF_NF_(i) : 2 * ((se - sb + 2) / 4);
    const int cut = sb + ((nf < se - sb) ? nf : se - sb);
    int j = sb + (h ? 1 : 0) * (cut - sb);                                   /* affine in the lane parity: one per-lane base, immediate offsets */
    const int fcnt = flast ? (h ? PHF_NI_(i) - PHF_NF_(i) : PHF_NF_(i)) : fper / 2;
    const int jend = fixed_n ? j + fcnt : PHF_PICK(h, cut, se);
    const int npairs = fixed_n ? fcnt / 2 : (jend - j) / 2;
    PHF_UNROLL
    for (int p = 0; p < npairs; ++p, j += 2) {                               /* two points at a time */
      const phf_ktab ke = k_exp;
      const double lc0 = fixed_n ? plc[poff + 2 * p] : lc[j], lc1 = fixed_n ? plc[poff + 2 * p + 1] : lc[j + 1];
      const double y0 = fixed_n ? py[poff + 2 * p] : y[j], y1 = fixed_n ? py[poff + 2 * p + 1] : y[j + 1];
      const double d0 = 1.0 + phf_exp_capped_k(__builtin_fmin(hill * (lc0 - ln_ic50), 40.0), ke);
      const double d1 = 1.0 + phf_exp_capped_k(__builtin_fmin(hill * (lc1 - ln_ic50), 40.0), ke);
      const double inv = phf_rcp(d0 * d1);
      const double pred0 = phf_fma(-PHF_K100(k_exp), inv * d1, PHF_K100(k_exp)), pred1 = phf_fma(-PHF_K100(k_exp), inv * d0, PHF_K100(k_exp));
      const double r0 = y0 - pred0, r1 = y1 - pred1;
      sse = phf_fma(r0, r0, sse); sse = phf_fma(r1, r1, sse);
      mass *= phf_trunc_mass_x2(pred0, pred1, inv_s, k_exp, skip_tails);
    }
    if (fixed_n ? (fcnt & 1) : (j < jend)) {                                 /* at most one left */
      const double lcs = fixed_n ? plc[poff + fcnt - 1] : lc[j], ys = fixed_n ? py[poff + fcnt - 1] : y[j];
      const double w = phf_rcp(1.0 + phf_exp_capped_k(__builtin_fmin(hill * (lcs - ln_ic50), 40.0), k_exp));
      const double pred = phf_fma(-PHF_K100(k_exp), w, PHF_K100(k_exp));
      const double r = ys - pred;
      sse = phf_fma(r, r, sse);
      mass *= phf_trunc_mass(pred, inv_s, k_exp);
    }
    poff += fcnt;
  }
#undef PHF_NI_
#undef PHF_NF_
  /* sum_j ln(Phi(b_j) - Phi(a_j)) as ONE logarithm of the product: every mass is in (0, 1], a half has a handful of points, so the
   * product cannot overflow and underflows only where sigma ~ 1e27 — there the half is -inf (a proposal to reject), never +inf */
  const double trunc = phf_log_pos_k(mass, k_log);
  const double r = part - phf_fma(sse, 0.5 * inv_s * inv_s, trunc);
  return (mass < PHF_DBL_MIN) ? -PHF_INF : r;
}

PHF_HD double phf_hier_log_target_n(int n_expts, int fixed_n, const int* expt_start, const double* lc, const double* y,
                                    const double* th, int ts, const phf_hier_prior* pr, phf_ktab k_exp, phf_ktab k_log) {
  const int bad = phf_hier_out_of_support(n_expts, th, ts, pr);
  const double p0 = phf_hier_target_half(0, n_expts, fixed_n, expt_start, lc, y, th, ts, pr, k_exp, k_log, 1);
  const double p1 = phf_hier_target_half(1, n_expts, fixed_n, expt_start, lc, y, th, ts, pr, k_exp, k_log, 1);
  return bad ? -PHF_INF : p0 + p1;
}

PHF_HD double phf_hier_log_target(int n_expts, const int* expt_start, const double* lc, const double* y,
                                  const double* th, int ts, const phf_hier_prior* pr, phf_ktab k_exp, phf_ktab k_log) {
  return phf_hier_log_target_n(n_expts, 0, expt_start, lc, y, th, ts, pr, k_exp, k_log);
}

/* ---- the same target, experiment by experiment --------------------------------------------------------------------
 * For pairs with many experiments (more than the kernels compiled per Ne cover: the reference's synthetic set has one
 * with 50) a wavefront works on ONE chain and its lanes take one experiment each.  An experiment's terms are then computed
 * on their own — no reciprocal shared across experiments — and the per-experiment sums are added in experiment order.
 * Same mathematics as phf_hier_log_target_n, a different (equally fixed) order of roundings; the scalar twin uses this
 * form for the same pairs, so the two still agree bit for bit.                                                      */
typedef struct {
  double inv_s, inv_sc;                 /* 1/sigma, 1/s */
  double log_sigma, ln_alpha, ln_beta, ln_s, beta, mu;
  double prior;                         /* the five shifted-Gamma log-priors (:187) */
  int bad;                              /* hyper-parameter outside its support (:176,182) */
} phf_hier_common;

/* the part that depends on (alpha, beta, mu, s, sigma) only: 9 logarithms, and 1/sigma, 1/s behind one division */
PHF_HD phf_hier_common phf_hier_common_terms(double alpha, double beta, double mu, double s, double sigma,
                                             const phf_hier_prior* pr, phf_ktab k_log) {
  phf_hier_common c;
  c.bad = (alpha <= pr->loc[0]) | (beta <= pr->loc[1]) | (mu <= pr->loc[2]) | (s <= pr->loc[3]) | (sigma <= pr->loc[4]);
  const double hv[5] = {alpha, beta, mu, s, sigma};
  double lx[9];
  lx[0] = sigma; lx[1] = alpha; lx[2] = beta; lx[3] = s;
  PHF_UNROLL
  for (int k = 0; k < 5; ++k) lx[4 + k] = hv[k] - pr->loc[k];
  double rc[2] = {sigma, s};
  phf_batch_recip(rc, 2);
  c.inv_s = rc[0]; c.inv_sc = rc[1];
  double lg[9];
  PHF_UNROLL
  for (int k = 0; k < 9; ++k) lg[k] = phf_log_fast_k(lx[k], k_log);
  c.log_sigma = lg[0]; c.ln_alpha = lg[1]; c.ln_beta = lg[2]; c.ln_s = lg[3]; c.beta = beta; c.mu = mu;
  double prior = 0.0;
  PHF_UNROLL
  for (int k = 0; k < 5; ++k) prior += phf_fma(pr->shape_m1[k], lg[4 + k], -lx[4 + k] * pr->inv_scale[k]);
  c.prior = prior;
  return c;
}

/* one experiment: its n points lc[0..n-1], y[0..n-1] and its (pIC50_i, Hill_i); returns through sse, trunc, hyper, bad */
PHF_HD void phf_hier_experiment_terms(const phf_hier_common* c, double pic50, double hill, const double* lc, const double* y,
                                      int n, phf_ktab k_exp, phf_ktab k_log, double* out_sse, double* out_trunc,
                                      double* out_hyper, int* out_bad) {
  *out_bad = (hill < 0.0) | (pic50 < PHF_HIER_PIC50_LOWER);
  const double ln_ic50 = PHF_LN10 * (6.0 - pic50);
  double sse = 0.0, trunc = 0.0;
  int j = 0;
  for (; j + 2 <= n; j += 2) {                                               /* :117-125, two points at a time */
    const phf_ktab ke = k_exp;
    const double d0 = 1.0 + phf_exp_capped_k(__builtin_fmin(hill * (lc[j] - ln_ic50), 40.0), ke);
    const double d1 = 1.0 + phf_exp_capped_k(__builtin_fmin(hill * (lc[j + 1] - ln_ic50), 40.0), ke);
    const double inv = phf_rcp(d0 * d1);
    const double pred0 = phf_fma(-PHF_K100(k_exp), inv * d1, PHF_K100(k_exp)), pred1 = phf_fma(-PHF_K100(k_exp), inv * d0, PHF_K100(k_exp));
    const double r0 = y[j] - pred0, r1 = y[j + 1] - pred1;
    sse = phf_fma(r0, r0, sse); sse = phf_fma(r1, r1, sse);
    trunc += phf_trunc_term(pred0, c->inv_s, k_exp, k_log) + phf_trunc_term(pred1, c->inv_s, k_exp, k_log);
  }
  for (; j < n; ++j) {
    const double w = phf_rcp(1.0 + phf_exp_capped_k(__builtin_fmin(hill * (lc[j] - ln_ic50), 40.0), k_exp));
    const double pred = phf_fma(-PHF_K100(k_exp), w, PHF_K100(k_exp));
    const double r = y[j] - pred;
    sse = phf_fma(r, r, sse);
    trunc += phf_trunc_term(pred, c->inv_s, k_exp, k_log);
  }
  /* log-logistic density of Hill_i (:134-142), logistic density of pIC50_i (:144-154): three logarithms */
  const double z = (pic50 - c->mu) * c->inv_sc;
  const double ln_h = phf_log_fast_k(hill, k_log);
  const double la1 = 1.0 + phf_exp_fast_k(-z, k_exp);
  const double v1 = phf_log_pos_k(la1, k_log);
  const double la0 = 1.0 + phf_exp_fast_k(c->beta * (ln_h - c->ln_alpha), k_exp);
  const double v0 = phf_log_pos_k(la0, k_log);
  double hyper = (c->ln_beta - c->beta * c->ln_alpha) + (c->beta - 1.0) * ln_h;
  hyper += (-z - c->ln_s);
  hyper -= 2.0 * ((la0 > 0x1p1000) ? PHF_INF : v0);                          /* overflowed power: log(inf) = inf */
  hyper -= 2.0 * ((la1 > 0x1p1000) ? PHF_INF : v1);
  *out_sse = sse; *out_trunc = trunc; *out_hyper = hyper;
}

PHF_HD double phf_hier_combine(const phf_hier_common* c, int n_pts, double sse, double trunc, double hyper, int bad) {
  double total = -(phf_fma((double)n_pts, c->log_sigma, sse * (0.5 * c->inv_s * c->inv_s)) + trunc);   /* :122-125 */
  total += hyper;
  total += c->prior;
  return (bad | c->bad) ? -PHF_INF : total;
}

/* sequential form: what one lane, or the scalar twin, computes for a whole parameter vector */
PHF_HD double phf_hier_log_target_by_experiment(int n_expts, const int* expt_start, const double* lc, const double* y,
                                                const double* th, int ts, const phf_hier_prior* pr, phf_ktab k_exp,
                                                phf_ktab k_log) {
  const int dim = 5 + 2 * n_expts;
  const phf_hier_common c = phf_hier_common_terms(th[0], th[1 * ts], th[2 * ts], th[3 * ts], th[(dim - 1) * ts], pr, k_log);
  double sse = 0.0, trunc = 0.0, hyper = 0.0;
  int bad = 0;
  for (int i = 0; i < n_expts; ++i) {
    double a, b, h;
    int bi;
    phf_hier_experiment_terms(&c, th[(4 + 2 * i) * ts], th[(5 + 2 * i) * ts], lc + expt_start[i], y + expt_start[i],
                              expt_start[i + 1] - expt_start[i], k_exp, k_log, &a, &b, &h, &bi);
    sse += a; trunc += b; hyper += h; bad |= bi;
  }
  return phf_hier_combine(&c, expt_start[n_expts], sse, trunc, hyper, bad);
}

/* which form a pair uses: the kernels compiled per Ne (and their twin) the batched one, larger Ne the per-experiment one */
#define PHF_HIER_BATCHED_MAX_EXPTS 8
PHF_HD double phf_hier_log_target_any(int n_expts, const int* expt_start, const double* lc, const double* y,
                                      const double* th, int ts, const phf_hier_prior* pr, phf_ktab k_exp, phf_ktab k_log) {
  if (n_expts > PHF_HIER_BATCHED_MAX_EXPTS)
    return phf_hier_log_target_by_experiment(n_expts, expt_start, lc, y, th, ts, pr, k_exp, k_log);
  return phf_hier_log_target_n(n_expts, 0, expt_start, lc, y, th, ts, pr, k_exp, k_log);
}

/* Draws of hierarchical MH iteration t: Philox blocks 0..NB-1, NB = ceil(dim/4); word j of block b is standard normal 4 b + j
 * (phf_normal_u32: the piecewise inverse CDF of one word, as in the single-level sampler) for 4 b + j < dim; dim = 5 + 2 Ne is odd,
 * so the last block always has a word to spare, and its LAST word is the accept uniform u = (w + 1/2) / 2^32 (32 bits, like
 * single-level model 2: log u >= -22.9).  Returns log(u); the normals go to z[i*zs].
 * (Rounds 1-3 first half: two Box-Muller pairs per block and one more block for a 53-bit uniform.)                      */
PHF_HD double phf_hier_draws_k(int dim, uint32_t chain_id, uint32_t problem_id, uint32_t t, uint32_t seed_lo,
                               uint32_t seed_hi, phf_ktab k_log, double* z, int zs) {
  const int nb = (dim + 3) / 4;
  uint32_t w_u = 0u;
  PHF_UNROLL
  for (int b = 0; b < nb; ++b) {
    const phf_u32x4 w = phf_philox_mh(chain_id, problem_id, t, (uint32_t)b, seed_lo, seed_hi);
    const int i = 4 * b;
    z[i * zs] = phf_normal_u32(w.w[0]);
    if (i + 1 < dim) z[(i + 1) * zs] = phf_normal_u32(w.w[1]);
    if (i + 2 < dim) z[(i + 2) * zs] = phf_normal_u32(w.w[2]);
    if (i + 3 < dim) z[(i + 3) * zs] = phf_normal_u32(w.w[3]);
    if (b == nb - 1) w_u = w.w[3];
  }
  return phf_log_pos_k(phf_unit_open32(w_u), k_log);
}

PHF_HD double phf_hier_draws(int dim, uint32_t chain_id, uint32_t problem_id, uint32_t t, uint32_t seed_lo,
                             uint32_t seed_hi, phf_ktab k_log, double* z, int zs) {
  return phf_hier_draws_k(dim, chain_id, problem_id, t, seed_lo, seed_hi, k_log, z, zs);
}

#endif /* PHF_HIER_MODEL_H */
