/* phf_model.h — the single-level Hill-curve log-target and the per-iteration random draws, written once
 * and compiled both into the gfx950 kernels and into the host twin (oracle/), so the two evaluate the very
 * same fp64 operation sequence.
 *
 * What it computes is the reference's (python/doseresponse.py):
 *   dose_response_model :84-85, pic50_to_ic50 :87-88, log_data_likelihood_model_{1,2}_capped :203-248,
 *   log_priors_model_{1,2} :166-184, log_pic50_exponential :151-156, log_gamma_prior :304-317.
 * How it computes it is shaped for the fp64 VALU (see phf_math.h): per-point work is done 4 (uncensored) or
 * 2 (censored) points at a time so independent polynomial chains interleave, and the IEEE divisions of a
 * group — 1/(1+x) of the Hill curves — are shared through a product tree (one division + a few multiplies
 * instead of one division each).  Logarithms, exponentials, the censored entries' log Phi and the proposals'
 * normals come from the LDS tables of phf_math.h (no division, no erfcx, no Box-Muller).
 *
 * Entry layout (include/pyhillfit_amd.h, phf_points): n_other uncensored entries first (0 < y < 100), then the
 * n_cens censored ones (y == 0, then y == 100).  Arrays must be readable up to index n_other + n_cens - 1.
 * An entry stands for w data points at one concentration.  The reference's sums over points (:244-247) regroup
 * exactly: the curve depends on the point only through its concentration, so for replicates at one concentration
 *     sum_j (y_j - pred)^2 = sum_j (y_j - ybar)^2 + w (ybar - pred)^2      (uncensored; first term constant in theta)
 *     sum_j log Phi(z(pred)) = w log Phi(z(pred))                            (censored)
 * and the per-iteration cost scales with the number of distinct concentrations (Crumb: 4 doses x 3-6 experiments).
 * w = 1, ybar = y, ss_within = 0 is the unmerged form.
 */
#ifndef PHF_MODEL_H
#define PHF_MODEL_H

#include "phf_math.h"
#include "phf_philox.h"

#define PHF_SIGMA_FLOOR 1e-3                    /* sigma_uniform_lower, doseresponse.py:12  */
#define PHF_PIC50_RATE 0.2                      /* doseresponse.py:14  */
#define PHF_PIC50_LOWER (-3.0)                  /* doseresponse.py:16  */
#define PHF_HILL_UPPER 10.0                     /* doseresponse.py:18  */
#define PHF_SIGMA_LOC 1e-3                      /* doseresponse.py:24  */
#define PHF_SIGMA_SHAPE_M1 4.0                  /* sigma_shape - 1, doseresponse.py:22 */
#define PHF_SIGMA_INV_SCALE (4.0 / (6.0 - 1e-3)) /* 1/sigma_scale, doseresponse.py:25 */
#define PHF_HILL_ARG_CAP 40.0                   /* exp(40): 100/(1+x) already rounds pred to exactly 100 */

/* Hill-curve denominator 1 + (dose/IC50)^hill = 1 + exp(hill (ln dose - ln IC50))   (doseresponse.py:84-88) */
PHF_HD double phf_hill_den(int model, double ln_conc, double hill, double ln_ic50, phf_ktab k_exp) {
  const double a = (model == 1) ? (ln_conc - ln_ic50) : hill * (ln_conc - ln_ic50);
  return 1.0 + phf_exp_capped_k(__builtin_fmin(a, PHF_HILL_ARG_CAP), k_exp);
}

/* percent block from w = 1/(1 + x):  100 (1 - w) */
PHF_HD double phf_hill_percent(double w, phf_ktab k_exp) { return phf_fma(-PHF_K100(k_exp), w, PHF_K100(k_exp)); }
#define PHF_PCT_(w_) phf_hill_percent((w_), k_exp)      /* inside the target: k_exp is in scope */

/* z-score of a censored point: y == 0 -> (0 - pred)/sigma (logcdf, :244); y == 100 -> (pred - 100)/sigma (logsf, :245) */
PHF_HD double phf_censored_z(double pred, double y, double inv_s) {
  const double sgn = (y > 50.0) ? 1.0 : -1.0;
  return (sgn * (pred - y)) * inv_s;
}

/* log-likelihood (tempered), log-prior and UNtempered log-likelihood of one parameter vector.
 * model 1: th = (pIC50, sigma), Hill = 1;  model 2: th = (pIC50, Hill, sigma).
 * out_ll1 = log_data_likelihood(..., t = 1): what python/compute_bayes_factors.py:18-21 re-evaluates for every saved
 * sample of every rung; it falls out of the same arithmetic here, so the samplers carry it along for free.      */
PHF_HD void phf_sl_log_target(int model, const double* lc, const double* y, const double* w, int n_other, int n_cens,
                              double n_other_points, double ss_within, double pi_bit, double temperature,
                              const double* th, phf_ktab k_exp, phf_ktab k_log,
                              double* out_lik, double* out_prior, double* out_ll1) {
  const double pic50 = th[0];
  const double hill = (model == 1) ? 1.0 : th[1];
  const double sigma = (model == 1) ? th[1] : th[2];
  const double ln_ic50 = PHF_LN10 * (6.0 - pic50);
  const double sl = sigma - PHF_SIGMA_LOC;
  const double inv_s = phf_rcp(sigma);
  const double log_sigma = phf_log_pos_k(sigma, k_log);
  const double log_sl = phf_log_pos_k(sl, k_log);

  double sse = ss_within, cens = 0.0;
  int j = 0;
  for (; j + 4 <= n_other; j += 4) {              /* uncensored entries, four at a time (:247) */
    const phf_ktab ke = k_exp;
    const double d0 = phf_hill_den(model, lc[j], hill, ln_ic50, ke), d1 = phf_hill_den(model, lc[j + 1], hill, ln_ic50, ke);
    const double d2 = phf_hill_den(model, lc[j + 2], hill, ln_ic50, ke), d3 = phf_hill_den(model, lc[j + 3], hill, ln_ic50, ke);
    const double p01 = d0 * d1, p23 = d2 * d3;
    const double inv = phf_rcp(p01 * p23);
    const double i01 = inv * p23, i23 = inv * p01;
    const double r0 = y[j] - PHF_PCT_(i01 * d1), r1 = y[j + 1] - PHF_PCT_(i01 * d0);
    const double r2 = y[j + 2] - PHF_PCT_(i23 * d3), r3 = y[j + 3] - PHF_PCT_(i23 * d2);
    sse = phf_fma(w[j] * r0, r0, sse); sse = phf_fma(w[j + 1] * r1, r1, sse);
    sse = phf_fma(w[j + 2] * r2, r2, sse); sse = phf_fma(w[j + 3] * r3, r3, sse);
  }
  const int rem = n_other - j;                    /* 0..3 left: still one division */
  if (rem == 3) {
    const phf_ktab ke = k_exp;
    const double d0 = phf_hill_den(model, lc[j], hill, ln_ic50, ke), d1 = phf_hill_den(model, lc[j + 1], hill, ln_ic50, ke);
    const double d2 = phf_hill_den(model, lc[j + 2], hill, ln_ic50, ke);
    const double p01 = d0 * d1;
    const double inv = phf_rcp(p01 * d2);
    const double i01 = inv * d2;
    const double r0 = y[j] - PHF_PCT_(i01 * d1), r1 = y[j + 1] - PHF_PCT_(i01 * d0);
    const double r2 = y[j + 2] - PHF_PCT_(inv * p01);
    sse = phf_fma(w[j] * r0, r0, sse); sse = phf_fma(w[j + 1] * r1, r1, sse); sse = phf_fma(w[j + 2] * r2, r2, sse);
  } else if (rem == 2) {
    const phf_ktab ke = k_exp;
    const double d0 = phf_hill_den(model, lc[j], hill, ln_ic50, ke), d1 = phf_hill_den(model, lc[j + 1], hill, ln_ic50, ke);
    const double inv = phf_rcp(d0 * d1);
    const double r0 = y[j] - PHF_PCT_(inv * d1), r1 = y[j + 1] - PHF_PCT_(inv * d0);
    sse = phf_fma(w[j] * r0, r0, sse); sse = phf_fma(w[j + 1] * r1, r1, sse);
  } else if (rem == 1) {
    const double r = y[j] - PHF_PCT_(phf_rcp(phf_hill_den(model, lc[j], hill, ln_ic50, k_exp)));
    sse = phf_fma(w[j] * r, r, sse);
  }
  j = n_other;
  const int n = n_other + n_cens;
  /* censored entries (:244-245): log Phi(z), z <= 0, from the table of phf_math.h — it covers every z a likelihood that is not
   * -inf anyway (sigma above its floor, below) can produce */
  for (; j + 2 <= n; j += 2) {                    /* two at a time: one division for the two Hill curves */
    const phf_ktab ke = k_exp;
    const double d0 = phf_hill_den(model, lc[j], hill, ln_ic50, ke), d1 = phf_hill_den(model, lc[j + 1], hill, ln_ic50, ke);
    const double inv = phf_rcp(d0 * d1);
    const double z0 = phf_censored_z(PHF_PCT_(inv * d1), y[j], inv_s);
    const double z1 = phf_censored_z(PHF_PCT_(inv * d0), y[j + 1], inv_s);
    cens = phf_fma(w[j], phf_log_ndtr_tab(z0, -z0 * PHF_INV_SQRT2), cens);
    cens = phf_fma(w[j + 1], phf_log_ndtr_tab(z1, -z1 * PHF_INV_SQRT2), cens);
  }
  for (; j < n; ++j) {
    const double pred = PHF_PCT_(phf_rcp(phf_hill_den(model, lc[j], hill, ln_ic50, k_exp)));
    const double z = phf_censored_z(pred, y[j], inv_s);
    cens = phf_fma(w[j], phf_log_ndtr_tab(z, -z * PHF_INV_SQRT2), cens);
  }
  double a = cens - pi_bit;
  a = phf_fma(-n_other_points, log_sigma, a);                            /* :246 */
  a = phf_fma(-sse, 0.5 * inv_s * inv_s, a);                             /* :247 */
  if (sigma <= PHF_SIGMA_FLOOR) a = -PHF_INF;                            /* :238-240 (one select: t * -inf = -inf for t > 0 ...) */
  double lik = temperature * a;                                          /* :248 */
  if (temperature == 0.0) lik = 0.0;                                     /* :230-231 (... and t = 0 is 0 whatever a is) */
  *out_lik = lik;
  *out_ll1 = a;

  /* the prior: -inf outside ANY of the supports, the sum of the two terms inside (one select on one combined predicate, not one per
   * term: the reference's -inf + x and -inf + -inf are -inf either way) */
  const int outside = (pic50 < PHF_PIC50_LOWER)                          /* :151-156 */
                      | (sigma <= PHF_SIGMA_LOC)                         /* :304-317: x < loc -> -inf; x == loc -> log 0 = -inf */
                      | ((model == 2) & ((hill < 0.0) | (hill > PHF_HILL_UPPER)));   /* :181-182 */
  const double g = phf_fma(PHF_SIGMA_SHAPE_M1, log_sl, -sl * PHF_SIGMA_INV_SCALE);
  const double lp = -PHF_PIC50_RATE * pic50 + g;
  *out_prior = outside ? -PHF_INF : lp;
}

/* The random numbers of MH iteration t of one chain — ONE Philox4x32-10 block (128 bits) per iteration:
 *   d == 2: z0, z1 from words 0, 1 (phf_normal_u32: piecewise inverse CDF of a 31-bit uniform + a sign bit, |z| <= 6.34),
 *           accept uniform from words 2, 3 (53 bits: numpy's random_sample construction);
 *   d == 3: z0, z1, z2 from words 0, 1, 2, accept uniform (w + 1/2) / 2^32 from word 3.
 * Stateless: a function of (chain, problem, t, seed) alone, so a launch, a quantum or the twin may start anywhere.
 * Returns log(u) (PyHillFit.py:834-835) and the d standard normals in z (PyHillFit.py:831); z[2] = 0 for d == 2.
 * (Rounds 1-3 drew the normals by Box-Muller from 24- / 32-bit fields; the inverse CDF costs a third of the fp64 operations.) */
PHF_HD double phf_mh_draws(int d, uint32_t chain_id, uint32_t problem_id, uint32_t t, uint32_t seed_lo,
                           uint32_t seed_hi, phf_ktab k_log, double* z) {
  const phf_u32x4 b = phf_philox_mh(chain_id, problem_id, t, 0u, seed_lo, seed_hi);
  z[0] = phf_normal_u32(b.w[0]);
  z[1] = phf_normal_u32(b.w[1]);
  if (d == 2) {
    z[2] = 0.0;
    const double u = phf_uniform53(b.w[2], b.w[3]);
    const double log_u = phf_log_pos_k(u, k_log);
    return (u < PHF_DBL_MIN) ? -PHF_INF : log_u;       /* u == 0 (probability 2^-53): log 0 = -inf, accept */
  }
  z[2] = phf_normal_u32(b.w[2]);
  return phf_log_pos_k(phf_unit_open32(b.w[3]), k_log);  /* u = (w + 1/2) / 2^32 is never 0 */
}

#endif /* PHF_MODEL_H */
