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

/* Phi(b) - Phi(a) for a <= 0 <= b:  1 - [Q(b) + Phi(a)], both tails through erfcx (no cancellation in the tails) */
PHF_HD double phf_norm_mass_between(double a, double b, phf_ktab k_erfcx, phf_ktab k_exp) {
  const double ya = -a * PHF_INV_SQRT2, yb = b * PHF_INV_SQRT2;
  const double qa = phf_erfcx_den(ya), qb = phf_erfcx_den(yb);
  const double iq = 1.0 / (qa * qb);
  const double ea = phf_erfcx_finish_k(ya, iq * qb, k_erfcx), eb = phf_erfcx_finish_k(yb, iq * qa, k_erfcx);
  const double ga = phf_exp_fast_k(-0.5 * a * a, k_exp), gb = phf_exp_fast_k(-0.5 * b * b, k_exp);
  return 1.0 - 0.5 * phf_fma(ea, ga, eb * gb);
}

PHF_HD double phf_hier_log_target(int n_expts, const int* expt_start, const double* lc, const double* y,
                                  const double* th, int ts, const phf_hier_prior* pr) {
  const int dim = 5 + 2 * n_expts;
  const double alpha = th[0], beta = th[1 * ts], mu = th[2 * ts], s = th[3 * ts], sigma = th[(dim - 1) * ts];
  int bad = (alpha <= pr->loc[0]) | (beta <= pr->loc[1]) | (mu <= pr->loc[2]) | (s <= pr->loc[3]) | (sigma <= pr->loc[4]);
  const double inv_s = 1.0 / sigma;
  const double log_sigma = phf_log_fast(sigma);
  const double ln_alpha = phf_log_fast(alpha), ln_beta = phf_log_fast(beta), ln_s = phf_log_fast(s);
  const double inv_sc = 1.0 / s;
  double sse = 0.0, trunc = 0.0, hyper = 0.0;
  PHF_UNROLL
  for (int i = 0; i < n_expts; ++i) {
    const double pic50 = th[(4 + 2 * i) * ts], hill = th[(5 + 2 * i) * ts];
    bad |= (hill < 0.0) | (pic50 < PHF_HIER_PIC50_LOWER);
    const double ln_ic50 = PHF_LN10 * (6.0 - pic50);
    for (int j = expt_start[i]; j < expt_start[i + 1]; ++j) {               /* :117-125 */
      const phf_ktab ke = PHF_KLOAD(phf_k_exp);
      const double a = hill * (lc[j] - ln_ic50);
      const double w = 1.0 / (1.0 + phf_exp_capped_k(__builtin_fmin(a, 40.0), ke, 1));
      const double pred = phf_fma(-100.0, w, 100.0);
      const double r = y[j] - pred;
      sse = phf_fma(r, r, sse);
      const double mass = phf_norm_mass_between(-pred * inv_s, (100.0 - pred) * inv_s, PHF_KLOAD(phf_k_erfcx), ke);
      trunc += phf_log_fast(mass);
    }
    /* log-logistic density of Hill_i (:134-142) and logistic density of pIC50_i (:144-154) */
    const double ln_h = phf_log_fast(hill);
    const double pw = phf_exp_fast(beta * (ln_h - ln_alpha));               /* (Hill_i/alpha)^beta */
    const double ll = (ln_beta - beta * ln_alpha) + (beta - 1.0) * ln_h - 2.0 * phf_log_fast(1.0 + pw);
    const double z = (pic50 - mu) * inv_sc;
    const double lg = (-z - ln_s) - 2.0 * phf_log_fast(1.0 + phf_exp_fast(-z));
    hyper += ll; hyper += lg;
  }
  const int n_pts = expt_start[n_expts];
  double total = -(phf_fma((double)n_pts, log_sigma, sse * (0.5 * inv_s * inv_s)) + trunc);   /* :122-125 */
  total += hyper;
  const double hv[5] = {alpha, beta, mu, s, sigma};
  for (int k = 0; k < 5; ++k) {                                              /* :187 */
    const double xl = hv[k] - pr->loc[k];
    total += phf_fma(pr->shape_m1[k], phf_log_fast(xl), -xl * pr->inv_scale[k]);
  }
  return bad ? -PHF_INF : total;
}

/* Draws of hierarchical MH iteration t: dim standard normals into z[i*zs] (Box-Muller, two pairs per Philox block,
 * blocks 0..ceil(dim/4)-1) and log(u) of the accept uniform (block ceil(dim/4)).                                 */
PHF_HD double phf_hier_draws(int dim, uint32_t chain_id, uint32_t problem_id, uint32_t t, uint32_t seed_lo,
                             uint32_t seed_hi, double* z, int zs) {
  const int nb = (dim + 3) / 4;
  PHF_UNROLL
  for (int b = 0; b < nb; ++b) {
    const phf_u32x4 w = phf_philox4x32_10(chain_id, problem_id, t, (uint32_t)b, seed_lo, seed_hi);
    double z0, z1, z2, z3;
    phf_box_muller(w.w[0], w.w[1], &z0, &z1);
    phf_box_muller(w.w[2], w.w[3], &z2, &z3);
    const int i = 4 * b;
    z[i * zs] = z0;
    if (i + 1 < dim) z[(i + 1) * zs] = z1;
    if (i + 2 < dim) z[(i + 2) * zs] = z2;
    if (i + 3 < dim) z[(i + 3) * zs] = z3;
  }
  const phf_u32x4 wu = phf_philox4x32_10(chain_id, problem_id, t, (uint32_t)nb, seed_lo, seed_hi);
  return phf_log_fast(phf_uniform53(wu.w[0], wu.w[1]));
}

#endif /* PHF_HIER_MODEL_H */
