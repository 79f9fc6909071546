/* phf_predictive_model.h — posterior-predictive curves of Hill and pIC50 from hierarchical samples
 * (python/construct_hierarchical_cdfs.py:32-58): for every saved (alpha, beta, mu, s) the reference adds, on a
 * 501-point grid each,
 *     fisk.cdf(x; c=beta, scale=alpha) = 1/(1 + (x/alpha)^-beta)          fisk.pdf = (beta/x) u/(1+u)^2, u = (x/alpha)^-beta
 *     logistic.cdf(x; mu, s)           = 1/(1 + e^-(x-mu)/s)              logistic.pdf = (1/s) v/(1+v)^2, v = e^-(x-mu)/s
 * and divides the four sums by the number of samples.  The per-sample and per-(sample, grid point) arithmetic lives
 * here, shared by the HIP kernel (phf_predictive.hip) and the scalar CPU twin (oracle/phf_oracle.c); both add the
 * terms in the same order (samples in order inside a chunk, chunks in order), so they agree bit for bit.
 *
 * u and v are exponentials of arguments capped at +170 (e^170 ~ 7e73: four factors (1+u)(1+v)(1+u')(1+v') of a
 * shared reciprocal stay below DBL_MAX); the cap changes a CDF value only below 1.5e-74.                      */
#ifndef PHF_PREDICTIVE_MODEL_H
#define PHF_PREDICTIVE_MODEL_H

#include "phf_math.h"

#define PHF_PRED_TILE 512          /* samples staged per LDS tile */
#define PHF_PRED_CURVES 4          /* hill cdf, pic50 cdf, hill pdf, pic50 pdf */
#define PHF_PRED_EXP_CAP 170.0

/* sample -> (ln alpha, beta, mu, 1/s) */
PHF_HD void phf_pred_prepare(double alpha, double beta, double mu, double s, double* lna, double* b, double* m,
                             double* inv_s) {
  *lna = phf_log_fast(alpha);
  *b = beta;
  *m = mu;
  *inv_s = 1.0 / s;
}

/* grid point -> (ln x, 1/x) of the Hill axis; x <= 0 gives (-inf, 0): its pdf terms vanish, its cdf is zeroed on output */
PHF_HD void phf_pred_hill_axis(double x, double* lnx, double* inv_x) {
  const int pos = x > 0.0;
  *lnx = pos ? phf_log_fast(x) : -PHF_INF;
  *inv_x = pos ? 1.0 / x : 0.0;
}

/* One sample against TWO grid points of each axis: acc[j][curve] += term.  k_exp: exp coefficient table. */
PHF_HD void phf_pred_accumulate2(double lna, double beta, double mu, double inv_s, const double lnx[2],
                                 const double inv_x[2], const double px[2], phf_ktab k_exp, double acc[2][PHF_PRED_CURVES]) {
  double u[2], v[2], a[2], b[2];
  PHF_UNROLL
  for (int j = 0; j < 2; ++j) {
    const double w = beta * (lna - lnx[j]);                 /* ln (x/alpha)^-beta */
    const double z = (mu - px[j]) * inv_s;                  /* -(x-mu)/s */
    u[j] = phf_exp_capped_k(__builtin_fmin(w, PHF_PRED_EXP_CAP), k_exp);
    v[j] = phf_exp_capped_k(__builtin_fmin(z, PHF_PRED_EXP_CAP), k_exp);
    a[j] = 1.0 + u[j];
    b[j] = 1.0 + v[j];
  }
  const double p0 = a[0] * b[0], p1 = a[1] * b[1];
  const double r = 1.0 / (p0 * p1);                         /* one division for the four reciprocals */
  const double r0 = r * p1, r1 = r * p0;
  const double ra[2] = {r0 * b[0], r1 * b[1]};              /* 1/(1+u) */
  const double rb[2] = {r0 * a[0], r1 * a[1]};              /* 1/(1+v) */
  PHF_UNROLL
  for (int j = 0; j < 2; ++j) {
    acc[j][0] += ra[j];
    acc[j][1] += rb[j];
    acc[j][2] = phf_fma((beta * inv_x[j]) * (u[j] * ra[j]), ra[j], acc[j][2]);
    acc[j][3] = phf_fma(inv_s * (v[j] * rb[j]), rb[j], acc[j][3]);
  }
}

#endif /* PHF_PREDICTIVE_MODEL_H */
