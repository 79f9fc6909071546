/* phf_oracle.c — scalar CPU twin of the HIP Metropolis-Hastings kernels.
 *
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg load the library built from this file (oracle/_build/).
 *
 * It restates the reference algorithm (mirams/PyHillFit) one chain at a time in plain C:
 *   censored-Gaussian Hill log-likelihood     python/doseresponse.py:203-248
 *   priors and log-target                     python/doseresponse.py:151-189,304-317
 *   adaptive Metropolis loop                  python/PyHillFit.py:796-856, python/PyHillTemp.py:76-123
 *   hierarchical target and loop              python/PyHillFit.py:113-193,429-511
 * The chain state machine (start, proposal assembly, Cholesky factor, accept rule, adaptation, thinning,
 * segment hand-over) is written independently of the device code.  The log-target and the per-iteration
 * draws come from the SAME source the kernels compile (phf_model.h on top of phf_math.h / phf_philox.h):
 * their evaluation order is tuned for the fp64 VALU and only a shared source can keep two builds bit-identical,
 * which is what makes "same seed => same accept sequence and bit-identical chain" testable on the GPU.
 * What pins that shared arithmetic to the reference is below, not the sharing.
 * PINNED (tests/test_oracle_c.py): log-targets, single-level and hierarchical, against the reference golden vectors
 * (<= 1e-12 relative), and the loop against reference traces by replaying the recorded proposals/uniforms draw by draw.  Compile: see oracle/Makefile (-ffp-contract=off -mfma).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../pyhillfit_amd/csrc/phf_model.h"
#include "../pyhillfit_amd/csrc/phf_hier_model.h"
#include "../pyhillfit_amd/csrc/phf_predictive_model.h"

typedef struct {
  int32_t model;                        /* 1: (pIC50, sigma), Hill = 1;  2: (pIC50, Hill, sigma) */
  int32_t n_other, n_zero, n_hundred;   /* entries are stored in that order */
  const double* ln_conc;                /* natural log of the dose */
  const double* response;               /* (mean) response of the entry's points */
  const double* weight;                 /* number of points the entry stands for */
  double n_other_points;                /* sum of the weights of the uncensored entries */
  double ss_within;                     /* within-entry sum of squares of merged replicate points */
  double pi_bit;                        /* doseresponse.py:299-301 */
  double temperature;
} phfo_problem;

typedef struct {
  int64_t t_begin, t_end;               /* run iterations t_begin+1 .. t_end */
  int32_t thinning;
  int32_t reset_mean;                   /* PyHillTemp.py:114-115 */
  int64_t adapt_start;                  /* 1000*d (single level) or 100*dim (hierarchical) */
  uint32_t seed_lo, seed_hi;
  uint32_t chain_id, problem_id;
  const double* gamma;                  /* gamma[s] = 1/(s+1)^0.6, s >= 1 */
} phfo_run;

/* ---------------------------------------------------------------- model library (shared source: phf_model.h) */
static void target_parts(const phfo_problem* pb, const double* th, double* lik, double* prior, double* ll1) {
  phf_sl_log_target(pb->model, pb->ln_conc, pb->response, pb->weight, pb->n_other, pb->n_zero + pb->n_hundred,
                    pb->n_other_points, pb->ss_within, pb->pi_bit, pb->temperature, th, phf_k_exp, phf_k_log, lik, prior, ll1);
}

double phfo_log_prior(const phfo_problem* pb, const double* th) { double l, p, a; target_parts(pb, th, &l, &p, &a); return p; }

double phfo_log_likelihood(const phfo_problem* pb, const double* th) { double l, p, a; target_parts(pb, th, &l, &p, &a); return l; }

/* log_data_likelihood at temperature 1 whatever the problem's own temperature (compute_bayes_factors.py:19-20) */
double phfo_log_likelihood_t1(const phfo_problem* pb, const double* th) { double l, p, a; target_parts(pb, th, &l, &p, &a); return a; }

double phfo_log_target(const phfo_problem* pb, const double* th) {  /* doseresponse.py:187-189 */
  double l, p, a; target_parts(pb, th, &l, &p, &a); return l + p;
}

/* ---------------------------------------------------------------- proposal factor */
/* lower Cholesky factor of a packed lower-triangular covariance (row-major), semi-definite safe:
 * a non-positive pivot zeroes its column (numpy's SVD-based sampler accepts such matrices too). */
static void chol_packed(int d, const double* c, double* l) {
  double inv[128];
  for (int i = 0; i < d; ++i) {
    for (int j = 0; j <= i; ++j) {
      double s = c[i * (i + 1) / 2 + j];
      for (int k = 0; k < j; ++k) s = phf_fma(-l[i * (i + 1) / 2 + k], l[j * (j + 1) / 2 + k], s);
      if (i == j) {
        const double sp = __builtin_fmax(s, 0.0);
        const double r = phf_sqrt(sp);
        l[i * (i + 1) / 2 + i] = r;
        inv[i] = (sp > 0.0) ? 1.0 / r : 0.0;
      } else {
        l[i * (i + 1) / 2 + j] = s * inv[j];
      }
    }
  }
}

/* state layout: th[d], lt, mean[d], cov[d(d+1)/2], loga, n_accepted, untempered log-likelihood of the current state */
int phfo_state_size(int d) { return 2 * d + d * (d + 1) / 2 + 4; }

void phfo_init_state(const phfo_problem* pb, int cov_identity, double cov_scale, const double* theta0, double* st) {
  const int d = (pb->model == 1) ? 2 : 3;
  double* th = st; double* lt = st + d; double* mean = st + d + 1; double* cov = mean + d;
  for (int i = 0; i < d; ++i) { th[i] = theta0[i]; mean[i] = theta0[i]; }   /* PyHillFit.py:750 */
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j)                                          /* PyHillFit.py:751 / PyHillTemp.py:80 */
      cov[i * (i + 1) / 2 + j] = (i != j) ? 0.0 : (cov_identity ? cov_scale : cov_scale * __builtin_fabs(theta0[i]));
  *lt = phfo_log_target(pb, th);                                          /* PyHillFit.py:789 */
  cov[d * (d + 1) / 2] = 0.0;      /* loga   :796 */
  cov[d * (d + 1) / 2 + 1] = 0.0;  /* accepted count */
  cov[d * (d + 1) / 2 + 2] = phfo_log_likelihood_t1(pb, th);
}

/* Advance one chain of any dimension.  out_rows receives (theta, log-target) for every t with
 * t % thinning == 0, densely.  If star_replay/u_replay are given the proposals and uniforms come from them
 * (reference trace replay) instead of Philox.  scaled_cov_trace, if given, receives exp(loga)*cov (d*d, full)
 * as handed to the proposal at each iteration (what the reference passes to multivariate_normal). */
#define PHFO_MAX_DIM 128
typedef double (*target_fn)(const void* ctx, const double* th, double* ll1);

static void advance_generic(int d, target_fn target, const void* ctx, const phfo_run* run, double* st,
                            double* out_rows, const double* star_replay, const double* u_replay, double* scaled_cov_trace) {
  const int ntri = d * (d + 1) / 2;
  double* th = st; double* lt = st + d; double* mean = st + d + 1; double* cov = mean + d;
  double* loga = cov + ntri; double* nacc = loga + 1; double* ll1 = loga + 2;
  static __thread double L[PHFO_MAX_DIM * (PHFO_MAX_DIM + 1) / 2];
  double z[PHFO_MAX_DIM + 4], star[PHFO_MAX_DIM], v[PHFO_MAX_DIM];
  chol_packed(d, cov, L);
  double sc = phf_exp_fast(0.5 * *loga);
  int64_t row = 0;
  for (int64_t t = run->t_begin + 1; t <= run->t_end; ++t) {
    if (scaled_cov_trace) {
      const double e = phf_exp(*loga);
      double* o = scaled_cov_trace + (t - run->t_begin - 1) * d * d;
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) o[i * d + j] = e * cov[(i >= j) ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i];
    }
    double log_u;
    if (star_replay) {
      for (int i = 0; i < d; ++i) star[i] = star_replay[(t - run->t_begin - 1) * d + i];
      log_u = phf_log(u_replay[t - run->t_begin - 1]);
    } else {
      /* PyHillFit.py:831/485 — theta* ~ N(theta, e^loga cov) drawn as theta + e^(loga/2) L z */
      log_u = phf_mh_draws(d, run->chain_id, run->problem_id, (uint32_t)t, run->seed_lo, run->seed_hi, phf_k_log, z);
      for (int i = 0; i < d; ++i) {
        double yv = L[i * (i + 1) / 2 + i] * z[i];
        for (int k = i - 1; k >= 0; --k) yv = phf_fma(L[i * (i + 1) / 2 + k], z[k], yv);
        star[i] = phf_fma(sc, yv, th[i]);
      }
    }
    double ll1_star;
    const double lt_star = target(ctx, star, &ll1_star);                     /* :833 */
    const int acc = log_u < lt_star - *lt;                                   /* :834-838 */
    if (acc) { for (int i = 0; i < d; ++i) th[i] = star[i]; *lt = lt_star; *ll1 = ll1_star; }
    *nacc += (double)acc;
    if (run->reset_mean && t == run->adapt_start)                            /* PyHillTemp.py:114-115 */
      for (int i = 0; i < d; ++i) mean[i] = th[i];
    if (t > run->adapt_start) {                                              /* :840-846 / :495-501 */
      const double g = run->gamma[t - run->adapt_start];
      const double omg = 1.0 - g;
      for (int i = 0; i < d; ++i) v[i] = th[i] - mean[i];
      for (int i = 0; i < d; ++i)
        for (int j = 0; j <= i; ++j) cov[i * (i + 1) / 2 + j] = phf_fma(g, v[i] * v[j], omg * cov[i * (i + 1) / 2 + j]);
      for (int i = 0; i < d; ++i) mean[i] = phf_fma(g, th[i], omg * mean[i]);
      *loga = phf_fma(g, (double)acc - 0.25, *loga);
      chol_packed(d, cov, L);
      sc = phf_exp_fast(0.5 * *loga);
    }
    if (t % run->thinning == 0) {                                            /* :847-848 / :502-503 */
      for (int i = 0; i < d; ++i) out_rows[row * (d + 1) + i] = th[i];
      out_rows[row * (d + 1) + d] = *lt;
      ++row;
    }
  }
}

static double sl_target(const void* ctx, const double* th, double* ll1) {
  double l, p; target_parts((const phfo_problem*)ctx, th, &l, &p, ll1); return l + p;
}

void phfo_advance(const phfo_problem* pb, const phfo_run* run, double* st, double* out_rows,
                  const double* star_replay, const double* u_replay, double* scaled_cov_trace) {
  advance_generic((pb->model == 1) ? 2 : 3, sl_target, pb, run, st, out_rows, star_replay, u_replay, scaled_cov_trace);
}

/* ---------------------------------------------------------------- hierarchical model (PyHillFit.py:113-193,429-511) */
typedef struct {
  int32_t n_expts;
  const int32_t* expt_start;            /* [n_expts+1] */
  const double* ln_conc;
  const double* response;
  phf_hier_prior prior;
} phfo_hier_problem;

double phfo_hier_log_target(const phfo_hier_problem* pb, const double* th) {
  return phf_hier_log_target_any(pb->n_expts, pb->expt_start, pb->ln_conc, pb->response, th, 1, &pb->prior, phf_k_exp, phf_k_log);
}

/* state: th[d], lt, mean[d], L[d(d+1)/2] (the adapted covariance as L diag(d) L': unit lower L packed row-major, d_i in the
 * diagonal slots), loga, n_accepted */
void phfo_hier_init_state(const phfo_hier_problem* pb, double cov_scale, const double* theta0, double* st) {
  const int d = 5 + 2 * pb->n_expts;
  double* th = st; double* lt = st + d; double* mean = st + d + 1; double* L = mean + d;
  for (int i = 0; i < d; ++i) { th[i] = theta0[i]; mean[i] = theta0[i]; }   /* PyHillFit.py:433,442 */
  for (int i = 0; i < d; ++i)                                               /* diag(0.01 |theta0|), :431: L = I, d = the diagonal */
    for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = (i != j) ? 0.0 : cov_scale * __builtin_fabs(theta0[i]);
  *lt = phfo_hier_log_target(pb, th);                                       /* :447 */
  L[d * (d + 1) / 2] = 0.0;
  L[d * (d + 1) / 2 + 1] = 0.0;
}

/* The hierarchical loop (PyHillFit.py:484-511).  The covariance recursion cov <- (1-g) cov + g v v' (:498-499) is
 * carried on the factors cov = L diag(d) L' (L unit lower): d scales by 1-g, then the rank-one update of Gill, Golub, Murray &
 * Saunders (1974, method C1) with weight g and vector v, column by column. */
void phfo_hier_advance(const phfo_hier_problem* pb, const phfo_run* run, double* st, double* out_rows,
                       const double* star_replay, const double* u_replay) {
  const int d = 5 + 2 * pb->n_expts;
  const int ntri = d * (d + 1) / 2;
  double* th = st; double* lt = st + d; double* mean = st + d + 1; double* L = mean + d;
  double* loga = L + ntri; double* nacc = loga + 1;
  double z[PHFO_MAX_DIM + 4], star[PHFO_MAX_DIM], w[PHFO_MAX_DIM];
  double sc = phf_exp_fast(0.5 * *loga);
  int64_t row = 0;
  for (int64_t t = run->t_begin + 1; t <= run->t_end; ++t) {
    double log_u;
    if (star_replay) {
      for (int i = 0; i < d; ++i) star[i] = star_replay[(t - run->t_begin - 1) * d + i];
      log_u = phf_log(u_replay[t - run->t_begin - 1]);
    } else {
      log_u = phf_hier_draws(d, run->chain_id, run->problem_id, (uint32_t)t, run->seed_lo, run->seed_hi, phf_k_log, z, 1);
      for (int i = 0; i < d; ++i) {                                        /* u = sqrt(d) z */
        const double di = L[i * (i + 1) / 2 + i];
        z[i] = ((di > 0.0) ? phf_sqrt(di) : 0.0) * z[i];
      }
      for (int i = 0; i < d; ++i) {                                        /* :485: theta* = theta + e^(loga/2) L u */
        double yv = 0.0;                                                   /* ascending k from +0, u_i last */
        for (int k = 0; k < i; ++k) yv = phf_fma(L[i * (i + 1) / 2 + k], z[k], yv);
        star[i] = phf_fma(sc, yv + z[i], th[i]);
      }
    }
    const double lt_star = phfo_hier_log_target(pb, star);                 /* :486 */
    const int acc = log_u < lt_star - *lt;                                 /* :487-492 */
    if (acc) { for (int i = 0; i < d; ++i) th[i] = star[i]; *lt = lt_star; }
    *nacc += (double)acc;
    if (t > run->adapt_start) {                                            /* :495-501 */
      const double g = run->gamma[t - run->adapt_start];
      const double omg = 1.0 - g;
      for (int i = 0; i < d; ++i) w[i] = th[i] - mean[i];
      for (int i = 0; i < d; ++i) mean[i] = phf_fma(g, th[i], omg * mean[i]);
      *loga = phf_fma(g, (double)acc - 0.25, *loga);
      double alpha = g;
      for (int k = 0; k < d; ++k) {
        const double p = w[k];
        const double dk = omg * L[k * (k + 1) / 2 + k];
        const double ap = alpha * p;
        const double dn = phf_fma(ap, p, dk);
        const double inv = (dn > 0.0) ? 1.0 / dn : 0.0;
        const double beta = ap * inv;
        alpha = (dn > 0.0) ? (alpha * dk) * inv : alpha;
        L[k * (k + 1) / 2 + k] = dn;
        for (int i = k + 1; i < d; ++i) {
          const double lik = L[i * (i + 1) / 2 + k];
          w[i] = phf_fma(-p, lik, w[i]);
          L[i * (i + 1) / 2 + k] = phf_fma(beta, w[i], lik);
        }
      }
      sc = phf_exp_fast(0.5 * *loga);
    }
    if (t % run->thinning == 0) {                                          /* :502-503 */
      for (int i = 0; i < d; ++i) out_rows[row * (d + 1) + i] = th[i];
      out_rows[row * (d + 1) + d] = *lt;
      ++row;
    }
  }
}

/* ---------------------------------------------------------------- leaf-function probes (tests) */
#define VEC1(name, fn) void name(int64_t n, const double* x, double* out) { for (int64_t i = 0; i < n; ++i) out[i] = fn(x[i]); }
VEC1(phfo_vec_exp, phf_exp)
VEC1(phfo_vec_log, phf_log)
VEC1(phfo_vec_erfcx, phf_erfcx_nonneg)
VEC1(phfo_vec_log_ndtr, phf_log_ndtr)
VEC1(phfo_vec_ndtr, phf_ndtr)
VEC1(phfo_vec_sqrt, phf_sqrt)
VEC1(phfo_vec_exp_fast, phf_exp_fast)
VEC1(phfo_vec_log_fast, phf_log_fast)
VEC1(phfo_vec_log_ndtr_nonpos, phf_log_ndtr_nonpos)
static inline double log_ndtr_tab1(double x) { return phf_log_ndtr_tab(x, -x * PHF_INV_SQRT2); }
VEC1(phfo_vec_log_ndtr_tab, log_ndtr_tab1)
VEC1(phfo_vec_erfc_tab, phf_erfc_tab)

void phfo_vec_log_ndtr_nonpos_x2(int64_t n, const double* x, double* out) {
  for (int64_t i = 0; i + 1 < n; i += 2) phf_log_ndtr_nonpos_x2(x[i], x[i + 1], &out[i], &out[i + 1]);
}

void phfo_vec_normal_u32(int64_t n, const uint32_t* w, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = phf_normal_u32(w[i]);
}

void phfo_vec_sincos(int64_t n, const uint32_t* w, double* sn, double* cs) {
  for (int64_t i = 0; i < n; ++i) phf_sincos_2pi_u32(w[i], &sn[i], &cs[i]);
}

/* rounds: 7, 10, or 0 = the rounds the samplers draw with (PHF_PHILOX_ROUNDS) */
int phfo_philox_rounds(void) { return PHF_PHILOX_ROUNDS; }
void phfo_philox(int rounds, int64_t n, const uint32_t* ctr_key /* [n][6] */, uint32_t* out /* [n][4] */) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* c = ctr_key + 6 * i;
    const phf_u32x4 r = rounds == 10 ? phf_philox4x32_10(c[0], c[1], c[2], c[3], c[4], c[5])
                      : rounds == 7 ? phf_philox4x32_7(c[0], c[1], c[2], c[3], c[4], c[5])
                                    : phf_philox_mh(c[0], c[1], c[2], c[3], c[4], c[5]);
    memcpy(out + 4 * i, r.w, 16);
  }
}

/* the normals and log(u) of iteration t of a chain, exactly as the samplers draw them */
void phfo_draws(int d, uint32_t chain_id, uint32_t problem_id, uint32_t t, uint32_t seed_lo, uint32_t seed_hi,
                double* z /* [4] */, double* log_u) {
  *log_u = phf_mh_draws(d, chain_id, problem_id, t, seed_lo, seed_hi, phf_k_log, z);
}

/* ---- posterior-predictive curves: twin of phf_predictive_accumulate (python/construct_hierarchical_cdfs.py:32-58) ----
 * rows host [num_rows][num_problems][row_stride][num_chains]; sums host [num_problems][4][grid_points], added to.
 * Same order of additions as the kernels: samples in order inside a chunk, then chunk sums left to right.       */
void phfo_predictive_accumulate(int num_problems, const double* rows, int64_t num_rows, int row_stride, int num_chains,
                                int chains_used, int grid_points, const double* hill_x, const double* pic50_x,
                                int chunk, double* sums) {
  const int64_t samples = num_rows * chains_used;
  const size_t col = (size_t)num_chains;
  /* a lane of the kernel owns grid points t and t+256 of its 512-point tile; they share one reciprocal, so the
     twin pairs them the same way */
  for (int q = 0; q < num_problems; ++q)
    for (int g0 = 0; g0 < (grid_points + 511) / 512 * 256; ++g0) {
      double lnx[2], inv_x[2], px[2];
      int g[2], live[2];
      for (int j = 0; j < 2; ++j) {
        const int gj = (g0 / 256) * 512 + j * 256 + g0 % 256;
        live[j] = gj < grid_points;
        g[j] = live[j] ? gj : grid_points - 1;
        phf_pred_hill_axis(hill_x[g[j]], &lnx[j], &inv_x[j]);
        px[j] = pic50_x[g[j]];
      }
      if (!live[0]) continue;
      double tot[2][PHF_PRED_CURVES];
      for (int j = 0; j < 2; ++j)
        for (int f = 0; f < PHF_PRED_CURVES; ++f) tot[j][f] = sums[((size_t)q * PHF_PRED_CURVES + f) * grid_points + g[j]];
      for (int64_t m0 = 0; m0 < samples; m0 += chunk) {
        const int64_t m1 = m0 + chunk < samples ? m0 + chunk : samples;
        double acc[2][PHF_PRED_CURVES] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        for (int64_t m = m0; m < m1; ++m) {
          const int64_t r = m / chains_used;
          const int c = (int)(m - r * chains_used);
          const double* p = rows + ((size_t)(r * num_problems + q) * row_stride) * col + c;
          double lna, b, mu, inv_s;
          phf_pred_prepare(p[0], p[col], p[2 * col], p[3 * col], &lna, &b, &mu, &inv_s);
          phf_pred_accumulate2(lna, b, mu, inv_s, lnx, inv_x, px, phf_k_exp, acc);
        }
        for (int j = 0; j < 2; ++j)
          for (int f = 0; f < PHF_PRED_CURVES; ++f) tot[j][f] += acc[j][f];
      }
      for (int j = 0; j < 2; ++j) {
        if (!live[j]) continue;
        for (int f = 0; f < PHF_PRED_CURVES; ++f) {
          double v = tot[j][f];
          if ((f == 0 || f == 2) && !(hill_x[g[j]] > 0.0)) v = 0.0;
          sums[((size_t)q * PHF_PRED_CURVES + f) * grid_points + g[j]] = v;
        }
      }
    }
}
