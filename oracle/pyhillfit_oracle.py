"""CPU oracle (numpy/scipy) for the PyHillFit Metropolis-Hastings hot path.

TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  Only tests/, __graft_entry__.smoke()
and bench.py's ``cpu_baseline`` leg may import this module.  The shipped sampler is the
HIP library behind include/pyhillfit_amd.h; it never routes through this file.

What this is: an independent restatement, in plain numpy + scipy.special, of the
reference algorithm (mirams/PyHillFit, python/doseresponse.py, python/PyHillFit.py,
python/PyHillTemp.py), each function citing the reference file:line it follows.
It is PINNED: tests/test_oracle_golden.py checks it against golden vectors produced by
executing the reference itself (tests/golden/make_golden.py) — log-targets, hierarchical
targets and draw-by-draw loop traces.

Third-party arithmetic behind the reference that is not under /root/reference:
scipy.stats.norm.logcdf/logsf/cdf (-> scipy.special.log_ndtr/ndtr) and numpy's legacy
RandomState.multivariate_normal/rand; versions used for the goldens are in
tests/golden/VERSIONS.json (the reference pins none).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
from scipy.special import log_ndtr, ndtr

# ---- constants: doseresponse.py:12-25 -------------------------------------------------
SIGMA_FLOOR = 1e-3            # sigma_uniform_lower (:12) — likelihood returns -inf at or below it
PIC50_RATE = 0.2              # pic50_exp_rate (:14)
PIC50_LOWER = -3.0            # pic50_exp_lower (:16)
HILL_LOWER, HILL_UPPER = 0.0, 10.0   # (:17-18)
SIGMA_SHAPE = 5.0             # (:22)
SIGMA_MODE = 6.0              # (:23)
SIGMA_LOC = 1e-3              # (:24)
SIGMA_SCALE = (SIGMA_MODE - SIGMA_LOC) / (SIGMA_SHAPE - 1.0)   # (:25)
LADDER_N, LADDER_C = 40, 3    # (:27-28)

# ---- hierarchical hyper-prior constants: PyHillFit.py:301,340-364 (Elkins et al. values) -
ELKINS_HILL_ALPHAS = np.array([1.188, 1.744, 1.530, 0.930, 0.605, 1.325, 1.179, 0.979, 1.790, 1.708, 1.586,
                               1.469, 1.429, 1.127, 1.011, 1.318, 1.063])
ELKINS_HILL_BETAS = 1. / np.array([0.0835, 0.1983, 0.2089, 0.1529, 0.1206, 0.2386, 0.2213, 0.2263, 0.1784,
                                   0.1544, 0.2486, 0.2031, 0.2025, 0.1510, 0.1837, 0.1677, 0.0862])
ELKINS_PIC50_MUS = np.array([5.235, 5.765, 6.060, 5.315, 5.571, 7.378, 7.248, 5.249, 6.408, 5.625, 7.321, 6.852,
                             6.169, 6.217, 5.927, 7.414, 4.860])
ELKINS_PIC50_SIGMAS = np.array([0.0760, 0.1388, 0.1459, 0.2044, 0.1597, 0.2216, 0.1856, 0.1560, 0.1034, 0.1033,
                                0.1914, 0.1498, 0.1464, 0.1053, 0.1342, 0.1808, 0.0860])
HIER_PIC50_LOWER = -2.0       # pic50_prior[0], PyHillFit.py:215


def hierarchical_prior_params():
    """(shapes, scales, locs) of the five Gamma hyper-priors — PyHillFit.py:301,349-364."""
    locs = np.array([0., 2., -4., 0.01, SIGMA_LOC])
    modes = np.array([ELKINS_HILL_ALPHAS.mean(), ELKINS_HILL_BETAS.mean() - 2., ELKINS_PIC50_MUS.mean(),
                      ELKINS_PIC50_SIGMAS.mean(), SIGMA_MODE])
    shapes = np.array([5., 2.5, 7.5, 2.5, SIGMA_SHAPE])
    return shapes, (modes - locs) / (shapes - 1.), locs


# ---- data ------------------------------------------------------------------------------
@dataclass
class PairData:
    """One (drug, channel) pair as the single-level drivers see it (PyHillFit.py:661-683)."""
    concs: np.ndarray
    responses: np.ndarray
    experiments: Optional[List[np.ndarray]] = None   # per-experiment (n_i,2) arrays, hierarchical path
    is0: np.ndarray = field(init=False)
    is100: np.ndarray = field(init=False)
    other: np.ndarray = field(init=False)
    pi_bit: float = field(init=False)

    def __post_init__(self):
        y = self.responses
        self.is0 = y == 0                         # PyHillFit.py:675
        self.is100 = y == 100                     # :676
        self.other = (0 < y) & (y < 100)          # :677  (y<0 or y>100 fall in no mask)
        # doseresponse.py:299-301 is handed the *mask*, so len() = N_total (PyHillFit.py:683)
        self.pi_bit = 0.5 * len(self.other) * np.log(2 * np.pi)


def pair_from_rows(experiment, dose, response):
    """Group rows like doseresponse.load_crumb_data (:60-67) + concatenate like PyHillFit.py:661-665."""
    experiment = np.asarray(experiment); dose = np.asarray(dose, float); response = np.asarray(response, float)
    seen = []
    for e in experiment:               # pandas .unique(): order of first appearance
        if e not in seen:
            seen.append(e)
    expts = [np.column_stack([dose[experiment == e], response[experiment == e]]) for e in seen]
    num_expts = max(seen)              # doseresponse.py:62 (max label, not count)
    expts_used = expts[:num_expts]     # `for i in xrange(num_expts): experiments[i]`
    return PairData(np.concatenate([x[:, 0] for x in expts_used]), np.concatenate([x[:, 1] for x in expts_used]), expts)


# ---- model library (doseresponse.py) -----------------------------------------------------
def hill_curve(dose, hill, ic50):
    """doseresponse.py:84-85 — percent block."""
    return 100. * (1. - 1. / (1. + (1. * dose / ic50) ** hill))


def ic50_of(pic50):
    """doseresponse.py:87-88 — IC50 in uM."""
    return 10 ** (6 - pic50)


def gamma_logprior(x, shape, scale, loc):
    """doseresponse.py:304-317 — unnormalised shifted-Gamma log density; -inf left of loc."""
    if np.any(x < loc):
        return -np.inf
    with np.errstate(divide="ignore"):
        return (shape - 1) * np.log(x - loc) - (x - loc) / scale


def pic50_logprior(x):
    """doseresponse.py:151-156."""
    return -np.inf if x < PIC50_LOWER else -PIC50_RATE * x


def log_prior(model, params):
    """doseresponse.py:166-172 (model 1) / :175-184 (model 2)."""
    if model == 1:
        pic50, sigma = params
    else:
        pic50, hill, sigma = params
        if hill < HILL_LOWER or hill > HILL_UPPER:
            return -np.inf
    return pic50_logprior(pic50) + gamma_logprior(sigma, SIGMA_SHAPE, SIGMA_SCALE, SIGMA_LOC)


def log_likelihood(model, pair, params, t):
    """Censored-Gaussian data log-likelihood x temperature.

    doseresponse.py:203-226 (model 1, Hill fixed to 1) and :229-248 (model 2)."""
    if t == 0:
        return 0
    if model == 1:
        pic50, sigma = params
        hill = 1
    else:
        pic50, hill, sigma = params
    if sigma <= SIGMA_FLOOR:
        return -np.inf
    pred = hill_curve(pair.concs, hill, ic50_of(pic50))
    # st.norm.logcdf(0, pred, sigma) = log_ndtr((0-pred)/sigma); st.norm.logsf(100, pred, sigma) = log_ndtr(-(100-pred)/sigma)
    lo = np.sum(log_ndtr((0 - pred[pair.is0]) / sigma))
    hi = np.sum(log_ndtr(-((100 - pred[pair.is100]) / sigma)))
    n_log_sigma = pair.other.sum() * np.log(sigma)
    sse = np.sum((pair.responses[pair.other] - pred[pair.other]) ** 2 / (2. * sigma ** 2))
    return t * (lo + hi - pair.pi_bit - n_log_sigma - sse)


def log_likelihood_as_reference(model, pair, params, t):
    """The same likelihood through the reference's own library calls — scipy.stats.norm.logcdf(0, pred, sigma) and
    .logsf(100, pred, sigma) on the masked predictions, doseresponse.py:218-219 / :244-245 — so that a loop built on it has the
    reference's COST profile (the generic-distribution argument checking of scipy.stats is ~60 % of a reference
    log_target call).  Values equal log_likelihood() to rounding (tests/test_oracle_golden.py)."""
    import scipy.stats as st
    if t == 0:
        return 0
    if model == 1:
        pic50, sigma = params
        hill = 1
    else:
        pic50, hill, sigma = params
    if sigma <= SIGMA_FLOOR:
        return -np.inf
    pred = hill_curve(pair.concs, hill, ic50_of(pic50))
    lo = np.sum(st.norm.logcdf(0, pred[pair.is0], sigma))
    hi = np.sum(st.norm.logsf(100, pred[pair.is100], sigma))
    n_log_sigma = len(pair.responses[pair.other]) * np.log(sigma)
    sse = np.sum((pair.responses[pair.other] - pred[pair.other]) ** 2 / (2. * sigma ** 2))
    return t * (lo + hi - pair.pi_bit - n_log_sigma - sse)


def log_target(model, pair, params, t, as_reference=False):
    """doseresponse.py:187-189 — likelihood is always evaluated; only it is tempered."""
    with np.errstate(all="ignore"):
        lik = log_likelihood_as_reference if as_reference else log_likelihood
        return lik(model, pair, params, t) + log_prior(model, params)


def temperature_ladder(n=LADDER_N, c=LADDER_C):
    """PyHillTemp.py:151."""
    return (np.arange(n + 1.) / n) ** c


# ---- hierarchical target (PyHillFit.py:113-193) ---------------------------------------------
def hier_data_loglik(hills, pic50s, sigma, experiments):
    """PyHillFit.py:113-132 — Gaussian truncated to [0,100], per experiment."""
    total = 0.
    for i, ex in enumerate(experiments):
        conc, data = ex[:, 0], ex[:, 1]
        pred = hill_curve(conc, hills[i], ic50_of(pic50s[i]))
        sse = np.sum((data - pred) ** 2) / (2 * sigma ** 2)
        trunc = np.sum(np.log(ndtr((100 - pred) / sigma) - ndtr((0 - pred) / sigma)))
        total -= (len(conc) * np.log(sigma) + sse + trunc)
    return total


def hier_hill_logdensity(x, alpha, beta):
    """PyHillFit.py:134-142 — log-logistic(alpha, beta)."""
    return np.log(beta) - beta * np.log(alpha) + (beta - 1.) * np.log(x) - 2 * np.log(1 + (x / alpha) ** beta)


def hier_pic50_logdensity(x, mu, s):
    """PyHillFit.py:144-154 — logistic(mu, s)."""
    z = (x - mu) / s
    return -z - np.log(s) - 2 * np.log(1 + np.exp(-z))


def hier_log_target(experiments, theta, shapes, scales, locs):
    """PyHillFit.py:173-193.  theta = [alpha, beta, mu, s, pIC50_1, Hill_1, ..., sigma]."""
    theta = np.asarray(theta, float)
    if np.any(theta[:4] <= locs[:4]):
        return -np.inf
    alpha, beta, mu, s = theta[:4]
    pic50s, hills, sigma = theta[4:-1:2], theta[5:-1:2], theta[-1]
    if np.any(hills < 0) or np.any(pic50s < HIER_PIC50_LOWER) or sigma <= locs[-1]:
        return -np.inf
    with np.errstate(all="ignore"):
        total = hier_data_loglik(hills, pic50s, sigma, experiments)
        total += np.sum(hier_hill_logdensity(hills, alpha, beta))
        total += np.sum(hier_pic50_logdensity(pic50s, mu, s))
        total += np.sum(gamma_logprior(theta[[0, 1, 2, 3, -1]], shapes, scales, locs))
    return total


# ---- draw sources ----------------------------------------------------------------------------
class LegacyNumpyDraws:
    """The reference's own RNG: numpy legacy global-state semantics (PyHillFit.py:825,831,834)."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def propose(self, theta, scaled_cov):
        return self.rs.multivariate_normal(theta, scaled_cov)

    def uniform(self):
        return self.rs.rand()


class RecordedDraws:
    """Replays proposals theta* and uniforms u captured from a reference run (golden G3)."""

    def __init__(self, stars, us):
        self.stars, self.us, self.k = stars, us, 0

    def propose(self, theta, scaled_cov):
        return np.array(self.stars[self.k])

    def uniform(self):
        u = self.us[self.k]
        self.k += 1
        return u


# ---- adaptive Metropolis loop --------------------------------------------------------------
def adaptive_mh(target, theta0, cov0, iterations, thinning, adapt_start, draws,
                reset_mean_at_adapt_start=False, trace=None):
    """The reference's adaptive-Metropolis loop, shared shape of
       PyHillFit.py:796-856 (single level), :429-511 (hierarchical), PyHillTemp.py:76-123 (tempered).

    target(theta)->float; row 0 of the returned chain is the start point (:814).
    reset_mean_at_adapt_start reproduces PyHillTemp.py:114-115.
    trace, if a dict, receives per-iteration scaled covariances under 'cov'."""
    theta = np.array(theta0, float)
    d = len(theta)
    lt = target(theta)
    n_saved = iterations // thinning + 1
    chain = np.zeros((n_saved, d + 1))
    chain[0] = np.concatenate((theta, [lt]))
    loga, acceptance = 0., 0.
    mean = theta.copy()
    cov = np.array(cov0, float)
    for t in range(1, iterations + 1):
        scaled = np.exp(loga) * cov
        if trace is not None:
            trace.setdefault("cov", []).append(scaled)
        star = draws.propose(theta, scaled)
        lt_star = target(star)
        u = draws.uniform()
        with np.errstate(all="ignore"):
            ok = np.log(u) < lt_star - lt            # NaN compares False -> reject
        if ok:
            theta, lt, accepted = star, lt_star, 1
        else:
            accepted = 0
        acceptance = ((t - 1.) * acceptance + accepted) / t
        if reset_mean_at_adapt_start and t == adapt_start:
            mean = theta.copy()
        if t > adapt_start:
            s = t - adapt_start
            g = 1 / (s + 1) ** 0.6
            v = (theta - mean)[None, :]
            cov = (1 - g) * cov + g * np.dot(v.T, v)
            mean = (1 - g) * mean + g * theta
            loga += g * (accepted - 0.25)
        if t % thinning == 0:
            chain[t // thinning] = np.concatenate((theta, [lt]))
    return chain, {"loga": loga, "acceptance": acceptance, "mean": mean, "cov": cov}


def single_level_chain(model, pair, theta0, iterations, thinning, draws, temperature=1, as_reference=False):
    """PyHillFit.py:748-751,787,796-856: cov0 = 0.05*diag|theta0|, adapt after 1000*d.
    as_reference: evaluate the likelihood through scipy.stats like the reference (bench.py's cpu_baseline)."""
    theta0 = np.asarray(theta0, float)
    cov0 = 0.05 * np.diag(np.abs(theta0))
    return adaptive_mh(lambda th: log_target(model, pair, th, temperature, as_reference), theta0, cov0, iterations, thinning,
                       1000 * len(theta0), draws)


def tempered_chain(model, pair, temperature, iterations, thinning, draws, trace=None):
    """PyHillTemp.py:57-125: start at ones(d), cov0 = I, mean reset at t == 1000*d."""
    d = 2 if model == 1 else 3
    return adaptive_mh(lambda th: log_target(model, pair, th, temperature), np.ones(d), np.eye(d), iterations,
                       thinning, 1000 * d, draws, reset_mean_at_adapt_start=True, trace=trace)


def hierarchical_chain(experiments, theta0, iterations, thinning, draws):
    """PyHillFit.py:429-511: cov0 = diag(0.01*|theta0|), adapt after 100*dim."""
    theta0 = np.asarray(theta0, float)
    shapes, scales, locs = hierarchical_prior_params()
    cov0 = np.diag(0.01 * np.abs(theta0))
    return adaptive_mh(lambda th: hier_log_target(experiments, th, shapes, scales, locs), theta0, cov0, iterations,
                       thinning, 100 * len(theta0), draws)


def drop_burn_in(chain, burn_in_fraction):
    """PyHillFit.py:861-864 / PyHillTemp.py:71,125 (Python-2 integer division)."""
    return chain[chain.shape[0] // burn_in_fraction:]


# ---- posterior-predictive curves (python/construct_hierarchical_cdfs.py:32-58,133-149) ---------------------------------
PRED_GRID_POINTS = 501
PRED_HILL_RANGE = (0., 4.)
PRED_PIC50_RANGE = (-2., 12.)


def predictive_grids():
    """construct_hierarchical_cdfs.py:33-39"""
    return (np.linspace(PRED_HILL_RANGE[0], PRED_HILL_RANGE[1], PRED_GRID_POINTS),
            np.linspace(PRED_PIC50_RANGE[0], PRED_PIC50_RANGE[1], PRED_GRID_POINTS))


def predictive_cdfs(alphas, betas, mus, ss, block=2000):
    """construct_posterior_predictive_cdfs (:32-58): sample means of fisk(c=beta, scale=alpha) and logistic(mu, s) CDFs and
    PDFs on the two grids.  The reference adds one sample at a time; here blocks of samples are evaluated at once and
    added in the same order (np.add.reduce over the sample axis is pairwise, so agreement is to rounding, ~1e-15)."""
    import scipy.stats as st
    hill_x, pic50_x = predictive_grids()
    n = len(alphas)
    sums = np.zeros((4, PRED_GRID_POINTS))
    for i in range(0, n, block):
        a, b, m, s = (np.asarray(v[i:i + block], dtype=float)[:, None] for v in (alphas, betas, mus, ss))
        sums[0] += st.fisk.cdf(hill_x[None, :], c=b, scale=a, loc=0).sum(axis=0)
        sums[2] += st.fisk.pdf(hill_x[None, :], c=b, scale=a, loc=0).sum(axis=0)
        sums[1] += st.logistic.cdf(pic50_x[None, :], m, s).sum(axis=0)
        sums[3] += st.logistic.pdf(pic50_x[None, :], m, s).sum(axis=0)
    sums /= n
    return hill_x, sums[0], pic50_x, sums[1], sums[2], sums[3]


def predictive_samples(hill_x, hill_cdf, pic50_x, pic50_cdf, num_samples, rng):
    """:133-137 — inverse-CDF draws by linear interpolation: all Hill uniforms first, then all pIC50 uniforms."""
    hu = rng.rand(num_samples)
    pu = rng.rand(num_samples)
    return np.interp(hu, hill_cdf, hill_x), np.interp(pu, pic50_cdf, pic50_x)
