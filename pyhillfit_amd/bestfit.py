"""Least-squares start point for the single-level chains — stands in for the reference's CMA-ES search
(python/PyHillFit.py:93-102,699-735; `cma` is an external dependency that is not part of the sampling path).

Same objective and parametrisation as the reference: minimise sum((model - data)^2) over x with
pIC50 = x0^2 + pic50_exp_lower, Hill = x1^2 + hill_uniform_lower (:722,726), then sigma0 = sqrt(SS/N) (:101-102,729).
Deterministic: coarse grid + Nelder-Mead polish.  Deviation: when SS = 0 (all responses
zero) the reference starts at sigma = 0, a point of -inf log-target with a zero-variance sigma proposal that its
chain can never leave; here such starts use sigma0 = 1."""
import numpy as np
from scipy.optimize import minimize
import scipy.stats as st

from . import doseresponse as dr


def _curve(concs, pic50, hill):
    return 100. * (1. - 1. / (1. + (concs / 10 ** (6 - pic50)) ** hill))


def sum_of_square_diffs(params, concs, responses):
    """PyHillFit.py:93-97."""
    pic50, hill = params
    return np.sum((_curve(concs, pic50, hill) - responses) ** 2)


def best_fit(concs, responses, model):
    """(theta0, SS) with theta0 = (pIC50, sigma) for model 1, (pIC50, Hill, sigma) for model 2."""
    concs = np.asarray(concs, float); responses = np.asarray(responses, float)
    p_grid = np.linspace(dr.pic50_exp_lower, 12.0, 76)
    h_grid = np.array([1.0]) if model == 1 else np.exp(np.linspace(np.log(0.05), np.log(10.0), 40))
    with np.errstate(all="ignore"):
        pred = _curve(concs[None, None, :], p_grid[:, None, None], h_grid[None, :, None])
        ss = np.sum((pred - responses) ** 2, axis=2)
    ip, ih = np.unravel_index(np.argmin(ss), ss.shape)
    x0 = np.array([np.sqrt(p_grid[ip] - dr.pic50_exp_lower), np.sqrt(h_grid[ih] - dr.hill_uniform_lower)])

    if model == 1:
        obj = lambda x: sum_of_square_diffs([x[0] ** 2 + dr.pic50_exp_lower, 1.], concs, responses)
        res = minimize(obj, x0[:1], method="Nelder-Mead", options={"xatol": 1e-10, "fatol": 1e-12, "maxiter": 4000})
        pic50, hill = res.x[0] ** 2 + dr.pic50_exp_lower, 1.0
    else:
        obj = lambda x: sum_of_square_diffs(x ** 2 + [dr.pic50_exp_lower, dr.hill_uniform_lower], concs, responses)
        res = minimize(obj, x0, method="Nelder-Mead", options={"xatol": 1e-10, "fatol": 1e-12, "maxiter": 8000})
        pic50, hill = res.x ** 2 + [dr.pic50_exp_lower, dr.hill_uniform_lower]
    ss_best = float(res.fun)
    sigma0 = np.sqrt(ss_best / len(responses))              # initial_sigma, PyHillFit.py:101-102
    if not sigma0 > dr.sigma_loc:
        sigma0 = 1.0
    theta0 = np.array([pic50, sigma0]) if model == 1 else np.array([pic50, hill, sigma0])
    return theta0, ss_best


# ---- hierarchical start point (replaces the CMA-ES / scipy fits of PyHillFit.py:243-257,310-336) ---------------------
def hierarchical_first_iteration(experiments, locs):
    """theta0 = [alpha, beta, mu, s, pIC50_1, Hill_1, ..., sigma] as the reference builds it (:336), with
    deterministic least-squares / maximum-likelihood fits instead of CMA-ES."""
    best_fits = []
    for ex in experiments:                                           # :243-257  per-experiment (pIC50, Hill) fit
        th, ss = _fit_pic50_hill(ex[:, 0], ex[:, 1])
        best_fits.append([th[0], th[1], np.sqrt(ss / len(ex))])      # initial_sigma, :101-102,255
    best_fits = np.array(best_fits)
    sigma_cur = np.mean(best_fits[:, -1])                            # :303-305
    if sigma_cur <= locs[3]:
        sigma_cur = locs[3] + 0.1
    hills = np.maximum(best_fits[:, 1], 1e-3)
    nll = lambda x: -np.sum(st.fisk.logpdf(hills, c=abs(x[1]) + 1e-9, scale=abs(x[0]) + 1e-9))   # :310-324 (product of pdfs)
    res = minimize(nll, np.array([max(np.median(hills), 0.1), 3.0]), method="Nelder-Mead", options={"xatol": 1e-8, "fatol": 1e-10})
    alpha_cur, beta_cur = abs(res.x[0]), min(abs(res.x[1]), 20.0)
    if alpha_cur <= locs[0]:
        alpha_cur = locs[0] + 0.1
    if beta_cur <= locs[1]:
        beta_cur = locs[1] + 0.1
    mu_cur, s_cur = st.logistic.fit(best_fits[:, 0])                 # :330
    if mu_cur <= locs[2]:
        mu_cur = locs[2] + 0.1
    if s_cur <= locs[3]:
        s_cur = locs[3] + 0.1
    return np.concatenate(([alpha_cur, beta_cur, mu_cur, s_cur], best_fits[:, :-1].flatten(), [sigma_cur]))


def _fit_pic50_hill(concs, responses):
    """sum-of-squares fit with pIC50 >= -2, Hill >= 0 (pic50_hill_priors_lowers, PyHillFit.py:218,253)."""
    lowers = np.array([-2., 0.])
    p_grid = np.linspace(-2.0, 12.0, 57); h_grid = np.exp(np.linspace(np.log(0.05), np.log(10.0), 30))
    with np.errstate(all="ignore"):
        pred = _curve(concs[None, None, :], p_grid[:, None, None], h_grid[None, :, None])
        ss = np.sum((pred - responses) ** 2, axis=2)
    ip, ih = np.unravel_index(np.argmin(ss), ss.shape)
    x0 = np.sqrt(np.array([p_grid[ip], h_grid[ih]]) - lowers)
    res = minimize(lambda x: sum_of_square_diffs(x ** 2 + lowers, concs, responses), x0, method="Nelder-Mead",
                   options={"xatol": 1e-9, "fatol": 1e-11, "maxiter": 6000})
    return res.x ** 2 + lowers, float(res.fun)
