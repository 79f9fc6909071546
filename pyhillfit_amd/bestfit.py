"""Least-squares start point for the single-level chains — stands in for the reference's CMA-ES search
(python/PyHillFit.py:93-102,699-735; `cma` is an external dependency that is not part of the sampling path).

Same objective and parametrisation as the reference: minimise sum((model - data)^2) over x with
pIC50 = x0^2 + pic50_exp_lower, Hill = x1^2 + hill_uniform_lower (:722,726), then sigma0 = sqrt(SS/N) (:101-102,729).
Deterministic: coarse grid + Nelder-Mead polish.  Deviation: when SS = 0 (all responses
zero) the reference starts at sigma = 0, a point of -inf log-target with a zero-variance sigma proposal that its
chain can never leave; here such starts use sigma0 = 1."""
import numpy as np

from . import doseresponse as dr


def minimize(*args, **kwargs):
    """scipy.optimize.minimize, imported at first use: only the scalar fits (tests, the cross-checks of the batched path) need it, and
    importing scipy.optimize + scipy.stats costs 0.3 s of a 3 s command-line run (profiles/r04/cli_host_profile_single_level.txt)"""
    from scipy.optimize import minimize as _minimize
    return _minimize(*args, **kwargs)


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


def chain_start(theta_fit, model):
    """The point a chain is STARTED from, given the least-squares fit (one row, or [P][d]).  The fit itself is what the best-fit file
    holds (PyHillFit.py:739-746) and is returned unchanged except for one case: a model-2 fit whose Hill coefficient lies above the
    prior's upper bound (doseresponse.py:18,181-182: Hill > 10 -> log-prior -inf).  The reference's search has no upper bound
    (PyHillFit.py:726: Hill = x^2 + lower), so its chain for such a pair starts at a log-target of -inf, accepts only proposals that
    land inside [0, 10] — for a steep curve fitted with Hill ~ 25 that is 13 proposal standard deviations away — and in that case
    never moves for the whole run (4 of the 210 Crumb pairs fit above 10; Mexiletine-Nav1.5-peak at 25 stays stuck).  Here the START
    is moved to the bound; the target, the proposal and the chain are the reference's.  """
    th = np.array(theta_fit, dtype=float, copy=True)
    if model == 2:
        th[..., 1] = np.minimum(th[..., 1], dr.hill_uniform_upper)
    return th


# ---- all pairs at once -------------------------------------------------------------------------------------------------
LN10 = np.log(10.0)


def _pad(pairs, width=None):
    """ragged list of (concs, responses) -> ln_conc[P][N], y[P][N], w[P][N] (1 for a point, 0 for padding), n[P]"""
    P, N = len(pairs), (max(len(c) for c, _ in pairs) if width is None else width)
    lnc = np.zeros((P, N)); y = np.zeros((P, N)); w = np.zeros((P, N))
    for k, (c, r) in enumerate(pairs):
        n = len(c)
        with np.errstate(divide="ignore"):
            lnc[k, :n] = np.log(np.asarray(c, float))
        y[k, :n] = r; w[k, :n] = 1.0
    return lnc, y, w, w.sum(axis=1)


def _curve_log(lnc, pic50, hill):
    """the Hill curve through the log-domain form the kernels use; broadcasting over leading axes"""
    with np.errstate(all="ignore"):
        a = hill * (lnc - LN10 * (6.0 - pic50))
        return 100.0 * (1.0 - 1.0 / (1.0 + np.exp(np.minimum(a, 700.0))))


def _lm(lnc, y, w, p, g, model, iterations, p_lower=None):
    """`iterations` Levenberg-Marquardt steps on (pIC50, ln Hill) for a batch of pairs; returns (p, g, SS)"""
    P = len(p)
    p_lower = dr.pic50_exp_lower if p_lower is None else p_lower

    def ss_of(p_, g_):
        return np.sum(w * (_curve_log(lnc, p_[:, None], np.exp(g_)[:, None]) - y) ** 2, axis=1)

    lam = np.full(P, 1e-3)
    cur = ss_of(p, g)
    lnc_f = np.where(w > 0, lnc, 0.0)
    for _ in range(iterations):
        hh = np.exp(g)[:, None]
        with np.errstate(all="ignore"):
            a = hh * (lnc_f - LN10 * (6.0 - p[:, None]))
            x = np.exp(np.clip(a, -700.0, 700.0))
            pred = 100.0 * x / (1.0 + x)
            s_ = np.where(np.isfinite(a), 100.0 * x / (1.0 + x) ** 2, 0.0)       # d pred / d a
            jp = s_ * hh * LN10                                                  # d a / d pIC50 = Hill ln 10
            jg = np.where(np.isfinite(a), s_ * a, 0.0)                           # d a / d ln Hill = a
        pred = np.where(np.isfinite(a), pred, np.where(a > 0, 100.0, 0.0))
        r = w * (pred - y)
        app, agg, apg = np.sum(w * jp * jp, 1), np.sum(w * jg * jg, 1), np.sum(w * jp * jg, 1)
        bp, bg = -np.sum(jp * r, 1), -np.sum(jg * r, 1)
        if model == 1:
            dp = bp / (app * (1 + lam) + 1e-300); dg = np.zeros(P)
        else:
            a11, a22 = app * (1 + lam) + 1e-300, agg * (1 + lam) + 1e-300
            det = a11 * a22 - apg * apg
            det = np.where(np.abs(det) > 1e-300, det, 1e-300)
            dp, dg = (a22 * bp - apg * bg) / det, (a11 * bg - apg * bp) / det
            # projected step: a pair sitting on the pIC50 bound with the gradient pushing outwards (flat, weakly blocking
            # data) moves along ln Hill alone — otherwise the clipped joint step stalls just above the constrained optimum
            pinned = ((p <= p_lower) & (dp < 0)) | ((p >= 20.0) & (dp > 0))
            dp = np.where(pinned, 0.0, dp); dg = np.where(pinned, bg / a22, dg)
        p_new = np.clip(p + np.clip(dp, -2.0, 2.0), p_lower, 20.0)
        g_new = np.clip(g + np.clip(dg, -1.0, 1.0), np.log(1e-4), np.log(1e3))
        trial = ss_of(p_new, g_new)
        better = trial <= cur
        p, g, cur = np.where(better, p_new, p), np.where(better, g_new, g), np.where(better, trial, cur)
        lam = np.clip(np.where(better, lam / 3.0, lam * 4.0), 1e-12, 1e12)
    return p, g, cur


def _least_squares_batch(pairs, model, p_lower):
    """(pIC50 [P], Hill [P], SS [P], n [P]) minimising sum((curve - response)^2) for every (concs, responses) of `pairs`.
    A pair's result does not depend on what else is in the batch (the multi-GPU partitions must start their chains from the
    same points): pairs are grouped by their own padded width, every reduction runs along a pair's own row, and a pair's
    iteration schedule is decided by its own progress."""
    P = len(pairs)
    width = [max(8, 1 << (len(c) - 1).bit_length()) for c, _ in pairs]
    p = np.empty(P); hill = np.empty(P); cur = np.empty(P); n = np.empty(P)
    for wd in sorted(set(width)):
        idx = [k for k in range(P) if width[k] == wd]
        p[idx], hill[idx], cur[idx], n[idx] = _least_squares_group([pairs[k] for k in idx], model, p_lower, wd)
    return p, hill, cur, n


def _least_squares_group(pairs, model, p_lower, wd):
    lnc, y, w, n = _pad(pairs, wd)
    P = len(pairs)
    p_grid = np.linspace(p_lower, 12.0, 61)
    h_grid = np.array([1.0]) if model == 1 else np.exp(np.linspace(np.log(0.05), np.log(10.0), 24))
    p = np.empty(P); h = np.empty(P)
    for k0 in range(0, P, 64):                                       # chunks bound the [pairs][61][24][N] temporary
        sl = slice(k0, min(P, k0 + 64))
        pred = _curve_log(lnc[sl, None, None, :], p_grid[None, :, None, None], h_grid[None, None, :, None])
        ss = np.sum(w[sl, None, None, :] * (pred - y[sl, None, None, :]) ** 2, axis=3)
        flat = np.argmin(ss.reshape(ss.shape[0], -1), axis=1)
        p[sl] = p_grid[flat // len(h_grid)]; h[sl] = h_grid[flat % len(h_grid)]
    g = np.log(h)
    cur = np.full(P, np.inf)
    active = np.ones(P, dtype=bool)
    for _ in range(16):                                              # blocks of 50 steps; a pair stops after a block without progress
        idx = np.nonzero(active)[0]
        if len(idx) == 0:
            break
        p2, g2, c2 = _lm(lnc[idx], y[idx], w[idx], p[idx], g[idx], model, 50, p_lower)
        active[idx] = (cur[idx] - c2) > 1e-11 * (c2 + 1e-30)
        p[idx], g[idx], cur[idx] = p2, g2, c2
    return p, np.exp(g), cur, n


def best_fit_batch(pairs, model):
    """least-squares start points of ALL pairs in one vectorised pass (replaces one scipy Nelder-Mead per pair: 1.9 s of the
    7 s `PyHillFit.py -a` run for the 210 Crumb pairs -> 0.3 s): coarse grid, then batched Levenberg-Marquardt on
    (pIC50, ln Hill) with the analytic Jacobian, every pair advancing in lock-step with its own damping.
    Same objective and conventions as best_fit (python/PyHillFit.py:93-102,699-735).  Returns (theta0 [P][d], SS [P]).
    Pinned by golden G8 (tests/golden/make_golden_bestfit.py): the reference's own objective on a dense grid."""
    p, hill, cur, n = _least_squares_batch(pairs, model, dr.pic50_exp_lower)
    sigma0 = np.sqrt(cur / n)                                        # initial_sigma, PyHillFit.py:101-102
    sigma0 = np.where(sigma0 > dr.sigma_loc, sigma0, 1.0)            # SS = 0 (all responses 0): not the reference's absorbing 0
    theta0 = np.column_stack([p, sigma0]) if model == 1 else np.column_stack([p, hill, sigma0])
    return theta0, cur


# ---- hierarchical start point (replaces the CMA-ES / scipy fits of PyHillFit.py:243-257,310-336) ---------------------
def _logistic_mle_batch(values, s_floor=1e-4, iterations=60):
    """Maximum-likelihood (mu, s) of a logistic distribution for every row of a ragged list of small samples, all rows at
    once: damped Newton on (mu, ln s) of  NLL = sum_i [ z_i + ln s + 2 ln(1 + exp(-z_i)) ],  z = (y - mu)/s.
    A sample that does not identify a scale (identical values, one value) runs into s_floor."""
    width = [max(8, 1 << (len(v) - 1).bit_length()) for v in values]
    if len(set(width)) > 1:                                          # a row's result must not depend on its batch: group by own width
        mu = np.empty(len(values)); sc = np.empty(len(values))
        for wd in sorted(set(width)):
            idx = [k for k in range(len(values)) if width[k] == wd]
            mu[idx], sc[idx] = _logistic_mle_batch([values[k] for k in idx], s_floor, iterations)
        return mu, sc
    P, N = len(values), width[0]
    y = np.zeros((P, N)); w = np.zeros((P, N))
    for k, v in enumerate(values):
        y[k, :len(v)] = v; w[k, :len(v)] = 1.0
    n = w.sum(axis=1)
    mu = (w * y).sum(axis=1) / n
    sd = np.sqrt((w * (y - mu[:, None]) ** 2).sum(axis=1) / n)
    ls = np.log(np.maximum(sd * np.sqrt(3.0) / np.pi, s_floor))

    def nll(mu_, ls_):
        s_ = np.exp(ls_)[:, None]
        z = (y - mu_[:, None]) / s_
        return (w * (z + 2.0 * np.logaddexp(0.0, -z))).sum(axis=1) + n * ls_

    cur = nll(mu, ls)
    for _ in range(iterations):
        s_ = np.exp(ls)
        z = (y - mu[:, None]) / s_[:, None]
        t = np.tanh(0.5 * z)                                         # 1 - 2/(1+e^z)
        q = 0.5 * (1.0 - t * t)                                      # 2 e^z/(1+e^z)^2
        g_mu = -(w * t).sum(axis=1) / s_                             # d NLL / d mu
        g_ls = n - (w * z * t).sum(axis=1)                           # d NLL / d ln s
        h_mm = (w * q).sum(axis=1) / s_ ** 2
        h_ml = ((w * (t + z * q)).sum(axis=1)) / s_
        h_ll = (w * (z * t + z * z * q)).sum(axis=1)
        lam = 1e-9 * (h_mm + h_ll) + 1e-300
        det = (h_mm + lam) * (h_ll + lam) - h_ml ** 2
        ok = det > 0
        d_mu = np.where(ok, -((h_ll + lam) * g_mu - h_ml * g_ls) / np.where(ok, det, 1.0), -g_mu / (h_mm + lam))
        d_ls = np.where(ok, -((h_mm + lam) * g_ls - h_ml * g_mu) / np.where(ok, det, 1.0), 0.0)
        d_ls = np.clip(d_ls, -1.0, 1.0)
        step = np.ones(P)
        for _h in range(8):                                          # backtracking, row by row
            mu_new = mu + step * d_mu
            ls_new = np.maximum(ls + step * d_ls, np.log(s_floor))
            trial = nll(mu_new, ls_new)
            worse = ~(trial <= cur)
            if not worse.any():
                break
            step = np.where(worse, 0.5 * step, step)
        better = trial <= cur
        mu, ls, cur = np.where(better, mu_new, mu), np.where(better, ls_new, ls), np.where(better, trial, cur)
    return mu, np.exp(ls)


def _hyper_start_batch(tables, locs):
    """(alpha, beta, mu, s, sigma) for every pair from its per-experiment fits (PyHillFit.py:303-334): the log-logistic fit of
    the Hill coefficients (:310-324) is the logistic fit of their logarithms — alpha = exp(location), beta = 1/scale, capped at
    20 like the scalar search it replaces — and the pIC50s get a logistic fit (:330); both for all pairs at once."""
    P = len(tables)
    sigma = np.array([np.mean(t[:, -1]) for t in tables])                                # :303-305
    sigma = np.where(sigma <= locs[3], locs[3] + 0.1, sigma)
    loc_h, scale_h = _logistic_mle_batch([np.log(np.maximum(t[:, 1], 1e-3)) for t in tables], s_floor=1.0 / 20.0)
    alpha, beta = np.exp(loc_h), np.minimum(1.0 / scale_h, 20.0)
    alpha = np.where(alpha <= locs[0], locs[0] + 0.1, alpha)
    beta = np.where(beta <= locs[1], locs[1] + 0.1, beta)
    mu, s_ = _logistic_mle_batch([t[:, 0] for t in tables])
    mu = np.where(mu <= locs[2], locs[2] + 0.1, mu)
    s_ = np.where(s_ <= locs[3], locs[3] + 0.1, s_)
    return np.column_stack([alpha, beta, mu, s_, sigma])


def _fisk_logpdf(x, c, scale):
    """log-logistic (scipy.stats.fisk) log-density, written out (the scalar cross-check of _hyper_start_batch)"""
    z = x / scale
    return np.log(c) - np.log(scale) + (c - 1.0) * np.log(z) - 2.0 * np.log1p(z ** c)


def _hyper_start(best_fits, locs):
    """one pair, with scipy's scalar optimisers: what _hyper_start_batch replaces (kept as its cross-check in the tests)"""
    sigma_cur = np.mean(best_fits[:, -1])                            # :303-305
    if sigma_cur <= locs[3]:
        sigma_cur = locs[3] + 0.1
    hills = np.maximum(best_fits[:, 1], 1e-3)
    nll = lambda x: -np.sum(_fisk_logpdf(hills, abs(x[1]) + 1e-9, abs(x[0]) + 1e-9))   # :310-324 (product of pdfs)
    res = minimize(nll, np.array([max(np.median(hills), 0.1), 3.0]), method="Nelder-Mead", options={"xatol": 1e-8, "fatol": 1e-10})
    alpha_cur, beta_cur = abs(res.x[0]), min(abs(res.x[1]), 20.0)
    if alpha_cur <= locs[0]:
        alpha_cur = locs[0] + 0.1
    if beta_cur <= locs[1]:
        beta_cur = locs[1] + 0.1
    import scipy.stats as st
    mu_cur, s_cur = st.logistic.fit(best_fits[:, 0])                 # :330
    if mu_cur <= locs[2]:
        mu_cur = locs[2] + 0.1
    if s_cur <= locs[3]:
        s_cur = locs[3] + 0.1
    return alpha_cur, beta_cur, mu_cur, s_cur, sigma_cur


def hierarchical_first_iteration_batch(experiments_per_pair, locs):
    """theta0 = [alpha, beta, mu, s, pIC50_1, Hill_1, ..., sigma] for every pair (PyHillFit.py:243-257,303-336): the
    per-experiment (pIC50, Hill) least-squares fits of ALL pairs run as one batch (700 fits for the Crumb set; they were
    85 % of the start-point time as one Nelder-Mead each), and so do the two distribution fits per pair
    (_hyper_start_batch; the scalar _hyper_start is their cross-check in the tests)."""
    for k, exs in enumerate(experiments_per_pair):
        if len(exs) == 0 or any(len(ex) == 0 for ex in exs):
            # sqrt(SS / 0) would hand the chain a NaN start sigma without a word
            raise ValueError("pair %d: %s" % (k, "no experiments" if len(exs) == 0 else "an experiment without points"))
    flat = [(ex[:, 0], ex[:, 1]) for exs in experiments_per_pair for ex in exs]
    p, hill, ss, n = _least_squares_batch(flat, 2, -2.0)           # pic50_hill_priors_lowers = (-2, 0), PyHillFit.py:218,253
    tables, k = [], 0
    for exs in experiments_per_pair:
        m = len(exs)
        tables.append(np.column_stack([p[k:k + m], hill[k:k + m], np.sqrt(ss[k:k + m] / n[k:k + m])]))   # initial_sigma, :101-102,255
        k += m
    hypers = _hyper_start_batch(tables, np.asarray(locs, float))
    return [np.concatenate((hy[:4], t[:, :-1].flatten(), [hy[4]])) for t, hy in zip(tables, hypers)]


def hierarchical_first_iteration(experiments, locs):
    """one pair (the batch of one)"""
    return hierarchical_first_iteration_batch([experiments], locs)[0]


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
