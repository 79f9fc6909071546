"""Start points ON THE GPU BOX (-m gpu): the least-squares finder that stands in for the reference's CMA-ES search
(python/PyHillFit.py:699-735 single level, :243-257,310-336 hierarchical) is host numpy by design; golden G8 pins it in the CPU suite.
Here the same pin runs on the GPU box's own host (another CPU, another BLAS / SIMD path than the build container's), and what it
produces is handed to the device the way the command lines do it: every pair's start point must be a point of FINITE log-target for the
HIP kernels, and the value the device gives it must be the one the reference's objective implies."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pyhillfit_amd import _lib
    _lib.load()
    return "cuda:0"


@pytest.fixture(scope="module")
def dr(gpu):
    from pyhillfit_amd import doseresponse as d
    d.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    return d


def test_single_level_start_points_of_every_pair_on_this_host_and_on_the_device(gpu, dr):
    """G8 = the reference's own sum_of_square_diffs (PyHillFit.py:93-97) on a dense (pIC50, Hill) grid for all 210 pairs and both
    models.  On this box: the batched fit's SS is <= the reference grid's minimum for every pair, sigma0 = sqrt(SS / N) (:101-102,729);
    the device evaluates the model-m log-likelihood at that point to exactly what SS implies for an uncensored pair —
    -N/2 ln 2 pi - N ln sigma0 - SS / (2 sigma0^2) (doseresponse.py:229-248) — and to a finite value for every pair.  The PRIOR at the
    start is -inf exactly where the least-squares Hill coefficient lies above the prior's upper bound of 10 (the reference's search
    has no upper bound either, PyHillFit.py:726: Hill = x^2 + lower): such chains start at a log-target of -inf like the
    reference's, and must leave it with their first acceptable proposal — checked by running them."""
    from pyhillfit_amd import bestfit
    from pyhillfit_amd.sampler import log_target_batch
    with open(os.path.join(GOLDEN, "g8_least_squares_grid.json")) as f:
        g8 = json.load(f)
    names = [(w["drug"], w["channel"]) for w in g8["pairs"]]
    assert len(names) == 210
    pairs = [dr.concatenate_experiments(*dr.load_crumb_data(d, c)[::2]) for d, c in names]
    packed = dr.pack_single_level(names)
    for model in (2, 1):
        theta, ss = bestfit.best_fit_batch(pairs, model)
        grid_min = np.array([w["model_%d" % model]["grid_min_ss"] for w in g8["pairs"]])
        assert np.all(ss <= grid_min * (1 + 1e-12) + 1e-20), (model, float(np.max(ss - grid_min)))
        n = np.array([w["n"] for w in g8["pairs"]])
        sig = np.sqrt(ss / n)
        assert np.all(theta[:, -1] == np.where(sig > dr.sigma_loc, sig, 1.0))
        lik, prior = log_target_batch(packed, model, list(range(210)), [1.0] * 210, theta, device=gpu)
        assert np.all(np.isfinite(lik)), (model, [names[k] for k in np.flatnonzero(~np.isfinite(lik))])
        outside = np.flatnonzero(~np.isfinite(prior))
        if model == 1:
            assert len(outside) == 0
        else:                                                                   # log_priors_model_2 (doseresponse.py:175-184): Hill outside [0, 10]
            assert np.array_equal(outside, np.flatnonzero(theta[:, 1] > 10.0)), ([names[k] for k in outside], theta[outside])
            assert np.all(prior[outside] == -np.inf) and 0 < len(outside) < 30  # 4 of the 210 pairs; Mexiletine-Nav1.5-peak fits Hill ~ 25
            from pyhillfit_amd.sampler import SingleLevelSampler
            sub = dr.pack_single_level([names[k] for k in outside])

            def run_from(th0):
                s = SingleLevelSampler(sub, 2, list(range(len(outside))), [1.0] * len(outside), 64, thinning=5, seed=25, device=gpu)
                s.init(th0, cov_identity=False, cov_scale=0.05)                 # PyHillFit.py:748-751
                first = s.log_target().cpu().numpy().copy()
                s.advance(5000, save=False)
                return first, s.log_target().cpu().numpy(), s.theta().cpu().numpy()
            # started AT the fit, like the reference: a log-target of -inf, and the steepest fit never finds its way into the support
            first, last, _ = run_from(theta[outside])
            assert np.all(first == -np.inf)
            stuck = [names[k] for q, k in enumerate(outside) if not np.isfinite(last[q]).any()]
            print("fits with Hill > 10:", [(names[k], round(float(theta[k, 1]), 2)) for k in outside], "stuck at -inf for 5 000 iterations:", stuck)
            assert ("Mexiletine", "Nav1.5-peak") in stuck
            # started where the command line starts them (bestfit.chain_start: Hill moved onto the prior's bound): finite from the start
            start = bestfit.chain_start(theta, model)
            assert np.array_equal(np.delete(start, outside, axis=0), np.delete(theta, outside, axis=0)) and np.all(start[outside, 1] == 10.0)
            first, last, th_end = run_from(start[outside])
            assert np.all(np.isfinite(first)) and np.all(np.isfinite(last)) and np.all(th_end[1] <= 10.0) and np.all(th_end[1] >= 0.0)
            assert np.array_equal(bestfit.chain_start(theta, 1), theta)        # model 1 has no Hill column: nothing to move
        checked = 0
        for k, (concs, y) in enumerate(pairs):
            y = np.asarray(y)
            if np.all((y > 0) & (y < 100)) and sig[k] > dr.sigma_loc:           # no censored and no ignored points: a plain Gaussian
                want = -0.5 * len(y) * np.log(2 * np.pi) - len(y) * np.log(sig[k]) - ss[k] / (2 * sig[k] ** 2)
                assert lik[k] == pytest.approx(want, rel=1e-9, abs=1e-9), (names[k], model)
                checked += 1
        assert checked >= 40                                                    # 52 of the 210 pairs have no censored response


def test_hierarchical_start_points_of_every_pair_are_in_the_device_targets_support(gpu, dr):
    """hierarchical_first_iteration_batch (PyHillFit.py:243-257 per-experiment fits, :310-336 the two distribution fits) for all 210
    pairs on this host, then the device's log_target_distribution (:173-193) at every one of them: finite — a start outside the support
    (alpha <= 0, s <= 0.01, a Hill_i < 0 ...) would pin its chains at -inf for the whole run."""
    from pyhillfit_amd import bestfit
    from pyhillfit_amd import hierarchical as H
    shapes, scales, locs = H.prior_params()
    groups = {}
    for d in dr.drugs:
        for c in dr.channels:
            ne, _, ex = dr.load_crumb_data(d, c)
            groups.setdefault(len(ex), []).append(((d, c), ex))
    total = 0
    for ne, members in sorted(groups.items()):
        exs = [m[1] for m in members]
        starts = np.array(bestfit.hierarchical_first_iteration_batch(exs, locs))
        assert starts.shape == (len(exs), 5 + 2 * ne)
        lt = H.log_target_batch(H.PackedHierPoints(exs), list(range(len(exs))), starts, device=gpu)
        bad = [members[k][0] for k in np.flatnonzero(~np.isfinite(lt))]
        assert not bad, bad
        for k in range(0, len(exs), 17):                                       # a pair's start does not depend on its batch
            assert np.array_equal(starts[k], bestfit.hierarchical_first_iteration(exs[k], locs))
        total += len(exs)
    assert total == 210
