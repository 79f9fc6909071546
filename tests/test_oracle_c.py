"""Pin the scalar C twin (oracle/phf_oracle.c) to the reference: golden log-targets (<= 1e-12) and the
reference's own loop traces replayed draw by draw; then its Philox-driven sampler statistically."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import c_oracle as co
from oracle import pyhillfit_oracle as orc


def _close(a, b, rtol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    inf = ~np.isfinite(a) | ~np.isfinite(b)
    assert np.array_equal(a[inf], b[inf], equal_nan=True)
    np.testing.assert_allclose(a[~inf], b[~inf], rtol=rtol, atol=0)


def test_log_target_against_reference_golden(golden_meta, oracle_pair):
    g = np.load(os.path.join(GOLDEN, "g1_log_target.npz"))
    packs = {}
    n = len(g["target"])
    lik, pri, tgt = np.empty(n), np.empty(n), np.empty(n)
    for k in range(n):
        model, t, ip = int(g["model"][k]), float(g["t"][k]), int(g["pair"][k])
        key = (ip, model, t)
        if key not in packs:
            m = golden_meta["g1_pairs"][ip]
            p = oracle_pair(m["drug"], m["channel"])
            packs[key] = co.PackedPair(p.concs, p.responses, model, t)
        th = g["theta"][k] if model == 2 else g["theta"][k][[0, 2]]
        lik[k], pri[k], tgt[k] = packs[key].log_likelihood(th), packs[key].log_prior(th), packs[key].log_target(th)
    # NaN policy differs harmlessly: reference gives nan for (lik=-inf)+(prior=+...)? no: both give -inf or finite
    _close(lik, g["lik"], 1e-12); _close(pri, g["prior"], 1e-12); _close(tgt, g["target"], 1e-12)


@pytest.mark.parametrize("run", ["amio_m2_t1", "amio_m1_t1", "amio_m2_t0125", "amio_m2_t0", "bepr_m2_t1", "moxi_m1_t1"])
def test_loop_replays_reference_trace(run, golden_meta, oracle_pair):
    """Feed the reference's recorded proposals and uniforms (PyHillTemp.do_mcmc) through the C loop:
    same accept/reject at every iteration, same chain, same adapted covariance."""
    g = np.load(os.path.join(GOLDEN, "g3_traces.npz"))
    m = next(r for r in golden_meta["g3_runs"] if r["name"] == run)
    p = oracle_pair(m["drug"], m["channel"])
    pk = co.PackedPair(p.concs, p.responses, m["model"], m["temperature"])
    T, d = m["iterations"], pk.d
    st = pk.init_state(np.ones(d), True, 1.0)                    # PyHillTemp.py:63,80
    want = g[run + "_chain"]
    assert st[d] == pytest.approx(want[0, d], rel=1e-12)
    rows, cov = pk.advance(st, 0, T, 1, 1000 * d, True, co.gamma_table(T), star_replay=g[run + "_star"],
                           u_replay=g[run + "_u"], trace_cov=True)
    chain = np.vstack([want[:1], rows])
    assert np.array_equal(np.diff(chain[:, 0]) != 0, np.diff(want[:, 0]) != 0)
    _close(chain[:, :d], want[:, :d], 0)                         # accepted proposals are copied: bit-exact
    _close(chain[:, d], want[:, d], 1e-12)
    np.testing.assert_allclose(cov[::20], g[run + "_cov_every20"], rtol=1e-10)
    np.testing.assert_allclose(cov[-1], g[run + "_cov_last"], rtol=1e-10)


def test_segmented_advance_is_bit_identical(oracle_pair):
    p = oracle_pair("Amiodarone", "hERG")
    pk = co.PackedPair(p.concs, p.responses, 2, 1.0)
    gam = co.gamma_table(6000)
    st1 = pk.init_state([6.0, 0.7, 8.0], False, 0.05)
    one = pk.advance(st1, 0, 6000, 5, 3000, False, gam, seed=25, chain_id=3, problem_id=1)
    st2 = pk.init_state([6.0, 0.7, 8.0], False, 0.05)
    parts = [pk.advance(st2, a, b, 5, 3000, False, gam, seed=25, chain_id=3, problem_id=1)
             for a, b in ((0, 1234), (1234, 3000), (3000, 3001), (3001, 6000))]
    assert np.array_equal(one, np.vstack(parts)) and np.array_equal(st1, st2)
    assert one.shape == (1200, 4)


def _run_chains(pk, theta0, n_chains, T, thin, adapt, reset, cov_identity, cov_scale, seed=25):
    gam = co.gamma_table(T)
    out = []
    for c in range(n_chains):
        st = pk.init_state(theta0, cov_identity, cov_scale)
        rows = pk.advance(st, 0, T, thin, adapt, reset, gam, seed=seed, chain_id=c)
        out.append(rows[rows.shape[0] // 4:])
    return np.concatenate(out)


def test_prior_only_rung_known_answer(oracle_pair):
    """t = 0: the target is the prior (doseresponse.py:230-231): pIC50 ~ -3+Exp(5), Hill ~ U(0,10),
    sigma ~ 1e-3 + Gamma(5, 1.49975) — a reference-independent known answer."""
    p = oracle_pair("Amiodarone", "hERG")
    pk = co.PackedPair(p.concs, p.responses, 2, 0.0)
    s = _run_chains(pk, np.ones(3), 24, 400000, 5, 3000, True, True, 1.0)   # exponential tail: needs long chains
    mean, sd = s[:, :3].mean(0), s[:, :3].std(0)
    np.testing.assert_allclose(mean, [2.0, 5.0, 7.49975], rtol=0.04)
    np.testing.assert_allclose(sd, [5.0, 10 / np.sqrt(12), np.sqrt(5) * 1.49975], rtol=0.05)


def test_posterior_matches_reference_long_run(oracle_pair):
    """G5: posterior moments of the reference do_mcmc (200k iterations) vs 16 Philox chains of the twin."""
    with open(os.path.join(GOLDEN, "g5_posteriors.json")) as f:
        g5 = json.load(f)
    for want in g5:
        if (want["drug"], want["model"], want["temperature"]) not in (("Amiodarone", 2, 1.0), ("Amiodarone", 1, 1.0), ("Bepridil", 2, 1.0)):
            continue
        p = oracle_pair(want["drug"], want["channel"])
        pk = co.PackedPair(p.concs, p.responses, want["model"], want["temperature"])
        s = _run_chains(pk, np.ones(pk.d), 16, 200000, 5, 1000 * pk.d, True, True, 1.0)
        # the reference figure is ONE chain of 30 001 autocorrelated rows: allow 1 % plus 3 of its own
        # Monte-Carlo standard errors (ESS ~ rows/20, measured spread of single-chain means)
        ref_se = np.array(want["sd"][:pk.d]) / np.sqrt(want["rows"] / 20.0)
        assert np.all(np.abs(s.mean(0)[:pk.d] - want["mean"][:pk.d]) <= 0.01 * np.abs(want["mean"][:pk.d]) + 3 * ref_se)
        # quantiles are the robust shape check; a single chain's sd moves by +-5..25 % with rare Hill excursions
        for q, key in ((0.05, "q05"), (0.5, "q50"), (0.95, "q95")):
            np.testing.assert_allclose(np.quantile(s[:, :pk.d], q, axis=0), want[key][:pk.d], rtol=0.02)
        np.testing.assert_allclose(s.std(0)[:pk.d], want["sd"][:pk.d], rtol=0.2)
        assert s.mean(0)[pk.d] == pytest.approx(want["mean"][pk.d], abs=0.08)


def test_twin_agrees_with_numpy_oracle_on_all_pairs(g4_pairs, oracle_pair):
    """every Crumb pair, both models, at the packed ordering the kernels use"""
    rng = np.random.default_rng(3)
    for (d, c) in list(g4_pairs)[::7]:
        p = oracle_pair(d, c)
        for model in (1, 2):
            pk = co.PackedPair(p.concs, p.responses, model, 1.0)
            for _ in range(5):
                th = np.array([rng.uniform(3, 9), rng.uniform(0.3, 3), rng.uniform(1, 20)])
                params = th if model == 2 else th[[0, 2]]
                assert pk.log_target(params) == pytest.approx(orc.log_target(model, p, params, 1.0), rel=1e-12)


# ------------------------------------------------------------------------------------------------ hierarchical
def test_hierarchical_target_against_reference_golden(golden_meta, oracle_pair):
    """G2: PyHillFit.log_target_distribution, Ne = 3..6 and the synthetic Ne = 5 / 50 sets, support edges included."""
    g = np.load(os.path.join(GOLDEN, "g2_hier_target.npz"))
    shapes, scales, locs = orc.hierarchical_prior_params()
    for ip, m in enumerate(golden_meta["g2_pairs"]):
        p = oracle_pair(m["drug"], m["channel"], m["file"])
        pk = co.PackedHierPair(p.experiments, shapes, scales, locs)
        got = np.array([pk.log_target(th) for th in g["theta_%d" % ip]])
        _close(got, g["target_%d" % ip], 1e-12)
        assert np.isinf(g["target_%d" % ip]).sum() >= 8


@pytest.mark.parametrize("drug,channel", [("Amiodarone", "hERG"), ("Amiodarone", "Kv4.3"), ("Dofetilide", "hERG"), ("Amitriptyline", "Kv4.3")])
def test_hierarchical_target_across_the_tail_cut_against_the_independent_oracle(drug, channel, oracle_pair):
    """The kernels' target takes an upper truncation tail with (100 - pred)/(sigma sqrt2) >= 6 as zero, logs the PRODUCT of a
    half's truncation masses and the product of the deferred log-logistic / logistic terms (phf_hier_model.h).  Against the numpy
    restatement of PyHillFit.py:113-193 — term by term, scipy's ndtr, no shared source — on parameter vectors whose sigma sweeps
    the points of every experiment across the cut (Ne = 3, 4, 5, 6), and on vectors whose powers overflow the product (the
    term-by-term fallback): 1e-12 like the goldens."""
    shapes, scales, locs = orc.hierarchical_prior_params()
    p = oracle_pair(drug, channel)
    pk = co.PackedHierPair(p.experiments, shapes, scales, locs)
    ne = len(p.experiments)
    rng = np.random.RandomState(20 + ne)
    crossed = [0, 0]
    for k in range(400):
        sigma = float(np.exp(rng.uniform(np.log(0.3), np.log(40.0))))
        pic50s, hills = rng.uniform(3.0, 8.0, ne), np.exp(rng.uniform(np.log(0.3), np.log(3.0), ne))
        th = np.concatenate([[rng.uniform(0.3, 3.0), rng.uniform(2.1, 9.0), rng.uniform(3.0, 8.0), rng.uniform(0.05, 1.5)],
                             np.column_stack([pic50s, hills]).ravel(), [sigma]])
        want = orc.hier_log_target(p.experiments, th, shapes, scales, locs)
        got = pk.log_target(th)
        assert got == pytest.approx(want, rel=1e-12, abs=1e-12), (k, th)
        for i, ex in enumerate(p.experiments):                                  # which side of the cut were this vector's points on?
            pred = orc.hill_curve(ex[:, 0], hills[i], orc.ic50_of(pic50s[i]))
            yb = (100.0 - pred) / sigma / np.sqrt(2.0)
            crossed[0] += int((yb >= 6.0).sum()); crossed[1] += int((yb < 6.0).sum())
    assert min(crossed) > 200                                                    # both sides well covered
    # every (Hill_i / alpha)^beta = 1e120 and every exp(-z_i) = e^300: each term finite, their products beyond 2^1000 -> the
    # term-by-term path
    th = np.concatenate([[0.5, 120.0, 8.0, 0.01 + 1.0 / 150.0], np.tile([6.0, 5.0], ne), [5.0]])
    want = orc.hier_log_target(p.experiments, th, shapes, scales, locs)
    assert np.isfinite(want) and want < -1000 and pk.log_target(th) == pytest.approx(want, rel=1e-12)


def test_hierarchical_factor_update_is_the_reference_covariance_recursion(oracle_pair):
    """the twin carries the covariance as L diag(d) L' and applies cov <- (1-g) cov + g v v' (PyHillFit.py:498-499) as a rank-one
    update of the factors: L D L' must equal the covariance obtained by iterating the reference formula on the same accepted states"""
    p = oracle_pair("Amiodarone", "hERG")
    shapes, scales, locs = orc.hierarchical_prior_params()
    pk = co.PackedHierPair(p.experiments, shapes, scales, locs)
    d = pk.dim
    theta0 = np.array([1., 5., 6., .3, 6., .8, 6.1, .7, 5.9, .9, 8.])
    T, adapt = 3000, 400
    gam = co.gamma_table(T)
    st = pk.init_state(theta0, 0.01)
    assert st[d] == pytest.approx(-28.722039448904166, rel=1e-13)              # SURVEY probe value
    rows = pk.advance(st, 0, T, 1, adapt, gam, seed=7, chain_id=2, problem_id=5)
    chain = np.vstack([np.concatenate([theta0, [st[d] * 0 + pk.log_target(theta0)]]), rows])
    cov = np.diag(0.01 * np.abs(theta0)); mean = theta0.copy()
    for t in range(1, T + 1):
        th = chain[t, :d]
        if t > adapt:
            gs = gam[t - adapt]
            v = (th - mean)[None, :]
            cov = (1 - gs) * cov + gs * np.dot(v.T, v)
            mean = (1 - gs) * mean + gs * th
    np.testing.assert_allclose(co.hier_state_covariance(st, d), cov, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(st[d + 1:2 * d + 1], mean, rtol=1e-12)
    assert 0.05 < st[-1] / T < 0.7 and (np.diff(chain[:, 0]) != 0).sum() > 200


def test_hierarchical_factor_update_with_a_zero_variance_direction(oracle_pair):
    """a start point with a component that is exactly 0 gives the reference's initial covariance 0.01 diag|theta0| (PyHillFit.py:431)
    a zero row and column: that component is never proposed away from its start, v_k stays 0 and the reference's recursion keeps the
    row at zero — a frozen component, in the reference as here.  The L diag(d) L' update must not divide by zero on the way
    (dn = 0 leaves the column and alpha alone) and must give the recursion's matrix everywhere else"""
    p = oracle_pair("Amiodarone", "hERG")
    shapes, scales, locs = orc.hierarchical_prior_params()
    pk = co.PackedHierPair(p.experiments, shapes, scales, locs)
    d = pk.dim
    theta0 = np.array([1., 5., 6., .3, 6., .8, 6.1, .7, 0.0, .9, 8.])             # pIC50_3 = 0: inside the support (pIC50_i >= -2)
    T, adapt = 1500, 50
    gam = co.gamma_table(T)
    st = pk.init_state(theta0, 0.01)
    assert np.isfinite(st[d]) and co.hier_state_covariance(st, d)[8, 8] == 0.0
    rows = pk.advance(st, 0, T, 1, adapt, gam, seed=11, chain_id=1, problem_id=3)
    chain = np.vstack([np.concatenate([theta0, [pk.log_target(theta0)]]), rows])
    assert np.all(chain[:, 8] == 0.0)                                             # no variance in that direction: it cannot move
    cov = np.diag(0.01 * np.abs(theta0)); mean = theta0.copy()
    for t in range(1, T + 1):
        th = chain[t, :d]
        if t > adapt:
            gs = gam[t - adapt]
            v = (th - mean)[None, :]
            cov = (1 - gs) * cov + gs * np.dot(v.T, v)
            mean = (1 - gs) * mean + gs * th
    got = co.hier_state_covariance(st, d)
    assert np.isfinite(got).all() and np.all(got[8] == 0.0) and np.all(cov[8] == 0.0)
    np.testing.assert_allclose(got, cov, rtol=1e-8, atol=1e-13)
    assert (np.diff(chain[:, 0]) != 0).sum() > 50


def test_hierarchical_posterior_is_consistent_with_reference_samples(oracle_pair):
    """coarse statistical pin (the reference's stored chaste/samples: 500 draws of an unseeded run):
    top-level (alpha, mu) of Amiodarone-hERG and Dofetilide-hERG"""
    with open(os.path.join(GOLDEN, "chaste_alpha_mu_stats.json")) as f:
        ref = json.load(f)
    shapes, scales, locs = orc.hierarchical_prior_params()
    for drug, start in (("Amiodarone", [1., 5., 6., .3]), ("Dofetilide", [1., 5., 8.5, .3])):
        p = oracle_pair(drug, "hERG")
        pk = co.PackedHierPair(p.experiments, shapes, scales, locs)
        ne = pk.n_expts
        theta0 = np.concatenate([start, np.tile([start[2], 0.8], ne), [8.0]])
        T = 150000
        gam = co.gamma_table(T)
        samples = []
        for c in range(4):
            st = pk.init_state(theta0, 0.01)
            rows = pk.advance(st, 0, T, 5, 100 * pk.dim, gam, seed=25, chain_id=c)
            samples.append(rows[len(rows) // 4:])
        s = np.concatenate(samples)
        w = ref["%s_hERG" % drug]
        assert abs(s[:, 0].mean() - w["alpha_mean"]) < 0.12 * w["alpha_mean"] + 3 * w["alpha_sd"] / np.sqrt(w["n"] / 10.)
        assert abs(s[:, 2].mean() - w["mu_mean"]) < 0.01 * abs(w["mu_mean"]) + 3 * w["mu_sd"] / np.sqrt(w["n"] / 10.)
        assert s[:, 0].std() == pytest.approx(w["alpha_sd"], rel=0.35) and s[:, 2].std() == pytest.approx(w["mu_sd"], rel=0.35)


@pytest.mark.parametrize("name", ["posterior_like", "wide_and_edges", "single_sample"])
def test_predictive_twin_matches_reference_function(name):
    """the C twin of phf_predictive_accumulate against golden G7 (reference function, scipy.stats arithmetic)"""
    g = np.load(os.path.join(GOLDEN, "g7_predictive_cdfs.npz"))
    s = g[name + "_samples"]
    rows = s[:, None, :, None]
    for chunk in (50, 4096):
        sums = co.predictive_accumulate(rows, 1, g[name + "_hill_x"], g[name + "_pic50_x"], chunk)[0] / len(s)
        for f, k in enumerate(("hill_cdf", "pic50_cdf", "hill_pdf", "pic50_pdf")):
            ref = g[name + "_" + k]
            assert np.all(np.abs(sums[f] - ref) <= 1e-70 + 1e-11 * np.abs(ref)), (name, k, chunk)
        assert sums[0][0] == 0.0 and sums[2][0] == 0.0


# ---- G9: the reference's inlined single-level and hierarchical loops (tests/golden/make_golden_loops.py) through the C twin -----
def _g9():
    import json
    with open(os.path.join(GOLDEN, "g9_loop_traces_meta.json")) as f:
        return np.load(os.path.join(GOLDEN, "g9_loop_traces.npz")), json.load(f)


@pytest.mark.parametrize("run", ["sl_amio_m2", "sl_amio_m1", "sl_moxi_m2"])
def test_single_level_loop_replays_reference_statements(run, oracle_pair):
    """python/PyHillFit.py:748-751,787-856 executed from the reference's own statements: its recorded proposals and uniforms
    through the C loop (start = given point, covariance 0.05 diag|theta0|, no mean reset: the kernels' PyHillFit mode) give the
    same accept/reject at every iteration, the same chain and the same adapted covariance"""
    g, meta = _g9()
    m = next(r for r in meta["single_level"] if r["name"] == run)
    p = oracle_pair(m["drug"], m["channel"])
    pk = co.PackedPair(p.concs, p.responses, m["model"], 1.0)
    T, d = m["iterations"], pk.d
    st = pk.init_state(np.array(m["theta0"]), False, 0.05)        # :748-751
    want = g[run + "_chain"]
    assert st[d] == pytest.approx(want[0, d], rel=1e-12)
    rows, cov = pk.advance(st, 0, T, 1, 1000 * d, False, co.gamma_table(T), star_replay=g[run + "_star"], u_replay=g[run + "_u"], trace_cov=True)
    chain = np.vstack([want[:1], rows])
    assert np.array_equal(np.diff(chain[:, 0]) != 0, np.diff(want[:, 0]) != 0)
    _close(chain[:, :d], want[:, :d], 0)
    _close(chain[:, d], want[:, d], 1e-12)
    np.testing.assert_allclose(cov[::20], g[run + "_cov_every20"], rtol=1e-10)
    np.testing.assert_allclose(cov[-1], g[run + "_cov_last"], rtol=1e-10)
    final = g[run + "_final"]                                     # cov_estimate, mean_estimate, loga, acceptance after the last iteration
    tri = st[2 * d + 1:2 * d + 1 + d * (d + 1) // 2]
    full = np.zeros((d, d)); full[np.tril_indices(d)] = tri; full = full + np.tril(full, -1).T
    np.testing.assert_allclose(full.ravel(), final[:d * d], rtol=1e-10)
    np.testing.assert_allclose(st[d + 1:2 * d + 1], final[d * d:d * d + d], rtol=1e-12)
    assert st[2 * d + 1 + d * (d + 1) // 2] == pytest.approx(final[-2], rel=1e-11)            # loga
    assert st[2 * d + 2 + d * (d + 1) // 2] / T == pytest.approx(final[-1], rel=1e-12)        # accepted count / T = running acceptance


@pytest.mark.parametrize("run", ["hier_amio", "hier_amit"])
def test_hierarchical_loop_replays_reference_statements(run, oracle_pair):
    """python/PyHillFit.py:431-511 from the reference's own statements (Ne = 3 and 6): same accept sequence and chain; the factors the
    twin carries (L, d) multiply out to the reference's final covariance"""
    g, meta = _g9()
    m = next(r for r in meta["hierarchical"] if r["name"] == run)
    p = oracle_pair(m["drug"], m["channel"])
    shapes, scales, locs = orc.hierarchical_prior_params()
    pk = co.PackedHierPair(p.experiments, shapes, scales, locs)
    T, d = m["iterations"], pk.dim
    theta0 = np.array(m["theta0"])
    st = pk.init_state(theta0, 0.01)                              # :431
    want = g[run + "_chain"]
    assert st[d] == pytest.approx(want[0, d], rel=1e-12)
    rows = pk.advance(st, 0, T, 1, 100 * d, co.gamma_table(T), star_replay=g[run + "_star"], u_replay=g[run + "_u"])
    chain = np.vstack([want[:1], rows])
    assert np.array_equal(np.diff(chain[:, 0]) != 0, np.diff(want[:, 0]) != 0) and (np.diff(want[:, 0]) != 0).sum() > 300
    _close(chain[:, :d], want[:, :d], 0)
    _close(chain[:, d], want[:, d], 1e-11)
    final = g[run + "_final"]
    np.testing.assert_allclose(co.hier_state_covariance(st, d), final[:d * d].reshape(d, d), rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(st[d + 1:2 * d + 1], final[d * d:d * d + d], rtol=1e-12)
    assert st[2 * d + 1 + d * (d + 1) // 2] == pytest.approx(final[-2], rel=1e-11)
