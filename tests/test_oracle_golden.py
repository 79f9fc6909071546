"""Pin the numpy oracle (oracle/pyhillfit_oracle.py) to vectors produced by the reference itself.

Golden files come from tests/golden/make_golden.py (reference executed in memory)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import pyhillfit_oracle as orc


def _same(a, b, rtol=1e-13):
    a, b = np.asarray(a, float), np.asarray(b, float)
    inf = np.isinf(a) | np.isinf(b) | np.isnan(a) | np.isnan(b)
    assert np.array_equal(a[inf], b[inf], equal_nan=True)
    np.testing.assert_allclose(a[~inf], b[~inf], rtol=rtol, atol=0)


def test_pairs_match_reference_loader(g4_pairs, oracle_pair):
    assert len(g4_pairs) == 210
    for (d, c), g in g4_pairs.items():
        p = oracle_pair(d, c)
        assert np.array_equal(p.concs, g["concs"]) and np.array_equal(p.responses, g["responses"])
        assert (int(p.is0.sum()), int(p.is100.sum()), int(p.other.sum())) == (g["n0"], g["n100"], g["n_other"])
        assert p.pi_bit == g["pi_bit"]
        assert [len(e) for e in p.experiments] == g["expt_sizes"]


def test_hill_curve_values():
    g = np.load(os.path.join(GOLDEN, "g1_log_target.npz"))
    for (pic50, hill), want in zip(g["curve_in"], g["curve_out"]):
        got = orc.hill_curve(g["curve_doses"], hill, orc.ic50_of(pic50))
        assert np.array_equal(got, want)


def test_log_target_single_level(golden_meta, oracle_pair):
    """G1: dr.log_target / log_data_likelihood / log_priors, models 1 and 2, 4 temperatures, 10 pairs."""
    g = np.load(os.path.join(GOLDEN, "g1_log_target.npz"))
    pairs = [oracle_pair(m["drug"], m["channel"]) for m in golden_meta["g1_pairs"]]
    n = len(g["target"])
    lik, pri, tgt = np.empty(n), np.empty(n), np.empty(n)
    for k in range(n):
        model, t, th = int(g["model"][k]), float(g["t"][k]), g["theta"][k]
        params = th if model == 2 else th[[0, 2]]
        p = pairs[int(g["pair"][k])]
        with np.errstate(all="ignore"):
            lik[k] = orc.log_likelihood(model, p, params, t)
            pri[k] = orc.log_prior(model, params)
        tgt[k] = orc.log_target(model, p, params, t)
    assert np.isinf(g["target"]).sum() > 100          # the edge rows are really exercised
    _same(lik, g["lik"]); _same(pri, g["prior"]); _same(tgt, g["target"])
    # the survey's probe values
    amio = pairs[0]
    assert orc.log_target(2, amio, np.array([6., 1., 5.]), 1) == pytest.approx(-58.39140921642633, rel=1e-14)
    assert orc.log_target(2, amio, np.array([6., 1., 5.]), 0) == pytest.approx(1.9037293660251158, rel=1e-14)


def test_reference_cost_profile_likelihood_is_the_same_function(golden_meta, oracle_pair):
    """bench.py's cpu_baseline times the loop through scipy.stats.norm.logcdf/logsf, the calls of doseresponse.py:244-245;
    it must be the same function as the oracle's lean form (every 7th G1 row, both models, all temperatures, edges)."""
    g = np.load(os.path.join(GOLDEN, "g1_log_target.npz"))
    pairs = [oracle_pair(m["drug"], m["channel"]) for m in golden_meta["g1_pairs"]]
    idx = np.arange(0, len(g["target"]), 7)
    got = np.empty(len(idx))
    for j, k in enumerate(idx):
        model, t, th = int(g["model"][k]), float(g["t"][k]), g["theta"][k]
        got[j] = orc.log_target(model, pairs[int(g["pair"][k])], th if model == 2 else th[[0, 2]], t, as_reference=True)
    _same(got, g["target"][idx], rtol=1e-12)


def test_hierarchical_prior_params():
    g = np.load(os.path.join(GOLDEN, "g2_hier_target.npz"))
    shapes, scales, locs = orc.hierarchical_prior_params()
    assert np.array_equal(shapes, g["shapes"]) and np.array_equal(scales, g["scales"]) and np.array_equal(locs, g["locs"])


def test_hierarchical_log_target(golden_meta, oracle_pair):
    """G2: PyHillFit.log_target_distribution incl. support edges, Ne = 3..6 and synthetic Ne = 5, 50."""
    g = np.load(os.path.join(GOLDEN, "g2_hier_target.npz"))
    shapes, scales, locs = orc.hierarchical_prior_params()
    for ip, m in enumerate(golden_meta["g2_pairs"]):
        p = oracle_pair(m["drug"], m["channel"], m["file"])
        assert len(p.experiments) == m["Ne"]
        got = np.array([orc.hier_log_target(p.experiments, th, shapes, scales, locs) for th in g["theta_%d" % ip]])
        _same(got, g["target_%d" % ip], rtol=1e-12)
    amio = oracle_pair("Amiodarone", "hERG")
    v = orc.hier_log_target(amio.experiments, [1, 5, 6, .3, 6, .8, 6.1, .7, 5.9, .9, 8], shapes, scales, locs)
    assert v == pytest.approx(-28.722039448904166, rel=1e-13)


@pytest.mark.parametrize("run", ["amio_m2_t1", "amio_m1_t1", "amio_m2_t0125", "amio_m2_t0", "bepr_m2_t1", "moxi_m1_t1"])
def test_loop_trace_recorded_draws(run, golden_meta, oracle_pair):
    """G3: replay the reference do_mcmc draw by draw (recorded theta*, u): identical accept sequence,
    chain and adaptation (scaled covariance handed to the proposal) at every iteration."""
    g = np.load(os.path.join(GOLDEN, "g3_traces.npz"))
    m = next(r for r in golden_meta["g3_runs"] if r["name"] == run)
    p = oracle_pair(m["drug"], m["channel"])
    trace = {}
    chain, fin = orc.tempered_chain(m["model"], p, m["temperature"], m["iterations"], 1,
                                    orc.RecordedDraws(g[run + "_star"], g[run + "_u"]), trace=trace)
    want = g[run + "_chain"]
    assert np.array_equal(np.diff(chain[:, 0]) != 0, np.diff(want[:, 0]) != 0)     # accept sequence
    _same(chain, want, rtol=1e-13)
    covs = np.array(trace["cov"])
    np.testing.assert_allclose(covs[::20], g[run + "_cov_every20"], rtol=1e-11)
    np.testing.assert_allclose(covs[-1], g[run + "_cov_last"], rtol=1e-11)
    assert (np.diff(want[:, 0]) != 0).sum() > 100


def test_loop_trace_numpy_legacy_rng(golden_meta, oracle_pair):
    """Same loop, but drawing from numpy's legacy RandomState(1) like the reference (PyHillTemp.py:16-17):
    reproduces the reference chain draw-for-draw on this platform's LAPACK."""
    g = np.load(os.path.join(GOLDEN, "g3_traces.npz"))
    m = next(r for r in golden_meta["g3_runs"] if r["name"] == "amio_m2_t1")
    p = oracle_pair(m["drug"], m["channel"])
    chain, _ = orc.tempered_chain(2, p, 1.0, 2000, 1, orc.LegacyNumpyDraws(1))
    _same(chain, g["amio_m2_t1_chain"][:2001], rtol=1e-9)


def test_thinning_and_burn_in(golden_meta, oracle_pair):
    g = np.load(os.path.join(GOLDEN, "g3_traces.npz"))
    p = oracle_pair("Amiodarone", "hERG")
    chain, _ = orc.tempered_chain(2, p, 1.0, 1000, 5, orc.LegacyNumpyDraws(1))
    out = orc.drop_burn_in(chain, 4)
    assert out.shape == g["thin5_burn4_chain"].shape == (151, 4)
    _same(out, g["thin5_burn4_chain"], rtol=1e-9)


def test_temperature_ladder():
    lad = orc.temperature_ladder()
    assert len(lad) == 41 and lad[0] == 0 and lad[-1] == 1 and lad[1] == (1 / 40) ** 3


@pytest.mark.parametrize("name", ["posterior_like", "wide_and_edges", "single_sample"])
def test_predictive_cdfs_restatement_equals_reference_function(name):
    """G7: construct_posterior_predictive_cdfs (construct_hierarchical_cdfs.py:32-58) executed on seeded samples"""
    g = np.load(os.path.join(GOLDEN, "g7_predictive_cdfs.npz"))
    s = g[name + "_samples"]
    hx, hc, px, pc, hp, pp = orc.predictive_cdfs(s[:, 0], s[:, 1], s[:, 2], s[:, 3], block=97)
    assert np.array_equal(hx, g[name + "_hill_x"]) and np.array_equal(px, g[name + "_pic50_x"])
    for got, k in ((hc, "hill_cdf"), (pc, "pic50_cdf"), (hp, "hill_pdf"), (pp, "pic50_pdf")):
        np.testing.assert_allclose(got, g[name + "_" + k], rtol=1e-13, atol=1e-300)


def test_predictive_samples_are_inverse_cdf_draws():
    g = np.load(os.path.join(GOLDEN, "g7_predictive_cdfs.npz"))
    hx, hc, px, pc = (g["posterior_like_" + k] for k in ("hill_x", "hill_cdf", "pic50_x", "pic50_cdf"))
    hs, ps = orc.predictive_samples(hx, hc, px, pc, 2000, np.random.RandomState(1))
    u = np.random.RandomState(1).rand(4000)
    for smp, x, c, uu in ((hs, hx, hc, u[:2000]), (ps, px, pc, u[2000:])):   # Hill uniforms first, then pIC50 (:133-134)
        inside = (uu > c[0] + 1e-9) & (uu < c[-1] - 1e-9)                   # outside the tabulated range np.interp clamps
        assert inside.sum() > 1900
        assert np.allclose(np.interp(smp[inside], x, c), uu[inside], atol=1e-9)


# ---- G9: the reference's two INLINED loops, executed from its own statements (tests/golden/make_golden_loops.py) ----------------
def _g9():
    with open(os.path.join(GOLDEN, "g9_loop_traces_meta.json")) as f:
        return np.load(os.path.join(GOLDEN, "g9_loop_traces.npz")), json.load(f)


@pytest.mark.parametrize("run", ["sl_amio_m2", "sl_amio_m1", "sl_moxi_m2"])
def test_single_level_loop_replays_the_reference_statements(run, oracle_pair):
    """python/PyHillFit.py:748-751,787-856 (covariance 0.05 diag|theta0|, NO mean reset, adaptation after 1000 d): the recorded
    proposals and uniforms through the oracle's loop give the reference's accept sequence, chain, adapted covariance at every
    iteration, and its final (cov, mean, loga, acceptance)"""
    g, meta = _g9()
    m = next(r for r in meta["single_level"] if r["name"] == run)
    assert m["reference_lines"] == [[748, 751], [787, 864]] and m["when_to_adapt"] == 1000 * (m["model"] + 1)
    p = oracle_pair(m["drug"], m["channel"])
    theta0 = np.array(m["theta0"])
    cov0 = 0.05 * np.diag(np.abs(theta0))
    trace = {}
    chain, fin = orc.adaptive_mh(lambda th: orc.log_target(m["model"], p, th, 1), theta0, cov0, m["iterations"], 1, m["when_to_adapt"],
                                 orc.RecordedDraws(g[run + "_star"], g[run + "_u"]), trace=trace)
    want = g[run + "_chain"]
    assert np.array_equal(np.diff(chain[:, 0]) != 0, np.diff(want[:, 0]) != 0) and (np.diff(want[:, 0]) != 0).sum() > 500
    _same(chain, want, rtol=1e-13)
    covs = np.array(trace["cov"])
    np.testing.assert_allclose(covs[::20], g[run + "_cov_every20"], rtol=1e-11)
    np.testing.assert_allclose(covs[-1], g[run + "_cov_last"], rtol=1e-11)
    d = len(theta0)
    final = g[run + "_final"]
    np.testing.assert_allclose(fin["cov"].ravel(), final[:d * d], rtol=1e-11)
    np.testing.assert_allclose(fin["mean"], final[d * d:d * d + d], rtol=1e-12)
    assert fin["loga"] == pytest.approx(final[-2], rel=1e-12) and fin["acceptance"] == pytest.approx(final[-1], rel=1e-12)
    # the convenience wrapper is that same loop
    chain2, _ = orc.single_level_chain(m["model"], p, theta0, m["iterations"], 1, orc.RecordedDraws(g[run + "_star"], g[run + "_u"]))
    assert np.array_equal(chain2, chain)


def test_single_level_loop_thinning_burn_in_and_seed(oracle_pair):
    """:805-814,847-848,861-864 (thinning 5, burn-in = saved/4 rows) and :824-825 (the loop seeds numpy with 25 itself)"""
    g, meta = _g9()
    m = next(r for r in meta["single_level"] if r["name"] == "sl_amio_m2_thin5_burn4")
    p = oracle_pair(m["drug"], m["channel"])
    chain, _ = orc.single_level_chain(2, p, np.array(m["theta0"]), m["iterations"], 5, orc.LegacyNumpyDraws(25))
    out = orc.drop_burn_in(chain, 4)
    want = g["sl_amio_m2_thin5_burn4_chain"]
    assert out.shape == want.shape == (751, 4) and m["rows"] == 751
    _same(out, want, rtol=1e-9)                                   # numpy's own generator: draw for draw up to LAPACK's SVD


@pytest.mark.parametrize("run", ["hier_amio", "hier_amit"])
def test_hierarchical_loop_replays_the_reference_statements(run, oracle_pair):
    """python/PyHillFit.py:431-511 (covariance 0.01 diag|theta0|, adaptation after 100 dim), Ne = 3 and Ne = 6"""
    g, meta = _g9()
    m = next(r for r in meta["hierarchical"] if r["name"] == run)
    assert m["reference_lines"] == [[431, 511]]
    p = oracle_pair(m["drug"], m["channel"])
    theta0 = np.array(m["theta0"])
    dim = len(theta0)
    assert m["when_to_adapt"] == 100 * dim and dim == 5 + 2 * m["Ne"]
    shapes, scales, locs = orc.hierarchical_prior_params()
    trace = {}
    chain, fin = orc.adaptive_mh(lambda th: orc.hier_log_target(p.experiments, th, shapes, scales, locs), theta0, np.diag(0.01 * np.abs(theta0)),
                                 m["iterations"], 1, 100 * dim, orc.RecordedDraws(g[run + "_star"], g[run + "_u"]), trace=trace)
    want = g[run + "_chain"]
    assert np.array_equal(np.diff(chain[:, 0]) != 0, np.diff(want[:, 0]) != 0) and (np.diff(want[:, 0]) != 0).sum() > 300
    _same(chain, want, rtol=1e-12)
    covs = np.array(trace["cov"])
    np.testing.assert_allclose(covs[::50], g[run + "_cov_every50"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(covs[-1], g[run + "_cov_last"], rtol=1e-10, atol=1e-300)
    final = g[run + "_final"]
    np.testing.assert_allclose(fin["cov"].ravel(), final[:dim * dim], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(fin["mean"], final[dim * dim:dim * dim + dim], rtol=1e-12)
    assert fin["loga"] == pytest.approx(final[-2], rel=1e-12)
    chain2, _ = orc.hierarchical_chain(p.experiments, theta0, m["iterations"], 1, orc.RecordedDraws(g[run + "_star"], g[run + "_u"]))
    assert np.array_equal(chain2, chain)


def test_hierarchical_loop_thinning_and_burn(oracle_pair):
    """:469-472,502-503: saved_iterations = T/thin + 1, burn = saved/4 (only used for the alpha/mu subsample), full chain kept"""
    g, meta = _g9()
    m = next(r for r in meta["hierarchical"] if r["name"] == "hier_amio_thin5")
    assert m["saved_iterations"] == m["iterations"] // 5 + 1 == m["rows"] and m["burn"] == m["saved_iterations"] // 4
    p = oracle_pair(m["drug"], m["channel"])
    chain, _ = orc.hierarchical_chain(p.experiments, np.array(m["theta0"]), m["iterations"], 5, orc.LegacyNumpyDraws(1))
    _same(chain, g["hier_amio_thin5_chain"], rtol=1e-8)


def test_g10_and_g5d_fixtures_are_what_their_generators_say():
    """G10 (tests/golden/make_golden_posteriors_hier.py): 6 pairs x 4 seeds of the reference's lifted hierarchical loop (PyHillFit.py:431-511),
    every column; the seeds of a pair agree with each other within their own standard errors (the fixture is self-consistent), the
    acceptance sits at the loop's target.  G10b (--per-drug): one pair of every drug by the rule in the generator, seeds 201 and 202.
    G5d (make_golden_posteriors_reseed.py, round-4 protocol): the SAME 96 reference chains (seeds 1..96) for every one of the six cases,
    and what the reseeding found — the seed-1 chain golden G5c holds for Ranolazine-Nav1.5-peak sits far out among its own siblings."""
    with open(os.path.join(GOLDEN, "g10_hier_posteriors.json")) as f:
        g10 = json.load(f)
    assert [(e["drug"], e["Ne"]) for e in g10] == [("Amiodarone", 3), ("Amiodarone", 4), ("Dofetilide", 5), ("Amitriptyline", 6), ("Sertindole", 3), ("Cibenzoline", 6)]
    for e in g10:
        assert e["iterations"] == 500000 and e["thinning"] == 5 and len(e["runs"]) == 4 and e["dim"] == 5 + 2 * e["Ne"]
        assert all(r["reference_lines"] == [[431, 511]] and r["rows"] == 75001 and 0.23 < r["acceptance"] < 0.27 for r in e["runs"])
        means = np.array([r["mean"] for r in e["runs"]]); ses = np.array([r["batch_means_se"] for r in e["runs"]])
        assert means.shape == (4, e["dim"] + 1)
        z = np.abs(means - means.mean(axis=0)) / np.maximum(ses, 1e-12)
        assert z.max() < 6.0, (e["drug"], e["channel"], float(z.max()))
        assert np.allclose(e["pooled"]["mean"], means.mean(axis=0))
    with open(os.path.join(GOLDEN, "g10b_hier_posteriors_per_drug.json")) as f:
        g10b = json.load(f)
    with open(os.path.join(GOLDEN, "g4_pairs.json")) as f:
        g4 = json.load(f)["pairs"]
    drugs, channels = [], []
    for w in g4:                                                    # the data file's order
        if w["drug"] not in drugs: drugs.append(w["drug"])
        if w["channel"] not in channels: channels.append(w["channel"])
    in_g10 = {(e["drug"], e["channel"]) for e in g10}
    want = []
    for i, d in enumerate(drugs):                                   # the generator's rule: drug i takes channel i mod 7, the next one if that pair is in G10
        k = i % len(channels)
        while (d, channels[k]) in in_g10:
            k = (k + 1) % len(channels)
        want.append((d, channels[k]))
    assert [(e["drug"], e["channel"]) for e in g10b] == want and len(want) == 30
    for e in g10b:
        assert e["iterations"] == 500000 and e["thinning"] == 5 and [r["seed"] for r in e["runs"]] == [201, 202] and e["dim"] == 5 + 2 * e["Ne"]
        assert all(r["reference_lines"] == [[431, 511]] and r["rows"] == 75001 and 0.23 < r["acceptance"] < 0.27 for r in e["runs"])
        means = np.array([r["mean"] for r in e["runs"]]); ses = np.array([r["batch_means_se"] for r in e["runs"]])
        z = np.abs(means[0] - means[1]) / np.sqrt(ses[0] ** 2 + ses[1] ** 2)
        assert z.max() < 5.0, (e["drug"], e["channel"], float(z.max()))     # the two seeds agree within their own errors (observed max 3.4)
        assert np.allclose(e["pooled"]["se_between_seeds"], np.abs(means[0] - means[1]) / 2)
    path = os.path.join(GOLDEN, "g10c_hier_posteriors_all_remaining.json")          # G10c: every pair in neither G10 nor G10b, seeds 301, 302
    if os.path.exists(path):
        with open(path) as f:
            g10c = json.load(f)
        pinned = in_g10 | set(want)
        rest = [(d, c) for d in drugs for c in channels if (d, c) not in pinned]
        mine = [(e["drug"], e["channel"]) for e in g10c]
        assert len(rest) == 174 and mine == [pc for pc in rest if pc in set(mine)]   # a subset of the remaining pairs, in the data file's order
        # the pairs whose two reference chains disagree with EACH OTHER (by more than 5 of their own batch-means standard errors: up to 55 —
        # a steep experiment's Hill_i lives in the heavy tail of the log-logistic level, and one 500 000-iteration chain does not cover it)
        # are among the eight the follow-up G10d gives ten seeds (make_golden_posteriors_hier.py: FOLLOW_UP)
        follow_up = {("Azithromycin", "Kir2.1"), ("Mexiletine", "Nav1.5-peak"), ("Moxifloxacin", "Cav1.2"), ("Nilotinib", "Kir2.1"),
                     ("Ondansetron", "Kir2.1"), ("Propafenone", "Kir2.1"), ("Ranolazine", "Cav1.2"), ("Dofetilide", "Cav1.2")}
        disagree = set()
        for e in g10c:
            assert e["iterations"] == 500000 and e["thinning"] == 5 and [r["seed"] for r in e["runs"]] == [301, 302] and e["dim"] == 5 + 2 * e["Ne"]
            assert all(r["reference_lines"] == [[431, 511]] and r["rows"] == 75001 and 0.23 < r["acceptance"] < 0.27 for r in e["runs"])
            means = np.array([r["mean"] for r in e["runs"]]); ses = np.array([r["batch_means_se"] for r in e["runs"]])
            z = np.abs(means[0] - means[1]) / np.sqrt(ses[0] ** 2 + ses[1] ** 2)
            if z.max() >= 5.0:
                disagree.add((e["drug"], e["channel"]))
        assert len(mine) == 174 and disagree and disagree <= follow_up, disagree - follow_up
        path = os.path.join(GOLDEN, "g10d_hier_posteriors_follow_up.json")         # G10d: those eight pairs, seeds 311..318
        if os.path.exists(path):
            with open(path) as f:
                g10d = json.load(f)
            assert {(e["drug"], e["channel"]) for e in g10d} <= follow_up
            assert all([r["seed"] for r in e["runs"]] == list(range(311, 319)) and e["iterations"] == 500000 for e in g10d)
    with open(os.path.join(GOLDEN, "g5d_posteriors_reseeded.json")) as f:
        g5d = json.load(f)
    assert len(g5d) == 6 and all(e["seeds"] == list(range(1, 97)) and e["iterations"] == 200000 for e in g5d)      # the same seeds for every case
    path = os.path.join(GOLDEN, "g5e_posterior_widths_reseeded.json")               # G5e: the width follow-up, 12 cases x seeds 1..32
    if os.path.exists(path):
        with open(path) as f:
            g5e = json.load(f)
        assert len(g5e) == 12 and all(e["seeds"] == list(range(1, 33)) and e["iterations"] == 200000 and e["temperature"] == 1.0 for e in g5e)
        assert sorted(e["model"] for e in g5e) == [1] * 3 + [2] * 9
        assert not {(e["drug"], e["channel"], e["model"]) for e in g5e} & {(e["drug"], e["channel"], e["model"]) for e in g5d}
    rano = [e for e in g5d if e["drug"] == "Ranolazine"][0]
    p = np.array([r["mean"][0] for r in rano["runs"]])
    with open(os.path.join(GOLDEN, "g5c_posteriors_all_pairs_model_2.json")) as f:
        g5c = [w for w in json.load(f) if (w["drug"], w["channel"]) == ("Ranolazine", "Nav1.5-peak")][0]
    assert abs(p[0] - g5c["mean"][0]) < 1e-9                        # seed 1 = the chain G5c holds ...
    assert np.mean(p >= p[0]) <= 0.10 and (p[0] - p[1:].mean()) / (p[1:].std(ddof=1) / np.sqrt(len(p) - 1)) > 3.0   # ... among the highest of the 96, far from their mean
