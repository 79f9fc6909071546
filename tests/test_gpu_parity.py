"""GPU parity tests proper (-m gpu): the HIP path, through the C ABI, against the oracle.

Bar: bit-exact against the scalar CPU twin (same Philox stream, same fixed-order fp64 arithmetic); <= 1e-12
relative against the reference's golden log-targets; posterior means within 1 % of the reference's long runs."""
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
    _lib.load()                      # fails loudly if the HIP library is missing
    return "cuda:0"


@pytest.fixture(scope="module")
def dr(gpu):
    from pyhillfit_amd import doseresponse as d
    d.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    return d


def _pair_arrays(dr, drug, channel):
    ne, _, ex = dr.load_crumb_data(drug, channel)
    return dr.concatenate_experiments(ne, ex)


# ------------------------------------------------------------------------------------------------- leaf numerics
def test_device_math_bit_identical_to_host_build(gpu):
    from oracle import c_oracle as co
    from pyhillfit_amd.sampler import debug_math
    rng = np.random.default_rng(11)
    n = 400000
    cases = {
        0: ("exp", np.concatenate([rng.uniform(-746, 710, n), rng.uniform(-3, 3, n), [0.0, -0.0, 709.782712893384, -745.13, -745.2, -744.4, np.inf, -np.inf]])),
        1: ("log", np.concatenate([np.exp(rng.uniform(-744, 709, n)), rng.uniform(0.4, 2.2, n), [0.0, 5e-324, 2.2e-308, 1.0, np.inf, -1.0]])),
        2: ("erfcx", np.concatenate([np.exp(rng.uniform(-30, 12, n)), rng.uniform(0, 10, n), [0.0, 4.0, 1e99, 1e101, 1e300]])),
        3: ("log_ndtr", np.concatenate([-np.exp(rng.uniform(-30, 12, n)), rng.uniform(-40, 40, n), [0.0, -1e5, 38.5]])),
        4: ("ndtr", rng.uniform(-40, 40, n)),
        5: ("sqrt", np.concatenate([np.exp(rng.uniform(-700, 700, n)), rng.uniform(0, 4, n), [0.0, 5e-324, 1e-310]])),
        9: ("exp_fast", np.concatenate([rng.uniform(-760, 720, n), rng.uniform(-3, 3, n), [np.nan, np.inf, -np.inf]])),
        10: ("log_fast", np.concatenate([np.exp(rng.uniform(-708, 709, n)), rng.uniform(-1, 2, n), [0.0, 1e-310]])),
        11: ("log_ndtr_nonpos", np.concatenate([-np.exp(rng.uniform(-30, 11.6, n)), rng.uniform(-40, 0, n), [0.0, -1e5]])),
        19: ("erfc_tab", np.concatenate([rng.uniform(0, 6.5, n), np.exp(rng.uniform(-30, 2, n)), np.arange(0, 26) / 4.0, np.arange(0, 26) / 4.0 + 0.125,
                          [0.0, -0.0, 5.999999, 6.0, 7.0, 1e300, -1.0, -1e300, np.nan, np.inf]])),
        18: ("log_ndtr_tab", np.concatenate([-np.exp(rng.uniform(-30, 12.1, n)), rng.uniform(-40, 0, n), [0.0, -0.0, -1e5, -1.85e5, -2e5, 3.0, 1e5, np.nan, np.inf]])),
    }
    for fn, (name, x) in cases.items():
        got, want = debug_math(fn, x, gpu), co.vec(name, x)
        assert np.array_equal(got, want, equal_nan=True), "%s: %d mismatches" % (name, int((got != want).sum()))
    x = np.concatenate([np.exp(rng.uniform(-700, 700, n)), -np.exp(rng.uniform(-5, 5, n))])
    assert np.array_equal(debug_math(6, x, gpu), 1.0 / x)                      # IEEE division
    # the MH loops' own division / square root: hipcc's iterations without the exponent-range handling (phf_math.h) must still
    # be the correctly rounded results — that is what the twin computes with / and sqrt — over 2^-600 .. 2^600, both signs
    x = np.concatenate([np.exp(rng.uniform(-415, 415, n)), -np.exp(rng.uniform(-415, 415, n)), rng.uniform(1, 2, n), 1 + np.arange(1, 4097) * 2.0 ** -52,
                        2 - np.arange(1, 4097) * 2.0 ** -52, [1.0, 2.0, 3.0, 4.0, 2.0 ** 52 + 1, 1.0 / 3, 1e70, 1e-70]])
    assert np.array_equal(debug_math(12, x, gpu), 1.0 / x), "phf_rcp"
    ln10 = float.fromhex("0x1.26bb1bbb55516p+1")
    assert np.array_equal(debug_math(14, x, gpu), ln10 / x), "phf_div (denominator varies)"
    assert np.array_equal(debug_math(16, x, gpu), x / ln10), "phf_div (numerator varies)"
    xp = np.abs(x)
    assert np.array_equal(debug_math(13, xp, gpu), np.sqrt(xp)), "phf_sqrt_pos"
    # a Cholesky pivot and its reciprocal from one hardware estimate (phf_sqrt_rcp_pos): the same two correctly rounded numbers
    assert np.array_equal(debug_math(21, xp, gpu), np.sqrt(xp)), "phf_sqrt_rcp_pos: square root"
    assert np.array_equal(debug_math(20, xp, gpu), 1.0 / np.sqrt(xp)), "phf_sqrt_rcp_pos: reciprocal of the rounded square root"
    xz = np.concatenate([xp[:1000], [0.0, -0.0, -1.0, -1e-300]])
    want = np.where(xz > 0, np.sqrt(np.abs(xz)), 0.0)
    assert np.array_equal(debug_math(15, xz, gpu), want), "phf_sqrt_nonneg"
    w = rng.integers(0, 2 ** 32, n, dtype=np.uint64)
    w[:6] = [0, 2 ** 28 - 1, 2 ** 28, 2 ** 31, 2 ** 32 - 1, 2 ** 32 - 2 ** 28]
    s, c = co.sincos(w.astype(np.uint32))
    assert np.array_equal(debug_math(7, w.astype(np.float64), gpu), s)
    assert np.array_equal(debug_math(8, w.astype(np.float64), gpu), c)
    w[:8] = [0, 1, 2 ** 31 - 1, 2 ** 31, 2 ** 31 + 1, 2 ** 32 - 1, 2 ** 30, 3 * 2 ** 29]
    assert np.array_equal(debug_math(17, w.astype(np.float64), gpu), co.normal_u32(w.astype(np.uint32))), "phf_normal_u32"


def test_device_philox_known_answers(gpu):
    from oracle import c_oracle as co
    from pyhillfit_amd.sampler import debug_philox
    ck = np.random.default_rng(5).integers(0, 2 ** 32, (10000, 6), dtype=np.uint64).astype(np.uint32)
    ck[0] = 0; ck[1] = 0xffffffff
    ck[2] = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0]
    from test_math_philox import PHILOX_KAT
    for rounds, kat in PHILOX_KAT.items():                  # Random123 kat_vectors, 7 and 10 rounds
        got = debug_philox(ck, gpu, rounds=rounds)
        for i, (_, want) in enumerate(kat):
            assert got[i].tolist() == want, (rounds, i)
        assert np.array_equal(got, co.philox(ck, rounds))
    assert np.array_equal(debug_philox(ck, gpu), co.philox(ck, 7))          # the samplers' own block: 7 rounds


# ------------------------------------------------------------------------------------------------- log-target
def test_log_target_batch_vs_reference_golden_and_twin(gpu, dr, golden_meta):
    from oracle import c_oracle as co
    from pyhillfit_amd.sampler import log_target_batch
    g = np.load(os.path.join(GOLDEN, "g1_log_target.npz"))
    names = [(m["drug"], m["channel"]) for m in golden_meta["g1_pairs"]]
    packed = dr.pack_single_level(names)
    for model in (1, 2):
        sel = g["model"] == model
        th = g["theta"][sel] if model == 2 else g["theta"][sel][:, [0, 2]]
        lik, pri = log_target_batch(packed, model, g["pair"][sel], g["t"][sel], th, gpu)
        for got, want in ((lik, g["lik"][sel]), (pri, g["prior"][sel]), (lik + pri, g["target"][sel])):
            bad = ~np.isfinite(want)
            assert np.array_equal(got[bad], want[bad], equal_nan=True)
            np.testing.assert_allclose(got[~bad], want[~bad], rtol=1e-12, atol=0)
        # and bit-identical to the CPU twin
        packs = {}
        tw = np.empty(sel.sum())
        for k, (ip, t, row) in enumerate(zip(g["pair"][sel], g["t"][sel], th)):
            key = (int(ip), float(t))
            if key not in packs:
                packs[key] = co.PackedPair(*_pair_arrays(dr, *names[int(ip)]), model, float(t))
            tw[k] = packs[key].log_target(row)
        assert np.array_equal(lik + pri, tw, equal_nan=True)


def test_reference_signature_log_target(gpu, dr):
    """dr.log_target(y, where_y_0, where_y_100, where_y_other, concs, params, t, pi_bit) as the reference's
    callers use it (doseresponse.py:187-189); probe values from SURVEY.md section 8c."""
    concs, y = _pair_arrays(dr, "Amiodarone", "hERG")
    w0, w100, wo = dr.response_masks(y)
    pi_bit = dr.compute_pi_bit_of_log_likelihood(wo)
    assert pi_bit == 11.027262398456072
    dr.define_model(2)
    probes = [((6, 1, 5), 1, -58.39140921642633), ((6.2, 0.7, 8), 1, -38.16481071511112),
              ((1, 1, 1), 1, -12946.405169647265), ((6, 1, 5), 0.125, -5.633162956781316), ((6, 1, 5), 0, 1.9037293660251158)]
    for th, t, want in probes:
        assert dr.log_target(y, w0, w100, wo, concs, np.array(th, float), t, pi_bit) == pytest.approx(want, rel=1e-12)
    for th in [(5.5, 11, 3), (-3.5, 1, 2), (6, 1, 5e-4)]:
        assert dr.log_target(y, w0, w100, wo, concs, np.array(th, float), 1, pi_bit) == -np.inf
    dr.define_model(1)
    assert np.isfinite(dr.log_target(y, w0, w100, wo, concs, np.array([6., 5.]), 1, pi_bit))


# ------------------------------------------------------------------------------------------------- the MH loop
CASES = [  # (model, pairs, temps-per-pair, chains, iterations, thinning, adapt_start, reset_mean, cov_identity, cov_scale, theta0)
    (2, [("Amiodarone", "hERG"), ("Bepridil", "hERG"), ("Moxifloxacin", "KvLQT1/mink")], [1.0, 0.125], 192, 1500, 5, 400, False, False, 0.05, [6.0, 0.8, 8.0]),
    (1, [("Amiodarone", "hERG"), ("Rufinamide", "Kir2.1")], [1.0, 0.0], 70, 1200, 3, 300, True, True, 1.0, [1.0, 1.0]),
    (2, [("Amitriptyline", "Kv4.3"), ("Lidocaine", "KvLQT1/mink")], [1.0], 64, 900, 1, 200, True, True, 1.0, [1.0, 1.0, 1.0]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_chains_bit_identical_to_cpu_twin(case, gpu, dr):
    """same seed => same accept sequence, same samples, same log-targets, same final state, bit for bit;
    the GPU run is cut into uneven launches to cover the state hand-over between kernels."""
    from oracle import c_oracle as co
    from pyhillfit_amd.sampler import SingleLevelSampler, gamma_table
    model, names, temps, C, T, thin, adapt, reset, cov_id, cov_scale, theta0 = CASES[case]
    packed = dr.pack_single_level(names)
    pair_index = [p for p in range(len(names)) for _ in temps]
    temperature = [t for _ in names for t in temps]
    s = SingleLevelSampler(packed, model, pair_index, temperature, C, thinning=thin, seed=1234567890123, adapt_start=adapt,
                           reset_mean_at_adapt_start=reset, chain_id_base=1000, problem_ids=np.arange(len(pair_index)) + 7,
                           device=gpu)
    s.init(theta0, cov_identity=cov_id, cov_scale=cov_scale)
    row0 = s.row0.cpu().numpy()
    parts = [s.advance(k).cpu().numpy() for k in (adapt - 7, 8, 1, T - adapt - 2)]
    chain = np.concatenate(parts)                                  # [T/thin][Q][d+1][C]
    assert chain.shape == (T // thin, len(pair_index), s.d + 1, C)
    state = s.state.cpu().numpy().reshape(s.S, len(pair_index), C)
    gam = gamma_table(T)
    rng = np.random.default_rng(case)
    for q in range(len(pair_index)):
        pk = co.PackedPair(*_pair_arrays(dr, *names[pair_index[q]]), model, temperature[q])
        for c in [0, C - 1] + rng.integers(0, C, 2).tolist():
            st = pk.init_state(theta0, cov_id, cov_scale)
            assert np.array_equal(row0[q, :, c], np.concatenate([theta0, [st[s.d]]]))
            rows = pk.advance(st, 0, T, thin, adapt, reset, gam, seed=1234567890123, chain_id=1000 + c, problem_id=7 + q)
            assert np.array_equal(chain[:, q, :, c], rows), (q, c)
            assert np.array_equal(state[:, q, c], st), (q, c)
    assert 0.01 < float(s.acceptance().mean()) < 0.95


def _shape_pair(ko, kc, rng):
    """a synthetic pair whose merged packing has ko uncensored and kc censored entries (1-3 replicate points each)"""
    concs, y = [], []
    doses = 10.0 ** np.linspace(-3, 3, 12)
    n_zero = (kc + 1) // 2
    for k in range(kc):                                   # zeros at the lowest doses, hundreds at the highest
        dose, val = (doses[k], 0.0) if k < n_zero else (doses[11 - (k - n_zero)], 100.0)
        for _ in range(int(rng.integers(1, 4))):
            concs.append(dose); y.append(val)
    for k in range(ko):
        for _ in range(int(rng.integers(1, 4))):
            concs.append(doses[3 + k]); y.append(float(np.clip(15.0 * (k + 1) + rng.normal(0, 4), 1.0, 99.0)))
    order = rng.permutation(len(y))
    return np.array(concs)[order], np.array(y)[order]


@pytest.mark.parametrize("model,chains", [(2, 64), (1, 64), (2, 1920)])
def test_every_entry_count_shape_bit_identical_to_cpu_twin(model, chains, gpu):
    """the advance kernel has one straight-line loop body per (uncensored, censored) entry-count shape and two register
    budgets (launches of at most / more than one wavefront per SIMD): every shape 0..5 x 0..5, through both variants,
    against the twin's run-time loops"""
    from oracle import c_oracle as co
    from pyhillfit_amd.doseresponse import PackedPoints
    from pyhillfit_amd.sampler import SingleLevelSampler, gamma_table
    rng = np.random.default_rng(100 + model)
    shapes = [(ko, kc) for ko in range(6) for kc in range(6) if ko + kc > 0]
    pairs = [_shape_pair(ko, kc, rng) for ko, kc in shapes]
    packed = PackedPoints(pairs)
    assert [(int(c[0]), int(c[1] + c[2])) for c in packed.counts] == shapes
    Q, T, thin, adapt = len(shapes), 260, 2, 60
    assert (Q * ((chains + 63) // 64) > 1024) == (chains > 64)            # 35 blocks: one wave per SIMD; 1 050: more
    s = SingleLevelSampler(packed, model, list(range(Q)), [1.0] * Q, chains, thinning=thin, seed=99, adapt_start=adapt, device=gpu)
    theta0 = [5.0, 1.0, 9.0] if model == 2 else [5.0, 9.0]
    s.init(theta0, cov_identity=False, cov_scale=0.05)
    chain = np.concatenate([s.advance(k).cpu().numpy() for k in (adapt + 1, T - adapt - 1)])
    state = s.state.cpu().numpy().reshape(s.S, Q, chains)
    gam = gamma_table(T)
    for q, (concs, y) in enumerate(pairs):
        pk = co.PackedPair(concs, y, model, 1.0)
        for c in (0, chains - 1, int(rng.integers(0, chains))):
            st = pk.init_state(theta0, False, 0.05)
            rows = pk.advance(st, 0, T, thin, adapt, False, gam, seed=99, chain_id=c, problem_id=q)
            assert np.array_equal(chain[:, q, :, c], rows), (shapes[q], c)
            assert np.array_equal(state[:, q, c], st), (shapes[q], c)
    # the moment-accumulating builds of the same bodies (what the command lines run): same chains bit for bit, and the device
    # accumulators are the sums of the rows saved after `after_iteration`
    m = SingleLevelSampler(packed, model, list(range(Q)), [1.0] * Q, chains, thinning=thin, seed=99, adapt_start=adapt, device=gpu)
    m.init(theta0, cov_identity=False, cov_scale=0.05)
    m.enable_moments(after_iteration=adapt)
    chain_m = np.concatenate([m.advance(k).cpu().numpy() for k in (adapt + 1, T - adapt - 1)])
    assert np.array_equal(chain_m, chain) and np.array_equal(m.state.cpu().numpy().reshape(s.S, Q, chains), state)
    mean, var, n = m.posterior_moments()
    keep = chain[adapt // thin:]
    assert n == len(keep)
    np.testing.assert_allclose(mean.cpu().numpy(), keep.mean(axis=0).transpose(1, 0, 2), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(var.cpu().numpy(), keep.var(axis=0, ddof=1).transpose(1, 0, 2), rtol=1e-7, atol=1e-10)


def test_launch_order_does_not_change_results(gpu, dr):
    """phf_problems.launch_order (ABI 3) only chooses which problem's wavefronts the GPU gets first (most expensive first by
    default: a shorter tail); rows and state land at the problem's own place whatever the order"""
    from pyhillfit_amd.sampler import SingleLevelSampler
    names = [(d, c) for d in dr.drugs[:9] for c in dr.channels]
    packed = dr.pack_single_level(names)
    Q = len(names)
    temps = [1.0 if q % 3 else 0.25 for q in range(Q)]
    runs = []
    for order in (None, "cost", np.random.default_rng(4).permutation(Q)):
        s = SingleLevelSampler(packed, 2, list(range(Q)), temps, 192, thinning=5, seed=8, adapt_start=40, device=gpu, launch_order=order)
        s.init([6.0, 0.8, 8.0])
        runs.append((s.run(120), s.state.clone()))
        if isinstance(order, str):
            o = s.launch_order.cpu().numpy()
            cnt = packed.counts
            cost = 28.0 * cnt[o, 0] + 115.0 * (cnt[o, 1] + cnt[o, 2])
            assert sorted(o.tolist()) == list(range(Q)) and np.all(np.diff(cost) <= 0)       # a permutation, most expensive first
    for chain, state in runs[1:]:
        assert torch.equal(chain, runs[0][0]) and torch.equal(state, runs[0][1])
    with pytest.raises(ValueError):
        SingleLevelSampler(packed, 2, list(range(Q)), temps, 64, device=gpu, launch_order=[0] * Q)


def test_shard_invariance_and_resume_full_width(gpu, dr):
    """BASELINE config 2 width (65 536 chains of Amiodarone-hERG): the same chains computed (a) in one launch,
    (b) as two half-batches with chain_id_base — the multi-GPU partition — and (c) through a checkpoint/restore
    give identical bits (size-independent property; the twin covers correctness at small sizes)."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    packed = dr.pack_single_level([("Amiodarone", "hERG")])
    C, T = 65536, 60
    kw = dict(thinning=5, seed=25, adapt_start=20, device=gpu)
    full = SingleLevelSampler(packed, 2, [0], [1.0], C, **kw)
    full.init([6.0, 0.8, 8.0])
    a = full.run(T)
    halves = []
    for h in range(2):
        s = SingleLevelSampler(packed, 2, [0], [1.0], C // 2, chain_id_base=h * (C // 2), **kw)
        s.init([6.0, 0.8, 8.0])
        halves.append(s.run(T))
    assert torch.equal(a, torch.cat(halves, dim=3))
    r = SingleLevelSampler(packed, 2, [0], [1.0], C, **kw)
    r.init([6.0, 0.8, 8.0])
    first = r.advance(35)
    sd = r.state_dict()
    r2 = SingleLevelSampler(packed, 2, [0], [1.0], C, **kw)
    r2.load_state_dict(sd)
    second = r2.advance(25)
    assert torch.equal(a[1:], torch.cat([first, second]))
    assert torch.isfinite(a).all()
    # distinct chains really are distinct streams
    assert torch.unique(a[-1, 0, 0, :]).numel() > 0.9 * C


# ------------------------------------------------------------------------------------------------- statistics
def test_posterior_means_within_one_percent_of_reference(gpu, dr):
    """north-star tolerance: posterior means within 1 % of the CPU reference (tests/golden/g5_posteriors.json,
    reference do_mcmc 200k iterations) — 2 048 chains x 200k iterations per problem (as long as the reference run:
    Hill has rare long excursions), pooled, moments accumulated on the device."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    with open(os.path.join(GOLDEN, "g5_posteriors.json")) as f:
        g5 = json.load(f)
    for model in (1, 2):
        want = [w for w in g5 if w["model"] == model and w["temperature"] in (1.0, 0.125)]
        names = [(w["drug"], w["channel"]) for w in want]
        packed = dr.pack_single_level(names)
        s = SingleLevelSampler(packed, model, list(range(len(names))), [w["temperature"] for w in want], 2048, thinning=5,
                               seed=25, reset_mean_at_adapt_start=True, device=gpu)
        s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)      # PyHillTemp start
        s.enable_moments(after_iteration=50000)
        s.advance(200000, save=False)
        mean, var, n = s.posterior_moments()
        pooled = mean.mean(dim=2).cpu().numpy()                      # [d+1][Q]
        pooled_sd = torch.sqrt(var.mean(dim=2) + mean.var(dim=2)).cpu().numpy()
        for q, w in enumerate(want):
            ref_se = np.array(w["sd"][:s.d]) / np.sqrt(w["rows"] / 20.0)     # the reference is ONE chain
            err = np.abs(pooled[:s.d, q] - w["mean"][:s.d])
            assert np.all(err <= 0.01 * np.abs(w["mean"][:s.d]) + 3 * ref_se), (w["drug"], model, pooled[:, q], w["mean"])
            np.testing.assert_allclose(pooled_sd[:s.d, q], w["sd"][:s.d], rtol=0.2)


def test_posteriors_of_other_pair_shapes_match_reference(gpu, dr):
    """G5b (tests/golden/make_golden_posteriors_extra.py, reference do_mcmc, 200k iterations each): 20 points with both kinds
    of censoring, the pair with the out-of-range response, a 6-point pair, an all-zero pair, model 1 with a saturated point.
    Pooled means of 1 024 chains within 1 % + 4 batch-means standard errors of the reference's single chain."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    with open(os.path.join(GOLDEN, "g5b_posteriors.json")) as f:
        g5b = json.load(f)
    for model in (1, 2):
        want = [w for w in g5b if w["model"] == model]
        names = [(w["drug"], w["channel"]) for w in want]
        packed = dr.pack_single_level(names)
        s = SingleLevelSampler(packed, model, list(range(len(names))), [1.0] * len(want), 1024, thinning=5, seed=11,
                               reset_mean_at_adapt_start=True, device=gpu)
        s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)
        s.enable_moments(after_iteration=50000)
        s.advance(200000, save=False)
        mean, var, n = s.posterior_moments()
        pooled = mean.mean(dim=2).cpu().numpy()
        pooled_sd = torch.sqrt(var.mean(dim=2) + mean.var(dim=2)).cpu().numpy()
        for q, w in enumerate(want):
            k = s.d + 1                                                    # parameters and the log-target column
            err = np.abs(pooled[:k, q] - w["mean"][:k])
            tol = 0.01 * np.abs(w["mean"][:k]) + 4 * np.array(w["batch_means_se"][:k])
            assert np.all(err <= tol), (w["drug"], w["channel"], model, pooled[:k, q], w["mean"], tol)
            np.testing.assert_allclose(pooled_sd[:s.d, q], w["sd"][:s.d], rtol=0.15)


@pytest.mark.parametrize("model", [2, 1])
def test_every_crumb_pair_posterior_within_one_percent_of_reference(model, gpu, dr):
    """BASELINE's correctness target on the full data set (config 3): G5c = the reference's own sampler (PyHillTemp.do_mcmc,
    200 000 iterations, tests/golden/make_golden_posteriors_all.py) run on every one of the 210 pairs, for both models; here
    256 chains per pair, same length and start, pooled on the device.  Tolerance per (pair, column), with NO entry allowed outside:
    1 % of the reference mean + 4 standard errors of the reference — the batch-means s.e. of its single chain, or, for the five
    weakly informative pairs whose single chain proved too short to say what the mean is (golden G5d: eight and more independent
    reference chains each; Ranolazine-Nav1.5-peak's seed-1 chain sits 3.5 of the others' standard errors from their mean), the
    pooled mean and the s.e. of the reseeded runs (conftest.reference_posteriors).  Posterior WIDTHS: every sd within [0.8, 1.25] of the
    reference's, no entry excepted, with golden G5e (32 reference seeds for each of the twelve (pair, model) cases whose single G5c chain
    gave a width outside that band; rule fixed before the runs) standing in for those cases' single chains."""
    from conftest import reference_posteriors
    from pyhillfit_amd.sampler import SingleLevelSampler
    names, want, se, want_sd, reseeded = reference_posteriors(model)
    have_g5e = os.path.exists(os.path.join(GOLDEN, "g5e_posterior_widths_reseeded.json"))
    assert len(names) == 210 and len(reseeded) == ((5 if model == 2 else 0) + ((9 if model == 2 else 3) if have_g5e else 0))
    packed = dr.pack_single_level(names)
    s = SingleLevelSampler(packed, model, list(range(len(names))), [1.0] * len(names), 256, thinning=5, seed=5,
                           reset_mean_at_adapt_start=True, device=gpu)
    s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)          # PyHillTemp.py:63,80
    s.enable_moments(after_iteration=50000)
    s.advance(200000, save=False)
    mean, var, n = s.posterior_moments()
    pooled = mean.mean(dim=2).cpu().numpy()                          # [d+1][210]
    pooled_sd = torch.sqrt(var.mean(dim=2) + mean.var(dim=2)).cpu().numpy()
    # the GPU's own Monte-Carlo error (scatter of its 256 chains' means / 16) joins the reference's in quadrature: against a single
    # reference chain it is negligible, against 32 or 96 pooled reference seeds of a long-tailed marginal it is not (Diltiazem-Kv4.3 Hill:
    # reference 0.2035 +- 0.0005 over 32 seeds, GPU 0.2077 +- 0.002)
    se_gpu = (mean.std(dim=2) / np.sqrt(mean.shape[2])).cpu().numpy()
    ratio = np.abs(pooled - want) / (0.01 * np.abs(want) + 4 * np.sqrt(se ** 2 + se_gpu ** 2))
    worst = np.unravel_index(np.argmax(ratio), ratio.shape)
    print("G5c+G5d: worst ratio %.2f at %s column %d; fraction below 0.5: %.4f" % (ratio.max(), names[worst[1]], worst[0], np.mean(ratio < 0.5)))
    order = np.dstack(np.unravel_index(np.argsort(-ratio, axis=None), ratio.shape))[0][:25]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)                # scratch record of the entries closest to the tolerance
    with open(os.path.join(REPO, "gpurun_out", "g5c_worst_model_%d.json" % model), "w") as f:
        json.dump([{"drug": names[q][0], "channel": names[q][1], "column": int(k), "ratio": float(ratio[k, q]), "gpu": float(pooled[k, q]),
                    "reference": float(want[k, q]), "reference_se": float(se[k, q]), "gpu_sd": float(pooled_sd[k, q])} for k, q in order], f, indent=1)
    # EVERY (pair, column): within 1 % of the reference's mean + 4 of its standard errors — no fraction allowed to fail
    assert ratio.max() < 1.0, (ratio.max(), names[worst[1]], worst[0], pooled[worst], want[worst], se[worst])
    # posterior widths.  The reference's single 200k chain estimates an sd poorly where the posterior has a long thin tail (Hill of weakly
    # informative pairs): against G5c alone 12 of the 1 050 ratios lay outside [0.8, 1.25] (tools/diag_sd_ratios.py).  Golden G5e gives each of
    # those (pair, model) cases 32 reference seeds (rule fixed before the runs): with it EVERY width is held to [0.8, 1.25]
    sd_ratio = pooled_sd[:s.d] / want_sd[:s.d]
    lo, hi = np.unravel_index(np.argmin(sd_ratio), sd_ratio.shape), np.unravel_index(np.argmax(sd_ratio), sd_ratio.shape)
    info = (sd_ratio.min(), names[lo[1]], lo[0], sd_ratio.max(), names[hi[1]], hi[0], np.mean((sd_ratio > 0.8) & (sd_ratio < 1.25)))
    print("posterior widths model %d: sd ratios %.3f (%s column %d) .. %.3f (%s column %d)" % (model, info[0], info[1], info[2], info[3], info[4], info[5]))
    if have_g5e:
        # G5e gave each of the twelve (pair, model) cases with a single-chain width outside the band 32 reference seeds; two Hill entries
        # (Diltiazem-Kv4.3, Lidocaine-Kv4.3) stayed outside and were "reported" in round 4.  Round 5: golden G5f — 256 FRESH reference
        # seeds (129..384) for exactly those two pairs, rule written into make_golden_posteriors_reseed.py before the runs — stands in
        # for their G5e entries (conftest.reference_posteriors), and NO entry is exempt any more.
        assert os.path.exists(os.path.join(GOLDEN, "g5f_posterior_tails_reseeded.json"))
        outside = [(names[q], int(k), round(float(sd_ratio[k, q]), 3)) for k, q in zip(*np.nonzero((sd_ratio <= 0.8) | (sd_ratio >= 1.25)))]
        print("posterior widths model %d: outside [0.8, 1.25]: %s" % (model, outside))
        assert outside == [], outside
    else:
        assert np.mean((sd_ratio > 0.8) & (sd_ratio < 1.25)) >= 0.97, info
        assert sd_ratio.min() > 0.5 and sd_ratio.max() < 3.0, info


def test_reseeded_cases_two_sample_against_the_reference_seeds(gpu, dr):
    """Golden G5d under its round-4 protocol (tests/golden/make_golden_posteriors_reseed.py: a closed list of six cases, the SAME 96
    independent reference chains for every one of them, fixed before any GPU number of the round was looked at): a two-sample test with
    no percentage slack.  GPU: 4 096 chains per case, the reference's length, start and burn-in; z = (GPU pooled mean - mean of the
    reference seeds' means) / sqrt(se_ref^2 + se_gpu^2), se_ref = scatter between the reference seeds / sqrt(seeds), se_gpu = scatter
    between the GPU's chains / sqrt(chains).  Every column of every case: |z| < 3 (24 entries: about one run in fifteen of a CORRECT
    sampler would show one beyond 3; none may reach 4)."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    with open(os.path.join(GOLDEN, "g5d_posteriors_reseeded.json")) as f:
        g5d = json.load(f)
    assert len(g5d) == 6 and len({len(e["seeds"]) for e in g5d}) == 1, [len(e["seeds"]) for e in g5d]      # the same seeds for every case
    n_seeds = len(g5d[0]["seeds"])
    assert n_seeds >= 32 and all(e["model"] == 2 and e["iterations"] == 200000 for e in g5d)
    names = [(e["drug"], e["channel"]) for e in g5d]
    packed = dr.pack_single_level(names)
    C = 4096
    s = SingleLevelSampler(packed, 2, list(range(len(g5d))), [e["temperature"] for e in g5d], C, thinning=5, seed=11,
                           reset_mean_at_adapt_start=True, device=gpu)
    s.init(np.ones(3), cov_identity=True, cov_scale=1.0)          # PyHillTemp.py:63,80
    s.enable_moments(after_iteration=50000)                       # first quarter of the saved rows dropped (:70-71)
    s.advance(200000, save=False)
    mean, var, n = s.posterior_moments()
    per_chain = mean.cpu().numpy()                                # [d+1][Q][C]
    worst = 0.0
    for q, e in enumerate(g5d):
        ref_means = np.array([r["mean"] for r in e["runs"]])      # [seeds][d+1]
        assert ref_means.shape == (n_seeds, 4) and np.allclose(ref_means.mean(axis=0), e["mean"])
        se_ref = ref_means.std(axis=0, ddof=1) / np.sqrt(n_seeds)
        gpu_mean = per_chain[:, q].mean(axis=1)
        se_gpu = per_chain[:, q].std(axis=1, ddof=1) / np.sqrt(C)
        z = (gpu_mean - ref_means.mean(axis=0)) / np.sqrt(se_ref ** 2 + se_gpu ** 2)
        print("G5d two-sample %s-%s t=%g: GPU %s  reference %s +- %s  z %s" % (e["drug"], e["channel"], e["temperature"], np.round(gpu_mean, 4).tolist(),
                                                                              np.round(ref_means.mean(axis=0), 4).tolist(), np.round(se_ref, 4).tolist(), np.round(z, 2).tolist()))
        worst = max(worst, float(np.abs(z).max()))
        assert np.abs(z).max() < 3.0, (e["drug"], e["channel"], e["temperature"], z.tolist())
    print("G5d two-sample: worst |z| %.2f over %d entries, %d reference seeds per case" % (worst, 4 * len(g5d), n_seeds))


def test_g5f_large_hill_tail_pairs_against_256_fresh_reference_seeds(gpu, dr):
    """Golden G5f (tests/golden/make_golden_posteriors_reseed.py --tails, rule fixed before the runs): Diltiazem-Kv4.3
    and Lidocaine-Kv4.3, model 2, reference seeds 129..384.  4 096 GPU chains per pair, the reference's length, start and burn-in:
    every column's pooled sd within [0.8, 1.25] of the 256 reference seeds' pooled sd, and the pooled means in a two-sample test,
    |z| < 3, standard errors from the scatter between reference seeds and between GPU chains (as G5d).  Also recorded: the share of
    chains that make the excursion to large Hill (per-chain Hill sd above three times the median), reference and GPU side by side."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    with open(os.path.join(GOLDEN, "g5f_posterior_tails_reseeded.json")) as f:
        g5f = json.load(f)
    assert [(e["drug"], e["channel"]) for e in g5f] == [("Diltiazem", "Kv4.3"), ("Lidocaine", "Kv4.3")]
    assert all(e["seeds"] == list(range(129, 385)) and e["model"] == 2 and e["iterations"] == 200000 for e in g5f)
    names = [(e["drug"], e["channel"]) for e in g5f]
    packed = dr.pack_single_level(names)
    C = 4096
    s = SingleLevelSampler(packed, 2, [0, 1], [1.0, 1.0], C, thinning=5, seed=11, reset_mean_at_adapt_start=True, device=gpu)
    s.init(np.ones(3), cov_identity=True, cov_scale=1.0)          # PyHillTemp.py:63,80
    s.enable_moments(after_iteration=50000)                       # first quarter of the saved rows dropped (:70-71)
    s.advance(200000, save=False)
    mean, var, n = s.posterior_moments()
    per_chain, per_var = mean.cpu().numpy(), var.cpu().numpy()    # [d+1][Q][C]
    report = []
    for q, e in enumerate(g5f):
        ref_means = np.array([r["mean"] for r in e["runs"]]); ref_sds = np.array([r["sd"] for r in e["runs"]])
        n_seeds = len(ref_means)
        assert n_seeds == 256
        gpu_mean = per_chain[:, q].mean(axis=1)
        gpu_sd = np.sqrt(per_var[:, q].mean(axis=1) + per_chain[:, q].var(axis=1))
        ref_sd = np.sqrt((ref_sds ** 2).mean(axis=0) + ref_means.var(axis=0))
        assert np.allclose(ref_sd, e["sd"])
        ratio = gpu_sd / ref_sd
        z = (gpu_mean - ref_means.mean(axis=0)) / np.sqrt(ref_means.var(axis=0, ddof=1) / n_seeds + per_chain[:, q].var(axis=1, ddof=1) / C)
        hill_sd_gpu = np.sqrt(per_var[1, q]); hill_sd_ref = ref_sds[:, 1]
        share = (float(np.mean(hill_sd_gpu > 3 * np.median(hill_sd_gpu))), float(np.mean(hill_sd_ref > 3 * np.median(hill_sd_ref))))
        report.append({"pair": "%s-%s" % names[q], "sd_ratio": ratio.tolist(), "z": z.tolist(), "gpu_sd": gpu_sd.tolist(), "reference_sd": ref_sd.tolist(),
                       "excursion_share_gpu_and_reference": share})
        print("G5f %s-%s: sd ratios %s  z %s  chains with a Hill excursion: GPU %.3f, reference %.3f" % (
            names[q][0], names[q][1], np.round(ratio, 3).tolist(), np.round(z, 2).tolist(), share[0], share[1]))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "g5f_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    for r in report:
        assert 0.8 < min(r["sd_ratio"]) and max(r["sd_ratio"]) < 1.25, r
        assert np.abs(r["z"]).max() < 3.0, r


def test_prior_only_rung_known_answer(gpu, dr):
    """t = 0 (first rung of the ladder): chains must sample the prior — pIC50 ~ -3+Exp(mean 5), Hill ~ U(0,10),
    sigma ~ 1e-3+Gamma(5, 1.49975): means (2, 5, 7.49975), sds (5, 2.887, 3.354)."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    packed = dr.pack_single_level([("Amiodarone", "hERG")])
    s = SingleLevelSampler(packed, 2, [0], [0.0], 4096, thinning=5, seed=3, reset_mean_at_adapt_start=True, device=gpu)
    s.init(np.ones(3), cov_identity=True, cov_scale=1.0)
    s.enable_moments(after_iteration=100000)      # the exponential pIC50 tail is reached slowly: run as long as the CPU test
    s.advance(400000, save=False)
    mean, var, n = s.posterior_moments()
    m = mean.mean(dim=2).cpu().numpy()[:3, 0]
    sd = torch.sqrt(var.mean(dim=2) + mean.var(dim=2)).cpu().numpy()[:3, 0]
    np.testing.assert_allclose(m, [2.0, 5.0, 7.49975], rtol=0.03)
    np.testing.assert_allclose(sd, [5.0, 10 / np.sqrt(12), np.sqrt(5) * 1.49975], rtol=0.04)


def test_thermodynamic_integration_against_the_reference_pipeline(gpu, dr):
    """G6 = the reference's own pipeline for two pairs and both models: PyHillTemp.do_mcmc on each of the 41 rungs, then
    compute_bayes_factors.py's per-rung mean of log L(theta; t=1), its trapezium rule and B12 (tests/golden/make_golden_ti.py).
    Here: one launch for all rungs of a model, 256 chains per rung, the per-rung mean accumulated inside the sampler kernel."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    path = os.path.join(GOLDEN, "g6_thermodynamic_integration.json")
    if not os.path.exists(path):
        pytest.skip("fixture not generated (tests/golden/make_golden_ti.py)")
    with open(path) as f:
        g6 = json.load(f)
    for entry in g6:
        temps = np.array(entry["temperatures"])
        assert np.array_equal(temps, dr.temperature_ladder()) and len(temps) == 41
        T, thin = entry["iterations"], 5
        burn = (T // thin + 1) // 4                                       # PyHillTemp.py:70-71
        packed = dr.pack_single_level([(entry["drug"], entry["channel"])])
        E = {}
        for model in (1, 2):
            ref = entry["models"][str(model)]
            s = SingleLevelSampler(packed, model, [0] * len(temps), temps.tolist(), 256, thinning=thin, seed=1,
                                   reset_mean_at_adapt_start=True, device=gpu)
            s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)
            s.enable_moments(after_iteration=max(burn * thin - 1, 0))
            s.advance(T, save=False)
            log_py = s.mean_log_likelihood_t1().mean(dim=1).cpu().numpy()                    # pooled over the chains of a rung
            want, se = np.array(ref["log_py"]), np.array(ref["batch_means_se"])
            ratio = np.abs(log_py - want) / (0.01 * np.abs(want) + 4 * se)
            print("G6 %s-%s model %d: per-rung ratios: max %.2f at rung %d, %d of %d above 1: %s" % (
                entry["drug"], entry["channel"], model, ratio.max(), int(ratio.argmax()), int((ratio >= 1).sum()), len(ratio),
                [(int(k), round(float(ratio[k]), 2), round(float(log_py[k]), 3), round(float(want[k]), 3), round(float(se[k]), 4)) for k in np.argsort(-ratio)[:4]]))
            assert ratio.max() < 1.0, (entry["drug"], model, ratio.max(), int(ratio.argmax()))         # every rung (observed: <= 0.69)
            E[model] = float(dr.trapezium_rule(temps, log_py))
            w = np.zeros(len(temps)); w[1:] += 0.5 * np.diff(temps); w[:-1] += 0.5 * np.diff(temps)   # trapezium weights
            se_E = float(np.sqrt(np.sum((w * se) ** 2)))
            assert abs(E[model] - ref["expectation"]) <= 0.01 * abs(ref["expectation"]) + 4 * se_E, (entry["drug"], model, E[model], ref["expectation"], se_E)
        # the Bayes factor itself (compute_bayes_factors.py:96): same order of magnitude and side of 1 as the reference's
        log_b12, ref_log_b12 = E[1] - E[2], entry["models"]["1"]["expectation"] - entry["models"]["2"]["expectation"]
        print("G6 %s-%s: E1 %.3f (ref %.3f)  E2 %.3f (ref %.3f)  log B12 %.3f (ref %.3f)" % (
            entry["drug"], entry["channel"], E[1], entry["models"]["1"]["expectation"], E[2], entry["models"]["2"]["expectation"], log_b12, ref_log_b12))
        assert abs(log_b12 - ref_log_b12) < 0.5, (entry["drug"], log_b12, ref_log_b12)
