"""Leaf numerics shared by the HIP kernels and the C twin (pyhillfit_amd/csrc/phf_math.h, phf_philox.h),
checked on the host build against numpy / scipy / mpmath and the Random123 known-answer vectors."""
import mpmath as mp
import numpy as np
import scipy.special as sp

from oracle import c_oracle as co

RNG = np.random.default_rng(7)


def _ulps(got, want):
    want = np.asarray(want, float)
    return np.max(np.abs(got - want) / np.spacing(np.abs(want)))


# Random123 kat_vectors: "philox4x32 10" and "philox4x32 7" on counter/key all zero, all ones, and the digits of pi
PHILOX_KAT = {
    10: [([0, 0, 0, 0, 0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
         ([0xffffffff] * 6, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
         ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])],
    7: [([0, 0, 0, 0, 0, 0], [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
        ([0xffffffff] * 6, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0], [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a])],
}


def philox_python(ck, rounds):
    """the round function written out once more, independently of phf_philox.h (Salmon et al., SC'11, section 3.3)"""
    c, k = [int(x) for x in ck[:4]], [int(x) for x in ck[4:]]
    for _ in range(rounds):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return c


def test_philox_known_answers():
    assert co.philox_rounds() == 7                         # what the samplers draw with (round 4; 10 before)
    for rounds, kat in PHILOX_KAT.items():
        for ck, want in kat:
            assert co.philox([ck], rounds)[0].tolist() == want
            assert philox_python(ck, rounds) == want
    ck = RNG.integers(0, 2 ** 32, (2000, 6), dtype=np.uint64).astype(np.uint32)
    for rounds in (7, 10):
        got = co.philox(ck, rounds)
        assert all(got[i].tolist() == philox_python(ck[i], rounds) for i in range(0, 2000, 7))
    assert np.array_equal(co.philox(ck), co.philox(ck, 7))    # rounds = 0: the samplers' own


def test_exp_log_within_an_ulp_or_two_of_libm():
    x = np.concatenate([RNG.uniform(-745, 709.7, 100000), RNG.uniform(-2, 2, 100000), [0.0, -0.0, 1.0, 709.782712893384]])
    assert _ulps(co.vec("exp", x), np.exp(x)) <= 1.0
    assert np.array_equal(co.vec("exp", [np.inf, -np.inf, 710.0, -746.0, -800.0]), [np.inf, 0, np.inf, 0, 0])
    assert np.isnan(co.vec("exp", [np.nan])[0])
    x = np.concatenate([np.exp(RNG.uniform(-700, 700, 100000)), RNG.uniform(0.5, 2, 100000), [5e-324, 1e-310, 1.0]])
    assert _ulps(co.vec("log", x), np.log(x)) <= 2.0      # table {1/c, log c} + degree-5 remainder: three roundings (phf_math.h)
    out = co.vec("log", [0.0, -0.0, -1.0, np.inf, np.nan])
    assert out[0] == -np.inf and out[1] == -np.inf and np.isnan(out[2]) and out[3] == np.inf and np.isnan(out[4])


def test_exp_log_against_mpmath():
    mp.mp.dps = 40
    for x in [-700.3, -37.2, -1e-5, 0.3, 1.0, 55.5, 709.7]:
        assert abs(mp.mpf(float(co.vec("exp", [x])[0])) / mp.exp(mp.mpf(x)) - 1) < 2.3e-16
    for x in [1e-300, 0.7071, 0.99999, 1.00001, 1.4143, 3.0, 1e300]:
        assert abs(mp.mpf(float(co.vec("log", [x])[0])) / mp.log(mp.mpf(x)) - 1) < 3.4e-16


def test_erfcx_and_normal_cdf():
    mp.mp.dps = 40
    for y in [0.0, 0.3, 1.0, 3.3, 10.0, 1e3, 7e4, 1e120]:
        ex = mp.exp(mp.mpf(y) ** 2) * mp.erfc(mp.mpf(y)) if y < 1e3 else None
        if ex is None:  # asymptotic series, exact enough at these sizes
            yy = mp.mpf(y); ex = 1 / (yy * mp.sqrt(mp.pi)) * (1 - 1 / (2 * yy * yy) + 3 / (4 * yy ** 4))
        assert abs(mp.mpf(float(co.vec("erfcx", [y])[0])) / ex - 1) < 6e-16
    y = np.concatenate([np.exp(RNG.uniform(-20, 12, 100000)), RNG.uniform(0, 8, 100000)])
    assert np.max(np.abs(co.vec("erfcx", y) / sp.erfcx(y) - 1)) < 4e-15
    # log Phi on the branch the censored likelihood uses (x <= 0), down to -1e5 (sigma = 1e-3)
    x = np.concatenate([-np.exp(RNG.uniform(-20, 11.6, 100000)), RNG.uniform(-40, 0, 100000), [0.0, -1e5]])
    assert np.max(np.abs(co.vec("log_ndtr", x) / sp.log_ndtr(x) - 1)) < 6e-15
    # the table form the single-level censored likelihood uses (y = -x/sqrt2 < 131 071) against the erfcx form and scipy
    x = np.concatenate([-np.exp(RNG.uniform(-20, 12.1, 100000)), RNG.uniform(-40, 0, 100000), [0.0, -1e5, -1.85e5]])
    assert np.max(np.abs(co.vec("log_ndtr_tab", x) / co.vec("log_ndtr", x) - 1)) < 2e-15
    assert np.max(np.abs(co.vec("log_ndtr_tab", x) / sp.log_ndtr(x) - 1)) < 6e-15
    # the hierarchical truncation masses' erfc table: ABSOLUTE accuracy (half an ulp of 1), zero from 6 on, interval joints included
    y = np.concatenate([RNG.uniform(0, 6, 200000), np.arange(0, 24) / 4.0 + 0.125, np.nextafter(np.arange(0, 24) / 4.0 + 0.125, 0), [0.0, 5.9999999]])
    assert np.max(np.abs(co.vec("erfc_tab", y) - sp.erfc(y))) < 4e-16          # scipy's own erfc is off by up to 3e-16 around y = 0.9
    assert np.array_equal(co.vec("erfc_tab", [6.0, 6.5, 1e300, np.inf]), [0.0, 0.0, 0.0, 0.0]) and co.vec("erfc_tab", [0.0])[0] == 1.0
    mp.mp.dps = 40
    for yy in np.concatenate([[0.01, 0.124, 0.126, 1.0, 2.37, 4.0, 5.9], RNG.uniform(0, 6, 2000)]):       # half an ulp of its own value at most
        assert abs(mp.mpf(float(co.vec("erfc_tab", [yy])[0])) - mp.erfc(mp.mpf(float(yy)))) < 1.2e-16
    x = RNG.uniform(0, 38, 50000)
    assert np.max(np.abs(co.vec("log_ndtr", x) - sp.log_ndtr(x))) < 4e-15
    x = RNG.uniform(-38, 10, 100000)
    got, want = co.vec("ndtr", x), sp.ndtr(x)
    assert np.max(np.abs(got - want)) < 1e-15
    assert np.max(np.abs(got / want - 1)[want > 0]) < 1e-12     # scipy's own tail error is ~x^2/2 ulp
    for xv in [-37.0, -20.5, -8.25, -3.0, -0.5, 0.7, 4.0]:         # the tail itself, against mpmath
        ex = mp.erfc(-mp.mpf(xv) / mp.sqrt(2)) / 2
        assert abs(mp.mpf(float(co.vec("ndtr", [xv])[0])) / ex - 1) < 1e-15


def test_kernel_fast_variants_match_the_full_functions():
    """the branch-free entry points the kernels use agree with the IEEE-complete ones on their domain"""
    x = np.concatenate([RNG.uniform(-760, 720, 100000), [-746.0, 710.0, 709.782712893384, np.inf, -np.inf]])
    assert np.array_equal(co.vec("exp_fast", x), co.vec("exp", x))
    assert co.vec("exp_fast", [np.nan])[0] == 0.0 and np.isnan(co.vec("exp", [np.nan])[0])
    x = np.concatenate([np.exp(RNG.uniform(-708, 709, 100000)), [2.2250738585072014e-308, 1.0, 0.5]])
    assert np.array_equal(co.vec("log_fast", x), co.vec("log", x))
    assert np.all(co.vec("log_fast", [0.0, -1.0, 1e-310, -np.inf]) == -np.inf)
    z = np.concatenate([-np.exp(RNG.uniform(-25, 11.6, 100000)), RNG.uniform(-40, 0, 100000), [0.0, -1e5]])
    one = co.vec("log_ndtr_nonpos", z)
    assert np.max(np.abs(one / sp.log_ndtr(z) - 1)) < 6e-15
    two = co.vec("log_ndtr_nonpos_x2", z[:200000])
    assert np.max(np.abs(two / one[:200000] - 1)) < 1.5e-15        # batched-reciprocal form: same value to ~3 ulp


def test_sincos_octants():
    w = RNG.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32)
    w[:10] = [0, 1, 2 ** 28 - 1, 2 ** 28, 2 ** 29, 2 ** 31, 2 ** 32 - 1, 2 ** 32 - 2 ** 28, 2 ** 32 - 2 ** 28 - 1, 3 * 2 ** 29]
    s, c = co.sincos(w)
    ang = 2 * np.pi * w.astype(float) / 2 ** 32
    assert np.max(np.abs(s - np.sin(ang))) < 1.5e-15 and np.max(np.abs(c - np.cos(ang))) < 1.5e-15
    assert s[0] == 0 and c[0] == 1 and abs(c[4] - np.sqrt(0.5)) < 2e-16 and s[5] == 0 and c[5] == -1


def test_normal_from_a_word_is_the_inverse_cdf():
    """phf_normal_u32: the piecewise inverse CDF against scipy's, interval ends and a dense sample; exact symmetry"""
    from scipy import stats
    w = np.concatenate([np.arange(0, 2 ** 31, 2 ** 31 // 40000, dtype=np.uint64), (2 ** np.arange(0, 31, dtype=np.uint64)) - 1, 2 ** np.arange(0, 31, dtype=np.uint64),
                        [0, 1, 2, 3, 2 ** 31 - 1]]).astype(np.uint32)
    z = co.normal_u32(w)
    want = stats.norm.ppf((w.astype(np.float64) + 0.5) / 2.0 ** 32)          # low half: negative quantiles ...
    assert np.max(np.abs(-z - want)) < 3e-7                                   # ... the table holds |z|; word w (top bit clear) is +|z|
    assert np.array_equal(co.normal_u32(w | np.uint32(0x80000000)), -z)        # the top bit is the sign: exactly symmetric
    assert z.max() < 6.34 and z.min() >= 0 and np.all(np.diff(z[:40000]) <= 1e-6)   # |z| falls as w rises


def test_draw_distributions():
    """The normals (inverse CDF of one word each) and the accept uniform drawn exactly as the sampler draws them."""
    z = np.array([co.draws(3, 0, 0, t)[0] for t in range(1, 20001)])
    u = np.exp(np.array([co.draws(3, 0, 0, t)[1] for t in range(1, 20001)]))       # draws() returns log u
    zz = z[:, :3].ravel()
    assert abs(zz.mean()) < 0.02 and abs(zz.std() - 1) < 0.02 and abs((zz ** 3).mean()) < 0.06 and abs((zz ** 4).mean() - 3) < 0.15
    assert np.abs(np.corrcoef(z[:, :3].T) - np.eye(3)).max() < 0.03
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.003 and u.min() >= 0 and u.max() < 1
    z2 = np.array([co.draws(2, 5, 9, t)[0] for t in range(1, 20001)])       # d = 2: 53-bit accept uniform
    assert np.all(z2[:, 2:] == 0) and abs(z2[:, :2].std() - 1) < 0.02 and abs(z2[:, :2].mean()) < 0.02
    assert np.abs(zz).max() < 6.34 and np.abs(z2).max() < 6.34               # documented truncation of the normals
    from scipy import stats
    assert stats.kstest(zz, "norm").pvalue > 1e-3 and stats.kstest(u, "uniform").pvalue > 1e-3
    # consecutive iterations are independent blocks: no correlation along t, in values or squares
    zc = z[:, 2]
    assert abs(np.corrcoef(zc[:-1], zc[1:])[0, 1]) < 0.02 and abs(np.corrcoef(zc[1:] ** 2, zc[:-1] ** 2)[0, 1]) < 0.03
    assert abs(np.corrcoef(zc[1:], z[:-1, 0])[0, 1]) < 0.02 and np.all(z[:, 3] == 0)
