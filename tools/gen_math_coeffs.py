#!/usr/bin/env python3
"""Derive the polynomial coefficients used by pyhillfit_amd/csrc/phf_math.h (mpmath, 60 digits).

Chebyshev interpolation (near-minimax) of each kernel function, converted to the monomial basis
and rounded to double; the reported max relative error is measured against mpmath on a dense grid
using exact rational arithmetic on the rounded coefficients (i.e. the approximation error only;
rounding error of the fma evaluation comes on top, < 1 ulp)."""
import sys
import mpmath as mp

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    n = deg + 1
    nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / n) for k in range(n)]
    xs = [(a + b) / 2 + (b - a) / 2 * t for t in nodes]
    fs = [f(x) for x in xs]
    c = []
    for j in range(n):
        s = mp.fsum(fs[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / n) for k in range(n))
        c.append(2 * s / n)
    c[0] /= 2
    # Chebyshev -> monomial in t, then t = (2x - a - b)/(b - a)
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for j in range(2, n):
        prev, prev2 = T[j - 1], T[j - 2]
        cur = [mp.mpf(0)] + [2 * v for v in prev]
        for i, v in enumerate(prev2):
            cur[i] -= v
        T.append(cur)
    mono_t = [mp.mpf(0)] * n
    for j in range(n):
        for i, v in enumerate(T[j]):
            mono_t[i] += c[j] * v
    # substitute t = alpha*x + beta
    alpha, beta = 2 / (b - a), -(a + b) / (b - a)
    poly = [mp.mpf(0)] * n
    # (alpha x + beta)^i expansion
    for i, coef in enumerate(mono_t):
        for k in range(i + 1):
            poly[k] += coef * mp.binomial(i, k) * alpha ** k * beta ** (i - k)
    return poly


def to_double(poly):
    return [float(v) for v in poly]


def horner(coefs, x):
    r = mp.mpf(0)
    for c in reversed(coefs):
        r = r * x + mp.mpf(c)
    return r


def max_rel_err(approx, exact, a, b, npts=4001):
    worst = mp.mpf(0)
    for k in range(npts):
        x = a + (b - a) * mp.mpf(k) / (npts - 1)
        e = exact(x)
        if e == 0:
            continue
        worst = max(worst, abs(approx(x) / e - 1))
    return worst


def show(name, coefs):
    print("/* %s */" % name)
    for i, c in enumerate(coefs):
        print("  %s,  /* x^%d */" % (float(c).hex(), i), "  // %.17g" % float(c))


def gen_exp():
    a = mp.log(2) / 2 * mp.mpf("1.0001")
    q = lambda r: (mp.exp(r) - 1 - r) / (r * r) if abs(r) > mp.mpf('1e-15') else mp.mpf(1) / 2 + r / 6
    for deg in (8, 9, 10):
        c = to_double(cheb_fit(q, -a, a, deg))
        err = max_rel_err(lambda r: 1 + r + r * r * horner(c, r), mp.exp, -a, a)
        print("exp: q deg", deg, "rel err", mp.nstr(err, 3))
    c = to_double(cheb_fit(q, -a, a, 9))
    show("exp: (exp(r)-1-r)/r^2, |r|<=ln2/2", c)


def gen_log():
    smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1) * mp.mpf("1.0001")
    zmax = smax ** 2
    # log((1+s)/(1-s)) = 2s + s*z*G(z),  G(z) = sum 2/(2k+3) z^k
    G = lambda z: (mp.log((1 + mp.sqrt(z)) / (1 - mp.sqrt(z))) - 2 * mp.sqrt(z)) / (mp.sqrt(z) * z) if z > mp.mpf('1e-20') else mp.mpf(2) / 3
    for deg in (5, 6, 7):
        c = to_double(cheb_fit(G, mp.mpf(0), zmax, deg))
        f = lambda s: 2 * s + s * s * s * horner(c, s * s)
        ex = lambda s: mp.log((1 + s) / (1 - s))
        err = max_rel_err(f, ex, mp.mpf("1e-6"), smax)
        print("log: G deg", deg, "rel err", mp.nstr(err, 3))
    c = to_double(cheb_fit(G, mp.mpf(0), zmax, 6))
    show("log: G(z), z=s^2<=0.02944", c)


def gen_erfcx():
    """(1+2y)*erfcx(y) as a polynomial in t=(y-K)/(y+K), y in [0,inf) -> t in [-1,1]; K=4, degree 22."""
    K = mp.mpf(4)

    def g(t):
        if t >= 1:
            return 2 / mp.sqrt(mp.pi)
        y = K * (1 + t) / (1 - t)
        return (1 + 2 * y) * mp.exp(y * y) * mp.erfc(y)
    c = to_double(cheb_fit(g, mp.mpf(-1), mp.mpf(1), 22))
    err = max_rel_err(lambda t: horner(c, t), g, mp.mpf(-1), mp.mpf(1), 4001)
    print("erfcx: K 4 deg 22 rel err", mp.nstr(err, 3))
    show("erfcx: (1+2y)erfcx(y) in t=(y-4)/(y+4)", c)


def gen_sincos():
    a = mp.pi / 8 * mp.mpf("1.0001")
    # sin(x) = x + x^3 * S(x^2); cos(x) = 1 - x^2/2 + x^4*C(x^2)
    S = lambda z: (mp.sin(mp.sqrt(z)) - mp.sqrt(z)) / (mp.sqrt(z) * z) if z > mp.mpf('1e-20') else -mp.mpf(1) / 6
    C = lambda z: (mp.cos(mp.sqrt(z)) - 1 + z / 2) / (z * z) if z > mp.mpf('1e-20') else mp.mpf(1) / 24
    for deg in (4, 5, 6):
        cs = to_double(cheb_fit(S, mp.mpf(0), a * a, deg))
        cc = to_double(cheb_fit(C, mp.mpf(0), a * a, deg))
        es = max_rel_err(lambda x: x + x ** 3 * horner(cs, x * x), mp.sin, mp.mpf("1e-5"), a)
        ec = max_rel_err(lambda x: 1 - x * x / 2 + x ** 4 * horner(cc, x * x), mp.cos, mp.mpf(0), a)
        print("sincos: deg", deg, "sin err", mp.nstr(es, 3), "cos err", mp.nstr(ec, 3))
    show("sin: S(z)", to_double(cheb_fit(S, mp.mpf(0), a * a, 4)))
    show("cos: C(z)", to_double(cheb_fit(C, mp.mpf(0), a * a, 4)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["exp", "log", "erfcx", "sincos"]
    for w in which:
        globals()["gen_" + w]()
