#!/usr/bin/env python3
"""Derive the tables and polynomial coefficients used by pyhillfit_amd/csrc/phf_math.h (mpmath, 60 digits).

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
    """exp(x) = 2^k * T[j] * (1 + p(r)), n = nearest integer to 64 x / ln2 = 64 k + j, r = x - n ln2/64 (|r| <= ln2/128),
    T[j] = 2^(j/64) (64 doubles, correctly rounded), p(r) = r + r^2/2 + r^3 q(r), q degree 2."""
    print("/* 2^(j/64), j = 0..63 */")
    row = [float(mp.mpf(2) ** (mp.mpf(j) / 64)).hex() for j in range(64)]
    for i in range(0, 64, 4):
        print("    " + ", ".join(row[i:i + 4]) + ",")
    a = mp.log(2) / 128 * mp.mpf("1.001")
    q = lambda r: (mp.expm1(r) - r - r * r / 2) / r ** 3 if abs(r) > mp.mpf('1e-12') else mp.mpf(1) / 6 + r / 24
    for deg in (1, 2, 3):
        c = to_double(cheb_fit(q, -a, a, deg))
        err = max_rel_err(lambda r: 1 + r + r * r / 2 + r ** 3 * horner(c, r), mp.exp, -a, a)
        print("exp: q deg", deg, "rel err", mp.nstr(err, 3))
    show("exp: (expm1(r) - r - r^2/2)/r^3, |r| <= ln2/128", to_double(cheb_fit(q, -a, a, 2)))


LOG_LO = 0x3fe6a09e667f3bcd            # bits of the smallest reduced mantissa m (sqrt(1/2) rounded up)


def gen_log():
    """log(x) = k ln2 + log c_j + log1p(r): x = 2^k m, m in [sqrt(1/2), sqrt 2); c_j = the double whose bits are (base + j) << 45
    nearest to m (129 grid points, c = 1 among them); r = m * invc_j - 1 by one fma (|r| <= 2^-8);
    table: invc_j = 1/c_j rounded, logc_j = -log(invc_j); log1p(r) = r + r^2 q(r), q degree 4 (q(0) = -1/2 exactly)."""
    import struct
    frombits = lambda u: struct.unpack('<d', struct.pack('<Q', u))[0]
    base = (LOG_LO + (1 << 44)) >> 45
    n = (((LOG_LO + (1 << 52)) + (1 << 44)) >> 45) - base + 1
    print("/* log table: base %s, %d entries {1/c, log c} */" % (hex(base), n))
    worst = mp.mpf(0)
    for j in range(n):
        c = frombits((base + j) << 45)
        invc = float(1 / mp.mpf(c))
        logc = float(-mp.log(mp.mpf(invc)))
        print("    {%s, %s},%s" % (invc.hex(), logc.hex(), "   /* c = %.10g */" % c if j % 16 == 0 or c == 1.0 else ""))
        lo = frombits(max(((base + j) << 45) - (1 << 44), LOG_LO))
        hi = frombits(min(((base + j) << 45) + (1 << 44) - 1, LOG_LO + (1 << 52) - 1))
        worst = max(worst, abs(mp.mpf(lo) * invc - 1), abs(mp.mpf(hi) * invc - 1))
    print("log: max |r|", mp.nstr(worst, 8))
    w = worst * mp.mpf("1.001")
    q = lambda r: (mp.log1p(r) - r) / (r * r) if abs(r) > mp.mpf('1e-20') else -mp.mpf(1) / 2 + r / 3
    for deg in (3, 4, 5):
        c = to_double(cheb_fit(q, -w, w, deg))
        err = max_rel_err(lambda r: r + r * r * horner(c, r), mp.log1p, -w, w, 4000)   # even count: r = 0 is not sampled
        print("log: q deg", deg, "rel err of log1p", mp.nstr(err, 3), "q(0) + 1/2 =", c[0] + 0.5)
    show("log: (log1p(r) - r)/r^2, |r| <= 2^-8", to_double(cheb_fit(q, -w, w, 4)))


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


def gen_normal():
    """Standard normal from 32 random bits by a piecewise inverse CDF: w = the low 31 bits, p = (w + 1/2) / 2^32 in (0, 1/2),
    |z| = -Phi^-1(p), sign = the top bit.  a = 2 w + 1 = 2^E m (E = 0..31, m in [1, 2)); interval 2 E + k, k = floor(2 (m - 1));
    on it |z| = P(s), s = m - 1, degree 5, coefficients of s^0..s^5 (monomials about s = 0).  64 x 6 doubles."""
    def z_of(E, s):
        p = mp.mpf(2) ** (E - 33) * (1 + s)
        return -mp.sqrt(2) * mp.erfinv(2 * p - 1)
    worst = mp.mpf(0)
    print("/* |z| on interval 2 E + k: coefficients of s^0 .. s^5 */")
    for E in range(32):
        for k in range(2):
            a, b = mp.mpf(k) / 2, mp.mpf(k + 1) / 2
            cf = to_double(cheb_fit(lambda s: z_of(E, s), a, b, 5))
            print("    {{%s}}," % ", ".join(float(v).hex() for v in cf))
            for i in range(41):
                s = a + (b - a) * i / 40
                worst = max(worst, abs(horner(cf, s) - z_of(E, s)))
    print("normal: max abs error", mp.nstr(worst, 3))


LOGPHI_BINADES = 17


def gen_logphi():
    """log Phi(x) for x <= 0 (the censored likelihood's only case) = -x^2/2 + g(y), y = -x/sqrt2, g(y) = log(erfcx(y)/2): smooth and
    slowly varying.  v = y + 1 = 2^E m; interval 8 E + k, k = floor(8 (m - 1)), E = 0..16 (y < 131071); on it g = P(s), s = m - 1,
    degree 9, coefficients of s^0..s^9 (monomials about s = 0).  136 x 10 doubles."""
    def g_of(E, s):
        y = mp.mpf(2) ** E * (1 + s) - 1
        if y > 40:            # erfcx(y) = 1/(y sqrt(pi)) sum_n (-1)^n (2n-1)!! / (2 y^2)^n: the terms fall below 1e-55 long before they grow
            u, ssum, term = 1 / (2 * y * y), mp.mpf(1), mp.mpf(1)
            for n in range(1, 60):
                term *= -(2 * n - 1) * u
                ssum += term
                if abs(term) < mp.mpf('1e-55'):
                    break
            return mp.log(ssum / (y * mp.sqrt(mp.pi)) / 2)
        return mp.log(mp.exp(y * y) * mp.erfc(y) / 2)
    worst = mp.mpf(0)
    print("/* g on interval 8 E + k: coefficients of s^0 .. s^9 */")
    for E in range(LOGPHI_BINADES):
        for k in range(8):
            a, b = mp.mpf(k) / 8, mp.mpf(k + 1) / 8
            cf = to_double(cheb_fit(lambda s: g_of(E, s), a, b, 9))
            print("    {{%s}}," % ", ".join(float(v).hex() for v in cf))
            for i in range(21):
                s = a + (b - a) * i / 20
                worst = max(worst, abs(horner(cf, s) - g_of(E, s)))
    print("logphi: max abs error of g", mp.nstr(worst, 3))


def gen_erfc():
    """erfc(y) on [0, 6) to an ABSOLUTE accuracy below half an ulp of 1 — what the hierarchical target's truncation masses
    1 - (erfc(ya) + erfc(yb))/2 need: centres c_j = j/4, j = 0..24 (the nearest to y), s = y - c_j in [-1/8, 1/8], degree 11,
    coefficients of s^0..s^11.  25 x 12 doubles.  (erfc(6) = 2.2e-17: from there on the tail counts as zero.)"""
    H = mp.mpf(1) / 8
    worst = mp.mpf(0)
    print("/* erfc about c_j = j/4: coefficients of s^0 .. s^11 */")
    for j in range(25):
        c = mp.mpf(j) / 4
        cf = to_double(cheb_fit(lambda s: mp.erfc(c + s), -H, H, 11))
        print("    {{%s}}," % ", ".join(float(v).hex() for v in cf))
        lo = -H if j > 0 else mp.mpf(0)
        for i in range(41):
            s = lo + (H - lo) * i / 40
            worst = max(worst, abs(horner(cf, s) - mp.erfc(c + s)))
    print("erfc: max abs error", mp.nstr(worst, 3))


if __name__ == "__main__":
    which = sys.argv[1:] or ["exp", "log", "erfcx", "sincos", "normal", "logphi", "erfc"]
    for w in which:
        globals()["gen_" + w]()
