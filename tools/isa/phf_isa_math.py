"""phf_math.h / phf_philox.h in gfx950 instructions, operation for operation (tools/isa/gfx950_asm.py builder).

Every function here emits EXACTLY the fp64 operation sequence of its C namesake in pyhillfit_amd/csrc/phf_math.h (the header the
hipcc kernels and the host twin compile) — same operations, same order, same constants — so that results are bit-identical; integer
bit manipulation is free to differ as long as the bits come out the same.  Functions take LISTS of arguments and advance them in
turn, instruction by instruction: independent dependency chains side by side.

LDS image of the tables (byte offsets inside the workgroup's LDS, all 16-byte aligned), copied from the constants blob by the
kernel's prologue; the blob itself is built by the host from the arrays of phf_math.h (phf_hier3_isa.cpp):
    exp2    64 doubles            0 ..  512      phf_t_exp2
    log     129 x {1/c, log c}  512 .. 2576      phf_t_log
    erfc    25 x 12 doubles    2576 .. 4976      phf_t_erfc
    normal  64 x 6 doubles     4976 .. 8048      phf_t_normal
"""
from gfx950_asm import Lit, Neg, Reg, VCC

EXP2_OFF, LOG_OFF, ERFC_OFF, NORMAL_OFF, TABLE_BYTES = 0, 512, 2576, 4976, 8048
LOG_TAB_BASE = 0x1ff35
ERFC_TAB_N = 25

# scalar constants of the blob, in this order behind the tables (doubles); the host writes them, the prologue loads them
CONSTS = [
    ("MAGIC", float.fromhex("0x1.8p52")), ("L2E64", float.fromhex("0x1.71547652b82fep+6")), ("NLN2HI64", float.fromhex("-0x1.62e42fee00000p-7")), ("NLN2LO64", float.fromhex("-0x1.a39ef35793c76p-39")),
    ("KE0", float.fromhex("0x1.5555555555555p-3")), ("KE1", float.fromhex("0x1.555565c3ff8a9p-5")), ("KE2", float.fromhex("0x1.11111a74dffd2p-7")), ("K100", 100.0),
    ("KL0", float.fromhex("0x1.55555555276f7p-2")), ("KL1", float.fromhex("-0x1.ffffffffafadap-3")), ("KL2", float.fromhex("0x1.999b080ce97c7p-3")), ("KL3", float.fromhex("-0x1.555695fa425fap-3")),
    ("LN2HI", float.fromhex("0x1.62e42fee00000p-1")), ("LN2LO", float.fromhex("0x1.a39ef35793c76p-33")), ("LN10", float.fromhex("0x1.26bb1bbb55516p+1")), ("ISQRT2", float.fromhex("0x1.6a09e667f3bcdp-1")),
    ("M746", -746.0), ("P710", 710.0), ("P40", 40.0), ("P6", 6.0),
    ("QUARTER", 0.25), ("DBLMIN", float.fromhex("0x1p-1022")), ("P2_1000", float.fromhex("0x1p1000")), ("LOGADD", None), ("MBITS", None),
]
CONST_BITS = {"LOGADD": (0x3ff0000000000000 - 0x3fe6a09e667f3bcd - (1023 << 52)) & 0xffffffffffffffff, "MBITS": 0x3fe6a09e667f3bcd}


def const_index(name):
    return [n for n, _ in CONSTS].index(name)


LOGPHI_TAB_N, LOGPHI_BYTES = 136, 136 * 80


class Ctx(object):
    """k: the Kernel; c: name -> SGPR pair (or VGPR pair) of a resident constant; off: LDS byte offset of each table in THIS kernel"""

    def __init__(self, k, c, off=None):
        self.k, self.c = k, c
        self.off = {"exp2": EXP2_OFF, "log": LOG_OFF, "erfc": ERFC_OFF, "normal": NORMAL_OFF}
        if off:
            self.off.update(off)


def _each(n, f):
    for i in range(n):
        f(i)


# ------------------------------------------------------------------------------------------------------------------ reciprocal, sqrt
def rcp(m, dsts, ds):
    """phf_rcp: hardware estimate, two Newton steps, the correcting step.  dsts[i] may not alias ds[i]."""
    k = m.k
    n = len(ds)
    e = [k.vd() for _ in range(n)]
    _each(n, lambda i: k.rcp_est(dsts[i], ds[i]))
    for _ in range(2):
        _each(n, lambda i: k.fma(e[i], Neg(ds[i]), dsts[i], 1.0))
        _each(n, lambda i: k.fma(dsts[i], dsts[i], e[i], dsts[i]))
    _each(n, lambda i: k.fma(e[i], Neg(ds[i]), dsts[i], 1.0))            # r = fma(-d, y, 1)
    _each(n, lambda i: k.fma(dsts[i], e[i], dsts[i], dsts[i]))           # fma(r, y, y)
    k.free(e)


def sqrt_nonneg(m, dsts, xs, masks):
    """phf_sqrt_nonneg: phf_sqrt_pos, then (x > 0) ? r : 0.  masks[i]: an SGPR pair that receives x > 0 (kept for the caller)."""
    k = m.k
    n = len(xs)
    g, h, r = [k.vd() for _ in range(n)], [k.vd() for _ in range(n)], [k.vd() for _ in range(n)]
    _each(n, lambda i: k.rsq_est(h[i], xs[i]))                           # y
    _each(n, lambda i: k.mul(g[i], xs[i], h[i]))                         # g = x y
    _each(n, lambda i: k.mul(h[i], h[i], 0.5))                           # h = y / 2
    _each(n, lambda i: k.fma(r[i], Neg(h[i]), g[i], 0.5))
    _each(n, lambda i: k.fma(g[i], g[i], r[i], g[i]))
    _each(n, lambda i: k.fma(h[i], h[i], r[i], h[i]))
    _each(n, lambda i: k.fma(r[i], Neg(g[i]), g[i], xs[i]))              # d
    _each(n, lambda i: k.fma(g[i], r[i], h[i], g[i]))
    _each(n, lambda i: k.fma(r[i], Neg(g[i]), g[i], xs[i]))
    _each(n, lambda i: k.fma(g[i], r[i], h[i], g[i]))
    _each(n, lambda i: k.cmp("gt", masks[i], xs[i], 0.0))
    _each(n, lambda i: k.cnd64(dsts[i], 0.0, g[i], masks[i]))
    k.free(g, h, r)


def sqrt_pos_chain(m, g, h, xs):
    """the Newton iteration of phf_sqrt_pos / phf_sqrt_rcp_pos: on return g[i] = sqrt(x) correctly rounded, h[i] ~ 1 / (2 sqrt x)"""
    k = m.k
    n = len(xs)
    r = [k.vd() for _ in range(n)]
    _each(n, lambda i: k.rsq_est(h[i], xs[i]))
    _each(n, lambda i: k.mul(g[i], xs[i], h[i]))
    _each(n, lambda i: k.mul(h[i], h[i], 0.5))
    _each(n, lambda i: k.fma(r[i], Neg(h[i]), g[i], 0.5))
    _each(n, lambda i: k.fma(g[i], g[i], r[i], g[i]))
    _each(n, lambda i: k.fma(h[i], h[i], r[i], h[i]))
    _each(n, lambda i: k.fma(r[i], Neg(g[i]), g[i], xs[i]))
    _each(n, lambda i: k.fma(g[i], r[i], h[i], g[i]))
    _each(n, lambda i: k.fma(r[i], Neg(g[i]), g[i], xs[i]))
    _each(n, lambda i: k.fma(g[i], r[i], h[i], g[i]))
    k.free(r)


def sqrt_pos(m, dsts, xs):
    """phf_sqrt_pos (no select: the caller guards x <= 0)"""
    k = m.k
    h = [k.vd() for _ in xs]
    sqrt_pos_chain(m, dsts, h, xs)
    k.free(h)


def sqrt_rcp_pos(m, dsts, invs, xs):
    """phf_sqrt_rcp_pos: g = sqrt(x) and 1 / g from ONE hardware estimate"""
    k = m.k
    n = len(xs)
    h = [k.vd() for _ in range(n)]
    e = [k.vd() for _ in range(n)]
    sqrt_pos_chain(m, dsts, h, xs)
    yi = invs
    _each(n, lambda i: k.add(yi[i], h[i], h[i]))
    _each(n, lambda i: k.fma(e[i], Neg(dsts[i]), yi[i], 1.0))
    _each(n, lambda i: k.fma(yi[i], yi[i], e[i], yi[i]))
    _each(n, lambda i: k.fma(e[i], Neg(dsts[i]), yi[i], 1.0))
    _each(n, lambda i: k.fma(yi[i], e[i], yi[i], yi[i]))
    k.free(h, e)


# ------------------------------------------------------------------------------------------------------------------ exp
def exp_core(m, dsts, xcs):
    """phf_exp_core_k on arguments already clamped; xcs[i] is destroyed (it becomes r); dsts[i] may not alias xcs[i]"""
    k, c = m.k, m.c
    n = len(xcs)
    t = [k.vd() for _ in range(n)]
    nd = [k.vd() for _ in range(n)]
    ad = [k.v1() for _ in range(n)]
    _each(n, lambda i: k.fma(t[i], xcs[i], c["L2E64"], c["MAGICV"]))
    _each(n, lambda i: k.add(nd[i], t[i], Neg(c["MAGICV"])))
    _each(n, lambda i: k.vop("v_lshlrev_b32_e32", ad[i], 3, t[i].lo()))
    _each(n, lambda i: k.vop("v_and_b32_e32", ad[i], Lit(0x1f8), ad[i]))
    tj = dsts
    _each(n, lambda i: k.ds_read(tj[i], ad[i], m.off["exp2"]))
    r = xcs
    _each(n, lambda i: k.fma(r[i], nd[i], c["NLN2HI64"], xcs[i]))
    _each(n, lambda i: k.fma(r[i], nd[i], c["NLN2LO64"], r[i]))
    r2, q = nd, [k.vd() for _ in range(n)]
    _each(n, lambda i: k.mul(r2[i], r[i], r[i]))
    _each(n, lambda i: k.fma(q[i], c["KE2V"], r[i], c["KE1"]))
    _each(n, lambda i: k.fma(q[i], q[i], r[i], c["KE0"]))
    _each(n, lambda i: k.fma(q[i], q[i], r[i], 0.5))
    _each(n, lambda i: k.fma(q[i], r2[i], q[i], r[i]))                   # p
    _each(n, lambda i: k.vop("v_ashrrev_i32_e32", ad[i], 6, t[i].lo()))
    _each(n, lambda i: k.fma(q[i], tj[i], q[i], tj[i]))
    _each(n, lambda i: k.ldexp(dsts[i], q[i], ad[i]))
    k.free(t, nd, ad, q)


def exp_fast(m, dsts, xs):
    """phf_exp_fast_k; xs[i] destroyed"""
    k, c = m.k, m.c
    _each(len(xs), lambda i: k.fmax(xs[i], xs[i], c["M746"]))
    _each(len(xs), lambda i: k.fmin(xs[i], xs[i], c["P710"]))
    exp_core(m, dsts, xs)


def exp_capped(m, dsts, xs):
    """phf_exp_capped_k; xs[i] destroyed"""
    k, c = m.k, m.c
    _each(len(xs), lambda i: k.fmax(xs[i], xs[i], c["M746"]))
    exp_core(m, dsts, xs)


# ------------------------------------------------------------------------------------------------------------------ log
def log_pos(m, dsts, xs):
    """phf_log_pos_k; xs[i] is kept; dsts[i] may alias xs[i]"""
    k, c = m.k, m.c
    n = len(xs)
    u = [k.vd() for _ in range(n)]
    e = [k.v1() for _ in range(n)]
    ad = [k.v1() for _ in range(n)]
    tab = [k.vq() for _ in range(n)]
    # u = bits(x) + (bits(1) - bits(sqrt 1/2)) - (1023 << 52): the exponent field then holds e = floor(log2(x / sqrt(1/2))) as a signed
    # number (one arithmetic shift), and the mantissa bits below it are what they would be without the last term
    _each(n, lambda i: k.emit("v_lshl_add_u64", [u[i]], [xs[i], 0, c["LOGADD"]], "valu", count="valu_int"))
    _each(n, lambda i: k.vop("v_ashrrev_i32_e32", e[i], 20, u[i].hi()))
    _each(n, lambda i: k.vop("v_and_b32_e32", u[i].hi(), Lit(0xfffff), u[i].hi()))
    _each(n, lambda i: k.emit("v_lshl_add_u64", [u[i]], [u[i], 0, c["MBITS"]], "valu", count="valu_int"))     # bits of m
    _each(n, lambda i: k.vop("v_add_u32_e32", ad[i], Lit(0x1000 - (LOG_TAB_BASE << 13)), u[i].hi()))
    _each(n, lambda i: k.vop("v_lshrrev_b32_e32", ad[i], 9, ad[i]))
    _each(n, lambda i: k.vop("v_and_b32_e32", ad[i], Lit(0x7ffff0), ad[i]))
    _each(n, lambda i: k.ds_read(tab[i], ad[i], m.off["log"]))
    dk = [k.vd() for _ in range(n)]
    _each(n, lambda i: k.cvt_f64_i32(dk[i], e[i]))
    r = u
    _each(n, lambda i: k.fma(r[i], u[i], tab[i].sub(0), -1.0))
    r2, q = [k.vd() for _ in range(n)], [k.vd() for _ in range(n)]
    _each(n, lambda i: k.mul(r2[i], r[i], r[i]))
    _each(n, lambda i: k.fma(q[i], c["KL3V"], r[i], c["KL2"]))
    _each(n, lambda i: k.fma(q[i], q[i], r[i], c["KL1"]))
    _each(n, lambda i: k.fma(q[i], q[i], r[i], c["KL0"]))
    _each(n, lambda i: k.fma(q[i], q[i], r[i], -0.5))
    hi = [tab[i].sub(2) for i in range(n)]
    _each(n, lambda i: k.fma(hi[i], dk[i], c["LN2HI"], hi[i]))            # fma(dk, LN2_HI, logc)
    _each(n, lambda i: k.mul(r2[i], r2[i], q[i]))
    _each(n, lambda i: k.fma(r2[i], dk[i], c["LN2LO"], r2[i]))            # lo
    _each(n, lambda i: k.add(r2[i], r[i], r2[i]))                        # r + lo
    _each(n, lambda i: k.add(dsts[i], hi[i], r2[i]))
    k.free(u, e, ad, tab, dk, r2, q)


def log_fast(m, dsts, xs):
    """phf_log_fast_k: log_pos, then -inf below 2^-1022.  dsts[i] may NOT alias xs[i]"""
    k, c = m.k, m.c
    log_pos(m, dsts, xs)
    for i in range(len(xs)):
        k.cmp_lit("ngt", 0x00100000, xs[i])                                # !(2^-1022 > x): keep r (also for a NaN, as the C select does)
        k.cnd32_vcc(dsts[i].lo(), 0, dsts[i].lo())
        k.cnd32_vcc(dsts[i].hi(), c["NINFHI"], dsts[i].hi())                # high word of -inf in a VGPR (VCC is the constant-bus read)


# ------------------------------------------------------------------------------------------------------------------ erfc table
def erfc_tab(m, dsts, ys):
    """phf_erfc_tab; ys kept"""
    k, c = m.k, m.c
    n = len(ys)
    t = [k.vd() for _ in range(n)]
    s = [k.vd() for _ in range(n)]
    ad = [k.v1() for _ in range(n)]
    tab = [[k.vq() for _ in range(6)] for _ in range(n)]
    _each(n, lambda i: k.fma(t[i], ys[i], 4.0, c["MAGICV"]))
    _each(n, lambda i: k.vop("v_min_u32_e32", ad[i], ERFC_TAB_N - 1, t[i].lo()))
    _each(n, lambda i: k.vop("v_mul_u32_u24_e32", ad[i], Lit(96), ad[i]))
    for j in (5, 4, 3, 2, 1, 0):
        _each(n, lambda i: k.ds_read(tab[i][j], ad[i], m.off["erfc"] + 16 * j))
    _each(n, lambda i: k.add(t[i], t[i], Neg(c["MAGICV"])))
    _each(n, lambda i: k.fma(s[i], t[i], Neg(c["QUARTER"]), ys[i]))
    p = t
    cf = lambda i, j: tab[i][j // 2].sub(2 * (j % 2))
    _each(n, lambda i: k.fma(p[i], cf(i, 11), s[i], cf(i, 10)))
    for j in range(9, -1, -1):
        _each(n, lambda i, j=j: k.fma(p[i], p[i], s[i], cf(i, j)))
    for i in range(n):
        k.cmp_lit("gt", 0x40180000, ys[i])                                 # y < 6
        k.cnd32_vcc(dsts[i].lo(), 0, p[i].lo())
        k.cnd32_vcc(dsts[i].hi(), 0, p[i].hi())
    k.free(t, s, ad, tab)


# ------------------------------------------------------------------------------------------------------------------ log Phi table
def log_ndtr_tab(m, dsts, xs):
    """phf_log_ndtr_tab(x, y = -x / sqrt 2): log Phi(x) for x <= 0 from the table; xs kept; dsts[i] may not alias xs[i]"""
    k, c = m.k, m.c
    n = len(xs)
    y = [k.vd() for _ in range(n)]
    vb = [k.vd() for _ in range(n)]
    ad = [k.v1() for _ in range(n)]
    tab = [[k.vq() for _ in range(5)] for _ in range(n)]
    _each(n, lambda i: k.mul(y[i], Neg(xs[i]), c["ISQRT2"]))                  # y = -x * (1 / sqrt 2)
    _each(n, lambda i: k.add(vb[i], y[i], 1.0))
    # jr = (hi >> 17) - (0x3ff << 3), clamped below 136 (unsigned: a negative jr is huge); 80 bytes per interval
    _each(n, lambda i: k.vop("v_lshrrev_b32_e32", ad[i], 17, vb[i].hi()))
    _each(n, lambda i: k.vop("v_add_u32_e32", ad[i], Lit(-(0x3ff << 3)), ad[i]))
    _each(n, lambda i: k.vop("v_min_u32_e32", ad[i], Lit(LOGPHI_TAB_N - 1), ad[i]))
    _each(n, lambda i: k.vop("v_mul_u32_u24_e32", ad[i], Lit(80), ad[i]))
    for j in (4, 3, 2, 1, 0):
        _each(n, lambda i: k.ds_read(tab[i][j], ad[i], m.off["logphi"] + 16 * j))
    _each(n, lambda i: k.vop("v_bfi_b32", vb[i].hi(), c["MANTHI"], vb[i].hi(), c["ONEHI"]))
    sft = vb
    _each(n, lambda i: k.add(sft[i], vb[i], -1.0))
    cf = lambda i, j: tab[i][j // 2].sub(2 * (j % 2))
    gq = y
    _each(n, lambda i: k.fma(gq[i], cf(i, 9), sft[i], cf(i, 8)))
    for j in range(7, -1, -1):
        _each(n, lambda i, j=j: k.fma(gq[i], gq[i], sft[i], cf(i, j)))
    _each(n, lambda i: k.mul(sft[i], xs[i], -0.5))                            # -0.5 * x
    _each(n, lambda i: k.fma(dsts[i], sft[i], xs[i], gq[i]))
    k.free(y, vb, ad, tab)


# ------------------------------------------------------------------------------------------------------------------ normals, uniforms
def normal_u32(m, dsts, ws):
    """phf_normal_u32 of 32-bit words ws[i] (VGPRs, kept)"""
    k, c = m.k, m.c
    n = len(ws)
    ab = [k.vd() for _ in range(n)]
    ad = [k.v1() for _ in range(n)]
    tab = [[k.vq() for _ in range(3)] for _ in range(n)]
    _each(n, lambda i: k.vop("v_lshl_or_b32", ad[i], ws[i], 1, 1))
    _each(n, lambda i: k.cvt_f64_u32(ab[i], ad[i]))
    _each(n, lambda i: k.vop("v_lshrrev_b32_e32", ad[i], 19, ab[i].hi()))
    _each(n, lambda i: k.vop("v_mad_u32_u24", ad[i], ad[i], 48, c["NORMBIAS"]))               # 48 (j + 2046) - 48 * 2046 (an SGPR)
    for j in (2, 1, 0):
        _each(n, lambda i: k.ds_read(tab[i][j], ad[i], m.off["normal"] + 16 * j))
    _each(n, lambda i: k.vop("v_bfi_b32", ab[i].hi(), c["MANTHI"], ab[i].hi(), c["ONEHI"]))    # mantissa bits under the exponent of 1.0
    sft = ab
    _each(n, lambda i: k.add(sft[i], ab[i], -1.0))
    cf = lambda i, j: tab[i][j // 2].sub(2 * (j % 2))
    z = dsts
    _each(n, lambda i: k.fma(z[i], cf(i, 5), sft[i], cf(i, 4)))
    for j in (3, 2, 1, 0):
        _each(n, lambda i, j=j: k.fma(z[i], z[i], sft[i], cf(i, j)))
    _each(n, lambda i: k.vop("v_bfi_b32", z[i].hi(), c["ABSMASK"], z[i].hi(), ws[i]))      # |P| with the sign bit of the word
    k.free(ab, ad, tab)


def unit_open32(m, dst, w):
    """phf_unit_open32: ((double)w + 0.5) * 2^-32"""
    k = m.k
    k.cvt_f64_u32(dst, w)
    k.add(dst, dst, 0.5)
    e = k.s1()
    k.sop("s_mov_b32", e, Lit(-32))
    k.ldexp(dst, dst, e)                                                  # the multiplication by 2^-32 is exact: same bits
    k.free(e)


# ------------------------------------------------------------------------------------------------------------------ Philox4x32-R
PHILOX_M0, PHILOX_M1, PHILOX_W0, PHILOX_W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox(m, c0, c1, c2, c3, keys, rounds=7):
    """phf_philox4x32_r.  c0..c3: counter words (VGPRs, SGPRs or inline integers; kept); keys[r] = (k0_r, k1_r), SGPRs, r = 0..rounds-1
    (the key schedule is wave-uniform: the caller advances it on the scalar unit); M0 / M1 in m.c["PM0"] / ["PM1"] (SGPRs).
    Returns (words, pairs): the four output words as VGPRs living inside `pairs` (two 64-bit products the caller frees when done).
    The two products of a round go to alternating register pairs, so that a round's low words stay readable as the next round's c1 / c3
    without a copy."""
    k, c = m.k, m.c
    pa = [(k.vd(), k.vd()), (k.vd(), k.vd())]
    a0, a1, a2, a3 = c0, c1, c2, c3
    for r in range(rounds):
        p0, p1 = pa[r % 2]
        k.mad_u64_u32(p0, a0, c["PM0"])
        k.mad_u64_u32(p1, a2, c["PM1"])
        k.xor3(p1.hi(), p1.hi(), a1, keys[r][0])                           # n0
        k.xor3(p0.hi(), p0.hi(), a3, keys[r][1])                           # n2
        a0, a1, a2, a3 = p1.hi(), p1.lo(), p0.hi(), p0.lo()
    last = pa[(rounds - 1) % 2]
    k.free(pa[rounds % 2])
    return [a0, a1, a2, a3], list(last)
