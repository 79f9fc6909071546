"""ctypes binding of oracle/_build/libphf_oracle.so — TEST INFRASTRUCTURE (see phf_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libphf_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


class Problem(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_other", C.c_int32), ("n_zero", C.c_int32), ("n_hundred", C.c_int32),
                ("ln_conc", C.c_void_p), ("response", C.c_void_p), ("weight", C.c_void_p), ("n_other_points", C.c_double),
                ("ss_within", C.c_double), ("pi_bit", C.c_double), ("temperature", C.c_double)]


class Run(C.Structure):
    _fields_ = [("t_begin", C.c_int64), ("t_end", C.c_int64), ("thinning", C.c_int32), ("reset_mean", C.c_int32),
                ("adapt_start", C.c_int64), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("chain_id", C.c_uint32), ("problem_id", C.c_uint32), ("gamma", C.c_void_p)]


class HierPrior(C.Structure):
    _fields_ = [("shape_m1", C.c_double * 5), ("inv_scale", C.c_double * 5), ("loc", C.c_double * 5)]


class HierProblem(C.Structure):
    _fields_ = [("n_expts", C.c_int32), ("expt_start", C.c_void_p), ("ln_conc", C.c_void_p), ("response", C.c_void_p),
                ("prior", HierPrior)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.phfo_log_target.restype = C.c_double
        L.phfo_log_likelihood.restype = C.c_double
        L.phfo_log_prior.restype = C.c_double
        L.phfo_state_size.restype = C.c_int
        L.phfo_log_likelihood_t1.restype = C.c_double
        L.phfo_hier_log_target.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gamma_table(n):
    """gamma[s] = 1/(s+1)**0.6 evaluated like the reference does (Python float pow), s = 0..n."""
    return np.array([1.0 / (s + 1.0) ** 0.6 for s in range(n + 1)], dtype=np.float64)


def _merged(concs, y):
    """replicates at one concentration -> (conc, mean, count) in first-appearance order, and the within-group sum of
    squares; correctly rounded sums (math.fsum), so any implementation of this grouping gives the same doubles"""
    import math
    seen = []
    for cv in concs:
        if cv not in seen:
            seen.append(cv)
    cc, yy, ww, terms = [], [], [], []
    for cv in seen:
        v = [float(b) for a, b in zip(concs, y) if a == cv]
        m = v[0] if len(v) == 1 else math.fsum(v) / len(v)
        cc.append(float(cv)); yy.append(m); ww.append(float(len(v)))
        if len(v) > 1:
            terms += [(b - m) * (b - m) for b in v]
    return cc, yy, ww, math.fsum(terms)


class PackedPair:
    """One pair in the kernel's order: 'other' entries, then y==0 entries, then y==100 entries (an entry = the points
    of that kind at one concentration, see include/pyhillfit_amd.h); merge=False keeps one entry per point."""

    def __init__(self, concs, responses, model, temperature=1.0, merge=True):
        concs = np.asarray(concs, float); y = np.asarray(responses, float)
        other = (0 < y) & (y < 100); zero = y == 0; hund = y == 100
        cc, yy, ww, ks, ss = [], [], [], [], 0.0
        for mask in (other, zero, hund):
            if merge:
                a, b, w, s_ = _merged(list(concs[mask]), list(y[mask]))
            else:
                a, b, w, s_ = list(concs[mask]), list(y[mask]), [1.0] * int(mask.sum()), 0.0
            cc += a; yy += b; ww += w; ks.append(len(a)); ss += s_
        with np.errstate(divide="ignore"):
            self.ln_conc = np.ascontiguousarray(np.log(np.array(cc, dtype=float)))
        self.response = np.ascontiguousarray(np.array(yy, dtype=float))
        self.weight = np.ascontiguousarray(np.array(ww, dtype=float))
        self.pb = Problem(model, ks[0], ks[1], ks[2], _p(self.ln_conc), _p(self.response), _p(self.weight),
                          float(other.sum()), ss, 0.5 * len(y) * np.log(2 * np.pi), float(temperature))
        self.d = 2 if model == 1 else 3

    def log_target(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        return lib().phfo_log_target(C.byref(self.pb), _p(th))

    def log_likelihood(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        return lib().phfo_log_likelihood(C.byref(self.pb), _p(th))

    def log_likelihood_t1(self, theta):
        """log_data_likelihood(..., t=1) regardless of this problem's temperature (compute_bayes_factors.py:19-20)"""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        return lib().phfo_log_likelihood_t1(C.byref(self.pb), _p(th))

    def log_prior(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        return lib().phfo_log_prior(C.byref(self.pb), _p(th))

    def init_state(self, theta0, cov_identity, cov_scale):
        st = np.zeros(lib().phfo_state_size(self.d))
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        lib().phfo_init_state(C.byref(self.pb), int(cov_identity), C.c_double(cov_scale), _p(th), _p(st))
        return st

    def advance(self, st, t_begin, t_end, thinning, adapt_start, reset_mean, gamma, seed=25, chain_id=0, problem_id=0,
                star_replay=None, u_replay=None, trace_cov=False):
        rows = t_end // thinning - t_begin // thinning
        out = np.zeros((rows, self.d + 1))
        run = Run(t_begin, t_end, thinning, int(reset_mean), adapt_start, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
                  chain_id, problem_id, _p(gamma))
        sr = np.ascontiguousarray(star_replay, dtype=np.float64) if star_replay is not None else None
        ur = np.ascontiguousarray(u_replay, dtype=np.float64) if u_replay is not None else None
        tr = np.zeros((t_end - t_begin, self.d, self.d)) if trace_cov else None
        lib().phfo_advance(C.byref(self.pb), C.byref(run), _p(st), _p(out), _p(sr) if sr is not None else None,
                           _p(ur) if ur is not None else None, _p(tr) if tr is not None else None)
        return (out, tr) if trace_cov else out


def vec(name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    getattr(lib(), "phfo_vec_" + name)(C.c_int64(x.size), _p(x), _p(out))
    return out


def sincos(w):
    w = np.ascontiguousarray(w, dtype=np.uint32)
    s = np.empty(w.size); c = np.empty(w.size)
    lib().phfo_vec_sincos(C.c_int64(w.size), _p(w), _p(s), _p(c))
    return s, c


def normal_u32(w):
    """the single-level sampler's standard normal of a 32-bit word (phf_math.h: phf_normal_u32)"""
    w = np.ascontiguousarray(w, dtype=np.uint32)
    out = np.empty(w.size)
    lib().phfo_vec_normal_u32(C.c_int64(w.size), _p(w), _p(out))
    return out


def philox_rounds():
    """rounds of the Philox4x32 block the samplers draw from (phf_philox.h: PHF_PHILOX_ROUNDS)"""
    return int(lib().phfo_philox_rounds())


def philox(ctr_key, rounds=0):
    """rounds: 7 or 10; 0 = the samplers' own"""
    ck = np.ascontiguousarray(ctr_key, dtype=np.uint32).reshape(-1, 6)
    out = np.empty((ck.shape[0], 4), dtype=np.uint32)
    lib().phfo_philox(C.c_int(rounds), C.c_int64(ck.shape[0]), _p(ck), _p(out))
    return out


def draws(d, chain_id, problem_id, t, seed=25):
    """(z[4], log u) of MH iteration t"""
    z = np.zeros(4); lu = C.c_double()
    lib().phfo_draws(d, chain_id, problem_id, t, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, _p(z), C.byref(lu))
    return z, lu.value


def hier_prior(shapes, scales, locs):
    pr = HierPrior()
    for k in range(5):
        pr.shape_m1[k] = shapes[k] - 1.0; pr.inv_scale[k] = 1.0 / scales[k]; pr.loc[k] = locs[k]
    return pr


class PackedHierPair:
    """One pair, points stored experiment by experiment (kernel order of the hierarchical sampler)."""

    def __init__(self, experiments, shapes, scales, locs):
        self.n_expts = len(experiments)
        self.dim = 5 + 2 * self.n_expts
        sizes = [len(e) for e in experiments]
        self.expt_start = np.ascontiguousarray(np.concatenate([[0], np.cumsum(sizes)]), dtype=np.int32)
        conc = np.concatenate([np.asarray(e)[:, 0] for e in experiments]).astype(float)
        with np.errstate(divide="ignore"):
            self.ln_conc = np.ascontiguousarray(np.log(conc))
        self.response = np.ascontiguousarray(np.concatenate([np.asarray(e)[:, 1] for e in experiments]).astype(float))
        self.pb = HierProblem(self.n_expts, _p(self.expt_start), _p(self.ln_conc), _p(self.response), hier_prior(shapes, scales, locs))

    def log_target(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.size == self.dim
        return lib().phfo_hier_log_target(C.byref(self.pb), _p(th))

    def init_state(self, theta0, cov_scale=0.01):
        st = np.zeros(2 * self.dim + self.dim * (self.dim + 1) // 2 + 3)
        th = np.ascontiguousarray(theta0, dtype=np.float64)
        lib().phfo_hier_init_state(C.byref(self.pb), C.c_double(cov_scale), _p(th), _p(st))
        return st

    def advance(self, st, t_begin, t_end, thinning, adapt_start, gamma, seed=25, chain_id=0, problem_id=0,
                star_replay=None, u_replay=None):
        rows = t_end // thinning - t_begin // thinning
        out = np.zeros((rows, self.dim + 1))
        run = Run(t_begin, t_end, thinning, 0, adapt_start, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, chain_id,
                  problem_id, _p(gamma))
        sr = np.ascontiguousarray(star_replay, dtype=np.float64) if star_replay is not None else None
        ur = np.ascontiguousarray(u_replay, dtype=np.float64) if u_replay is not None else None
        lib().phfo_hier_advance(C.byref(self.pb), C.byref(run), _p(st), _p(out), _p(sr) if sr is not None else None,
                                _p(ur) if ur is not None else None)
        return out


def hier_state_covariance(st, dim):
    """the adapted covariance a hierarchical chain's state carries: cov = L diag(d) L' with the packed lower triangle of the state
    holding the unit lower L and, in its diagonal slots, d (pyhillfit_amd/csrc/phf_hierarchical.hip: PHF_LDL_COLUMN)"""
    tri = np.zeros((dim, dim))
    tri[np.tril_indices(dim)] = np.asarray(st)[2 * dim + 1:2 * dim + 1 + dim * (dim + 1) // 2]
    dvec = np.diag(tri).copy()
    np.fill_diagonal(tri, 1.0)
    return (tri * dvec[None, :]) @ tri.T


def predictive_accumulate(rows, chains_used, hill_x, pic50_x, chunk, sums=None):
    """twin of phf_predictive_accumulate: rows [num_rows][Q][row_stride][C] -> sums [Q][4][G] (added to `sums`)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    nr, Q, rs, Cn = rows.shape
    hill_x = np.ascontiguousarray(hill_x, dtype=np.float64); pic50_x = np.ascontiguousarray(pic50_x, dtype=np.float64)
    G = len(hill_x)
    out = np.zeros((Q, 4, G)) if sums is None else np.ascontiguousarray(sums, dtype=np.float64).copy()
    L = lib()
    L.phfo_predictive_accumulate.restype = None
    L.phfo_predictive_accumulate.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p]
    L.phfo_predictive_accumulate(Q, _p(rows), nr, rs, Cn, int(chains_used), G, _p(hill_x), _p(pic50_x), int(chunk), _p(out))
    return out
