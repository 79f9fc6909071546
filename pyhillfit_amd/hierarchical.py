"""Host side of the hierarchical sampler (python/PyHillFit.py --hierarchical: run_hierarchical :213-642).

Mirrors the reference's set-up — per-experiment least-squares fits (:243-257), initial (alpha, beta) and (mu, s)
from those fits (:310-336), Gamma hyper-priors from the Elkins constants (:340-364), first covariance
diag(0.01*|theta0|) (:431), adaptation after 100*dim iterations (:440) — and hands the loop (:484-511) to the HIP
kernels.  Pairs are grouped by their number of experiments Ne (one kernel instantiation per Ne)."""
import ctypes as C
import json
import os
import time

import numpy as np
import torch

from . import _lib, bestfit, chainio
from . import doseresponse as dr
from .sampler import _ptr, _stream_ptr, gamma_table, raise_if_drained

# ---- Gamma hyper-priors: python/PyHillFit.py:301,340-364 (numbers from Elkins et al., as in the reference) ----------
ELKINS_HILL_ALPHAS = np.array([1.188, 1.744, 1.530, 0.930, 0.605, 1.325, 1.179, 0.979, 1.790, 1.708, 1.586, 1.469,
                               1.429, 1.127, 1.011, 1.318, 1.063])
ELKINS_HILL_BETAS = 1. / np.array([0.0835, 0.1983, 0.2089, 0.1529, 0.1206, 0.2386, 0.2213, 0.2263, 0.1784, 0.1544,
                                   0.2486, 0.2031, 0.2025, 0.1510, 0.1837, 0.1677, 0.0862])
ELKINS_PIC50_MUS = np.array([5.235, 5.765, 6.060, 5.315, 5.571, 7.378, 7.248, 5.249, 6.408, 5.625, 7.321, 6.852, 6.169,
                             6.217, 5.927, 7.414, 4.860])
ELKINS_PIC50_SIGMAS = np.array([0.0760, 0.1388, 0.1459, 0.2044, 0.1597, 0.2216, 0.1856, 0.1560, 0.1034, 0.1033, 0.1914,
                                0.1498, 0.1464, 0.1053, 0.1342, 0.1808, 0.0860])
MAX_EXPTS = 64    # PHF_HIER_MAX_EXPTS (Ne <= 8: kernels compiled per Ne; above: generic kernel, state in HBM)


def prior_params():
    """(shapes, scales, locs) for (alpha, beta, mu, s, sigma): PyHillFit.py:301,349-364."""
    locs = np.array([0., 2., -4, 0.01, dr.sigma_loc])
    modes = np.array([np.mean(ELKINS_HILL_ALPHAS), np.mean(ELKINS_HILL_BETAS) - 2., np.mean(ELKINS_PIC50_MUS),
                      np.mean(ELKINS_PIC50_SIGMAS), dr.sigma_mode])
    shapes = np.array([5., 2.5, 7.5, 2.5, dr.sigma_shape])
    scales = (modes - locs) / (shapes - 1.)
    return shapes, scales, locs


class HierPrior(C.Structure):
    _fields_ = [("shape_m1", C.c_double * 5), ("inv_scale", C.c_double * 5), ("loc", C.c_double * 5)]


class HierPoints(C.Structure):
    _fields_ = [("num_pairs", C.c_int32), ("stride", C.c_int32), ("n_expts", C.c_int32), ("points_per_expt", C.c_int32),
                ("ln_conc", C.c_void_p), ("response", C.c_void_p), ("expt_start", C.c_void_p)]


def make_prior(shapes=None, scales=None, locs=None):
    if shapes is None:
        shapes, scales, locs = prior_params()
    pr = HierPrior()
    for k in range(5):
        pr.shape_m1[k] = shapes[k] - 1.0; pr.inv_scale[k] = 1.0 / scales[k]; pr.loc[k] = locs[k]
    return pr


def shape_code(sizes):
    """PHF_HIER_SHAPE(per, last) of a pair's point shape (pyhillfit_amd/csrc/phf_hier_model.h; phf_hier_points.points_per_expt): `per` points in
    every experiment but the last, `last` there (coded only if it differs) — or, where the shape is not of that form, bit 30 + a nibble per
    experiment (up to 7 experiments of 1..15 points); 0 for a shape neither expresses"""
    sizes = [int(n) for n in sizes]
    per, last = sizes[0], sizes[-1]
    if len(sizes) >= 2 and all(n == per for n in sizes[:-1]) and 0 < per < 16 and 0 < last < 16:
        return per if last == per else per | (last << 4)
    if 2 <= len(sizes) <= 7 and all(0 < n < 16 for n in sizes):      # the list form: bit 30 + a nibble per experiment
        return (1 << 30) | sum(n << (4 * i) for i, n in enumerate(sizes))
    return 0


# (experiments, shape code) of the point shapes the hand-allocated gfx950 code object has a kernel for (tools/gen_hier_isa_main.py:
# HIER_KERNELS — the library looks the launch's shape up in its own table and runs the hipcc kernels for any other)
ISA_SHAPES = {(3, shape_code((4, 4, 4))), (3, shape_code((2, 2, 2))), (3, shape_code((5, 5, 4))),
              (4, shape_code((4, 4, 4, 1))), (4, shape_code((4, 4, 4, 2))), (4, shape_code((4, 4, 4, 3))),
              (4, shape_code((2, 2, 2, 1))), (4, shape_code((5, 5, 5, 1))),
              (5, shape_code((4, 4, 4, 1, 1))), (5, shape_code((4, 4, 4, 2, 1))), (5, shape_code((4, 4, 4, 4, 4))), (5, shape_code((5, 5, 4, 2, 2))),
              (6, shape_code((4, 4, 4, 1, 1, 1))), (6, shape_code((4, 4, 4, 4, 2, 1)))}


# ... and the shapes whose pairs get a launch group of their own.  The Ne = 4 kernels are measured and NOT grouped by default: alone they
# beat the hipcc kernel by 11 % (15.3 against 17.2 ms per 2 000 iterations of 32 pairs x 1 024 chains), beside the other groups of a Crumb
# run their 256-thread workgroups pack worse than the hipcc kernel's lone wavefronts (profiles/r05/c4_ne4_assembly_vs_hipcc.txt);
# PHF_HIER_ISA_NE4=1 groups them.
GROUPED_SHAPES = {k for k in ISA_SHAPES if k[0] == 3 or os.environ.get("PHF_HIER_ISA_NE4") == "1"}


def group_key(experiments, shapes=None):
    """Which launch group a pair belongs to: pairs of one group share a kernel instantiation.  (Ne, 0) in general; (Ne, shape code) for the
    pairs whose point shape the gfx950 assembly build has a kernel for (three experiments of 4 + 4 + 4 points: 147 of the Crumb set's 210
    pairs; 2 + 2 + 2: 6; 5 + 5 + 4: 1) — a group of its own each, so that the launch can state its shape (phf_hier_points.points_per_expt)."""
    key = (len(experiments), shape_code([len(x) for x in experiments]))
    return key if key in (GROUPED_SHAPES if shapes is None else shapes) else (key[0], 0)


class PackedHierPoints(object):
    """numpy image of `phf_hier_points`: pairs that all have Ne experiments, points experiment by experiment."""

    def __init__(self, experiments_per_pair):
        ne = {len(e) for e in experiments_per_pair}
        if len(ne) != 1:
            raise ValueError("all pairs of one PackedHierPoints must have the same number of experiments")
        self.n_expts = ne.pop()
        self.num_pairs = len(experiments_per_pair)
        self.stride = max(sum(len(x) for x in e) for e in experiments_per_pair)
        self.ln_conc = np.zeros((self.num_pairs, self.stride))
        self.response = np.zeros((self.num_pairs, self.stride))
        self.expt_start = np.zeros((self.num_pairs, self.n_expts + 1), dtype=np.int32)
        for p, expts in enumerate(experiments_per_pair):
            conc = np.concatenate([np.asarray(x)[:, 0] for x in expts]).astype(float)
            y = np.concatenate([np.asarray(x)[:, 1] for x in expts]).astype(float)
            with np.errstate(divide="ignore"):
                self.ln_conc[p, :len(conc)] = np.log(conc)
            self.response[p, :len(y)] = y
            self.expt_start[p] = np.concatenate([[0], np.cumsum([len(x) for x in expts])])
        # phf_hier_points.points_per_expt (ABI 6): the shape code of EVERY pair (n if every experiment has n points), 0 if the pairs differ
        codes = {shape_code([len(x) for x in e]) for e in experiments_per_pair}
        self.points_per_expt = codes.pop() if len(codes) == 1 else 0


class DeviceHierPoints(object):
    def __init__(self, packed, device):
        self.packed = packed
        self.device = torch.device(device)
        self.ln_conc = torch.from_numpy(packed.ln_conc).to(self.device)
        self.response = torch.from_numpy(packed.response).to(self.device)
        self.expt_start = torch.from_numpy(packed.expt_start).to(self.device)
        self.struct = HierPoints(packed.num_pairs, packed.stride, packed.n_expts, packed.points_per_expt, self.ln_conc.data_ptr(),
                                 self.response.data_ptr(), self.expt_start.data_ptr())


def _bind(lib):
    if getattr(lib, "_phf_hier_bound", False):
        return
    vp, i64, f64 = C.c_void_p, C.c_int64, C.c_double
    lib.phf_hierarchical_state_size.argtypes = [C.c_int]
    lib.phf_hierarchical_init.argtypes = [C.POINTER(HierPoints), C.POINTER(_lib.Problems), C.POINTER(HierPrior), f64, vp, vp, vp, vp]
    lib.phf_hierarchical_advance.argtypes = [C.POINTER(HierPoints), C.POINTER(_lib.Problems), C.POINTER(HierPrior),
                                             C.POINTER(_lib.MhConfig), i64, i64, vp, vp, vp, i64, vp]
    lib.phf_hierarchical_advance_queued.argtypes = [C.POINTER(HierPoints), C.POINTER(_lib.Problems), C.POINTER(HierPrior),
                                                    C.POINTER(_lib.MhConfig), i64, i64, vp, vp, vp, i64, C.c_int32, vp, vp]
    lib.phf_hierarchical_log_target.argtypes = [C.POINTER(HierPoints), C.POINTER(HierPrior), i64, vp, vp, vp, vp]
    lib.phf_hierarchical_queue_words.argtypes = [C.POINTER(HierPoints), C.POINTER(_lib.Problems)]
    lib.phf_hierarchical_queue_words.restype = i64
    lib._phf_hier_bound = True


def set_kernel_policy(lanes=0, wps=0):
    """PROCESS-WIDE: which kernel runs Ne = 3..6 groups (phf_hierarchical_set_kernel_policy): lanes per chain 1 | 2, register build of
    the two-lane kernel 1 | 2 (wavefronts per SIMD), 0 = not forced (a sampler's own kernel_hint, else the launch size, decides).  Every
    choice gives the same numbers bit for bit; used by the bit-identity tests and for A/B timing.  (PHF_HIER_LANES / PHF_HIER_WPS in the
    environment set the process's initial values; they are read by the library, once.)"""
    lib = _lib.load(); _bind(lib)
    _lib.check(lib.phf_hierarchical_set_kernel_policy(int(lanes), int(wps)), "phf_hierarchical_set_kernel_policy")


def last_kernel():
    """which kernel this thread's last advance launched (phf_hierarchical_last_kernel): 1 one lane per chain (hipcc), 2 two lanes,
    3 one wavefront per chain, 4 the hand-allocated gfx950 build of the Ne = 3 iteration"""
    return int(_lib.load().phf_hierarchical_last_kernel())


def simd_count():
    """SIMDs of the current device (phf_simd_count)."""
    return int(_lib.load().phf_simd_count())


def hint_side_by_side(samplers):
    """Called by whoever runs SEVERAL samplers side by side (one per Ne group, one stream each): the library decides one launch
    at a time — two lanes per chain when that group's wavefronts would leave SIMDs idle —, but when the groups TOGETHER give every
    SIMD a one-lane wavefront, a two-lane wavefront (half the latency, 1.7x the SIMD time) only takes slots from the others:
    C4 15.15 -> 14.5 ms with one lane everywhere.  The choice travels with each sampler's launches (phf_problems.kernel_hint) — no
    process-wide state is touched, later samplers of the process decide for themselves again — and a process-wide
    policy (set_kernel_policy, PHF_HIER_LANES) keeps the last word inside the library."""
    samplers = list(samplers)
    chains = sum(s.Q * s.C for s in samplers)
    lanes = 1 if -(-int(chains) // 64) >= simd_count() else 0
    for s in samplers:
        s.set_kernel_hint(lanes=lanes)
    return lanes


def log_target_batch(packed, pair_index, theta, prior=None, device="cuda"):
    """log_target_distribution (PyHillFit.py:173-193) of M parameter vectors theta[M][dim] on the GPU."""
    lib = _lib.load(); _bind(lib)
    dev = torch.device(device)
    pts = packed if isinstance(packed, DeviceHierPoints) else DeviceHierPoints(packed, dev)
    prior = prior or make_prior()
    theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
    m, dim = theta.shape
    if dim != 5 + 2 * pts.packed.n_expts:
        raise ValueError("theta must have 5 + 2*Ne columns")
    th = torch.from_numpy(np.ascontiguousarray(theta.T)).to(dev)
    pi = torch.from_numpy(np.ascontiguousarray(pair_index, dtype=np.int32)).to(dev)
    out = torch.empty(m, dtype=torch.float64, device=dev)
    _lib.check(lib.phf_hierarchical_log_target(C.byref(pts.struct), C.byref(prior), m, _ptr(pi), _ptr(th), _ptr(out),
                                               _stream_ptr(dev)), "phf_hierarchical_log_target")
    return out.cpu().numpy()


class HierarchicalSampler(object):
    """Q pairs (all with Ne experiments) x C chains of the hierarchical adaptive-Metropolis sampler on one GPU."""

    def __init__(self, points, pair_index, chains_per_problem, thinning=5, seed=25, adapt_start=None, prior=None,
                 problem_ids=None, chain_id_base=0, chain_offsets=None, device="cuda"):
        self.lib = _lib.load(); _bind(self.lib)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PhfError("pyhillfit_amd samplers run on a HIP device only (got %s)" % self.device)
        self.points = points if isinstance(points, DeviceHierPoints) else DeviceHierPoints(points, self.device)
        self.n_expts = self.points.packed.n_expts
        self.d = 5 + 2 * self.n_expts
        self.Q, self.C = len(pair_index), int(chains_per_problem)
        self.thinning, self.seed = int(thinning), int(seed)
        self.adapt_start = 100 * self.d if adapt_start is None else int(adapt_start)      # PyHillFit.py:440
        self.prior = prior or make_prior()
        dev = self.device
        self.pair_index = torch.tensor(np.asarray(pair_index, dtype=np.int32), device=dev)
        self.temperature = torch.ones(self.Q, dtype=torch.float64, device=dev)
        ids = np.arange(self.Q) if problem_ids is None else np.asarray(problem_ids)
        self.problem_ids = torch.tensor(ids.astype(np.int64), device=dev).to(torch.int32)
        # pairs with more points first (the cost of an iteration grows with the points; include/pyhillfit_amd.h: launch_order)
        npts = self.points.packed.expt_start[np.asarray(pair_index, dtype=np.int64), -1]
        self.launch_order = torch.tensor(np.argsort(-npts, kind="stable").astype(np.int32), device=dev)
        # chain_offsets[q]: global number of problem q's chain 0 (on top of chain_id_base) — a shard made of (pair, 64-chain block) units
        self.chain_offsets = None if chain_offsets is None else torch.tensor(np.asarray(chain_offsets, dtype=np.int64), device=dev).to(torch.int32)
        self.prob = _lib.Problems(self.Q, self.C, self.pair_index.data_ptr(), self.temperature.data_ptr(),
                                  self.problem_ids.data_ptr(), int(chain_id_base) & 0xFFFFFFFF, 0, self.launch_order.data_ptr(),
                                  None if self.chain_offsets is None else self.chain_offsets.data_ptr())
        self.S = self.lib.phf_hierarchical_state_size(self.n_expts)
        if self.S < 0:
            raise _lib.PhfError(self.lib.phf_last_error().decode())
        self.state = torch.zeros((self.S, self.Q * self.C), dtype=torch.float64, device=dev)
        self.moments, self.moments_after, self.t, self.row0, self._gamma = None, 0, 0, None, None
        # work-queue workspace of phf_hierarchical_advance_queued (used by launches that run the gfx950 assembly build on more blocks
        # than the chip holds wavefronts; ignored by every other launch); quantum 0 = the library's choice
        # (ABI 7: the library says how many words — 2 + blocks, plus the device-memory scratch of the Ne = 4 assembly kernels — and
        # kernel_hint bit 6 states that the workspace has them; the sticky fault word is word 1 + blocks)
        # — allocated at the sampler's first launch of its own: as a member of a FusedSamplers it never needs one)
        self.nblocks = self.Q * (-(-self.C // 64))
        self._queue = None
        self.prob.kernel_hint |= 64
        self.quantum = 0

    @property
    def queue(self):
        if self._queue is None:
            words = int(self.lib.phf_hierarchical_queue_words(C.byref(self.points.struct), C.byref(self.prob)))
            if words < 0:
                raise _lib.PhfError(self.lib.phf_last_error().decode())
            self._queue = torch.zeros(words, dtype=torch.int32, device=self.device)
        return self._queue

    def set_kernel_hint(self, lanes=0, wps=0, isa=None):
        """which kernel THIS sampler's launches should get (phf_problems.kernel_hint: lanes per chain 1 | 2, register build of the two-lane
        kernel 1 | 2; 0 = the library decides from the launch size; isa=False: not the hand-allocated gfx950 build of the Ne = 3
        iteration).  Same numbers either way; a process-wide policy overrides lanes / wps."""
        if lanes not in (0, 1, 2) or wps not in (0, 1, 2):
            raise ValueError("kernel hint: lanes and wps must be 0 (automatic), 1 or 2")
        keep = (self.prob.kernel_hint & 16) if isa is None else (0 if isa else 16)      # isa=None: leave that bit as it is
        self.prob.kernel_hint = int(lanes) | (int(wps) << 2) | keep | (self.prob.kernel_hint & 64)

    def init(self, theta0, cov_scale=0.01):
        """theta0: [dim], [Q][dim] or [Q][C][dim]"""
        th = torch.as_tensor(np.asarray(theta0, dtype=np.float64), device=self.device)
        if th.dim() == 1:
            th = th.view(1, 1, self.d).expand(self.Q, self.C, self.d)
        elif th.dim() == 2:
            th = th.view(self.Q, 1, self.d).expand(self.Q, self.C, self.d)
        th = th.permute(2, 0, 1).reshape(self.d, self.Q * self.C).contiguous()
        self.row0 = torch.empty((self.Q, self.d + 1, self.C), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.phf_hierarchical_init(C.byref(self.points.struct), C.byref(self.prob), C.byref(self.prior),
                                                  float(cov_scale), _ptr(th), _ptr(self.state), _ptr(self.row0),
                                                  _stream_ptr(self.device)), "phf_hierarchical_init")
        self.t = 0
        return self.row0

    def enable_moments(self, after_iteration=0):
        self.moments = torch.zeros((2 * (self.d + 1), self.Q * self.C), dtype=torch.float64, device=self.device)
        self.moments_after = int(after_iteration)

    def _config(self, t_end):
        need = max(0, t_end - self.adapt_start)
        if self._gamma is None or self._gamma.numel() < need + 1:
            self._gamma = torch.from_numpy(gamma_table(max(need, 1))).to(self.device)
        return _lib.MhConfig(2, self.thinning, self.adapt_start, 0, 0, self.seed, self._gamma.data_ptr())

    def reserve(self, total_iterations):
        self._config(int(total_iterations))

    def rows_between(self, t_begin, t_end):
        return t_end // self.thinning - t_begin // self.thinning

    def advance(self, n_iterations, out=None, save=True):
        t_end = self.t + int(n_iterations)
        cfg = self._config(t_end)
        rows = None
        if save:
            shape = (self.rows_between(self.t, t_end), self.Q, self.d + 1, self.C)
            rows = torch.empty(shape, dtype=torch.float64, device=self.device) if out is None else out
            if tuple(rows.shape) != shape or not rows.is_contiguous():
                raise ValueError("out must be contiguous with shape %s" % (shape,))
        _lib.check(self.lib.phf_hierarchical_advance_queued(C.byref(self.points.struct), C.byref(self.prob), C.byref(self.prior),
                                                            C.byref(cfg), self.t, t_end, _ptr(self.state), _ptr(rows),
                                                            _ptr(self.moments), self.moments_after, int(self.quantum), _ptr(self.queue),
                                                            _stream_ptr(self.device)),
                   "phf_hierarchical_advance_queued")
        self.t = t_end
        return rows

    def check_queue(self):
        """The gfx950 assembly build's queued launches raise the workspace's sticky fault word (its last int32) when a wavefront gives
        up waiting for its block's previous quantum: the launch drains and leaves stale chains behind a PHF_OK.  Read wherever the host
        hands results on (acceptance, posterior_moments, state_dict, the end of a run) — PhfError instead of such results."""
        if self._queue is not None:
            raise_if_drained(self._queue[:2 + self.nblocks], "hierarchical")

    def acceptance(self):
        self.check_queue()
        return self.state[-1].view(self.Q, self.C) / max(self.t, 1)

    def state_dict(self):
        """the sampler's position (iteration count) and chain state, for a bit-identical continuation (load_state_dict)"""
        self.check_queue()
        return {"t": self.t, "state": self.state.clone(), "n_expts": self.n_expts, "chains": self.C, "problems": self.Q,
                "philox_rounds": int(self.lib.phf_philox_rounds()), "abi_version": int(self.lib.phf_version())}

    def load_state_dict(self, sd):
        have = (int(self.lib.phf_philox_rounds()), int(self.lib.phf_version()))
        got = (sd.get("philox_rounds"), sd.get("abi_version"))
        if got != have:
            raise _lib.PhfError("checkpoint written by generator / ABI %s, this library is %s: the chains would not continue bit-identically"
                                % (got, have))
        if (sd["n_expts"], sd["chains"], sd["problems"]) != (self.n_expts, self.C, self.Q):
            raise ValueError("checkpoint of another batch shape")
        self.state.copy_(sd["state"]); self.t = int(sd["t"])

    def posterior_moments(self):
        self.check_queue()
        n = self.t // self.thinning - self.moments_after // self.thinning
        k = self.d + 1
        s1 = self.moments[:k].view(k, self.Q, self.C); s2 = self.moments[k:].view(k, self.Q, self.C)
        mean = s1 / n
        return mean, (s2 - s1 * mean) / max(n - 1, 1), n


class HierGroup(C.Structure):
    """phf_hier_group (include/pyhillfit_amd.h, ABI 7)"""
    _fields_ = [("pts", C.POINTER(HierPoints)), ("prob", C.POINTER(_lib.Problems)), ("cfg", C.POINTER(_lib.MhConfig)),
                ("state", C.c_void_p), ("rows", C.c_void_p), ("moments", C.c_void_p)]


class FusedSamplers(object):
    """Several HierarchicalSampler's — the launch groups of one run, each with a (Ne, point shape) the gfx950 code object has a kernel for,
    no two alike — advanced by ONE launch (phf_hierarchical_advance_fused): one persistent grid pulls every group's blocks from one work
    queue.  Every chain's numbers are those of the samplers' own advance(), bit for bit; the samplers keep their state, moments and
    iteration counters, so fused and separate launches can follow each other."""

    def __init__(self, samplers):
        self.samplers = list(samplers)
        if not self.samplers:
            raise ValueError("no samplers")
        s0 = self.samplers[0]
        self.lib, self.device = s0.lib, s0.device
        _bind(self.lib)
        self.lib.phf_hierarchical_fused_queue_words.argtypes = [C.c_int32, C.POINTER(HierGroup)]
        self.lib.phf_hierarchical_fused_queue_words.restype = C.c_int64
        self.lib.phf_hierarchical_advance_fused.argtypes = [C.c_int32, C.POINTER(HierGroup), C.POINTER(HierPrior), C.c_int64, C.c_int64, C.c_int64,
                                                            C.c_int32, C.c_void_p, C.c_void_p]
        kinds = [(s.n_expts, s.points.packed.points_per_expt) for s in self.samplers]
        if len(set(kinds)) != len(kinds) or any(kd not in ISA_SHAPES for kd in kinds):
            raise ValueError("fused launch: every sampler needs a (Ne, point shape) of its own that the gfx950 code object has a kernel for; got %s" % (kinds,))
        if any(s.device != self.device or s.thinning != s0.thinning or s.seed != s0.seed for s in self.samplers):
            raise ValueError("fused launch: one device, one thinning, one seed")
        self.nblocks = sum(s.nblocks for s in self.samplers)
        words = int(self.lib.phf_hierarchical_fused_queue_words(len(self.samplers), self._groups([None] * len(self.samplers), [None] * len(self.samplers))[0]))
        if words < 0:
            raise _lib.PhfError(self.lib.phf_last_error().decode())
        self.queue = torch.zeros(words, dtype=torch.int32, device=self.device)
        self.quantum = 0

    def _groups(self, cfgs, rows):
        arr = (HierGroup * len(self.samplers))()
        for g, s, cfg, r in zip(arr, self.samplers, cfgs, rows):
            g.pts = C.pointer(s.points.struct); g.prob = C.pointer(s.prob)
            g.cfg = C.pointer(cfg) if cfg is not None else None
            g.state = s.state.data_ptr(); g.rows = None if r is None else r.data_ptr()
            g.moments = None if s.moments is None else s.moments.data_ptr()
        return arr, cfgs

    def advance(self, n_iterations, out=None, save=True):
        """out: one tensor per sampler (as HierarchicalSampler.advance's), or None"""
        t0 = self.samplers[0].t
        if any(s.t != t0 for s in self.samplers):
            raise ValueError("fused launch: the samplers stand at different iterations")
        if len({s.moments_after for s in self.samplers if s.moments is not None}) > 1:
            raise ValueError("fused launch: one moments_after")
        t_end = t0 + int(n_iterations)
        cfgs = [s._config(t_end) for s in self.samplers]
        rows = [None] * len(self.samplers)
        if save:
            for j, s in enumerate(self.samplers):
                shape = (s.rows_between(t0, t_end), s.Q, s.d + 1, s.C)
                rows[j] = torch.empty(shape, dtype=torch.float64, device=self.device) if out is None else out[j]
                if tuple(rows[j].shape) != shape or not rows[j].is_contiguous():
                    raise ValueError("out[%d] must be contiguous with shape %s" % (j, shape))
        arr, keep = self._groups(cfgs, rows)
        after = max([s.moments_after for s in self.samplers if s.moments is not None] or [0])
        rc = self.lib.phf_hierarchical_advance_fused(len(self.samplers), arr, C.byref(self.samplers[0].prior), t0, t_end, after,
                                                     int(self.quantum), _ptr(self.queue), _stream_ptr(self.device))
        if rc == _lib.PHF_ERR_UNSUPPORTED:
            # what the one grid does not take (a launch that starts between two saved rows, PHF_HIER_ISA=0 ...): the samplers' own launches,
            # one after the other on this stream — the same numbers
            return [s.advance(n_iterations, out=r, save=save) for s, r in zip(self.samplers, rows)]
        _lib.check(rc, "phf_hierarchical_advance_fused")
        for s in self.samplers:
            s.t = t_end
        return rows

    def check_queue(self):
        raise_if_drained(self.queue[:2 + self.nblocks], "fused hierarchical")


# ---- start point: bestfit.hierarchical_first_iteration (torch-free, so that it can run in the worker processes) --------
first_iteration = bestfit.hierarchical_first_iteration


def cdf_chains(args, num_chains):
    """how many chains of every pair feed the posterior-predictive curves: --cdf-chains, default all"""
    n = getattr(args, "cdf_chains", 0)
    return num_chains if n <= 0 else min(n, num_chains)


# ---- driver ------------------------------------------------------------------------------------------------------------
def run_hierarchical(pairs, args, device, rank=0, world=1):
    """All pairs of this rank — replaces python/PyHillFit.py:213-642 run per pair."""
    t_begin = time.time()
    shapes, scales, locs = prior_params()
    prior = make_prior(shapes, scales, locs)
    groups, loaded = {}, []
    all_pairs = [(a, b) for a in dr.drugs for b in dr.channels]
    for drug, channel in pairs:
        try:
            num_expts, experiment_numbers, experiments = dr.load_crumb_data(drug, channel)
        except Exception:
            print("Problem loading data for {} + {} --- skipping".format(drug, channel)); continue
        fitted_all = not (0 < args.num_expts < num_expts)
        if not fitted_all:                                            # :228-231
            num_expts = args.num_expts
            experiments = experiments[:num_expts]
        experiments = experiments[:num_expts] if len(experiments) > num_expts else experiments
        ne = len(experiments)
        if ne > MAX_EXPTS:
            print("{} + {}: {} experiments exceed the {} supported by the hierarchical kernels --- skipping".format(drug, channel, ne, MAX_EXPTS))
            continue
        loaded.append((drug, channel, experiments, all_pairs.index((drug, channel)), fitted_all))
    # one persistent grid for the groups the gfx950 code object has kernels for (FusedSamplers) — when the run's chains give every SIMD a
    # one-lane wavefront (below that the groups run the two-lane kernels, whose wall time is the latency of one wavefront-iteration)
    mode = getattr(args, "fused_launch", "auto")
    one_lane = -(-args.num_chains // 64) * len(loaded) >= simd_count()
    use_fused = mode == "on" or (mode == "auto" and one_lane)
    for item in loaded:
        groups.setdefault(group_key(item[2], ISA_SHAPES if use_fused else None), []).append(item)
    summaries = []
    total_iterations, thinning = args.iterations, args.thinning
    saved_iterations = total_iterations // thinning + 1                # :469
    burn = saved_iterations // 4                                       # :472
    rng = np.random.RandomState(args.seed)
    rng_pred = np.random.RandomState(1)                                # construct_hierarchical_cdfs.py:12-13
    writers = chainio.WriterPool(getattr(args, "write_workers", 0))    # one pool for the start-point fits and the file formatting
    chain_streams = chainio.StreamWriters(getattr(args, "write_workers", 0))   # the chain files, written segment by segment
    # one sampler and one HIP stream per Ne group: the groups are independent, their launches overlap on the GPU
    runs = []
    ordered = sorted(groups.items(), reverse=True)
    # all start points first, in one sweep over the worker pool (the file-writer processes start after it)
    # (the least-squares fits of every experiment of every pair as ONE batch in this process, the two small distribution fits
    # per pair over the pool)
    fits = bestfit.hierarchical_first_iteration_batch([m[2] for _, members in ordered for m in members], locs)
    for (ne, _), members in ordered:
        packed = PackedHierPoints([m[2] for m in members])
        theta0 = np.array(fits[:len(members)]); fits = fits[len(members):]
        Q, C, d = len(members), args.num_chains, 5 + 2 * ne
        s = HierarchicalSampler(packed, list(range(Q)), C, thinning=thinning, seed=args.seed, prior=prior,
                                problem_ids=[m[3] for m in members], device=device)
        s.init(theta0, cov_scale=0.01)                                 # :431
        s.enable_moments(after_iteration=max(burn * thinning - 1, 0))
        s.reserve(total_iterations)
        kept = chainio.host_buffer((saved_iterations, Q, d + 1))   # pinned: the per-segment copies are asynchronous
        kept[0] = s.row0[:, :, 0].cpu()
        files = []
        for q, m in enumerate(members):                               # chain files grow while the GPU samples (:423-426,514-515)
            paths = dr.hierarchical_output_dirs_and_chain_file(m[0], m[1], ne)
            files.append(paths)
            chain_streams.create(paths[5], chainio.HIERARCHICAL_HEADER, kept[0:1, q].numpy())
        seg = max(thinning, args.segment - args.segment % thinning)
        buf = torch.empty((seg // thinning, Q, d + 1, C), dtype=torch.float64, device=device)
        curves = None
        if getattr(args, "predictive_cdfs", False):                    # construct_hierarchical_cdfs.py fused into the run
            from .predictive import PredictiveCurves
            curves = PredictiveCurves(Q, device)
            if burn == 0:
                curves.accumulate(s.row0.unsqueeze(0).contiguous(), cdf_chains(args, C))
        runs.append(dict(ne=ne, members=members, theta0=theta0, s=s, kept=kept, buf=buf, seg=seg, r=1, curves=curves, files=files,
                         stream=torch.cuda.Stream(device=device)))
    hint_side_by_side(r["s"] for r in runs)
    fused_runs = [r for r in runs if use_fused and (r["s"].n_expts, r["s"].points.packed.points_per_expt) in ISA_SHAPES]
    fused = None
    if len(fused_runs) > 1:
        fused = FusedSamplers([r["s"] for r in fused_runs])
        for r in fused_runs:
            r["s"].set_kernel_hint(lanes=1)                                # (their own launches, if any, stay on the kernels the fused grid runs)
            r["stream"] = fused_runs[0]["stream"]
    else:
        fused_runs = []
    if fused is None and sum(1 for r in runs if (r["s"].n_expts, r["s"].points.packed.points_per_expt) in ISA_SHAPES) > 1:
        # Several assembly-capable groups, each a launch of its own on its own stream: NOT side by side.  A queued assembly launch (a persistent
        # grid whose blocks chain their quanta through a progress word) beside another assembly kernel has been seen to wait for a progress word
        # that its XCD never showed it (tools/diag_drain.py: a poller read 1 for thirty seconds while the block went on to 14; the launch then
        # drains and the sticky fault word raises) — the one fused grid is the way to run them together; without it the groups take the hipcc kernels
        for r in runs:
            r["s"].set_kernel_hint(lanes=r["s"].prob.kernel_hint & 3, isa=False)
    torch.cuda.synchronize(device)
    start = time.time()
    done = 0
    # The groups' streams never join between segments: segment k+1 of every group is queued behind its segment k (and behind the
    # asynchronous copy of segment k's chain-0 rows to pinned host memory) BEFORE the host waits for segment k's copies and hands
    # them to the file writers.  A join per segment made every group wait for the slowest one and left the chip with the ragged
    # tail of a launch (jobs of one whole segment per wavefront) once per segment instead of once per run.
    in_flight = []                                  # (event, run, first row, number of rows) of the segment whose copies are under way
    join_segments = os.environ.get("PHF_JOIN_SEGMENTS", "0") == "1"
    def hand_over(items):
        for ev, run, r0, nr in items:
            ev.synchronize()
            for q, paths in enumerate(run["files"]):
                chain_streams.append(paths[5], run["kept"][r0:r0 + nr, q].numpy())
    while done < total_iterations:
        k = min(runs[0]["seg"], total_iterations - done) if runs else total_iterations
        queued = []
        fused_rows = {}
        if fused is not None:                  # the fused groups' segment: one launch, on their common stream
            nrs = [r["s"].rows_between(r["s"].t, r["s"].t + k) for r in fused_runs]
            with torch.cuda.stream(fused_runs[0]["stream"]):
                out = fused.advance(k, out=[r["buf"][:nr] for r, nr in zip(fused_runs, nrs)])
            fused_rows = {id(r): (o, nr) for r, o, nr in zip(fused_runs, out, nrs)}
        for run in runs:                       # queue this segment of every group (asynchronous, one stream each) ...
            s = run["s"]
            nr = fused_rows[id(run)][1] if id(run) in fused_rows else s.rows_between(s.t, s.t + k)
            with torch.cuda.stream(run["stream"]):
                rows = fused_rows[id(run)][0] if id(run) in fused_rows else s.advance(k, out=run["buf"][:nr])
                first = max(0, burn - run["r"])                        # saved rows before `burn` are the burn-in (:84-86 of the CDF script)
                if run["curves"] is not None and first < nr:
                    run["curves"].accumulate(rows[first:], cdf_chains(args, args.num_chains))
                run["kept"][run["r"]:run["r"] + nr].copy_(rows[:, :, :, 0], non_blocking=True)   # chain 0 of each pair
                ev = torch.cuda.Event()
                ev.record(run["stream"])
            queued.append((ev, run, run["r"], nr))
            run["r"] += nr
        hand_over(in_flight)                   # ... and, while the GPU works on it, give the PREVIOUS segment's rows to the file writers
        in_flight = queued
        if join_segments:                      # diagnostic (PHF_JOIN_SEGMENTS=1): wait for this segment before queueing the next, as round 1 did
            hand_over(in_flight); in_flight = []
        done += k
    hand_over(in_flight)
    torch.cuda.synchronize(device)
    if fused is not None:
        fused.check_queue()
    elapsed = time.time() - start
    total_chains = sum(len(r_["members"]) for r_ in runs) * args.num_chains
    for run in runs:
        ne, members, theta0, s, kept = run["ne"], run["members"], run["theta0"], run["s"], run["kept"]
        Q, C = len(members), args.num_chains
        mean, var, _ = s.posterior_moments()
        mean, var = mean.cpu().numpy(), var.cpu().numpy()
        acc = s.acceptance().cpu().numpy()
        for q, (drug, channel, experiments, _, fitted_all) in enumerate(members):
            d_clean, c_clean, output_dir, chain_dir, figs_dir, chain_file = run["files"][q]
            chain0 = kept[:, q].numpy()
            writers.submit(chainio.save_alpha_mu_samples, dr.alpha_mu_downsampling(d_clean, c_clean),
                           chainio.pick_alpha_mu_rows(chain0, args.num_APs, burn, rng), d_clean, c_clean)   # :519-525
            if run["curves"] is not None:
                from .predictive import save_cdfs_and_samples
                save_cdfs_and_samples(writers, d_clean, c_clean, ne, run["curves"].result(q), args.num_APs, rng_pred, fitted_all)
            summ = {"drug": d_clean, "channel": c_clean, "num_expts": ne, "chains": C, "iterations": total_iterations,
                    "pooled_mean": mean[:, q].mean(axis=1).tolist(),
                    "pooled_sd": np.sqrt(var[:, q].mean(axis=1) + mean[:, q].var(axis=1)).tolist(),
                    "acceptance": float(acc[q].mean()), "first_iteration": theta0[q].tolist(),
                    "mh_samples_per_second": total_chains * total_iterations / elapsed}
            with open(chain_file[:-4] + "_summary.json", "w") as f:
                json.dump(summ, f, indent=1)
            summaries.append(summ)
    writers.close()
    chain_streams.close()
    print("timing [rank {}]: data + start points {:.1f} s, sampling {:.1f} s ({} chains x {} iterations), chain files {:.1f} s".format(
        rank, start - t_begin, elapsed, total_chains, total_iterations, time.time() - start - elapsed))
    return summaries
