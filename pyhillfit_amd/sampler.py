"""Host driver of the HIP Metropolis-Hastings kernels (single-level / tempered models).

PyTorch is used for device memory and streams only; every number is produced by the kernels behind
include/pyhillfit_amd.h.  Mirrors the state machine of the reference loops
(python/PyHillFit.py:748-856, python/PyHillTemp.py:57-125) for Q problems x C chains at once."""
import ctypes as C

import numpy as np
import torch

from . import _lib

_GAMMA_CACHE = {}


def gamma_table(n):
    """gamma[s] = 1/(s+1)**0.6, s = 0..n, evaluated with Python float pow exactly like the reference
    (PyHillFit.py:842 `1/(s+1)**0.6`, PyHillTemp.py:117) — so the adaptation weights are the reference's."""
    n = int(n)
    have = _GAMMA_CACHE.get("t")
    if have is None or len(have) < n + 1:
        size = max(n + 1, 1024)
        have = np.array([1.0 / (s + 1.0) ** 0.6 for s in range(size)], dtype=np.float64)
        _GAMMA_CACHE["t"] = have
    return have[:n + 1]


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DevicePoints(object):
    """PackedPoints resident in HBM + the ctypes `phf_points` view of it."""

    def __init__(self, packed, device):
        self.packed = packed
        self.device = torch.device(device)
        self.ln_conc = torch.from_numpy(packed.ln_conc).to(self.device)
        self.response = torch.from_numpy(packed.response).to(self.device)
        self.weight = torch.from_numpy(packed.weight).to(self.device)
        self.counts = torch.from_numpy(packed.counts).to(self.device)
        self.pi_bit = torch.from_numpy(packed.pi_bit).to(self.device)
        self.extra = torch.from_numpy(packed.extra).to(self.device)
        self.struct = _lib.Points(packed.num_pairs, packed.stride, self.ln_conc.data_ptr(), self.response.data_ptr(),
                                  self.weight.data_ptr(), self.counts.data_ptr(), self.pi_bit.data_ptr(), self.extra.data_ptr())


def log_target_batch(packed, model, pair_index, temperature, theta, device="cuda"):
    """(log-likelihood, log-prior) of M parameter vectors on the GPU (phf_single_level_log_target).
    theta: [M][d] host array.  Returns two numpy arrays [M]."""
    lib = _lib.load()
    dev = torch.device(device)
    pts = packed if isinstance(packed, DevicePoints) else DevicePoints(packed, dev)
    theta = np.ascontiguousarray(np.asarray(theta, dtype=np.float64))
    m, d = theta.shape
    if d != (2 if model == 1 else 3):
        raise ValueError("theta must have %d columns for model %d" % (2 if model == 1 else 3, model))
    th = torch.from_numpy(np.ascontiguousarray(theta.T)).to(dev)              # [d][M]
    pi = torch.from_numpy(np.ascontiguousarray(pair_index, dtype=np.int32)).to(dev)
    tt = torch.from_numpy(np.ascontiguousarray(temperature, dtype=np.float64)).to(dev)
    lik = torch.empty(m, dtype=torch.float64, device=dev)
    pri = torch.empty(m, dtype=torch.float64, device=dev)
    _lib.check(lib.phf_single_level_log_target(C.byref(pts.struct), model, m, _ptr(pi), _ptr(tt), _ptr(th), _ptr(lik),
                                               _ptr(pri), _stream_ptr(dev)), "phf_single_level_log_target")
    return lik.cpu().numpy(), pri.cpu().numpy()


# A quantum of a queued launch is at most this many iterations, however long the advance() call: a wavefront that pulled quantum k + 1 of a
# block waits for the wavefront still running quantum k (at most about one quantum: ~10 ms of a C3-like launch at this cap), and
# its give-up limit is a fixed number of polls (a few seconds) that must stay far above any correct wait.
MAX_QUANTUM_ITERATIONS = 4000


def queue_quantum(n_iterations, queue_quanta, thinning):
    """iterations per quantum of a queued launch of n_iterations: about n / queue_quanta, a whole number of thinning periods, at least
    100 (the state goes through HBM between quanta) and at most MAX_QUANTUM_ITERATIONS (more quanta instead).  0: plain launch."""
    n, k = int(n_iterations), int(queue_quanta)
    if k <= 1 or n < 100 * k:
        return 0
    cap = MAX_QUANTUM_ITERATIONS - MAX_QUANTUM_ITERATIONS % thinning
    quantum = -(-n // k)
    quantum = max(100, -(-quantum // thinning) * thinning)
    return min(quantum, max(cap, thinning))


def raise_if_drained(queue, what="single-level"):
    """queue: the int32 workspace of phf_{single_level,hierarchical}_advance_queued ([2 + blocks], last word = sticky fault flag) or None"""
    if queue is not None and int(queue[-1].item()) != 0:
        raise _lib.PhfError("a queued " + what + " launch drained (a wavefront's wait for its block's previous quantum did not end): "
                            "chains, moments and state written since are stale — discard them")


class SingleLevelSampler(object):
    """Q problems x C chains of the adaptive-Metropolis sampler, advanced in lock-step on one GPU.

    problem q = (pair_index[q], temperature[q]); global ids (problem_ids, chain_id_base) select the Philox
    streams, so a shard of the batch draws the same numbers as the same chains in a bigger launch."""

    def __init__(self, points, model, pair_index, temperature, chains_per_problem, thinning=5, seed=25,
                 adapt_start=None, reset_mean_at_adapt_start=False, problem_ids=None, chain_id_base=0, device="cuda",
                 launch_order="cost", queue_quanta=4, chain_offsets=None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PhfError("pyhillfit_amd samplers run on a HIP device only (got %s)" % self.device)
        self.points = points if isinstance(points, DevicePoints) else DevicePoints(points, self.device)
        self.model = int(model)
        self.d = 2 if self.model == 1 else 3
        self.Q = len(pair_index)
        self.C = int(chains_per_problem)
        self.thinning = int(thinning)
        self.seed = int(seed)
        self.adapt_start = 1000 * self.d if adapt_start is None else int(adapt_start)   # PyHillFit.py:787
        self.reset_mean = bool(reset_mean_at_adapt_start)
        dev = self.device
        self.pair_index = torch.tensor(np.asarray(pair_index, dtype=np.int32), device=dev)
        self.temperature = torch.tensor(np.asarray(temperature, dtype=np.float64), device=dev)
        ids = np.arange(self.Q) if problem_ids is None else np.asarray(problem_ids)
        self.problem_ids = torch.tensor(ids.astype(np.int64), device=dev).to(torch.int32)   # bit pattern of uint32
        # The order in which the problems' wavefronts go to the GPU: most expensive first (longest-processing-time list
        # scheduling by the hardware dispatcher).  An iteration costs ~525 + 28 per uncensored + 115 per censored entry
        # instructions (tools/isa_stats.py), 550..1 500 over the Crumb pairs; in file order the launch ends with a ragged
        # tail — 46.4 ms against 43.4 for the 210 pairs x 4 096 chains.  Results do not depend on it.  None = as given.
        self.launch_order = None
        if isinstance(launch_order, str) and launch_order == "cost":
            cnt = self.points.packed.counts[np.asarray(pair_index, dtype=np.int64)]
            cost = 525.0 + 28.0 * cnt[:, 0] + 115.0 * (cnt[:, 1] + cnt[:, 2])
            self.launch_order = torch.tensor(np.argsort(-cost, kind="stable").astype(np.int32), device=dev)
        elif launch_order is not None:
            order = np.asarray(launch_order, dtype=np.int32)
            if sorted(order.tolist()) != list(range(self.Q)):
                raise ValueError("launch_order must be a permutation of the problems")
            self.launch_order = torch.tensor(order, device=dev)
        # chain_offsets[q]: global number of problem q's chain 0 (on top of chain_id_base) — a shard made of (pair, 64-chain block)
        # units (distributed.shard_blocks) lists each block as a problem of 64 chains and keeps every chain's Philox stream
        self.chain_offsets = None if chain_offsets is None else torch.tensor(np.asarray(chain_offsets, dtype=np.int64), device=dev).to(torch.int32)
        self.prob = _lib.Problems(self.Q, self.C, self.pair_index.data_ptr(), self.temperature.data_ptr(),
                                  self.problem_ids.data_ptr(), int(chain_id_base) & 0xFFFFFFFF, 0,
                                  self.launch_order.data_ptr() if self.launch_order is not None else None,
                                  None if self.chain_offsets is None else self.chain_offsets.data_ptr())
        # Launches of a few rounds of the chip run as a work queue of `queue_quanta` quanta per block (phf_single_level_advance_queued:
        # same results, shorter tail: the 210 pairs x 4 096 chains launch 44.2 ms plain, 42.4 with 4 quanta, 42.8 with 8, 44.2 with 16);
        # 0 = always the plain launch.  The workspace is one int per block + 1.
        self.queue_quanta = int(queue_quanta)
        self._queue = None
        self.nblocks = self.Q * ((self.C + 63) // 64)
        self.S = self.lib.phf_single_level_state_size(self.model)
        self.state = torch.zeros((self.S, self.Q * self.C), dtype=torch.float64, device=dev)
        self.moments = None
        self.moments_after = 0
        self.t = 0
        self.row0 = None
        self._gamma = None

    def set_kernel_hint(self, isa=False):
        """isa=True: this sampler's launches run the hand-allocated gfx950 build of the model-2 iteration where they are eligible
        (phf_problems.kernel_hint bit 5; opt-in: it is bit-identical but not faster than the hipcc kernel);
        isa=False: the hipcc kernel, even with PHF_SL_ISA=1 in the environment (bit 4).  For A/B timing and the bit-identity tests."""
        self.prob.kernel_hint = 32 if isa else 16

    def last_kernel(self):
        """which kernel this thread's last advance launched (phf_single_level_last_kernel): 1 hipcc, 2 gfx950 assembly, 3 the same queued"""
        return int(self.lib.phf_single_level_last_kernel())

    # -- start: PyHillFit.py:748-751,789,814 / PyHillTemp.py:63-80 --------------------------------------------
    def init(self, theta0, cov_identity=False, cov_scale=0.05):
        """theta0: [d] (every chain), [Q][d] (per problem) or [Q][C][d]."""
        th = torch.as_tensor(np.asarray(theta0, dtype=np.float64), device=self.device)
        if th.dim() == 1:
            th = th.view(1, 1, self.d).expand(self.Q, self.C, self.d)
        elif th.dim() == 2:
            th = th.view(self.Q, 1, self.d).expand(self.Q, self.C, self.d)
        th = th.permute(2, 0, 1).reshape(self.d, self.Q * self.C).contiguous()             # SoA [d][Q*C]
        self.row0 = torch.empty((self.Q, self.d + 1, self.C), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.phf_single_level_init(C.byref(self.points.struct), C.byref(self.prob), self.model,
                                                  int(bool(cov_identity)), float(cov_scale), _ptr(th), _ptr(self.state),
                                                  _ptr(self.row0), _stream_ptr(self.device)), "phf_single_level_init")
        self.t = 0
        return self.row0

    def enable_moments(self, after_iteration=0):
        """accumulate sum x, sum x^2 of the saved samples with t > after_iteration, per chain, on the device"""
        self.moments = torch.zeros((2 * (self.d + 1) + 1, self.Q * self.C), dtype=torch.float64, device=self.device)
        self.moments_after = int(after_iteration)

    def _config(self, t_end):
        need = max(0, t_end - self.adapt_start)
        if self._gamma is None or self._gamma.numel() < need + 1:
            self._gamma = torch.from_numpy(gamma_table(max(need, 1))).to(self.device)
        return _lib.MhConfig(self.model, self.thinning, self.adapt_start, int(self.reset_mean), 0, self.seed,
                             self._gamma.data_ptr())

    def reserve(self, total_iterations):
        """size the gamma table once for a whole run (keeps advance() free of allocations)"""
        self._config(int(total_iterations))

    def rows_between(self, t_begin, t_end):
        return t_end // self.thinning - t_begin // self.thinning

    def advance(self, n_iterations, out=None, save=True):
        """Run n_iterations more MH iterations on every chain.  Returns the saved rows [R][Q][d+1][C]
        (None if save=False).  `out` may be a preallocated tensor of that shape."""
        t_end = self.t + int(n_iterations)
        cfg = self._config(t_end)
        rows = None
        if save:
            r = self.rows_between(self.t, t_end)
            shape = (r, self.Q, self.d + 1, self.C)
            if out is None:
                rows = torch.empty(shape, dtype=torch.float64, device=self.device)
            else:
                if tuple(out.shape) != shape or not out.is_contiguous():
                    raise ValueError("out must be contiguous with shape %s" % (shape,))
                rows = out
        quantum = queue_quantum(n_iterations, self.queue_quanta, self.thinning)
        if quantum:
            # Whether the launch really runs as a queue is the library's decision (it knows the chip: phf_simd_count()): only launches
            # of 1..16 rounds of the chip's wavefront slots do, anything else falls back to the plain launch there.
            if self._queue is None:
                self._queue = torch.zeros(2 + self.nblocks, dtype=torch.int32, device=self.device)   # last word: sticky fault flag
            _lib.check(self.lib.phf_single_level_advance_queued(C.byref(self.points.struct), C.byref(self.prob), C.byref(cfg),
                                                                self.t, t_end, _ptr(self.state), _ptr(rows), _ptr(self.moments),
                                                                self.moments_after, quantum, _ptr(self._queue), _stream_ptr(self.device)),
                       "phf_single_level_advance_queued")
        else:
            _lib.check(self.lib.phf_single_level_advance(C.byref(self.points.struct), C.byref(self.prob), C.byref(cfg),
                                                         self.t, t_end, _ptr(self.state), _ptr(rows), _ptr(self.moments),
                                                         self.moments_after, _stream_ptr(self.device)),
                       "phf_single_level_advance")
        self.t = t_end
        return rows

    def check_queue(self):
        """A queued launch whose wavefronts gave up waiting for each other drains and leaves stale chains behind a PHF_OK (the
        launch is asynchronous).  Called wherever the host synchronises anyway — state_dict, posterior_moments,
        mean_log_likelihood_t1, acceptance, the end of run() — it reads the workspace's sticky fault word (a stream-ordered
        device-to-host copy, like phf_single_level_queue_status) and raises PhfError instead of handing such results on."""
        raise_if_drained(self._queue)

    def run(self, iterations, segment=None):
        """Whole chain like the reference keeps it: [iterations/thinning + 1][Q][d+1][C], row 0 = start point."""
        if iterations % self.thinning:
            raise ValueError("iterations must be a multiple of thinning (PyHillFit.py:805)")
        self.reserve(self.t + iterations)
        n_rows = iterations // self.thinning + 1
        chain = torch.empty((n_rows, self.Q, self.d + 1, self.C), dtype=torch.float64, device=self.device)
        chain[0] = self.row0
        seg = iterations if segment is None else int(segment)
        seg = max(self.thinning, seg - seg % self.thinning)
        done, r = 0, 1
        while done < iterations:
            k = min(seg, iterations - done)
            nr = self.rows_between(self.t, self.t + k)
            self.advance(k, out=chain[r:r + nr])
            done += k; r += nr
        self.check_queue()
        return chain

    # -- views of the state ---------------------------------------------------------------------------------
    def theta(self):
        return self.state[:self.d].view(self.d, self.Q, self.C)

    def log_target(self):
        return self.state[self.d].view(self.Q, self.C)

    def loga(self):
        return self.state[2 * self.d + 1 + self.d * (self.d + 1) // 2].view(self.Q, self.C)

    def acceptance(self):
        """running acceptance rate (PyHillFit.py:839) per chain"""
        self.check_queue()
        return self.state[2 * self.d + 2 + self.d * (self.d + 1) // 2].view(self.Q, self.C) / max(self.t, 1)

    def posterior_moments(self):
        """(mean, variance, n) per chain from the on-device accumulators: [d+1][Q][C]"""
        if self.moments is None:
            raise _lib.PhfError("enable_moments() was not called")
        self.check_queue()
        n = self.t // self.thinning - self.moments_after // self.thinning
        k = self.d + 1
        s1 = self.moments[:k].view(k, self.Q, self.C); s2 = self.moments[k:2 * k].view(k, self.Q, self.C)
        mean = s1 / n
        var = (s2 - s1 * mean) / max(n - 1, 1)
        return mean, var, n

    def mean_log_likelihood_t1(self):
        """[Q][C]: per chain, the mean over the saved post-burn samples of log_data_likelihood(theta, t=1) —
        compute_log_py_approxn of python/compute_bayes_factors.py:11-27, accumulated inside the sampler kernel"""
        if self.moments is None:
            raise _lib.PhfError("enable_moments() was not called")
        self.check_queue()
        n = self.t // self.thinning - self.moments_after // self.thinning
        return self.moments[2 * (self.d + 1)].view(self.Q, self.C) / n

    def state_dict(self):
        """checkpoint: everything needed to continue bit-identically"""
        self.check_queue()
        return {"state": self.state.clone(), "t": self.t, "moments": None if self.moments is None else self.moments.clone(),
                "moments_after": self.moments_after, "seed": self.seed,
                # what "continue bit-identically" depends on besides the state: the generator's rounds (7 since ABI 5, 10 before or with
                # -DPHF_PHILOX_ROUNDS=10) and the ABI the state's layout belongs to
                "philox_rounds": int(self.lib.phf_philox_rounds()), "abi_version": int(self.lib.phf_version())}

    def load_state_dict(self, sd):
        """refuses a checkpoint taken with another generator or ABI (it would continue on another random stream, silently);
        checkpoints from before these fields existed (rounds 1-4) are refused too: their generator is not known"""
        have = (int(self.lib.phf_philox_rounds()), int(self.lib.phf_version()))
        got = (sd.get("philox_rounds"), sd.get("abi_version"))
        if got != have:
            raise _lib.PhfError("checkpoint was taken with Philox rounds / ABI %s, this library has %s: the chains would not continue "
                                "bit-identically" % (got, have))
        self.state.copy_(sd["state"]); self.t = int(sd["t"]); self.seed = int(sd["seed"])
        if sd.get("moments") is not None:
            self.moments = sd["moments"].clone().to(self.device); self.moments_after = int(sd["moments_after"])


def debug_math(fn, x, device="cuda"):
    lib = _lib.load()
    dev = torch.device(device)
    xin = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)
    out = torch.empty_like(xin)
    _lib.check(lib.phf_debug_math(int(fn), xin.numel(), _ptr(xin), _ptr(out), _stream_ptr(dev)), "phf_debug_math")
    return out.cpu().numpy()


def debug_isa(fn, x, device="cuda"):
    """phf_debug_isa: the unit kernels of the hand-allocated gfx950 code object (include/pyhillfit_amd.h lists fn).  x: float64 array
    (fn 0..6), uint32 array (fn 7, 8) or uint32 [n][6] (fn 9: Philox4x32-7 with the key of element 0; returns uint32 [n][4])."""
    lib = _lib.load()
    dev = torch.device(device)
    if fn <= 6:
        xin = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)
        out = torch.empty_like(xin)
        n = xin.numel()
    elif fn <= 8:
        xin = torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint32).view(np.int32)).to(dev)
        out = torch.empty(xin.numel(), dtype=torch.float64, device=dev)
        n = xin.numel()
    else:
        ck = np.ascontiguousarray(x, dtype=np.uint32).reshape(-1, 6)
        xin = torch.from_numpy(ck.view(np.int32)).to(dev)
        out = torch.empty((ck.shape[0], 4), dtype=torch.int32, device=dev)
        n = ck.shape[0]
    _lib.check(lib.phf_debug_isa(int(fn), n, _ptr(xin), _ptr(out), _stream_ptr(dev)), "phf_debug_isa")
    res = out.cpu().numpy()
    return res.view(np.uint32) if fn == 9 else res


def debug_philox(counter_key, device="cuda", rounds=0):
    """Philox4x32-R blocks on the device; rounds 7 or 10, 0 = the samplers' own (phf_philox_rounds)"""
    lib = _lib.load()
    dev = torch.device(device)
    ck = np.ascontiguousarray(counter_key, dtype=np.uint32).reshape(-1, 6)
    cin = torch.from_numpy(ck.view(np.int32)).to(dev)
    out = torch.empty((ck.shape[0], 4), dtype=torch.int32, device=dev)
    if rounds:
        _lib.check(lib.phf_debug_philox_rounds(int(rounds), ck.shape[0], _ptr(cin), _ptr(out), _stream_ptr(dev)), "phf_debug_philox_rounds")
    else:
        _lib.check(lib.phf_debug_philox(ck.shape[0], _ptr(cin), _ptr(out), _stream_ptr(dev)), "phf_debug_philox")
    return out.cpu().numpy().view(np.uint32)
