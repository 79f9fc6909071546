"""Chain files: the output contract of the sampling step (python/PyHillFit.py:861-867,423-426,514-525;
python/PyHillTemp.py:165-169).  Text via np.savetxt's default '%.18e', '#' comment headers, so the reference's
downstream readers (np.loadtxt with usecols, last column = log-target) work unchanged."""
import ctypes
import multiprocessing as mp
import os
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
from concurrent.futures.process import BrokenProcessPool

import numpy as np


class WriterPool(object):
    """np.savetxt at '%.18e' formats ~60 MB of text per second and core, which is the bottleneck once sampling takes
    seconds (7.5 MB of text per single-level chain, 10-14 MB per hierarchical one, 630 files for the
    three -a runs).  Files are therefore written — and the start-point fits run — by `workers` processes that are
    SPAWNED (fresh interpreters: the parent has initialised HIP, a forked copy of it must not exist) and never import
    torch; arrays travel pickled.  workers <= 0 works synchronously in the caller, and so does a pool whose processes
    cannot start (spawn re-imports the main script, which an interactive session does not have): every job is a pure
    function of its arguments, so it is simply redone here.  close() waits and re-raises the first failure."""

    def __init__(self, workers, processes=None):
        # with the native formatter (libphf_textio.so) a job spends its time inside one C call that has released the GIL:
        # threads of THIS process do it — no interpreter start-up, no pickling of the rows (0.5 GB per `-a` run) to a worker
        if workers > 0 and _textio() and not processes:
            self.pool = ThreadPoolExecutor(workers)
        else:
            self.pool = ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) if workers > 0 else None
        self.pending = []

    def _broken(self):
        print("chainio.WriterPool: worker processes unavailable, continuing in the main process")
        self.pool.shutdown(wait=False)
        self.pool = None

    def submit(self, fn, *args):
        if self.pool is not None:
            try:
                self.pending.append((self.pool.submit(fn, *args), fn, args))
                return
            except BrokenProcessPool:
                self._broken()
        fn(*args)

    def map(self, fn, arg_tuples):
        """[fn(*args) for args in arg_tuples], spread over the workers (the start-point fits: pure numpy/scipy functions)"""
        arg_tuples = list(arg_tuples)
        if self.pool is not None and len(arg_tuples) > 1:
            try:
                return [f.result() for f in [self.pool.submit(fn, *a) for a in arg_tuples]]
            except BrokenProcessPool:
                self._broken()
        return [fn(*a) for a in arg_tuples]

    def close(self):
        try:
            for k, (f, fn, args) in enumerate(self.pending):
                try:
                    f.result()
                except BrokenProcessPool:
                    if self.pool is not None:
                        self._broken()
                    fn(*args)
        finally:
            self.pending = []
            if self.pool is not None:
                self.pool.shutdown()
                self.pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class StreamWriters(object):
    """Chain files written WHILE the GPU samples: every file belongs to one of `workers` single-process executors (spawned,
    torch-free like WriterPool's), which executes its jobs in submission order — create the file with its header lines, then
    append the rows of each segment as the sampler delivers them.  The text is what one np.savetxt call would have
    produced.  workers <= 0, or processes that cannot start, write in the caller.

    A worker that dies may have written any part of its last job.  Its files are therefore not patched but REWRITTEN by the
    caller from the job history (create truncates the file, then every segment delivered so far, in order); the healthy
    lanes keep their own files and go on.  The history holds references to the caller's row arrays, not copies."""

    def __init__(self, workers, processes=None):
        ctx = mp.get_context("spawn")
        if _textio() and not processes:                         # native formatter: one thread per lane (see WriterPool)
            self.lanes = [ThreadPoolExecutor(1) for _ in range(max(workers, 0))]
        else:
            self.lanes = [ProcessPoolExecutor(1, mp_context=ctx) for _ in range(max(workers, 0))]
        self.dead = set()
        self.pending = []                                       # (future, lane)
        self.history = {}                                       # path -> [(fn, args), ...] in submission order

    def _lane_of(self, path):
        return hash(path) % len(self.lanes)

    def _submit(self, path, fn, *args):
        self.history.setdefault(path, []).append((fn, args))
        if self.lanes:
            lane = self._lane_of(path)
            if lane not in self.dead:
                try:
                    self.pending.append((self.lanes[lane].submit(fn, *args), lane))
                    return
                except BrokenProcessPool:
                    self._lane_broke(lane)                      # rewrites the lane's files from the history, this job included
                    return
        fn(*args)

    def _lane_broke(self, lane):
        if lane in self.dead:
            return
        print("chainio.StreamWriters: a writer process died, its files are rewritten in the main process")
        self.dead.add(lane)
        self.lanes[lane].shutdown(wait=False)
        self.pending = [(f, l) for f, l in self.pending if l != lane]
        for path, jobs in self.history.items():
            if self._lane_of(path) == lane:
                for fn, args in jobs:                           # the first job of a file is its create ('w': truncates)
                    fn(*args)

    def create(self, path, header_lines, first_rows=None):
        self.history.pop(path, None)
        self._submit(path, _create_file, path, header_lines, first_rows)

    def append(self, path, rows):
        self._submit(path, _append_rows, path, rows)

    def close(self):
        try:
            while self.pending:
                f, lane = self.pending.pop(0)
                try:
                    f.result()
                except BrokenProcessPool:
                    self._lane_broke(lane)
        finally:
            self.pending = []
            for k, lane in enumerate(self.lanes):
                if k not in self.dead:
                    lane.shutdown()
            self.lanes, self.history = [], {}


_TEXTIO = None


def _textio():
    """pyhillfit_amd/lib/libphf_textio.so (csrc/phf_textio.cpp, built by pyhillfit_amd.build): np.savetxt's '%.18e' text, byte
    for byte, 3-4x faster than Python's per-row formatting.  Not built (a source checkout without build()): numpy writes."""
    global _TEXTIO
    if _TEXTIO is None:
        try:
            lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libphf_textio.so"))
            lib.phf_savetxt.restype = ctypes.c_int
            lib.phf_savetxt.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64]
            _TEXTIO = lib
        except OSError:
            _TEXTIO = False
    return _TEXTIO


def write_text(path, append, header_lines, rows):
    """header lines (only when creating the file) + rows as np.savetxt(outfile, rows) writes them"""
    lib = _textio()
    arr = None if rows is None else np.asarray(rows)
    if lib and (arr is None or (arr.ndim == 2 and arr.dtype == np.float64)):
        header = b"" if append else "".join(header_lines).encode()
        n, k = (0, 0) if arr is None else arr.shape
        a = None if arr is None else np.ascontiguousarray(arr)
        rc = lib.phf_savetxt(path.encode(), 1 if append else 0, header, len(header), None if a is None else a.ctypes.data, n, k, k)
        if rc:
            raise OSError(rc, os.strerror(rc), path)
        return
    with open(path, 'a' if append else 'w') as outfile:
        if not append:
            for line in header_lines:
                outfile.write(line)
        if rows is not None:
            np.savetxt(outfile, rows)


def _create_file(path, header_lines, first_rows):
    write_text(path, False, header_lines, first_rows)


def _append_rows(path, rows):
    write_text(path, True, (), rows)


HIERARCHICAL_HEADER = ("# Hill ~ log-logistic(alpha,beta), pIC50 ~ logistic(mu,s)\n",
                       "# alpha, beta, mu, s, pic50_1, hill_1, pic50_2, hill_2, ..., pic50_Ne, hill_Ne, sigma, log-target\n")


def default_write_workers(world=1):
    """host cores of this rank's share, minus the one driving the GPU; at most 16"""
    return max(0, min(16, (os.cpu_count() or 1) // max(world, 1) - 1))


def load_chain(chain_file, usecols=None):
    """Reader side of the contract (compute_bayes_factors.py:14, construct_hierarchical_cdfs.py:79: np.loadtxt with
    usecols): rows of the reference-format text file, '#' headers skipped."""
    return np.loadtxt(chain_file, usecols=usecols, ndmin=2)


def load_all_chains(chain_file):
    """The binary side file written by --save-all-chains: [rows][columns][chains], or None when it does not exist."""
    path = chain_file[:-4] + "_all_chains.npy"
    return np.load(path, mmap_mode="r") if os.path.exists(path) else None


def drop_burn_in(chain, burn_in_fraction):
    """PyHillFit.py:861-864 — Python-2 integer division: burn = saved_iterations / burn_fraction."""
    return chain[chain.shape[0] // int(burn_in_fraction):]


def save_single_level_chain(chain_file, chain, drug, channel, model):
    """PyHillFit.py:865-867.  Columns: model 2 -> pIC50, Hill, sigma, log-target; model 1 -> pIC50, sigma, log-target
    (the reference's header text says "(Hill,pIC50,sigma,log-target)"; its code writes the order used here)."""
    _create_file(chain_file, (single_level_header(drug, channel, model),), chain)


def single_level_header(drug, channel, model):
    cols = "(pIC50,Hill,sigma,log-target)" if model == 2 else "(pIC50,sigma,log-target)"
    return '# Nonhierarchical MCMC output for {} + {}: {}\n'.format(drug, channel, cols)


def save_tempered_chain(chain_file, chain):
    """PyHillTemp.py:169 — no header."""
    write_text(chain_file, False, (), chain)


def save_hierarchical_chain(chain_file, chain):
    """PyHillFit.py:423-426,514-515 — two header lines, then the FULL chain (burn-in included).
    Parameter order as the code stores it: alpha, beta, mu, s, pIC50_1, Hill_1, ..., sigma, log-target."""
    _create_file(chain_file, HIERARCHICAL_HEADER, chain)


def pick_alpha_mu_rows(chain, num_samples, burn, rng):
    """PyHillFit.py:519-521 — num_samples random post-burn rows, columns (alpha, mu); drawn in the caller so that the
    RandomState advances there whoever writes the file."""
    indices = rng.randint(burn, chain.shape[0], num_samples)
    return chain[indices][:, [0, 2]]


def save_alpha_mu_samples(samples_file, rows, drug, channel):
    """PyHillFit.py:522-525."""
    write_text(samples_file, False, ('# {} (alpha,mu) samples from hierarchical MCMC for {} + {}\n'.format(len(rows), drug, channel),), rows)


def save_table(path, table, header_line):
    """np.savetxt with an optional literal first line (construct_hierarchical_cdfs.py:130-131,144-149)."""
    write_text(path, False, (header_line,) if header_line else (), table)


def save_best_fit_params(path, theta0, model):
    """PyHillFit.py:739-746 (read back by assemble_BFs.py:62-63 with np.loadtxt)."""
    with open(path, "w") as outfile:
        outfile.write("# least-squares best fit params (start point of the MCMC)\n")
        outfile.write("# pIC50, sigma, (Hill=1, not included)\n" if model == 1 else "# pIC50, Hill, sigma\n")
        np.savetxt(outfile, [theta0])


def host_buffer(shape):
    """float64 host tensor for the rows the command lines keep (chain 0 of every pair): pinned, so that the per-segment copies out of
    the GPU are asynchronous (the next segment is launched behind them at once); pageable if the host refuses that much pinned
    memory (0.7-2 GB at the reference's defaults) - the copies are then synchronous, the results the same."""
    import torch
    try:
        return torch.empty(shape, dtype=torch.float64, pin_memory=True)
    except RuntimeError:
        return torch.empty(shape, dtype=torch.float64)
