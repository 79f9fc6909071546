"""Posterior-predictive CDFs of Hill and pIC50 from hierarchical samples (python/construct_hierarchical_cdfs.py).

The reference loops over the ~75 000 post-burn rows of ONE chain file per pair and calls scipy.stats fisk/logistic
cdf+pdf on a 501-point grid for each row (:32-58, ~30 s per pair).  Here the sums are accumulated on the GPU
(phf_predictive_accumulate) straight from the hierarchical sampler's row buffer — all chains of all pairs, segment by
segment while the sampler runs — or from chains read back from reference-format files (the drop-in script
pyhillfit_amd/construct_hierarchical_cdfs.py)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .sampler import _ptr, _stream_ptr

GRID_POINTS = 501                      # construct_hierarchical_cdfs.py:33
HILL_RANGE = (0., 4.)                  # :34-35
PIC50_RANGE = (-2., 12.)               # :36-37
DEFAULT_CHUNK = 4096
CURVES = ("hill_cdf", "pic50_cdf", "hill_pdf", "pic50_pdf")


def predictive_grids():
    """:38-39"""
    return np.linspace(HILL_RANGE[0], HILL_RANGE[1], GRID_POINTS), np.linspace(PIC50_RANGE[0], PIC50_RANGE[1], GRID_POINTS)


class PredictiveCurves(object):
    """Running sums [Q][4][G] in HBM for Q problems; accumulate() may be called once per segment of sampler rows."""

    def __init__(self, num_problems, device, hill_x=None, pic50_x=None, chunk=DEFAULT_CHUNK):
        self.lib = _lib.load()
        self.device = torch.device(device)
        gx, px = predictive_grids()
        self.hill_x_host = np.asarray(gx if hill_x is None else hill_x, dtype=np.float64)
        self.pic50_x_host = np.asarray(px if pic50_x is None else pic50_x, dtype=np.float64)
        if self.hill_x_host.shape != self.pic50_x_host.shape or self.hill_x_host.ndim != 1:
            raise ValueError("the two grids must be 1-d and of equal length")
        self.Q, self.G, self.chunk = int(num_problems), len(self.hill_x_host), int(chunk)
        self.hill_x = torch.from_numpy(self.hill_x_host).to(self.device)
        self.pic50_x = torch.from_numpy(self.pic50_x_host).to(self.device)
        self.sums = torch.zeros((self.Q, 4, self.G), dtype=torch.float64, device=self.device)
        self.count = 0
        self.scratch = None

    def accumulate(self, rows, chains_used=None):
        """rows: device tensor [num_rows][Q][row_stride][C], contiguous, columns 0..3 = alpha, beta, mu, s (the buffer
        HierarchicalSampler.advance returns, or a slice of it along the first axis).  Asynchronous on the current stream."""
        if rows.dim() != 4 or rows.shape[1] != self.Q or rows.dtype != torch.float64 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous float64 tensor [num_rows][%d][row_stride][chains]" % self.Q)
        if rows.device != self.device:
            raise ValueError("rows live on %s, the curves on %s" % (rows.device, self.device))
        nr, _, rs, Cn = rows.shape
        used = Cn if chains_used is None else int(chains_used)
        if nr == 0:
            return
        need = self.lib.phf_predictive_scratch_bytes(self.Q, nr * used, self.G, self.chunk)
        if self.scratch is None or self.scratch.numel() * 8 < need:
            self.scratch = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.phf_predictive_accumulate(self.Q, _ptr(rows), nr, rs, Cn, used, self.G, _ptr(self.hill_x),
                                                      _ptr(self.pic50_x), self.chunk, _ptr(self.sums), _ptr(self.scratch),
                                                      C.c_size_t(self.scratch.numel() * 8), _stream_ptr(self.device)),
                   "phf_predictive_accumulate")
        self.count += nr * used

    def means(self):
        """[Q][4][G] tensor: sums / number of samples (:54-57)"""
        if self.count == 0:
            raise ValueError("no samples accumulated")
        return self.sums / float(self.count)

    def result(self, q):
        """(hill_x, hill_cdf, pic50_x, pic50_cdf, hill_pdf, pic50_pdf) of problem q — the reference function's return (:58)"""
        m = self.means()[q].cpu().numpy()
        return self.hill_x_host, m[0], self.pic50_x_host, m[1], m[2], m[3]


def curves_from_chains(chains, device, chunk=DEFAULT_CHUNK):
    """chains: list of arrays [rows_i][>=4] (alpha, beta, mu, s first; burn-in already dropped, equal row counts batch
    into one launch).  Returns a list of result tuples, one per chain."""
    out = [None] * len(chains)
    by_len = {}
    for i, c in enumerate(chains):
        by_len.setdefault(len(c), []).append(i)
    for n, idx in by_len.items():
        rows = np.stack([np.asarray(chains[i], dtype=np.float64)[:, :4] for i in idx], axis=1)[:, :, :, None]   # [n][Q][4][1]
        pc = PredictiveCurves(len(idx), device, chunk=chunk)
        pc.accumulate(torch.from_numpy(np.ascontiguousarray(rows)).to(pc.device))
        for k, i in enumerate(idx):
            out[i] = pc.result(k)
    return out


def draw_predictive_samples(hill_x, hill_cdf, pic50_x, pic50_cdf, num_samples, rng):
    """:133-137 — inverse-CDF samples by linear interpolation; the Hill uniforms are drawn first, then the pIC50 ones"""
    hill_u = rng.rand(num_samples)
    pic50_u = rng.rand(num_samples)
    return np.interp(hill_u, hill_cdf, hill_x), np.interp(pic50_u, pic50_cdf, pic50_x)


def save_cdfs_and_samples(writers, drug, channel, num_expts, result, num_samples, rng, save_samples=True):
    """:130-131,139-149 — the two CDF files (x, cdf) and, when all experiments were fitted, the (Hill, pIC50) sample file
    that the action-potential step reads (chaste/TestCrumbPredictions.hpp:185-195).  drug/channel already cleaned."""
    from . import chainio
    from . import doseresponse as dr
    hill_x, hill_cdf, pic50_x, pic50_cdf, _, _ = result
    hill_file, pic50_file = dr.hierarchical_posterior_predictive_cdf_files(drug, channel, num_expts)
    writers.submit(chainio.save_table, hill_file, np.vstack((hill_x, hill_cdf)).T, None)
    writers.submit(chainio.save_table, pic50_file, np.vstack((pic50_x, pic50_cdf)).T, None)
    hill_s, pic50_s = draw_predictive_samples(hill_x, hill_cdf, pic50_x, pic50_cdf, num_samples, rng)
    if save_samples:
        header = '# {} samples of (Hill,pIC50) drawn from their posterior predictive distributions, as defined by MCMC samples\n'.format(num_samples)
        writers.submit(chainio.save_table, dr.hierarchical_hill_and_pic50_samples_for_AP_file(drug, channel),
                       np.vstack((hill_s, pic50_s)).T, header)
    return hill_s, pic50_s
