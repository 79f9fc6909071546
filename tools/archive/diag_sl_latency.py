"""Diagnostic: the single-level kernel in the command-line regime (210 pairs x 64 chains: one wavefront per pair, latency-bound) and
at the C2 shape, ms per 20 000 / 2 000 iterations."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler
dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
names = [(d, c) for d in dr.drugs for c in dr.channels]
for label, nm, C, I, mom in (("cli  210 x 64  ", names, 64, 20000, True), ("c2   1 x 65536 ", [("Amiodarone", "hERG")], 65536, 2000, False)):
    packed = dr.pack_single_level(nm)
    s = SingleLevelSampler(packed, 2, list(range(len(nm))), [1.0] * len(nm), C, thinning=5, seed=25, device="cuda:0")
    s.init([6.0, 0.8, 8.0], cov_identity=False, cov_scale=0.05)
    if mom:
        s.enable_moments(0)
    s.advance(4000, save=False)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.advance(I, save=not mom)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%s %d iterations: %.3f ms (min of 5; median %.3f)  = %.3f us per iteration" % (label, I, min(ts) * 1e3, np.median(ts) * 1e3, min(ts) / I * 1e6), flush=True)
