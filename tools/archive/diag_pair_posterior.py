"""Diagnostic: pooled posterior mean of one pair from many GPU chains run exactly like the reference's PyHillTemp.do_mcmc
(start ones, identity covariance, mean reset, 200 000 iterations, first quarter of the saved rows dropped), per seed.
    python tools/diag_pair_posterior.py Amitriptyline Kir2.1 [model] [chains] [seeds...]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler

drug, channel = sys.argv[1], sys.argv[2]
model = int(sys.argv[3]) if len(sys.argv) > 3 else 2
C = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
seeds = [int(x) for x in sys.argv[5:]] or [5, 6, 7]
dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
packed = dr.pack_single_level([(drug, channel)])
for seed in seeds:
    s = SingleLevelSampler(packed, model, [0], [1.0], C, thinning=5, seed=seed, reset_mean_at_adapt_start=True, device="cuda:0")
    s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)
    s.enable_moments(after_iteration=50000)
    s.advance(200000, save=False)
    mean, var, n = s.posterior_moments()
    m = mean[:, 0, :]                                         # [d+1][C]
    print("seed %d: %d chains, pooled mean %s  s.e. %s  (per-chain sd of the mean %s)" % (
        seed, C, np.round(m.mean(dim=1).cpu().numpy(), 5), np.round((m.std(dim=1) / np.sqrt(C)).cpu().numpy(), 5),
        np.round(m.std(dim=1).cpu().numpy(), 4)), flush=True)
