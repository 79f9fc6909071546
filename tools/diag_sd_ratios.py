"""Diagnostic: posterior WIDTHS of all 210 pairs (256 chains, the reference's 200 000-iteration protocol) against the reference sampler's
(G5c single chains; G5d pooled seeds where they exist): which (pair, column) entries lie outside [0.8, 1.25]?"""
import os, sys, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from conftest import reference_posteriors
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler
dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
for model in (2, 1):
    names, want, se, want_sd, reseeded = reference_posteriors(model)
    packed = dr.pack_single_level(names)
    s = SingleLevelSampler(packed, model, list(range(210)), [1.0] * 210, 256, thinning=5, seed=5, reset_mean_at_adapt_start=True, device="cuda:0")
    s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)
    s.enable_moments(after_iteration=50000)
    s.advance(200000, save=False)
    mean, var, n = s.posterior_moments()
    pooled_sd = torch.sqrt(var.mean(dim=2) + mean.var(dim=2)).cpu().numpy()
    r = pooled_sd[:s.d] / want_sd[:s.d]
    out = [(names[q], int(k), round(float(r[k, q]), 3), round(float(pooled_sd[k, q]), 3), round(float(want_sd[k, q]), 3), names[q] in reseeded)
           for k, q in zip(*np.nonzero((r < 0.8) | (r > 1.25)))]
    print("model %d: %d of %d entries outside [0.8, 1.25]:" % (model, len(out), r.size))
    for o in sorted(out, key=lambda o: o[2]):
        print("   ", o)
