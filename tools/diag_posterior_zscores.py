"""Diagnostic: ALL 210 pairs x both single-level models, 1 024 GPU chains each at the reference's protocol (200 000 iterations): the z-scores
(GPU pooled mean - reference mean) / sqrt(se_ref^2 + se_gpu^2) of every (pair, column) entry — a calibration summary of the parity
(the reference's s.e.: batch means of its single chain, G5d / G5e pooled seeds where they exist)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from conftest import reference_posteriors
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler
dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
lines = []
for model in (2, 1):
    names, want, se, want_sd, reseeded = reference_posteriors(model)
    packed = dr.pack_single_level(names)
    C = 1024
    s = SingleLevelSampler(packed, model, list(range(210)), [1.0] * 210, C, thinning=5, seed=123, reset_mean_at_adapt_start=True, device="cuda:0")
    s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)
    s.enable_moments(after_iteration=50000)
    s.advance(200000, save=False)
    mean, var, n = s.posterior_moments()
    pooled = mean.mean(dim=2).cpu().numpy(); se_gpu = (mean.std(dim=2) / np.sqrt(C)).cpu().numpy()
    z = (pooled - want) / np.sqrt(se ** 2 + se_gpu ** 2)
    rel = np.abs(pooled - want) / np.abs(want)
    lines.append("model %d: %d entries: z mean %.3f, sd %.2f, |z| > 2: %.3f, > 3: %.3f, > 4: %.4f, max %.2f (standard normal: 0, 1, 0.046, 0.003, 0.0001); "
                 "relative difference of the means: median %.2e, 99th percentile %.2e, max %.2e"
                 % (model, z.size, z.mean(), z.std(), np.mean(np.abs(z) > 2), np.mean(np.abs(z) > 3), np.mean(np.abs(z) > 4), np.abs(z).max(),
                    np.median(rel), np.quantile(rel, 0.99), rel.max()))
print("\n".join(lines))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "posterior_zscores.txt"), "w").write("\n".join(lines) + "\n")
