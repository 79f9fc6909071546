"""Diagnostic: one pair's single-level posterior (model 2) on the GPU, chain by chain: where does a pooled width come from?"""
import os, sys, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler
drug, channel = sys.argv[1], sys.argv[2]
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
packed = dr.pack_single_level([(drug, channel)])
s = SingleLevelSampler(packed, 2, [0], [1.0], C, thinning=5, seed=5, reset_mean_at_adapt_start=True, device="cuda:0")
s.init(np.ones(3), cov_identity=True, cov_scale=1.0)
s.enable_moments(after_iteration=50000)
s.advance(200000, save=False)
mean, var, n = s.posterior_moments()
m = mean[:, 0].cpu().numpy(); v = var[:, 0].cpu().numpy()
for k, name in enumerate(["pIC50", "Hill", "sigma", "log-target"]):
    sd = np.sqrt(v[k])
    print("%-10s pooled mean %.4f sd %.4f | per-chain mean quantiles 1 50 99 99.9 %%: %s | per-chain sd quantiles 50 90 99 99.9 %% max: %s"
          % (name, m[k].mean(), np.sqrt(v[k].mean() + m[k].var()), np.round(np.quantile(m[k], [0.01, 0.5, 0.99, 0.999]), 4).tolist(),
             np.round(list(np.quantile(sd, [0.5, 0.9, 0.99, 0.999])) + [sd.max()], 4).tolist()))
k = 1
big = np.sqrt(v[k]) > 3 * np.median(np.sqrt(v[k]))
print("chains with a Hill sd above 3 x the median: %d of %d (%.2f %%); they carry %.1f %% of the pooled variance" % (big.sum(), C, 100.0 * big.mean(),
      100.0 * (v[k][big].sum() + ((m[k][big] - m[k].mean()) ** 2).sum()) / (v[k].sum() + ((m[k] - m[k].mean()) ** 2).sum())))
