"""Diagnostic: hierarchical posteriors of all Crumb pairs vs the reference's stored (alpha, mu) samples."""
import json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import bestfit, doseresponse as dr, hierarchical as H

T = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
ref = json.load(open(os.path.join(REPO, "tests", "golden", "chaste_alpha_mu_stats.json")))
shapes, scales, locs = H.prior_params()
groups = {}
for d in dr.drugs:
    for c in dr.channels:
        ne, _, ex = dr.load_crumb_data(d, c)
        groups.setdefault(len(ex), []).append((d, c, ex))
rows = []
for ne, members in sorted(groups.items(), reverse=True):
    packed = H.PackedHierPoints([m[2] for m in members])
    theta0 = np.array([bestfit.hierarchical_first_iteration(m[2], locs) for m in members])
    s = H.HierarchicalSampler(packed, list(range(len(members))), C, thinning=5, seed=7, device="cuda:0")
    s.init(theta0, cov_scale=0.01)
    s.enable_moments(after_iteration=T // 4)
    s.advance(T, save=False)
    mean, var, n = s.posterior_moments()
    pooled = mean.mean(dim=2).cpu().numpy(); between = mean.std(dim=2).cpu().numpy(); within = var.mean(dim=2).sqrt().cpu().numpy()
    acc = s.acceptance().mean(dim=1).cpu().numpy()
    for q, (d, c, ex) in enumerate(members):
        w = ref["%s_%s" % (d.replace("/", "_"), c.replace("/", "_"))]
        za = abs(pooled[0, q] - w["alpha_mean"]) / (3 * w["alpha_sd"] / np.sqrt(w["n"]) + 0.005 * w["alpha_mean"])
        zm = abs(pooled[2, q] - w["mu_mean"]) / (3 * w["mu_sd"] / np.sqrt(w["n"]) + 0.005 * abs(w["mu_mean"]))
        rows.append((max(za, zm), d, c, ne, [len(e) for e in ex], pooled[0, q], between[0, q], within[0, q], w["alpha_mean"], w["alpha_sd"],
                     pooled[2, q], between[2, q], within[2, q], w["mu_mean"], w["mu_sd"], acc[q]))
rows.sort(reverse=True)
print("z  drug channel Ne pts | alpha: mine(between-chain sd, within sd) ref(sd) | mu: mine(...) ref(sd) | acc")
for r in rows[:24]:
    print("%5.1f %-14s %-12s %d %s | %.3f (%.3f, %.3f) %.3f (%.3f) | %.3f (%.3f, %.3f) %.3f (%.3f) | %.3f" % r)
print("fraction z<4:", np.mean([r[0] < 4 for r in rows]))
