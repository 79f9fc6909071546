"""Diagnostic: how often are the truncation tails of the hierarchical likelihood negligible for ALL 64 chains of a wavefront?
ln(Phi(b) - Phi(a)) with a = -pred/sigma, b = (100-pred)/sigma: a tail with argument/sqrt2 >= 6 contributes < 2^-54 to 1 - x.
Per (pair, point pair, saved row): A = lower tails of both points negligible on every chain, B = upper tails."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import bestfit, doseresponse as dr, hierarchical as H

dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
shapes, scales, locs = H.prior_params()
groups = {}
for d in dr.drugs:
    for c in dr.channels:
        ne, _, ex = dr.load_crumb_data(d, c)
        groups.setdefault(len(ex), []).append(ex)
C = 64
tot = np.zeros(4); cnt = 0
for ne, exs in sorted(groups.items()):
    s = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(len(exs))), C, thinning=100, seed=3, device="cuda:0")
    s.init(np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs]), cov_scale=0.01)
    s.advance(30000, save=False)
    rows = s.advance(5000, save=True).cpu().numpy()          # [R][Q][D+1][C]
    g = np.zeros(4); n = 0
    for q, ex in enumerate(exs):
        th = rows[:, q]                                       # [R][D+1][C]
        sigma = th[:, 4 + 2 * ne]
        for i, e in enumerate(ex):
            pic50, hill = th[:, 4 + 2 * i], th[:, 5 + 2 * i]
            conc = np.asarray(e)[:, 0]
            npts = len(conc); nf = 2 * ((npts + 2) // 4)
            for lo, hi in ((0, min(nf, npts)), (min(nf, npts), npts)):      # the two halves' shares
                j = lo
                while j + 2 <= hi:
                    ok = []
                    for jj in (j, j + 1):
                        x = (conc[jj] / 10 ** (6 - pic50)) ** hill
                        pred = 100 * (1 - 1 / (1 + x))
                        ok.append((pred / sigma / np.sqrt(2) >= 6, (100 - pred) / sigma / np.sqrt(2) >= 6))
                    A = (ok[0][0] & ok[1][0]).all(axis=1); B = (ok[0][1] & ok[1][1]).all(axis=1)
                    g += [A.mean(), B.mean(), (A & B).mean(), 1]; j += 2
    print("Ne=%d: point pairs %d: lower tails skippable %.3f, upper tails %.3f, both %.3f" % (ne, g[3], g[0] / g[3], g[1] / g[3], g[2] / g[3]), flush=True)
    tot += g * len(exs) ** 0  # per group equal weight of point pairs
print("all: lower %.3f upper %.3f both %.3f" % (tot[0] / tot[3], tot[1] / tot[3], tot[2] / tot[3]))
