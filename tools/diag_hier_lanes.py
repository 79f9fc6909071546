"""Diagnostic: hierarchical kernels with one lane per chain against two lanes per chain (PHF_HIER_LANES), per Ne group alone,
in the throughput regime (1024 chains per pair: the C4 shape) and in the command-line regime (64 chains per pair)."""
import os, sys, time
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
only = [int(x) for x in os.environ.get("PHF_DIAG_NE", "3,4,5,6").split(",")]
shapes = [int(x) for x in os.environ.get("PHF_DIAG_SHAPES", "1024,64").split(",")]
for C, I in [(c_, 500 if c_ >= 512 else 5000) for c_ in shapes]:
    for ne, exs in sorted(groups.items()):
        if ne not in only:
            continue
        s = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(len(exs))), C, thinning=5, seed=1, device="cuda:0")
        s.init(np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs]), cov_scale=0.01)
        s.advance(2000, save=False)
        out = []
        # PHF_DIAG_VARIANTS: lanes:wps pairs beyond the two defaults, e.g. "1:2" = the one-lane kernel built for two wavefronts per SIMD
        variants = [(1, 0), (2, 0)] + [tuple(int(x) for x in v.split(":")) for v in os.environ.get("PHF_DIAG_VARIANTS", "").split(",") if v]
        for lanes, wps in variants:
            H.set_kernel_policy(lanes=lanes, wps=wps)
            s.advance(I, save=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                s.advance(I, save=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            out.append(dt)
        H.set_kernel_policy(0, 0)
        extra = "".join("  | lanes %d wps %d: %.2f ms" % (l, w, o * 1e3) for (l, w), o in zip(variants[2:], out[2:]))
        print("chains/pair %4d  Ne=%d pairs=%3d  %d iterations: one lane %.2f ms, two lanes %.2f ms  (x%.2f)  | %.2f / %.2f us per iteration%s"
              % (C, ne, s.Q, I, out[0] * 1e3, out[1] * 1e3, out[0] / out[1], out[0] / I * 1e6, out[1] / I * 1e6, extra), flush=True)
