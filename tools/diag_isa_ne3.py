"""Timing: the hand-allocated gfx950 build of the Ne = 3 iteration (phf_hier3_advance) against the hipcc one-lane kernel, on the
147 Crumb pairs with 3 x 4 points (and, for reference, on all 154 Ne = 3 pairs through the hipcc kernel), 1 024 chains per pair,
launches of 500 and 2 000 iterations.  Produced profiles/r05/c4_isa_ne3_timing.txt.   python tools/diag_isa_ne3.py [chains]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import bestfit, doseresponse as dr, hierarchical as H

dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
shapes, scales, locs = H.prior_params()
exs = []
for d in dr.drugs:
    for c in dr.channels:
        ne, _, ex = dr.load_crumb_data(d, c)
        if len(ex) == 3 and all(len(x) == 4 for x in ex):
            exs.append(ex)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
packed = H.PackedHierPoints(exs)
theta0 = np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs])
res = {}
only = os.environ.get("PHF_DIAG_ONLY")                     # "isa" / "hipcc": one kernel only, 2 000-iteration launches only (PMC passes: every dispatch the same)
for isa in ((False, True, False, True) if not only else ((only == "isa"),)):
    s = H.HierarchicalSampler(packed, list(range(len(exs))), C, thinning=5, seed=1, device="cuda:0")
    s.set_kernel_hint(lanes=1, isa=isa)
    s.init(theta0, cov_scale=0.01)
    s.advance(2000, save=False)
    for I in ((500, 2000) if not only else (2000,)):
        s.advance(I, save=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            s.advance(I, save=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("%d pairs x %d chains, %4d iterations, %s: %.2f ms  (%.2f us per iteration; kernel %d)" % (
            len(exs), C, I, "gfx950 assembly" if isa else "hipcc one lane  ", dt * 1e3, dt / I * 1e6, H.last_kernel()), flush=True)
        res[(isa, I)] = dt
if not only:
    print("ratio hipcc / assembly: 500 its %.3f, 2000 its %.3f" % (res[(False, 500)] / res[(True, 500)], res[(False, 2000)] / res[(True, 2000)]))
