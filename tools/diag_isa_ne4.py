"""Diagnostic: the 32 Crumb pairs with 4 + 4 + 4 + 1 points x C chains, launches of 2 000 iterations: phf_hier4_advance_s4441 (the gfx950 assembly
build of the Ne = 4 iteration: two wavefronts per SIMD, a device-memory scratch tier) against hier_advance_kernel<4> (hipcc, one
wavefront per SIMD).  -> profiles/r05/c4_ne4_assembly_vs_hipcc.txt"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import bestfit, doseresponse as dr, hierarchical as H

dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
shapes, scales, locs = H.prior_params()
exs = []
for d in dr.drugs:
    for c in dr.channels:
        ne, _, ex = dr.load_crumb_data(d, c)
        if H.group_key(ex, H.ISA_SHAPES) == (4, H.shape_code((4, 4, 4, 1))):
            exs.append(ex)
packed = H.PackedHierPoints(exs)
theta0 = np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs])
I = 2000
for C in [int(x) for x in (sys.argv[1:] or ["1024", "4096", "64"])]:
    for isa in (False, True):
        s = H.HierarchicalSampler(packed, list(range(len(exs))), C, thinning=5, seed=5, device="cuda:0")
        s.set_kernel_hint(lanes=1, isa=isa)
        s.init(theta0, cov_scale=0.01)
        s.reserve(12 * I)
        rows = torch.empty((I // 5, s.Q, s.d + 1, C), dtype=torch.float64, device="cuda:0")
        for _ in range(3):
            s.advance(I, out=rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            s.advance(I, out=rows)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        blocks = s.Q * -(-C // 64)
        print("%d pairs x %5d chains (%5d blocks)  %-8s kernel %d: %8.2f ms per %d iterations = %.2f ns per 64-chain block-iteration, %.3g samples/s"
              % (s.Q, C, blocks, "assembly" if isa else "hipcc", H.last_kernel(), dt * 1e3, I, dt * 1e9 / I / blocks, s.Q * C * I / dt), flush=True)
        s.check_queue()
        del s
