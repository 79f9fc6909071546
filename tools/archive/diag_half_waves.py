"""Diagnostic: does a wavefront with only 32 active lanes cost less vector-ALU time than a full one on gfx950?  C2 (one pair x 65 536
chains = 1 024 wavefronts, one per SIMD) as 1 024 problems-blocks of 64 chains against 2 048 blocks of 32 chains (two half-filled
wavefronts per SIMD) and 4 096 blocks of 16."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler

dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
dr.define_model(2)
packed = dr.pack_single_level([("Amiodarone", "hERG")])
I = 2000
for lanes in (64, 32, 16):
    Q = 65536 // lanes
    s = SingleLevelSampler(packed, 2, [0] * Q, [1.0] * Q, lanes, thinning=5, seed=25, device="cuda:0")
    s.init([6.0, 0.8, 8.0], cov_identity=False, cov_scale=0.05)
    s.reserve(20 * I)
    rows = torch.empty((s.rows_between(0, I), s.Q, s.d + 1, s.C), dtype=torch.float64, device="cuda:0")
    for _ in range(4):
        s.advance(I, out=rows)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        s.advance(I, out=rows)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%2d active lanes per wavefront, %4d wavefronts: %.3f ms per %d iterations, %.3g samples/s" % (lanes, Q, dt * 1e3, I, 65536 * I / dt), flush=True)
