"""Diagnostic: C4 (210 pairs x 1 024 chains, four Ne groups side by side) with different kernels per group: one lane per chain everywhere
(what hint_side_by_side picks) against two lanes per chain for the groups with Ne >= 6, >= 5, >= 4."""
import os, sys, time, argparse
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from pyhillfit_amd import doseresponse as dr, distributed as D

dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
dev = torch.device("cuda", 0)
names = [(d, c) for d in dr.drugs for c in dr.channels]
I = 2000
for two_from in (99, 6, 5, 4):
    b = bench.HierarchicalBatch(dr, names, 1024, 5, 0, dev, torch)
    for h in b.samplers:
        h.set_kernel_hint(lanes=2 if h.n_expts >= two_from else 1)
    b.reserve(20 * I)
    rows = b.make_rows(I)
    for _ in range(3):
        b.advance(I, out=rows)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        b.advance(I, out=rows, join=False)
    b.join(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print("two lanes per chain for Ne >= %2d: %.2f ms per %d iterations = %.3g samples/s" % (two_from, dt * 1e3, I, b.chains * I / dt), flush=True)
    del b; torch.cuda.empty_cache()
