#!/usr/bin/env python3
"""C3 launch (210 pairs x 4 096 chains, 2 000 iterations) with the problems' wavefronts handed to the GPU in file order, most
expensive first (the default, phf_problems.launch_order) and cheapest first: what the order of a ragged launch costs."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import pyhillfit_amd  # noqa: F401
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler

dr.setup("data/crumb_dataset.json"); dr.define_model(2)
names = [(d, c) for d in dr.drugs for c in dr.channels]
packed = dr.pack_single_level(names)
cost = 525.0 + 28.0 * packed.counts[:, 0] + 115.0 * (packed.counts[:, 1] + packed.counts[:, 2])
dev = torch.device("cuda", 0)
for tag, order in (("file order", None), ("most expensive first", "cost"), ("cheapest first", np.argsort(cost, kind="stable"))):
    s = SingleLevelSampler(packed, 2, list(range(len(names))), [1.0] * len(names), 4096, thinning=5, seed=25, device=dev, launch_order=order)
    s.init([6.0, 0.8, 8.0], cov_identity=False, cov_scale=0.05)
    I = 2000; s.reserve(12 * I)
    rows = torch.empty((s.rows_between(0, I), len(names), 4, 4096), dtype=torch.float64, device=dev)
    for _ in range(4):
        s.advance(I, out=rows)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        s.advance(I, out=rows)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print("%-22s ms/launch %.2f  samples/s %.4g" % (tag, dt * 1e3, len(names) * 4096 * I / dt))
    del s, rows
