"""Measurement for SURVEY 8f-4 (posterior-predictive curves): the GPU accumulation on the BASELINE-C4 shape next to the
reference's arithmetic (scipy.stats, one sample at a time, restated below) on the host.  One JSON line.

    python tools/bench_predictive.py [--pairs 210] [--rows 75001] [--chains 1] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def reference_arithmetic(alphas, betas, mus, ss, grid_points=501):
    """what construct_hierarchical_cdfs.py:32-58 does per saved sample: four scipy.stats calls on the two grids, summed
    (kept inside this tool: oracle/ is for tests, smoke() and bench.py's cpu_baseline only)"""
    import scipy.stats as st
    hill_x, pic50_x = np.linspace(0., 4., grid_points), np.linspace(-2., 12., grid_points)
    sums = np.zeros((4, grid_points))
    for a_, b_, m_, s_ in zip(alphas, betas, mus, ss):
        sums[0] += st.fisk.cdf(hill_x, c=b_, scale=a_, loc=0)
        sums[2] += st.fisk.pdf(hill_x, c=b_, scale=a_, loc=0)
        sums[1] += st.logistic.cdf(pic50_x, m_, s_)
        sums[3] += st.logistic.pdf(pic50_x, m_, s_)
    sums /= len(alphas)
    return hill_x, sums[0], pic50_x, sums[1], sums[2], sums[3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=210)
    ap.add_argument("--rows", type=int, default=75001, help="post-burn rows per chain (reference: 75 001)")
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-samples", type=int, default=20000)
    a = ap.parse_args()
    import torch
    from pyhillfit_amd.predictive import PredictiveCurves
    dev = "cuda:0"
    g = torch.Generator(device=dev); g.manual_seed(1)
    rows = torch.empty((a.rows, a.pairs, 4, a.chains), dtype=torch.float64, device=dev)
    rows[:, :, 0] = 0.3 + torch.rand((a.rows, a.pairs, a.chains), generator=g, device=dev, dtype=torch.float64) * 1.5
    rows[:, :, 1] = 2.1 + torch.rand((a.rows, a.pairs, a.chains), generator=g, device=dev, dtype=torch.float64) * 8
    rows[:, :, 2] = 3.0 + torch.rand((a.rows, a.pairs, a.chains), generator=g, device=dev, dtype=torch.float64) * 6
    rows[:, :, 3] = 0.02 + torch.rand((a.rows, a.pairs, a.chains), generator=g, device=dev, dtype=torch.float64) * 0.5
    s = rows[:max(a.cpu_samples, 2000), 0, :, 0].cpu().numpy()
    check = reference_arithmetic(s[:2000, 0], s[:2000, 1], s[:2000, 2], s[:2000, 3])
    pc2 = PredictiveCurves(1, dev)
    pc2.accumulate(rows[:2000, :1, :, :1].contiguous())
    got = pc2.result(0)
    err = max(float(np.max(np.abs(got[i] - check[i]) / np.maximum(np.abs(check[i]), 1e-60))) for i in (1, 3, 4, 5))
    pc = PredictiveCurves(a.pairs, dev)
    pc.accumulate(rows)                                        # warm-up (allocates the scratch)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(a.steps):
        pc.accumulate(rows)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / a.steps
    evals = a.rows * a.chains * a.pairs * pc.G                 # (sample, grid point) pairs; each feeds 4 curves
    # host: the reference's arithmetic (scipy.stats fisk/logistic cdf+pdf per sample) on a bounded sample, 1 core
    n = a.cpu_samples
    t0 = time.perf_counter()
    reference_arithmetic(s[:n, 0], s[:n, 1], s[:n, 2], s[:n, 3])
    cpu = time.perf_counter() - t0
    print(json.dumps({"what": "posterior-predictive curves (construct_hierarchical_cdfs.py:32-58)", "pairs": a.pairs,
                      "samples_per_pair": a.rows * a.chains, "grid_points": pc.G, "ms_per_pass": ms,
                      "sample_grid_evaluations_per_s": evals / (ms * 1e-3),
                      "samples_per_s": a.rows * a.chains * a.pairs / (ms * 1e-3),
                      "cpu_reference_arithmetic": {"samples": n, "seconds": cpu, "samples_per_s": n / cpu, "cores": 1,
                                                   "kind": "port (scipy.stats per sample, as the reference loops)"},
                      "speedup_vs_1_core": (a.rows * a.chains * a.pairs / (ms * 1e-3)) / (n / cpu),
                      "max_rel_err_vs_port_2000_samples": err}))


if __name__ == "__main__":
    main()
