"""Diagnostic: per-iteration latency of the hierarchical kernels in the command-line regime (64 chains per pair)."""
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
C, I = 64, 20000
samplers = {}
for ne, exs in sorted(groups.items(), reverse=True):
    s = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(len(exs))), C, thinning=5, seed=1, device="cuda:0")
    s.init(np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs]), cov_scale=0.01)
    s.advance(2000, save=False)
    samplers[ne] = s
torch.cuda.synchronize()
for label, kw in (("no rows, no moments", dict(save=False)), ("rows", dict(save=True))):
    for ne, s in samplers.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.advance(I, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-22s Ne=%d pairs=%3d alone: %.1f us/iteration" % (label, ne, s.Q, dt / I * 1e6))
for ne, s in samplers.items():
    s.enable_moments(after_iteration=0)
for ne, s in samplers.items():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.advance(I, save=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-22s Ne=%d pairs=%3d alone: %.1f us/iteration" % ("rows + moments", ne, s.Q, dt / I * 1e6))
streams = {ne: torch.cuda.Stream() for ne in samplers}
torch.cuda.synchronize(); t0 = time.perf_counter()
for ne, s in samplers.items():
    with torch.cuda.stream(streams[ne]):
        s.advance(I, save=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("all four groups concurrently on 4 streams: %.1f us/iteration" % (dt / I * 1e6))

# ---- what slows concurrent groups down: sharing SIMDs, or different code competing for the instruction cache? ----
def timed(pairs_of_sampler_stream):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s, st in pairs_of_sampler_stream:
        with torch.cuda.stream(st):
            s.advance(I, save=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / I * 1e6

exs3 = groups[3]
half = len(exs3) // 2
def mk(exs, seed):
    s = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(len(exs))), C, thinning=5, seed=seed, device="cuda:0")
    s.init(np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs]), cov_scale=0.01)
    s.advance(2000, save=False)
    return s
a3, b3 = mk(exs3[:half], 3), mk(exs3[half:], 4)
st = [torch.cuda.Stream() for _ in range(4)]
print("Ne=3 first half alone (77 waves): %.1f us/it" % timed([(a3, st[0])]))
print("Ne=3 two halves on two streams (same code): %.1f us/it" % timed([(a3, st[0]), (b3, st[1])]))
print("Ne=3 half + Ne=4 (41 waves) on two streams: %.1f us/it" % timed([(a3, st[0]), (samplers[4], st[1])]))
print("Ne=4 alone: %.1f us/it" % timed([(samplers[4], st[1])]))
print("Ne=6 + Ne=5: %.1f us/it" % timed([(samplers[6], st[0]), (samplers[5], st[1])]))
print("Ne=6 + Ne=5 + Ne=4: %.1f us/it" % timed([(samplers[6], st[0]), (samplers[5], st[1]), (samplers[4], st[2])]))
