"""Diagnostic: the parts of a fused C4 step (210 pairs x 1 024 chains, 2 000 iterations) on their own: the fused launch of every assembly
group, of the Ne = 3 / Ne = 4 / Ne = 5 groups only, the hipcc rest (Ne = 6) on its stream, and all of it.
-> profiles/r05/c4_fused_launch.txt"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from pyhillfit_amd import doseresponse as dr, hierarchical as H

dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
dev = torch.device("cuda", 0)
names = [(d, c) for d in dr.drugs for c in dr.channels]
I, STEPS = 2000, 8
quantum = int(os.environ.get("PHF_DIAG_QUANTUM", "0"))


def run(label, keep):
    b = bench.HierarchicalBatch(dr, names, 1024, 5, 0, dev, torch)
    idx = [j for j, h in enumerate(b.samplers) if keep(h, j in b.fused_index)]
    fus = [j for j in idx if j in b.fused_index]
    b.fused = H.FusedSamplers([b.samplers[j] for j in fus]) if len(fus) > 1 else None
    if b.fused is not None and quantum:
        b.fused.quantum = quantum
    b.fused_index = fus if b.fused is not None else []
    b.single_index = [j for j in idx if j not in b.fused_index]
    b.streams = [torch.cuda.Stream(device=dev) for _ in range(len(b.single_index) + (1 if b.fused is not None else 0))]
    b.reserve((STEPS + 4) * I * 2)
    rows = b.make_rows(I)
    out = []
    for _ in range(2):
        for _ in range(2):
            b.advance(I, out=rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(STEPS):
            b.advance(I, out=rows, join=False)
        b.join(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / STEPS * 1e3)
    blocks = sum(b.samplers[j].nblocks for j in idx)
    print("%-64s %5d blocks  %6.2f / %6.2f ms per step" % (label, blocks, out[0], out[1]), flush=True)
    del b; torch.cuda.empty_cache()


run("fused: every assembly group (Ne = 3, 4, 5: twelve bodies)", lambda h, f: f)
run("fused: the three Ne = 3 groups", lambda h, f: f and h.n_expts == 3)
run("fused: the five Ne = 4 groups", lambda h, f: f and h.n_expts == 4)
run("fused: the four Ne = 5 groups", lambda h, f: f and h.n_expts == 5)
run("hipcc rest: Ne = 6", lambda h, f: not f)
run("everything (a C4 step)", lambda h, f: True)
