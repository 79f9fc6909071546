"""Diagnostic: C4 (210 pairs x 1 024 chains, five groups side by side, streams not joined between steps) under the launch-side knobs that
cost nothing to turn: HIP stream priorities of the groups, the order in which a step's launches are issued, the quantum of the assembly
kernel's work queue.  One line per variant: ms per 2 000-iteration step over 10 steps, two rounds (-> profiles/r05/c4_launch_knobs.txt)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from pyhillfit_amd import doseresponse as dr

dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
dev = torch.device("cuda", 0)
names = [(d, c) for d in dr.drugs for c in dr.channels]
I, STEPS = 2000, 10


def is_asm(h):
    return h.n_expts == 3 and h.points.packed.points_per_expt == 4


def run(label, prio=None, order=None, quantum=0):
    b = bench.HierarchicalBatch(dr, names, 1024, 5, 0, dev, torch)
    if prio is not None:                                   # prio(h) -> -1 (high) | 0
        b.streams = [torch.cuda.Stream(device=dev, priority=prio(h)) for h in b.samplers]
    if order is not None:
        idx = sorted(range(len(b.samplers)), key=lambda i: order(b.samplers[i]))
        b.samplers = [b.samplers[i] for i in idx]; b.streams = [b.streams[i] for i in idx]
    for h in b.samplers:
        if is_asm(h):
            h.quantum = quantum
    b.reserve((STEPS + 4) * I * 2)
    rows = b.make_rows(I)
    out = []
    for _ in range(2):
        for _ in range(3):
            b.advance(I, out=rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(STEPS):
            b.advance(I, out=rows, join=False)
        b.join(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / STEPS * 1e3)
    print("%-70s %6.2f / %6.2f ms per step" % (label, out[0], out[1]), flush=True)
    del b; torch.cuda.empty_cache()


run("as shipped (equal priorities, launches Ne 6, 5, 4, asm, irregular 3)")
run("assembly kernel's stream LOW, hipcc groups HIGH priority", prio=lambda h: 0 if is_asm(h) else -1)
run("assembly kernel's stream HIGH, hipcc groups LOW", prio=lambda h: -1 if is_asm(h) else 0)
run("Ne >= 5 HIGH, the rest LOW", prio=lambda h: -1 if h.n_expts >= 5 else 0)
run("assembly kernel launched FIRST in a step", order=lambda h: (0 if is_asm(h) else 1, -h.n_expts))
run("assembly kernel launched LAST in a step", order=lambda h: (1 if is_asm(h) else 0, -h.n_expts))
for q in (70, 280, 500, 1000):
    run("assembly queue quantum %d iterations (library's choice: 140)" % q, quantum=q)
run("as shipped, again")
