#!/usr/bin/env python3
"""Benchmark of the hot path: MH samples/sec of the HIP sampler (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3] [--iters-per-step I]

A "step" is ONE launch of the sampler kernel advancing every chain of the batch by I Metropolis-Hastings
iterations (thinned samples of all chains written to HBM).  N=1 workload = BASELINE.json configs[1]:
Amiodarone-hERG, model 2, 65 536 chains.  With N>1 (launched by torch.distributed.run, one rank per GPU) every
rank runs its own 65 536-chain shard of the chain batch (weak scaling, no collective in the data path; the
Philox chain ids continue across ranks), timing is barrier + synchronize on both sides, MAX over ranks.
Rank 0 prints ONE JSON line."""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import pyhillfit_amd  # noqa: E402,F401  (sets the ROCm runtime defaults — hardware queues — before the first GPU call)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 64 fp64 FMA lanes x 2 x 2.4 GHz (vector, non-MFMA)
FP64_VALU_MEASURED_TFLOPS = 58.3  # tools/microbench.hip: v_fma_f64 saturates at 2.25 ns per wave-instruction per SIMD


def profile_facts(workload, chains, iters, thinning):
    """Per-launch PMC facts measured with rocprofv3 on this kernel and committed under profiles/ (PMC counters cannot be
    read from inside the process): fp64 flop per MH iteration (SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64) and HBM traffic
    (WRITE_SIZE + 2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md), valid for the exact launch shape they
    were taken on — otherwise traffic is null."""
    path = os.path.join(REPO, "profiles", "pmc_facts.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        facts = json.load(f).get(workload)
    if not facts:
        return {}
    out = {"flop_per_iteration": facts.get("flop_per_iteration"), "source": facts.get("source")}
    if (facts.get("chains"), facts.get("iterations_per_launch"), facts.get("thinning")) == (chains, iters, thinning):
        out["traffic_bytes_per_launch"] = facts.get("traffic_bytes_per_launch")
    return out


def cpu_baseline(iterations=300000):
    """rank 0 only, N=1 only: the oracle timed on the host (about 10 s + 5 s of CPU work).
    'port' = the numpy/scipy restatement of the reference
    loop (python/PyHillTemp.py:57-125 with numpy's legacy RNG, like the reference), one core, same pair/model.
    The scalar C twin's rate is reported next to it."""
    from oracle import pyhillfit_oracle as orc
    from oracle import c_oracle as co
    from pyhillfit_amd import doseresponse as dr
    ne, _, ex = dr.load_crumb_data("Amiodarone", "hERG")
    concs, y = dr.concatenate_experiments(ne, ex)
    pair = orc.PairData(concs, y)
    t0 = time.perf_counter()
    orc.single_level_chain(2, pair, [6.0, 0.8, 8.0], iterations, 5, orc.LegacyNumpyDraws(25))
    dt = time.perf_counter() - t0
    pk = co.PackedPair(concs, y, 2, 1.0)
    st = pk.init_state([6.0, 0.8, 8.0], False, 0.05)
    n_c = 20000000
    gam = co.gamma_table(n_c)
    t0 = time.perf_counter()
    pk.advance(st, 0, n_c, 5, 3000, False, gam, seed=25)
    dtc = time.perf_counter() - t0
    return {"value": iterations / dt, "unit": "MH samples/s", "cores": 1, "kind": "port",
            "sample": "%d iterations of 1 chain, Amiodarone-hERG model 2, numpy/scipy restatement of the reference loop "
                      "(oracle/pyhillfit_oracle.py), %.1f s" % (iterations, dt),
            "c_twin_value": n_c / dtc, "c_twin_sample": "%d iterations, scalar C twin (oracle/phf_oracle.c), 1 core, %.1f s" % (n_c, dtc),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps before the timed ones (the first ~6 launches of a process run ~10 % slow while the clocks settle)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--iters-per-step", type=int, default=2000)
    ap.add_argument("--chains", type=int, default=None, help="chains per problem (default: 65536 c2, 4096 c3, 1024 c4/c5)")
    ap.add_argument("--thinning", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the sampler has no CPU path)")
    # one rank per GPU over RCCL.  Rehearsal on a single-GPU box: PHF_BENCH_BACKEND=gloo lets several ranks share
    # device 0 (RCCL refuses two ranks on one GPU); the timing protocol (barrier, MAX over ranks) is the same.
    backend = os.environ.get("PHF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from pyhillfit_amd import doseresponse as dr
    from pyhillfit_amd.sampler import SingleLevelSampler
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    dr.define_model(2)
    all_names = [(d, c) for d in dr.drugs for c in dr.channels]
    kernel_name = "mh_advance_kernel<2>"
    if a.workload == "c2":
        names = [("Amiodarone", "hERG")]
        C = a.chains or 65536
        label = "Amiodarone-hERG, non-hierarchical model 2 (pIC50, Hill, sigma), %d chains per GPU (BASELINE configs[1])" % C
    elif a.workload == "c3":
        names, C = all_names, a.chains or 4096
        label = "all %d Crumb drug x channel pairs, non-hierarchical model 2, %d chains each per GPU (BASELINE configs[2])" % (len(names), C)
    elif a.workload == "c5":
        names, C = all_names, a.chains or 1024
        label = "thermodynamic-integration ladder: 32 rungs x %d pairs x %d chains per GPU, model 2 (BASELINE configs[4])" % (len(names), C)
    else:
        names, C = all_names, a.chains or 1024
        label = "hierarchical model, all Crumb pairs, %d chains each per GPU (BASELINE configs[3])" % C
    if a.workload == "c4":
        # hierarchical: one sampler per Ne group, stepped back to back inside a "step"
        from pyhillfit_amd import hierarchical as H
        groups = {}
        for d_, c_ in names:
            ne, _, ex = dr.load_crumb_data(d_, c_)
            groups.setdefault(len(ex), []).append(ex)
        shapes, scales, locs = H.prior_params()
        samplers = []
        for ne, exs in sorted(groups.items(), reverse=True):
            hs = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(len(exs))), C, thinning=a.thinning, seed=25,
                                       chain_id_base=rank * C, device=dev)
            hs.init(np.array([H.first_iteration(e, locs) for e in exs]), cov_scale=0.01)
            samplers.append(hs)
        kernel_name = "hier_advance_kernel<Ne=3..6>"

        class Multi(object):
            d = None
            adapt_start = max(h.adapt_start for h in samplers)
            bytes_per_iter = sum(h.Q * C * 8.0 * (h.d + 1) for h in samplers) / a.thinning
            chains = sum(h.Q * C for h in samplers)

            def reserve(self, n):
                [h.reserve(n) for h in samplers]

            def make_rows(self, I):
                return [torch.empty((h.rows_between(0, I), h.Q, h.d + 1, C), dtype=torch.float64, device=dev) for h in samplers]

            streams = [torch.cuda.Stream(device=dev) for _ in samplers]

            def advance(self, I, out):
                # one HIP stream per Ne group: the small groups (Ne = 5, 6: tens of wavefronts) are latency-bound
                # and hide under the big Ne = 3 launch instead of queueing behind it
                cur = torch.cuda.current_stream(dev)
                for h, o, st in zip(samplers, out, self.streams):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        h.advance(I, out=o)
                for st in self.streams:
                    cur.wait_stream(st)
        s = Multi()
        Q = len(names)
    else:
        packed = dr.pack_single_level(names)
        if a.workload == "c5":
            ladder = dr.temperature_ladder(31)                           # 32 rungs (BASELINE configs[4]); reference ladder has 41
            pair_index = [p for p in range(len(names)) for _ in ladder]
            temps = [float(t) for _ in names for t in ladder]
        else:
            pair_index, temps = list(range(len(names))), [1.0] * len(names)
        Q = len(pair_index)
        # weak scaling: rank r owns chains [r*C, (r+1)*C) of every problem
        s = SingleLevelSampler(packed, 2, pair_index, temps, C, thinning=a.thinning, seed=25, chain_id_base=rank * C,
                               reset_mean_at_adapt_start=(a.workload == "c5"), device=dev)
        if a.workload == "c5":
            s.init(np.ones(3), cov_identity=True, cov_scale=1.0)          # PyHillTemp.py:63,80 start
        else:
            s.init([6.0, 0.8, 8.0], cov_identity=False, cov_scale=0.05)   # PyHillFit.py:748-751 start
        s.bytes_per_iter = float(Q) * C * 8 * (s.d + 1) / a.thinning      # 8(d+1)/thin B per iteration
        s.chains = Q * C
        s.make_rows = lambda I: torch.empty((s.rows_between(0, I), Q, s.d + 1, C), dtype=torch.float64, device=dev)
    I = a.iters_per_step
    s.reserve((a.warmup + a.steps) * I)
    rows = s.make_rows(I)

    for _ in range(a.warmup):
        s.advance(I, out=rows)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for b_, e_ in ev:                                # events are created lazily at their first record: not inside the timed region
        b_.record(); e_.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    gc.collect(); gc.disable()                      # no collector pause between launches of the timed region
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record()
        s.advance(I, out=rows)
        ev[k][1].record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kernel_ms = float(np.mean([b.elapsed_time(e) for b, e in ev]))   # HIP events on the launch stream
    for r_ in (rows if isinstance(rows, list) else [rows]):
        assert torch.isfinite(r_).all()

    if rank == 0:
        samples_per_step = float(s.chains) * I * world
        value = samples_per_step * a.steps / dt
        alg_bytes = s.bytes_per_iter * I
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        prof = profile_facts(a.workload, s.chains, I, a.thinning)
        flop_per_iter = prof.get("flop_per_iteration")
        tflops = None if flop_per_iter is None else float(s.chains) * I * flop_per_iter / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "MCMC samples/sec (whole node)", "value": value, "unit": "MH samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "real Crumb dose-response rows (data/crumb_dataset.json), synthetic chain batch",
            "config": {"workload": label, "iterations_per_step": I, "thinning": a.thinning, "chains_per_gpu": s.chains,
                       "adaptation": "on (steps start after iteration %d > adapt_start %d)" % (a.warmup * I, s.adapt_start)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": prof.get("traffic_bytes_per_launch"), "kernel": kernel_name, "kernel_ms": kernel_ms,
                         "traffic_source": prof.get("source"),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "fp64-VALU-bound scalar-per-chain arithmetic; the HBM fraction is reported as BASELINE asks, the binding roof is fp64_valu"},
            "fp64_valu": {"achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": None if tflops is None else tflops / FP64_VALU_PEAK_TFLOPS,
                          "measured_achievable_peak": FP64_VALU_MEASURED_TFLOPS, "flop_per_iteration": flop_per_iter,
                          "flop_source": prof.get("source")},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
