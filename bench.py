#!/usr/bin/env python3
"""Benchmark of the hot path: MH samples/sec of the HIP sampler (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--iters-per-step I] [--model 1|2] [--moments]

A "step" is ONE pass of the sampler over the whole batch: every chain advanced by I Metropolis-Hastings iterations
(thinned samples of all chains written to HBM).  The N=1 workload is the one BASELINE.json's metric is quoted on,
configs[2]: ALL 210 Crumb drug x channel pairs, non-hierarchical model 2, 4 096 chains each (c3); c2 / c4 / c5 are the
other configurations.

--gpus N > 1: this process starts `python -m torch.distributed.run --nproc-per-node N` as a child BEFORE it makes any
GPU call (one rank per GPU over RCCL; never a re-exec) and passes the child's output and exit code through; under
torchrun (the driver's way of starting it) RANK / LOCAL_RANK / WORLD_SIZE are read from the environment.  The path
shards without a data-path collective (python/PyHillFit.py:978-1003 maps pairs over a process pool):
  weak scaling (default): every rank runs the full workload on its own range of Philox chain ids
                          (rank r owns chains [r C, (r+1) C) of every pair), value = all ranks' iterations / time;
  strong scaling:         the (pair, 64-chain block) units of ONE full-size workload are partitioned over the ranks by cost
                          (distributed.shard_blocks: 13 440 blocks -> 1 680 per GPU at 8), every chain keeping its Philox stream.
With N > 1 BOTH are measured in the same run, one timed region each: `value` is the region --scaling names (weak by default),
`strong_value` (or `weak_value`) the other.  With N = 1 the c3 run also times short regions of c2, c4, c5 and of the kernels the command
lines and the thermodynamic-integration path launch — c5 with the on-device <log L> and moments (`c5_moments`), c3 with model 1
(`c3_model1`) — (`other_workloads`).
c4 (and a hierarchical strong-scaling share) launches the groups the gfx950 code object has kernels for — every pair with three or four
experiments — as ONE persistent grid per step (phf_hierarchical_advance_fused), Ne = 5 / 6 beside it on streams of their own;
PHF_BENCH_HIER_FUSED=0 gives every group a launch and a stream of its own (the arrangement before ABI 7), =3 fuses the Ne = 3 groups only — A/B
modes: assembly launches side by side carry the hazard of profiles/r05/queue_progress_word_hazard.txt (the fault word is checked).
RCCL is used outside the timed regions only: rank 0 reads and packs the data and broadcasts it, the per-problem
acceptance summaries are gathered to rank 0.  Timing: barrier + synchronize on both sides, MAX over ranks.
Rank 0 prints ONE JSON line."""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import pyhillfit_amd  # noqa: E402,F401  (sets the ROCm runtime defaults — hardware queues — before the first GPU call)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 64 fp64 FMA lanes x 2 x 2.4 GHz (vector, non-MFMA)
FP64_VALU_MEASURED_TFLOPS = 58.3  # tools/microbench.hip: v_fma_f64 saturates at 2.25 ns per wave-instruction per SIMD

DEFAULT_CHAINS = {"c2": 65536, "c3": 4096, "c4": 1024, "c5": 1024, "s3": 1024, "s3h": 1024}
# s3: SURVEY 8(d)'s synthetic scaling set (pyhillfit_amd/synthetic.py): P generated pairs, model 2; N > 1 splits ONE 1 680 x 4 096 batch
S3_PAIRS, S3_STRONG_CHAINS = 1680, 4096
# c3: 24 000 iterations per step = one queued launch of ~0.3 s (6 quanta of 4 000), so that the driver's 20 timed steps last > 6 s;
# c4: 2 000 per step — the command line runs 20 000 per launch, and at 500 a third of the HBM traffic was the state going in and out
DEFAULT_ITERS = {"c2": 2000, "c3": 24000, "c4": 2000, "c5": 500, "s3": 4000, "s3h": 2000}
# (timed steps, warm-up steps) of the short regions after the headline; each region lasts 0.1 .. 0.5 s
# (steps, warm-up) of the short regions: every one of them at least half a second of timed work (c2 2.3 ms, c4 ~48, c5 ~50, s3 ~100,
# c3 model 1 ~270 ms per step)
OTHER_STEPS = {"c2": (250, 10), "c4": (12, 3), "c5": (12, 4), "c5_moments": (12, 4), "c3_model1": (3, 2), "s3": (6, 2), "s3h": (3, 1)}
OTHER_SPECS = {"c2": ("c2", 2, False), "c4": ("c4", 2, False), "c5": ("c5", 2, False), "s3": ("s3", 2, False), "s3h": ("s3h", 2, False),          # name -> (workload, model, moments)
               "c5_moments": ("c5", 2, True), "c3_model1": ("c3", 1, False)}


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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s=12.0):
    """rank 0 only, N=1 only: the oracle timed on the host, one core (the reference loop is single-threaded per pair).

    value = the reference's COST PROFILE: the restated loop (python/PyHillTemp.py:57-125 / PyHillFit.py:830-848) calling
    scipy.stats.norm.logcdf/logsf per iteration exactly as python/doseresponse.py:244-245 does and
    RandomState.multivariate_normal / rand as PyHillFit.py:831,834 do.  Next to it: the lean numpy port (scipy.special
    directly) and the scalar C twin of the kernels."""
    from oracle import pyhillfit_oracle as orc
    from oracle import c_oracle as co
    from pyhillfit_amd import doseresponse as dr
    ne, _, ex = dr.load_crumb_data("Amiodarone", "hERG")
    concs, y = dr.concatenate_experiments(ne, ex)
    pair = orc.PairData(concs, y)
    th0 = [6.0, 0.8, 8.0]
    # a short probe sizes the sample to ~budget_s of CPU work
    t0 = time.perf_counter()
    orc.single_level_chain(2, pair, th0, 2000, 5, orc.LegacyNumpyDraws(25), as_reference=True)
    rate = 2000 / (time.perf_counter() - t0)
    iterations = int(max(5000, min(400000, rate * budget_s)) // 5 * 5)
    t0 = time.perf_counter()
    orc.single_level_chain(2, pair, th0, iterations, 5, orc.LegacyNumpyDraws(25), as_reference=True)
    dt = time.perf_counter() - t0
    n_lean = int(max(5000, min(400000, 5 * iterations)) // 5 * 5)
    t0 = time.perf_counter()
    orc.single_level_chain(2, pair, th0, n_lean, 5, orc.LegacyNumpyDraws(25))
    dt_lean = time.perf_counter() - t0
    pk = co.PackedPair(concs, y, 2, 1.0)
    st = pk.init_state(th0, False, 0.05)
    n_c = 20000000
    gam = co.gamma_table(n_c)
    t0 = time.perf_counter()
    pk.advance(st, 0, n_c, 5, 3000, False, gam, seed=25)
    dtc = time.perf_counter() - t0
    host_cpus = os.cpu_count() or 1
    pool = max(1, min(210, host_cpus - 1))                 # mp.Pool(min(num_cores, cpu_count() - 1)) over the pairs, PyHillFit.py:997-1003
    return {"value": iterations / dt, "unit": "MH samples/s", "cores": 1, "kind": "port",
            "pool_equivalent_value": iterations / dt * pool, "pool_equivalent_processes": pool,
            "pool_equivalent_note": "the 1-core figure x min(210 pairs, host cpus - 1): what the reference's process pool over pairs "
                                    "(python/PyHillFit.py:997-1003) would reach on this host if it scaled perfectly — computed, not measured",
            "sample": "%d iterations of 1 chain, Amiodarone-hERG model 2: restatement of the reference loop with the reference's "
                      "own library calls (scipy.stats.norm.logcdf/logsf per iteration, RandomState.multivariate_normal/rand; "
                      "oracle/pyhillfit_oracle.py as_reference=True), %.1f s" % (iterations, dt),
            "lean_port_value": n_lean / dt_lean,
            "lean_port_sample": "%d iterations, same loop calling scipy.special.log_ndtr directly, %.1f s" % (n_lean, dt_lean),
            "c_twin_value": n_c / dtc, "c_twin_sample": "%d iterations, scalar C twin (oracle/phf_oracle.c), 1 core, %.1f s" % (n_c, dtc),
            "cpu_model": _cpu_model(), "host_cpus": os.cpu_count()}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps before the timed ones (the first launches of a process run ~10 %% slow while the clocks settle)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5", "s3", "s3h"])
    ap.add_argument("--pairs", type=int, default=None, help="s3: number of generated pairs (default %d; SURVEY 8(d): 210 or 1 680)" % S3_PAIRS)
    ap.add_argument("--strong-chains", type=int, default=None, help="N > 1, default c3 run: chains per pair of the S3 batch that the strong-scaling "
                    "region splits (default %d)" % S3_STRONG_CHAINS)
    ap.add_argument("--iters-per-step", type=int, default=None, help="MH iterations per step (default: 8000 c3, 2000 c2, 500 c4/c5)")
    ap.add_argument("--chains", type=int, default=None, help="chains per problem (default: 65536 c2, 4096 c3, 1024 c4/c5)")
    ap.add_argument("--thinning", type=int, default=5)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--moments", action="store_true", help="also accumulate posterior moments and <log L> on the device (what the CLIs and the thermodynamic-integration path run)")
    ap.add_argument("--model", type=int, default=2, choices=[1, 2], help="single-level workloads: 2 = (pIC50, Hill, sigma) (BASELINE configs), 1 = (pIC50, sigma), Hill = 1")
    ap.add_argument("--queue-quanta", type=int, default=None, help="quanta per block of the work-queue launch (0: plain launch; default: the sampler's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="N = 1, c3: skip the short c2 / c4 / c5 regions after the headline")
    ap.add_argument("--single-region", action="store_true", help="N > 1: only the region --scaling names (no second region of the other kind)")
    return ap.parse_args(argv)


def spawn_ranks(a):
    """--gpus N without a torchrun environment: start the N ranks as a fresh child (no GPU call has been made by this
    process; it is a child process, not an exec) and hand back its exit code."""
    from pyhillfit_amd import distributed as D
    backend = os.environ.get("PHF_BENCH_BACKEND", "nccl")
    # the count comes from the driver's topology files (a child interpreter only where they are unreadable): the HIP runtime stays
    # untouched in this process; the gloo rehearsal (several ranks on one GPU) does not need it at all
    have = D.visible_gpu_count() if backend == "nccl" else a.gpus
    if backend == "nccl" and have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for but %d GPU(s) visible (RCCL needs one GPU per rank; "
                         "PHF_BENCH_BACKEND=gloo rehearses several ranks on one GPU)\n" % (a.gpus, have))
        return 2
    cmd = D.torchrun_command(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:])   # --standalone: the launcher picks the rendezvous port
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


class HierarchicalBatch(object):
    """c4: one sampler per group of pairs with equal numbers of experiments, one HIP stream each, inside a 'step'.
    units: None = every pair with chains [chain_id_base, chain_id_base + C); or this rank's (pair, 64-chain block) units
    (distributed.shard_blocks over all 210 pairs) — each block then is a problem of 64 chains with its own chain offset."""
    is_hier = True
    kernel_name = ("phf_hier3_advance (hand-allocated gfx950 build: the pairs of 3 x 4 points) + hier_advance_kernel<Ne=3..6> for the other groups, "
                   "one stream each (hier_advance2_kernel for groups that do not fill the chip together)")

    def __init__(self, dr, names, C, thinning, chain_id_base, dev, torch, units=None, experiments=None):
        """experiments: the pairs' experiments when they are not Crumb pairs (s3h: the generated set)"""
        from pyhillfit_amd import hierarchical as H
        self.torch, self.dev = torch, dev
        # PHF_BENCH_HIER_FUSED=0: every group a launch of its own on its own stream (what rounds 1-5 measured); default: the groups the
        # gfx950 code object has kernels for — Ne = 3 and the 4 + 4 + 4 + {1, 2, 3} shapes of Ne = 4 — in ONE persistent grid
        # (phf_hierarchical_advance_fused), the rest beside it on streams of their own
        mode = os.environ.get("PHF_BENCH_HIER_FUSED", "1")   # "3": only the Ne = 3 groups in the fused grid (A/B)
        self.use_fused = mode != "0"
        shapes = None if not self.use_fused else ({k for k in H.ISA_SHAPES if k[0] == 3} if mode == "3" else H.ISA_SHAPES)
        groups = {}
        for p, (d_, c_) in enumerate(names):
            ex = dr.load_crumb_data(d_, c_)[2] if experiments is None else experiments[p]
            groups.setdefault(H.group_key(ex, shapes), []).append((p, ex))
        shapes, scales, locs = H.prior_params()
        self.samplers = []
        for (ne, _), members in sorted(groups.items(), reverse=True):
            exs = [ex for _, ex in members]
            start = np.array([H.first_iteration(e, locs) for e in exs])
            if units is None:
                hs = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(len(exs))), C, thinning=thinning, seed=25,
                                           problem_ids=[p for p, _ in members], chain_id_base=chain_id_base, device=dev)
                hs.init(start, cov_scale=0.01)
            else:
                local = {p: k for k, (p, _) in enumerate(members)}
                mine = [(local[int(p)], int(b)) for p, b in units if int(p) in local]
                if not mine:
                    continue
                hs = H.HierarchicalSampler(H.PackedHierPoints(exs), [k for k, _ in mine], 64, thinning=thinning, seed=25,
                                           problem_ids=[members[k][0] for k, _ in mine], chain_offsets=[64 * b for _, b in mine], device=dev)
                hs.init(start[[k for k, _ in mine]], cov_scale=0.01)
            self.samplers.append(hs)
        if not self.samplers:
            raise SystemExit("bench.py: this rank's share of the batch is empty (more ranks than (pair, 64-chain block) units)")
        self.adapt_start = max(h.adapt_start for h in self.samplers)
        self.bytes_per_iter = sum(h.Q * h.C * 8.0 * (h.d + 1) for h in self.samplers) / thinning
        self.chains = sum(h.Q * h.C for h in self.samplers)
        if set(groups) == {(3, 4)}:                        # every pair has three experiments of four points (s3h): one launch of the assembly kernel
            self.kernel_name = "phf_hier3_advance (hand-allocated gfx950 build of the Ne = 3 iteration, work queue)"
        H.hint_side_by_side(self.samplers)                 # the groups run side by side: one lane per chain once they fill the chip together
        # launch units: the fused set (one launch, one stream), then every other sampler (a launch and a stream each)
        in_fused = [h for h in self.samplers if self.use_fused and (h.n_expts, h.points.packed.points_per_expt) in H.ISA_SHAPES]
        self.fused = H.FusedSamplers(in_fused) if len(in_fused) > 1 else None
        if self.fused is None:
            in_fused = []
        self.fused_index = [self.samplers.index(h) for h in in_fused]
        self.single_index = [j for j in range(len(self.samplers)) if j not in self.fused_index]
        if self.fused is not None:
            self.kernel_name = ("phf_hier_fused_advance (one persistent grid, a body per (Ne, point shape): %d of the %d launch groups) + hier_advance_kernel<Ne> "
                                "for the other groups on streams of their own" % (len(in_fused), len(self.samplers)))
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(len(self.single_index) + (1 if self.fused is not None else 0))]

    def enable_moments(self):
        [h.enable_moments(after_iteration=0) for h in self.samplers]

    def reserve(self, n):
        [h.reserve(n) for h in self.samplers]

    def make_rows(self, I):
        return [self.torch.empty((h.rows_between(0, I), h.Q, h.d + 1, h.C), dtype=self.torch.float64, device=self.dev)
                for h in self.samplers]

    def advance(self, I, out, join=True):
        # one HIP stream per Ne group: the small groups (Ne = 5, 6: tens of wavefronts) are latency-bound
        # and hide under the big Ne = 3 launch instead of queueing behind it.  join=False: the groups' streams are not joined
        # after the step — consecutive steps of a group follow each other on its stream, the groups drift apart, and the chip sees
        # the ragged tail of a launch once per run instead of once per step: how pyhillfit_amd.hierarchical.run_hierarchical
        # drives its segments.  The caller joins once at the end (self.join()).
        torch = self.torch
        cur = torch.cuda.current_stream(self.dev)
        streams = list(self.streams)
        for j in self.single_index:                       # the hipcc groups first: their lone wavefronts take whole SIMDs, the persistent grid what is left
            st = streams.pop(0)
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self.samplers[j].advance(I, out=out[j])
        if self.fused is not None:
            st = streams.pop(0)
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self.fused.advance(I, out=[out[j] for j in self.fused_index])
        if join:
            self.join()

    def join(self):
        cur = self.torch.cuda.current_stream(self.dev)
        for st in self.streams:
            cur.wait_stream(st)

    def mark(self, which):
        """HIP events on every group's stream (the streams the kernels are launched on): start and end of the timed region"""
        torch = self.torch
        cur = torch.cuda.current_stream(self.dev)
        evs = []
        for st in self.streams:
            if which == "start":
                st.wait_stream(cur)
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            evs.append(e)
        setattr(self, "ev_" + which, evs)

    def region_ms(self):
        return max(b.elapsed_time(e) for b, e in zip(self.ev_start, self.ev_end))

    def acceptance_summary(self):
        if self.fused is not None:
            self.fused.check_queue()
        return self.torch.cat([h.acceptance().mean(dim=1) for h in self.samplers])


class SingleLevelBatch(object):
    """c2 / c3 / c5: one SingleLevelSampler.  units as for HierarchicalBatch (indices into the problem list)."""
    is_hier = False

    def __init__(self, packed, pair_index, temps, C, a, chain_id_base, dev, torch, tempered, units=None):
        from pyhillfit_amd.sampler import SingleLevelSampler
        self.torch, self.dev = torch, dev
        model = a.model
        self.kernel_name = "mh_advance_kernel<%d, %s>" % (model, "moments" if a.moments else "no moments")
        kw = {} if a.queue_quanta is None else {"queue_quanta": a.queue_quanta}
        if units is not None and len(units) == 0:
            raise SystemExit("bench.py: this rank's share of the batch is empty (more ranks than (pair, 64-chain block) units)")
        if units is None:
            s = SingleLevelSampler(packed, model, pair_index, temps, C, thinning=a.thinning, seed=25, chain_id_base=chain_id_base,
                                   reset_mean_at_adapt_start=tempered, device=dev, **kw)
        else:
            s = SingleLevelSampler(packed, model, [pair_index[q] for q, _ in units], [temps[q] for q, _ in units], 64, thinning=a.thinning,
                                   seed=25, problem_ids=[int(q) for q, _ in units], chain_offsets=[64 * int(b) for _, b in units],
                                   reset_mean_at_adapt_start=tempered, device=dev, **kw)
        if tempered:
            s.init(np.ones(s.d), cov_identity=True, cov_scale=1.0)        # PyHillTemp.py:63,80 start
        else:
            s.init([6.0, 0.8, 8.0] if model == 2 else [6.0, 8.0], cov_identity=False, cov_scale=0.05)   # PyHillFit.py:748-751 start
        self.s = s
        self.adapt_start = s.adapt_start
        self.bytes_per_iter = float(s.Q) * s.C * 8 * (s.d + 1) / a.thinning   # 8(d+1)/thin B per iteration
        self.chains = s.Q * s.C

    def enable_moments(self):
        self.s.enable_moments(after_iteration=0)

    def reserve(self, n):
        self.s.reserve(n)

    def make_rows(self, I):
        s = self.s
        return self.torch.empty((s.rows_between(0, I), s.Q, s.d + 1, s.C), dtype=self.torch.float64, device=self.dev)

    def advance(self, I, out, join=True):
        self.s.advance(I, out=out)

    def acceptance_summary(self):
        return self.s.acceptance().mean(dim=1)


def workload_label(workload, n_pairs, C, per, model=2):
    m = "model 2 (pIC50, Hill, sigma)" if model == 2 else "model 1 (pIC50, sigma; Hill = 1: doseresponse.py:203-226)"
    if workload == "c2":
        return "Amiodarone-hERG, non-hierarchical %s, %d chains %s (BASELINE configs[1])" % (m, C, per)
    if workload == "c3":
        return "all %d Crumb drug x channel pairs, non-hierarchical %s, %d chains each %s (BASELINE configs[2])" % (n_pairs, m, C, per)
    if workload == "s3":
        return ("synthetic scaling set S3 (SURVEY 8d; pyhillfit_amd/synthetic.py, seed 12345): %d generated pairs of 3 x 4 points, non-hierarchical %s, "
                "%d chains each %s" % (n_pairs, m, C, per))
    if workload == "s3h":
        return ("synthetic scaling set S3 (SURVEY 8d), HIERARCHICAL model: %d generated pairs of 3 x 4 points, %d chains each %s — every pair has the "
                "shape the hand-allocated gfx950 kernel takes: one launch per step, no mixing of Ne groups" % (n_pairs, C, per))
    if workload == "c5":
        return "thermodynamic-integration ladder: 32 rungs x %d pairs x %d chains %s, %s (BASELINE configs[4])" % (n_pairs, C, per, m)
    return ("hierarchical model, all %d Crumb pairs, %d chains each %s (BASELINE configs[3]); one stream per Ne group, "
            "not joined between steps" % (n_pairs, C, per))


def make_batch(workload, scaling, C, a, ctx):
    """The batch of one rank.  weak: the full workload on this rank's own range of Philox chain ids.  strong: this rank's share of
    the (problem, 64-chain block) units of ONE workload (distributed.shard_blocks: LPT on block cost) — or, when the chains per
    problem are not whole blocks, of its pairs / chains."""
    dr, D, torch, dev, rank, world = ctx["dr"], ctx["D"], ctx["torch"], ctx["dev"], ctx["rank"], ctx["world"]
    all_names = [(d, c) for d in dr.drugs for c in dr.channels]
    names = [("Amiodarone", "hERG")] if workload == "c2" else all_names
    if workload in ("s3", "s3h"):
        names = [("synthetic", str(p)) for p in range(a.pairs or S3_PAIRS)]
    strong = scaling == "strong" and world > 1
    chain_id_base = 0 if scaling == "strong" else rank * C
    if workload in ("c4", "s3h"):
        units, exs = None, None
        if workload == "s3h":                                         # generated, deterministic (seed 12345): every rank makes the same set
            from pyhillfit_amd import synthetic
            exs = synthetic.generate(len(names))[0]
        if strong:
            if C % 64:
                raise SystemExit("bench.py: --scaling strong splits a batch by 64-chain blocks: --chains must be a multiple of 64")
            if exs is None:
                costs = [len(dr.concatenate_experiments(*dr.load_crumb_data(d_, c_)[::2])[0]) + 20.0 * len(dr.load_crumb_data(d_, c_)[2]) for d_, c_ in names]
            else:
                costs = [sum(len(e) for e in ex) + 20.0 * len(ex) for ex in exs]
            units = D.shard_blocks(costs, C // 64, world)[rank]
        b = HierarchicalBatch(dr, names, C, a.thinning, chain_id_base, dev, torch, units=units, experiments=exs)
    else:
        if workload == "s3":                                          # generated, deterministic: rank 0 makes it, the others get it by broadcast
            from pyhillfit_amd import synthetic
            packed = dr.PackedPoints(synthetic.single_level_pairs(synthetic.generate(len(names))[0])) if rank == 0 or ctx["backend_is_local"] else None
        else:
            packed = dr.pack_single_level(names) if rank == 0 or ctx["backend_is_local"] else None
        packed = D.broadcast_packed_points(packed, dev)               # RCCL: "scatter the dataset", a few tens of KB
        if workload == "c5":
            ladder = dr.temperature_ladder(31)                        # 32 rungs (BASELINE configs[4]); reference ladder has 41
            pair_index = [p for p in range(len(names)) for _ in ladder]
            temps = [float(t) for _ in names for t in ladder]
        else:
            pair_index, temps = list(range(len(names))), [1.0] * len(names)
        units = None
        if strong:
            if C % 64:
                raise SystemExit("bench.py: --scaling strong splits a batch by 64-chain blocks: --chains must be a multiple of 64")
            cnt = packed.counts[np.asarray(pair_index)]
            costs = 525.0 + 28.0 * cnt[:, 0] + 115.0 * (cnt[:, 1] + cnt[:, 2])     # instructions per iteration (sampler.py: launch_order)
            units = D.shard_blocks(costs, C // 64, world)[rank]
        b = SingleLevelBatch(packed, pair_index, temps, C, a, chain_id_base, dev, torch, workload == "c5", units=units)
    b.label = workload_label(workload, len(names), C, "in all, split by 64-chain blocks over the GPUs" if scaling == "strong" else "per GPU",
                             2 if workload in ("c4", "s3h") else a.model)
    if a.moments:
        b.enable_moments()
        b.label += " + on-device moments and <log L(t=1)> (the sums the command lines and compute_bayes_factors.py:11-27 need)"
    return b


def facts_key(workload, a):
    """the entry of profiles/pmc_facts.json a launch shape goes by: c2 | c3 | c4 | c5, + _model1 / _moments for the other kernels"""
    return workload + ("_model1" if a.model == 1 and workload not in ("c4", "s3h") else "") + ("_moments" if a.moments else "")


def timed_region(b, I, steps, warmup, ctx):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; time = MAX over ranks.
    Returns (seconds, kernel ms per step from HIP events on the launch stream(s), chains of all ranks)."""
    torch, dist, dev, world = ctx["torch"], ctx["dist"], ctx["dev"], ctx["world"]
    b.reserve((warmup + steps) * I)
    rows = b.make_rows(I)
    for _ in range(warmup):
        b.advance(I, out=rows)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for b_, e_ in ev:                                # events are created lazily at their first record: not inside the timed region
        b_.record(); e_.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    gc.collect(); gc.disable()                      # no collector pause between launches of the timed region
    t0 = time.perf_counter()
    if b.is_hier:                                   # four kernels per step on four streams, not joined between steps
        b.mark("start")
        for k in range(steps):
            b.advance(I, out=rows, join=False)
        b.mark("end")
        b.join()
    else:
        for k in range(steps):
            ev[k][0].record()
            b.advance(I, out=rows)
            ev[k][1].record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gc.enable()
    chains_total = float(b.chains)
    if world > 1:
        cdev = dev if ctx["backend"] == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ct = torch.tensor([chains_total], dtype=torch.float64, device=cdev)
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        chains_total = float(ct.item())
    if b.is_hier:
        kernel_ms = b.region_ms() / steps                            # HIP events on the groups' streams: longest stream / steps
    else:
        kernel_ms = float(np.mean([x.elapsed_time(y) for x, y in ev]))   # HIP events on the launch stream
    for r_ in (rows if isinstance(rows, list) else [rows]):
        assert torch.isfinite(r_[-1]).all() and torch.isfinite(r_[0]).all()
    return dt, kernel_ms, chains_total


def rooflines(workload, b, I, thinning, kernel_ms):
    alg_bytes = b.bytes_per_iter * I
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    prof = profile_facts(workload, b.chains, I, thinning)
    flop_per_iter = prof.get("flop_per_iteration")
    tflops = None if flop_per_iter is None else float(b.chains) * I * flop_per_iter / (kernel_ms * 1e-3) / 1e12
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": prof.get("traffic_bytes_per_launch"), "kernel": b.kernel_name, "kernel_ms": kernel_ms,
            "traffic_source": prof.get("source"), "algorithmic_bytes_per_launch": alg_bytes,
            "note": "fp64-VALU-bound scalar-per-chain arithmetic; the HBM fraction is reported as BASELINE asks, the binding roof is fp64_valu"}
    fp64 = {"achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": None if tflops is None else tflops / FP64_VALU_PEAK_TFLOPS,
            "measured_achievable_peak": FP64_VALU_MEASURED_TFLOPS, "flop_per_iteration": flop_per_iter, "flop_source": prof.get("source")}
    return roof, fp64


def release(torch):
    gc.collect()
    torch.cuda.empty_cache()


def main():
    a = parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
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

    from pyhillfit_amd import distributed as D
    from pyhillfit_amd import doseresponse as dr
    # ONE reader: rank 0 parses the data file, the table reaches the other ranks by broadcast (outside the timed region)
    D.setup_data_file(os.path.join(REPO, "data", "crumb_dataset.json"))
    dr.define_model(2)
    ctx = {"dr": dr, "D": D, "torch": torch, "dist": dist, "dev": dev, "rank": rank, "world": world, "backend": backend,
           "backend_is_local": world == 1}
    C = a.chains or DEFAULT_CHAINS[a.workload]
    I = a.iters_per_step or DEFAULT_ITERS[a.workload]

    # ---- the headline region: the scaling asked for (weak by default) ----
    b = make_batch(a.workload, a.scaling, C, a, ctx)
    dt, kernel_ms, chains_total = timed_region(b, I, a.steps, a.warmup, ctx)
    # "gather samples/summaries" (RCCL, outside the timed region): per-problem mean acceptance of every rank to rank 0
    acc = b.acceptance_summary().reshape(-1, 1)
    gathered = D.gather_rows(acc if backend == "nccl" or world == 1 else acc.cpu())
    out = None
    if rank == 0:
        acc_all = np.concatenate(gathered)
        roof, fp64 = rooflines(facts_key(a.workload, a), b, I, a.thinning, kernel_ms)
        out = {
            "metric": "MCMC samples/sec (whole node)", "value": chains_total * I * a.steps / dt, "unit": "MH samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "f64",
            "data": ("synthetic dose-response rows (SURVEY 8d S3, pyhillfit_amd/synthetic.py), synthetic chain batch" if a.workload in ("s3", "s3h")
                     else "real Crumb dose-response rows (data/crumb_dataset.json), synthetic chain batch"),
            "config": {"workload": b.label, "iterations_per_step": I, "thinning": a.thinning, "chains_per_gpu": b.chains,
                       "chains_all_gpus": chains_total,
                       "adaptation": "on (steps start after iteration %d > adapt_start %d)" % (a.warmup * I, b.adapt_start),
                       "mean_acceptance": float(acc_all.mean()), "problems_reporting": int(acc_all.shape[0])},
            "roofline": roof, "fp64_valu": fp64,
        }
    del b
    release(torch)

    # ---- N > 1: a second timed region in the same run with the OTHER kind of scaling.  The metric is "the full Crumb set at
    # 1/2/4/8 GPUs": `value` keeps the per-GPU work fixed (weak), `strong_value` splits ONE full-size batch over the GPUs by
    # (pair, 64-chain block) units.  (N = 1: the two coincide.) ----
    if world > 1 and not a.single_region:
        other = "strong" if a.scaling == "weak" else "weak"
        w2, C2, I2 = a.workload, C, I
        if other == "strong" and a.workload == "c3" and a.chains is None:
            # the strong-scaling batch is SURVEY 8(d)'s S3: ONE batch of 1 680 generated pairs x 4 096 chains = 107 520 blocks of 64 chains,
            # 13 440 per GPU at 8 — the Crumb batch's 13 440 blocks leave 1 680 per GPU there, one ragged round of the chip's 2 048 slots
            w2, C2, I2 = "s3", a.strong_chains or S3_STRONG_CHAINS, a.iters_per_step or DEFAULT_ITERS["s3"]
        b2 = make_batch(w2, other, C2, a, ctx)
        dt2, kms2, chains2 = timed_region(b2, I2, a.steps, a.warmup, ctx)
        I_other = I2
        if rank == 0:
            out[other + "_value"] = chains2 * I_other * a.steps / dt2
            out[other + "_region"] = {"scaling": other, "ms_per_step": dt2 / a.steps * 1e3, "kernel_ms_rank0": kms2, "chains_all_gpus": chains2,
                                      "chains_rank0": b2.chains, "workload": b2.label, "iterations_per_step": I_other}
        del b2
        release(torch)

    # ---- N = 1: the other BASELINE configurations, short regions of the same protocol, in the same JSON line ----
    if world == 1 and a.workload == "c3" and a.chains is None and a.iters_per_step is None and not a.no_other_workloads \
            and a.model == 2 and not a.moments:
        others = {}
        for name in ("c2", "c4", "c5", "c5_moments", "c3_model1", "s3", "s3h"):
            w, model_w, moments_w = OTHER_SPECS[name]
            aw = argparse.Namespace(**vars(a))
            aw.model, aw.moments = model_w, moments_w
            Cw, Iw = DEFAULT_CHAINS[w], DEFAULT_ITERS[w]
            bw = make_batch(w, "weak", Cw, aw, ctx)
            steps_w, warm_w = OTHER_STEPS[name]
            dtw, kmsw, chw = timed_region(bw, Iw, steps_w, warm_w, ctx)
            roof, fp64 = rooflines(facts_key(w, aw), bw, Iw, a.thinning, kmsw)
            others[name] = {"value": chw * Iw * steps_w / dtw, "unit": "MH samples/s", "ms_per_step": dtw / steps_w * 1e3, "kernel_ms": kmsw,
                            "steps": steps_w, "warmup": warm_w, "iterations_per_step": Iw, "chains": bw.chains, "workload": bw.label,
                            "roofline_frac": roof["frac"], "fp64_frac": fp64["frac"], "flop_per_iteration": fp64["flop_per_iteration"],
                            "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"], "traffic": roof["traffic"],
                            "kernel": bw.kernel_name, "mean_acceptance": float(bw.acceptance_summary().mean().item())}
            del bw
            release(torch)
        out["other_workloads"] = others

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
