"""Drop-in for the sampling step of the reference's python/PyHillTemp.py (thermodynamic integration), on MI355X.

    python -m pyhillfit_amd.PyHillTemp --data-file ../data/crumb_data.csv -m 2 -d 0 -c 0
           [-i 500000] [-t 5] [-b 4] [-nc 1] [--num-chains 64] [--seed 1] [--rungs 40] [--all-pairs]

The reference runs do_mcmc(temperature) (python/PyHillTemp.py:57-125) once per rung of the ladder
t_i = (i/n)^c, n = 40, c = 3 (:151, doseresponse.py:27-28) through a process pool (:155-161).  Here all rungs
(times --num-chains chains, times all selected pairs) advance in one batch of HIP kernel launches; each rung's
chain 0 is written, burn-in removed, headerless, to the reference's temperature_<t> chain file (:165-169), where
python/compute_bayes_factors.py expects it.  Start point ones(d), identity covariance, mean reset at 1000*d
(:63,80,114-115).  `--rungs N` changes n (BASELINE config 5 uses 32 rungs = --rungs 31).  `-nc N` — the reference's pool size —
starts min(N, visible GPUs) ranks, one per GPU, which share the (pair, rung) units; or launch under torchrun."""
import argparse
import json
import sys
import time

import numpy as np

from . import chainio
from . import distributed as phfdist
from . import doseresponse as dr


def build_parser():
    parser = argparse.ArgumentParser(prog="PyHillTemp.py")
    # python/PyHillTemp.py:21-37
    parser.add_argument("-i", "--iterations", type=int, help="number of MCMC iterations", default=500000)
    parser.add_argument("-t", "--thinning", type=int, help="save every t-th iteration", default=5)
    parser.add_argument("-b", "--burn-in-fraction", type=int, help="discard the first N/b saved rows as burn-in", default=4)
    parser.add_argument("-a", "--all", action='store_true', default=False, help="accepted, unused (as in the reference)")
    parser.add_argument("-nc", "--num-cores", type=int, default=1, help="the reference's pool size over the rungs (PyHillTemp.py:25,155-159): here GPUs — N > 1 starts min(N, visible GPUs) ranks that share the (pair, rung) units")
    parser.add_argument("-Ne", "--num_expts", type=int, default=0, help="accepted, unused (as in the reference)")
    parser.add_argument("--num-APs", type=int, default=500, help="accepted, unused")
    parser.add_argument("--single", action='store_true', default=True)
    parser.add_argument("--hierarchical", action='store_true', default=False, help="accepted, unused (as in the reference)")
    parser.add_argument("--fix-hill", action='store_true', default=False, help="accepted, unused (-m selects the model)")
    parser.add_argument("-bfo", "--best-fit-only", action='store_true', default=False, help="accepted, unused")
    req = parser.add_argument_group('required arguments')
    req.add_argument("--data-file", type=str, required=True)
    req.add_argument("-m", "--model", type=int, required=True, help="1. fix Hill=1; 2. vary Hill")
    req.add_argument("-d", "--drug", type=int, help="drug index", required=True)
    req.add_argument("-c", "--channel", type=int, help="channel index", required=True)
    new = parser.add_argument_group('MI355X options')
    new.add_argument("--num-chains", type=int, default=64)
    new.add_argument("--seed", type=int, default=1, help="Philox seed (the reference seeds numpy with 1, PyHillTemp.py:16-17)")
    new.add_argument("--rungs", type=int, default=None, help="n of the ladder (i/n)^c, default doseresponse.n = 40 (41 rungs)")
    new.add_argument("--all-pairs", action='store_true', default=False, help="run every drug x channel pair instead of -d/-c")
    new.add_argument("--device", type=str, default=None)
    new.add_argument("--write-workers", type=int, default=None, help="processes formatting the chain text files (default: host cores - 1, at most 16; 0 = main process)")
    new.add_argument("--segment", type=int, default=20000)
    new.add_argument("--output-root", type=str, default="output")
    return parser


def partition_units(points_per_pair, num_rungs, world):
    """The work units of a tempered run are (pair, rung): the reference maps the RUNGS of one pair over its process pool
    (python/PyHillTemp.py:151-161), so `-d D -c C` on 8 GPUs must spread the 41 rungs, and --all-pairs spreads pairs x rungs.
    Returns, per rank, a sorted array of unit numbers u = pair * num_rungs + rung (cost of a unit ~ the pair's points)."""
    costs = np.repeat(np.asarray(points_per_pair, dtype=np.float64), num_rungs)
    return phfdist.shard_problems(costs, world)


def assemble_thermodynamic_integration(unit_rows, pairs, temperatures, model, run_facts):
    """unit_rows: [n][4 + d + 1 (+ 1)] = (pair, rung, log_py_pooled, log_py_chain0, pooled means..., and optionally the standard
    deviation of the per-chain log_py over the rung's chains) from ALL ranks, any order.
    Returns (per-rung records in (pair, rung) order, one thermodynamic-integration record per pair with all its rungs):
    what python/compute_bayes_factors.py:67-83 recomputes from the chain files."""
    unit_rows = np.asarray(unit_rows, dtype=np.float64)
    R = len(temperatures)
    n_means = (2 if model == 1 else 3) + 1
    has_sd = unit_rows.shape[1] == 4 + n_means + 1
    order = np.lexsort((unit_rows[:, 1], unit_rows[:, 0]))
    unit_rows = unit_rows[order]
    keys = unit_rows[:, 0].astype(int) * R + unit_rows[:, 1].astype(int)
    if not np.array_equal(keys, np.arange(len(pairs) * R)):
        raise RuntimeError("tempered run: %d (pair, rung) units expected, %d distinct gathered" % (len(pairs) * R, len(np.unique(keys))))
    rungs, tis = [], []
    for ip, (drug, channel) in enumerate(pairs):
        d_clean, c_clean = drug.replace('/', '_'), channel.replace('/', '_')
        rows = unit_rows[ip * R:(ip + 1) * R]
        for ir, temperature in enumerate(temperatures):
            rungs.append({"drug": d_clean, "channel": c_clean, "temperature": float(temperature), "pooled_mean": rows[ir, 4:4 + n_means].tolist(),
                          "chain_file": dr.nonhierarchical_chain_file_and_figs_dir(model, drug, channel, temperature, make_dirs=False)[2],
                          "log_py_pooled": float(rows[ir, 2]), "log_py_chain0": float(rows[ir, 3])})
        ti = {"drug": d_clean, "channel": c_clean, "model": model, "temperatures": [float(t) for t in temperatures],
              "log_py_pooled": rows[:, 2].tolist(), "log_py_chain0": rows[:, 3].tolist()}
        ti.update(run_facts)
        ti["expectation_pooled"] = float(dr.trapezium_rule(ti["temperatures"], ti["log_py_pooled"]))
        ti["expectation_chain0"] = float(dr.trapezium_rule(ti["temperatures"], ti["log_py_chain0"]))
        if has_sd:
            # Monte-Carlo error of ONE chain's estimate (what the reference computes from its one chain file per rung): the scatter of the
            # per-chain rung means over this run's chains, carried through the trapezium weights (rungs are independent runs)
            tt = np.asarray(ti["temperatures"])
            w = np.zeros(R); w[1:] += 0.5 * np.diff(tt); w[:-1] += 0.5 * np.diff(tt)
            ti["log_py_chain_sd"] = rows[:, 4 + n_means].tolist()
            ti["expectation_chain_sd"] = float(np.sqrt(np.sum((w * rows[:, 4 + n_means]) ** 2)))
        tis.append(ti)
    return rungs, tis


def run_tempered(pairs, temperatures, args, device, rank=0, world=1):
    """pairs: [(drug, channel)] of the WHOLE run; one problem per (pair, rung), this rank's share of them sampled here.
    Every rank writes the chain files of its own units; rank 0 gathers the per-unit expectations, writes one
    thermodynamic_integration.json per pair and ONE run summary, and returns the per-rung records (other ranks: [])."""
    import torch
    from .sampler import SingleLevelSampler
    model = args.model
    d = dr.num_params
    R = len(temperatures)
    loaded = []
    for drug, channel in pairs:
        num_expts, _, experiments = dr.load_crumb_data(drug, channel)
        loaded.append((drug, channel) + tuple(dr.concatenate_experiments(num_expts, experiments)))   # PyHillTemp.py:130-136
    mine = partition_units([len(c) for _, _, c, _ in loaded], R, world)[rank]
    my_pairs = sorted(set(int(u) // R for u in mine))
    local_of = {ip: k for k, ip in enumerate(my_pairs)}
    C = args.num_chains
    total_iterations, thinning = args.iterations, args.thinning
    if total_iterations % thinning:
        raise SystemExit("iterations must be a multiple of thinning")
    num_saved = total_iterations // thinning + 1                                                    # :70
    burn = num_saved // args.burn_in_fraction                                                       # :71
    unit_rows = np.zeros((len(mine), 4 + d + 1 + 1))
    mcmc_time = 0.0
    if len(mine):
        packed = dr.PackedPoints([(loaded[ip][2], loaded[ip][3]) for ip in my_pairs])
        pair_index = [local_of[int(u) // R] for u in mine]
        temps = [float(temperatures[int(u) % R]) for u in mine]
        all_pairs = [(a, b) for a in dr.drugs for b in dr.channels]
        # Philox problem ids are global — (pair's number in the data file, rung) — so a unit draws the same numbers on any rank
        pids = [all_pairs.index((loaded[int(u) // R][0], loaded[int(u) // R][1])) * 1024 + int(u) % R for u in mine]
        Q = len(mine)
        s = SingleLevelSampler(packed, model, pair_index, temps, C, thinning=thinning, seed=args.seed, adapt_start=1000 * d,
                               reset_mean_at_adapt_start=True, problem_ids=pids, device=device)        # :83,114-115
        s.init(np.ones(d), cov_identity=True, cov_scale=1.0)                                            # :63,80
        s.enable_moments(after_iteration=max(burn * thinning - 1, 0))
        s.reserve(total_iterations)
        kept = chainio.host_buffer((num_saved, Q, d + 1))   # pinned: chain 0 leaves the GPU asynchronously
        kept[0] = s.row0[:, :, 0].cpu()
        writers = chainio.WriterPool(args.write_workers if args.write_workers is not None else chainio.default_write_workers(world))
        seg = max(thinning, args.segment - args.segment % thinning)
        buf = torch.empty((seg // thinning, Q, d + 1, C), dtype=torch.float64, device=device)
        done, r = 0, 1
        start = time.time()
        while done < total_iterations:
            k = min(seg, total_iterations - done)
            nr = k // thinning
            rows = s.advance(k, out=buf[:nr])
            kept[r:r + nr].copy_(rows[:, :, :, 0], non_blocking=True)   # stream-ordered; the next segment is queued behind it at once
            done += k; r += nr
        torch.cuda.synchronize(device)
        mcmc_time = time.time() - start
        print("\nMCMC time: {} s\n".format(int(mcmc_time)))                                             # :162-163
        mean, var, _ = s.posterior_moments()
        mean = mean.cpu().numpy()
        ll1 = s.mean_log_likelihood_t1().cpu().numpy()            # [Q][C]  E_rung[log L(theta; t=1)], fused into the sampler
        for q, u in enumerate(mine):
            ip, ir = int(u) // R, int(u) % R
            drug, channel = loaded[ip][0], loaded[ip][1]
            _, _, chain_file, _ = dr.nonhierarchical_chain_file_and_figs_dir(model, drug, channel, temperatures[ir])
            print("chain_file:", chain_file)
            writers.submit(chainio.save_tempered_chain, chain_file, kept[burn:, q].numpy())         # :125,169
            unit_rows[q, :4] = ip, ir, ll1[q].mean(), ll1[q, 0]
            unit_rows[q, 4:4 + d + 1] = mean[:, q].mean(axis=1)
            unit_rows[q, 4 + d + 1] = ll1[q].std(ddof=1) if C > 1 else 0.0
        writers.close()
    # ---- the one exchange step of the run: per-unit expectations to rank 0 (a few KB over RCCL / gloo) ----
    gathered = phfdist.gather_rows(torch.as_tensor(unit_rows, device=phfdist.collective_device(device)), dst=0)
    if rank != 0:
        return []
    facts = {"chains": C, "iterations": total_iterations, "thinning": thinning, "burn_in_fraction": args.burn_in_fraction, "ranks": world}
    out, tis = assemble_thermodynamic_integration(np.concatenate(gathered), [(l[0], l[1]) for l in loaded], temperatures, model, facts)
    for (drug, channel, _, _), ti in zip(loaded, tis):
        with open(thermodynamic_integration_file(model, drug, channel), "w") as f:
            json.dump(ti, f, indent=1)
    with open(dr.output_root + "/" + dr.dir_name + "/tempered_summary_model_%d.json" % model, "w") as f:
        json.dump({"mcmc_seconds_rank0": mcmc_time, "chains": C, "ranks": world, "rungs": out}, f, indent=1)
    return out


def thermodynamic_integration_file(model, drug, channel):
    """<output>/<csv>/single-level/<drug>/<channel>/model_<m>/thermodynamic_integration.json"""
    import os
    d_clean, c_clean = drug.replace('/', '_'), channel.replace('/', '_')
    base = '{}/{}/single-level/{}/{}/model_{}/'.format(dr.output_root, dr.dir_name, d_clean, c_clean, model)
    os.makedirs(base, exist_ok=True)
    return base + "thermodynamic_integration.json"


def main(argv=None):
    parser = build_parser()
    if argv is None and len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args(argv)
    n = phfdist.ranks_for_cores(args.num_cores)                         # -nc N: the reference's pool over the rungs (:155-159) -> N ranks
    if n:
        sys.exit(phfdist.spawn_ranks("pyhillfit_amd.PyHillTemp", sys.argv[1:] if argv is None else argv, n))
    rank, local_rank, world = phfdist.init()
    try:
        device = args.device or "cuda:%d" % local_rank
        dr.define_model(args.model)                                      # PyHillTemp.py:45
        phfdist.setup_data_file(args.data_file)                          # :48
        dr.output_root = args.output_root
        if args.all_pairs:
            pairs = [(a, b) for a in dr.drugs for b in dr.channels]
        else:
            pairs = [(dr.drugs[args.drug], dr.channels[args.channel])]   # :52-53
        temperatures = dr.temperature_ladder(args.rungs)                 # :151
        print("\nDoing temperatures: {}\n".format(temperatures))
        return run_tempered(pairs, temperatures, args, device, rank, world)
    finally:
        phfdist.finalize()       # also when this rank fails: the others' next collective raises instead of hanging


if __name__ == "__main__":
    main()
