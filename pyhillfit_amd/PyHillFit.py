"""Drop-in for the sampling step of the reference's python/PyHillFit.py, on MI355X.

    python -m pyhillfit_amd.PyHillFit --data-file ../data/crumb_data.csv -m 2 -a [--hierarchical]
           [-i 500000] [-t 5] [-b 4] [-c N] [-Ne 0] [--num-APs 500] [-bfo]
           [--num-chains 64 | 128 with --hierarchical] [--seed 25] [--device cuda:0] [--save-all-chains] [--segment 20000]

Same command-line flags, same output files in the same places (python/PyHillFit.py:33-65,645-971; chain-file
contract: doseresponse.py:70-82,115-128), but every selected (drug, channel) pair is sampled AT ONCE by the HIP
kernels, `--num-chains` independent chains per pair.  What lands in the reference's chain file is chain 0 of the
pair (burn-in removed exactly like PyHillFit.py:861-864); with --save-all-chains every chain is also written to
`<chain file minus .txt>_all_chains.npy` ([rows][d+1][chains]); posterior moments of all chains, accumulated on the
device, go to `<...>_summary.json`.  The CMA-ES start point is replaced by a deterministic least-squares fit
(bestfit.py); figures are not produced (plotting is outside the sampling step).

Multi-GPU: `-c/--num-cores N` — the reference's pool size (python/PyHillFit.py:40,997-1003) — starts min(N, visible GPUs) ranks,
one per GPU (this process runs `torch.distributed.run` as a child before it touches a GPU and passes the exit code on); or
launch under torchrun yourself.  Pairs are partitioned over the ranks by cost, each rank writes the files of its pairs, rank 0
reads the data file and broadcasts it first and gathers the summaries at the end (RCCL)."""
import argparse
import itertools as it
import json
import sys
import time

import numpy as np

from . import bestfit, chainio
from . import distributed as phfdist
from . import doseresponse as dr


def build_parser():
    parser = argparse.ArgumentParser(prog="PyHillFit.py")
    # flags of the reference, python/PyHillFit.py:35-48
    parser.add_argument("-i", "--iterations", type=int, help="number of MCMC iterations", default=500000)
    parser.add_argument("-t", "--thinning", type=int, help="how often to thin the MCMC, i.e. save every t-th iteration", default=5)
    parser.add_argument("-b", "--burn-in-fraction", type=int, help="given N saved MCMC iterations, discard the first N/b as burn-in", default=4)
    parser.add_argument("-a", "--all", action='store_true', help='run MCMC on all drugs and channels', default=False)
    parser.add_argument('-ppp', '--plot-parameter-paths', action='store_true', help='accepted, ignored (no figures)', default=False)
    parser.add_argument("-c", "--num-cores", type=int, help="number of cores to parallelise drug/channel combinations: here GPUs — N > 1 starts min(N, visible GPUs) ranks, one per GPU", default=1)
    parser.add_argument("-Ne", "--num-expts", type=int, help="how many experiments to fit to", default=0)
    parser.add_argument("--num-APs", type=int, help="how many (alpha,mu) samples to take for AP simulations", default=500)
    parser.add_argument("--hierarchical", action='store_true', help="run hierarchical MCMC algorithm", default=False)
    parser.add_argument("-bfo", "--best-fit-only", action='store_true', help="only do the best fit, then quit", default=False)
    req = parser.add_argument_group('required arguments')
    req.add_argument("--data-file", type=str, help="csv file in the format of crumb_data.csv", required=True)
    req.add_argument("-m", "--model", type=int, help="For non-hierarchical: 1. fix Hill=1; 2. vary Hill", required=True)
    # new, GPU-side options (defaults keep old command lines working)
    new = parser.add_argument_group('MI355X options')
    new.add_argument("--num-chains", type=int, default=None, help="independent chains per (drug, channel) pair (default: 64; --hierarchical: 128 — "
                     "measured on the full Crumb set, 128 chains per pair take the wall time of 64: 6.24 s against 6.21, the run being the latency of "
                     "one wavefront-iteration on a chip the 420 wavefronts of 64 chains leave 59 %% idle; profiles/r05/cli_host_profile_hierarchical.txt)")
    new.add_argument("--seed", type=int, default=25, help="Philox seed (the reference seeds numpy with 25)")
    new.add_argument("--device", type=str, default=None, help="HIP device, default cuda:<LOCAL_RANK>")
    new.add_argument("--predictive-cdfs", action='store_true', default=False, help="hierarchical: also write the posterior-predictive CDFs and (Hill,pIC50) samples of construct_hierarchical_cdfs.py, accumulated on the GPU during sampling")
    new.add_argument("--cdf-chains", type=int, default=0, help="chains per pair feeding --predictive-cdfs (0 = all; 1 = chain 0 only, what the reference's script computes from the chain file)")
    new.add_argument("--write-workers", type=int, default=None, help="processes formatting the chain text files (default: this rank's host cores - 1, at most 16; 0 = write in the main process)")
    new.add_argument("--save-all-chains", action='store_true', default=False, help="also write every chain to a .npy next to the chain file")
    new.add_argument("--segment", type=int, default=20000, help="MH iterations per kernel launch")
    new.add_argument("--fused-launch", choices=["auto", "on", "off"], default="auto",
                     help="--hierarchical: the launch groups the gfx950 code object has kernels for (Ne = 3; Ne = 4 with 4 + 4 + 4 + 1 / 2 / 3 points) through "
                          "ONE persistent grid per segment instead of a launch each (auto: when the run's chains give every SIMD a wavefront); same numbers")
    new.add_argument("--output-root", type=str, default="output", help="root of the output tree (reference: ./output)")
    new.add_argument("--drugs", type=str, default=None, help="comma-separated drug names instead of -a / the menu")
    new.add_argument("--channels", type=str, default=None, help="comma-separated channel names instead of -a / the menu")
    return parser


def select_pairs(args):
    if args.drugs or args.channels:
        drugs = args.drugs.split(",") if args.drugs else list(dr.drugs)
        channels = args.channels.split(",") if args.channels else list(dr.channels)
    else:
        drugs, channels = dr.list_drug_channel_options(args.all)      # PyHillFit.py:65
    return list(it.product(drugs, channels))                          # PyHillFit.py:978


def load_single_level_pairs(pairs):
    """PyHillFit.py:653-669: skip pairs without data or with missing responses."""
    out = []
    for drug, channel in pairs:
        try:
            num_expts, _, experiments = dr.load_crumb_data(drug, channel)
            concs, responses = dr.concatenate_experiments(num_expts, experiments)
        except Exception:
            print("Problem loading data, guessing there are no entries for {} + {} --- skipping".format(drug, channel))
            continue
        if np.any(np.isnan(responses)):
            print("Skipping {} because of empty responses / missing data".format((drug, channel)))
            continue
        out.append((drug, channel, concs, responses))
    return out


def run_single_level(pairs, args, device, rank=0, world=1):
    """All pairs of this rank at once — replaces python/PyHillFit.py:645-971 run per pair."""
    import torch
    from .sampler import SingleLevelSampler
    model, temperature = args.model, 1                                 # PyHillFit.py:57
    t_begin = time.time()
    loaded = load_single_level_pairs(pairs)
    if not loaded:
        return []
    # ---- start points + best-fit files (PyHillFit.py:699-746) ----
    writers = chainio.WriterPool(args.write_workers)                   # one pool for the start-point fits and the file formatting
    fit_theta, fit_ss = bestfit.best_fit_batch([(concs, responses) for _, _, concs, responses in loaded], model)   # all pairs at once
    theta0, files = [], []
    for (drug, channel, concs, responses), th0, ss in zip(loaded, fit_theta, fit_ss):
        d_clean, c_clean, chain_file, images_dir = dr.nonhierarchical_chain_file_and_figs_dir(model, drug, channel, temperature)
        chainio.save_best_fit_params(images_dir + "{}_{}_best_fit_params.txt".format(d_clean, c_clean), th0, model)
        start0 = bestfit.chain_start(th0, model)                       # the fit, with a Hill coefficient above the prior's bound moved onto it
        if not np.array_equal(start0, th0):
            print("{} + {}: least-squares Hill {:.3g} is outside the prior's support; chains start at Hill = {:g}".format(drug, channel, th0[1], start0[1]))
        theta0.append(start0); files.append((d_clean, c_clean, chain_file))
    if args.best_fit_only:
        writers.close()
        return []
    total_iterations, thinning = args.iterations, args.thinning
    assert total_iterations % thinning == 0                            # PyHillFit.py:805
    packed = dr.PackedPoints([(c, y) for _, _, c, y in loaded])
    Q, C = len(loaded), args.num_chains
    s = SingleLevelSampler(packed, model, list(range(Q)), [1.0] * Q, C, thinning=thinning, seed=args.seed,
                           adapt_start=1000 * dr.num_params, problem_ids=[p[4] for p in pairs_with_ids(pairs, loaded)],
                           device=device)
    s.init(np.array(theta0), cov_identity=False, cov_scale=0.05)       # PyHillFit.py:748-751
    saved_iterations = total_iterations // thinning + 1                # :810
    burn = saved_iterations // args.burn_in_fraction                   # :862
    s.enable_moments(after_iteration=max(burn * thinning - 1, 0))      # moments over exactly the rows that are written
    s.reserve(total_iterations)
    keep_all = args.save_all_chains
    if keep_all:                                                       # every saved row of every chain stays in HBM until it is written
        need = saved_iterations * Q * (s.d + 1) * C * 8
        free = torch.cuda.mem_get_info(device)[0]
        if need > 0.8 * free:
            raise SystemExit("--save-all-chains needs {:.1f} GB of device memory for {} pairs x {} chains x {} saved rows, {:.1f} GB are free: "
                             "select fewer pairs (--drugs/--channels), fewer chains or a larger thinning".format(need / 1e9, Q, C, saved_iterations, free / 1e9))
    d = s.d
    kept = (torch.empty((saved_iterations, Q, d + 1, C), dtype=torch.float64, device=device) if keep_all else
            chainio.host_buffer((saved_iterations, Q, d + 1, 1)))   # pinned: chain 0 leaves the GPU asynchronously
    kept[0] = s.row0 if keep_all else s.row0[:, :, :1].cpu()
    seg = max(thinning, args.segment - args.segment % thinning)
    buf = torch.empty((seg // thinning, Q, d + 1, C), dtype=torch.float64, device=device)
    done, r = 0, 1
    torch.cuda.synchronize(device)
    start = time.time()
    while done < total_iterations:
        k = min(seg, total_iterations - done)
        nr = k // thinning
        rows = s.advance(k, out=buf[:nr])
        # stream-ordered and asynchronous: the next segment is queued behind this copy while the host moves on (a blocking copy
        # here left the GPU idle for the gather + transfer + launch latency of every segment)
        kept[r:r + nr].copy_(rows if keep_all else rows[:, :, :, :1], non_blocking=True)
        done += k; r += nr
    torch.cuda.synchronize(device)
    elapsed = time.time() - start
    mean, var, n_mom = s.posterior_moments()
    mean, var = mean.cpu().numpy(), var.cpu().numpy()
    acc = s.acceptance().cpu().numpy()
    summaries = []
    for q, (d_clean, c_clean, chain_file) in enumerate(files):
        chain0 = kept[:, q, :, 0].cpu().numpy()
        writers.submit(chainio.save_single_level_chain, chain_file, chainio.drop_burn_in(chain0, args.burn_in_fraction), d_clean, c_clean, model)
        if keep_all:
            np.save(chain_file[:-4] + "_all_chains.npy", chainio.drop_burn_in(kept[:, q].cpu().numpy(), args.burn_in_fraction))   # binary: no formatting to spread
        pooled_mean = mean[:, q].mean(axis=1)
        pooled_sd = np.sqrt(var[:, q].mean(axis=1) + mean[:, q].var(axis=1))
        summ = {"drug": d_clean, "channel": c_clean, "model": model, "chains": C, "iterations": total_iterations,
                "thinning": thinning, "saved_rows_after_burn_in": int(saved_iterations - burn),
                "columns": dr.file_labels + ["log-target"], "pooled_mean": pooled_mean.tolist(), "pooled_sd": pooled_sd.tolist(),
                "per_chain_mean_sd": mean[:, q].std(axis=1).tolist(), "acceptance": float(acc[q].mean()),
                "start_point": np.asarray(theta0[q]).tolist(), "seed": args.seed,
                "mh_samples_per_second": Q * C * total_iterations / elapsed}
        with open(chain_file[:-4] + "_summary.json", "w") as f:
            json.dump(summ, f, indent=1)
        summaries.append(summ)
        print("\n\n{} + {} complete!\n\n".format(d_clean, c_clean))      # PyHillFit.py:970
    writers.close()
    print("timing [rank {}]: data + start points {:.1f} s, sampling {:.1f} s ({} chains x {} iterations), chain files {:.1f} s".format(
        rank, start - t_begin, elapsed, Q * C, total_iterations, time.time() - start - elapsed))
    return summaries


def pairs_with_ids(all_pairs, loaded):
    """global problem number of every loaded pair = its index in the full drug x channel product of the data file
    (so a pair's Philox streams do not depend on which other pairs were selected or on the rank layout)"""
    index = {(d, c): i for i, (d, c) in enumerate(it.product(dr.drugs, dr.channels))}
    return [(d, c, cc, y, index.get((d, c), 0)) for d, c, cc, y in loaded]


def main(argv=None):
    parser = build_parser()
    if argv is None and len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args(argv)
    n = phfdist.ranks_for_cores(args.num_cores)                        # -c N: the reference's pool (PyHillFit.py:997-1003) -> N ranks
    if n:
        sys.exit(phfdist.spawn_ranks("pyhillfit_amd.PyHillFit", sys.argv[1:] if argv is None else argv, n))
    rank, local_rank, world = phfdist.init()
    try:
        return _run(args, rank, local_rank, world)
    finally:
        phfdist.finalize()       # also on an error or SystemExit of this rank: the others' next collective fails instead of hanging


DEFAULT_CHAINS, DEFAULT_CHAINS_HIERARCHICAL = 64, 128


def _run(args, rank, local_rank, world):
    device = args.device or "cuda:%d" % local_rank
    if args.num_chains is None:
        args.num_chains = DEFAULT_CHAINS_HIERARCHICAL if args.hierarchical else DEFAULT_CHAINS
    if args.write_workers is None:
        args.write_workers = chainio.default_write_workers(world)
    dr.define_model(args.model)                                        # PyHillFit.py:56
    phfdist.setup_data_file(args.data_file)                            # :61 (rank 0 reads, broadcast to the other GPUs' ranks)
    dr.output_root = args.output_root
    pairs = select_pairs(args)
    if world > 1:                                                      # partition pairs over the GPUs of the node
        costs = []
        for d, c in pairs:
            try:
                costs.append(sum(len(e) for e in dr.load_crumb_data(d, c)[2]))
            except Exception:
                costs.append(0)
        mine = phfdist.shard_problems(costs, world)[rank]
        pairs = [pairs[i] for i in mine]
    if args.hierarchical:
        from .hierarchical import run_hierarchical
        summaries = run_hierarchical(pairs, args, device, rank, world)
    else:
        summaries = run_single_level(pairs, args, device, rank, world)
    if world > 1:
        import torch
        # gather a compact numeric summary on rank 0 (pooled means of the first 3 columns + acceptance)
        rows = torch.tensor([[s_["pooled_mean"][0], s_["pooled_mean"][-1], s_["acceptance"]] for s_ in summaries] or
                            np.zeros((0, 3)), dtype=torch.float64, device=phfdist.collective_device(device)).reshape(-1, 3)
        allrows = phfdist.gather_rows(rows, dst=0)
        if rank == 0:
            print("gathered summaries from %d ranks: %d pairs" % (world, sum(len(a) for a in allrows)))
    return summaries


if __name__ == "__main__":
    main()
