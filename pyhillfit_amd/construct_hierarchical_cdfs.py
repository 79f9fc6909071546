"""Drop-in for the reference's python/construct_hierarchical_cdfs.py: posterior-predictive CDFs of Hill and pIC50 and
the (Hill, pIC50) samples handed to the action-potential step.

    python construct_hierarchical_cdfs.py --data-file ../data/crumb_data.csv -a [-s 500] [-Ne N] [--num-cores K]

Same inputs and outputs as the reference: reads <hierarchical chain file> columns 0..3 (alpha, beta, mu, s), drops the
first quarter (:79-86), writes .../cdfs/*_posterior_predictive_{hill,pic50}_cdf.txt (:128-131) and
.../posterior_predictive_hill_pic50_samples/*_hill_pic50_samples.txt (:139-149).  The average over the samples — a
Python loop with four scipy.stats calls per row in the reference (:47-57) — runs on the GPU for all pairs at once
(pyhillfit_amd/predictive.py).  No plots are made (-np is accepted and is the only mode).  `PyHillFit.py --hierarchical
--predictive-cdfs` writes the same files during sampling, from all chains, without reading anything back."""
import argparse
import itertools as it
import sys
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp

import numpy as np

from . import chainio
from . import doseresponse as dr


def build_parser():
    parser = argparse.ArgumentParser(prog="construct_hierarchical_cdfs.py")                       # flags: :18-26
    parser.add_argument("-s", "--samples", type=int, help="number of Hill and pIC50 samples for use in AP model", default=500)
    parser.add_argument("-a", "--all", action='store_true', help='construct posterior predictive CDFs for Hill and pIC50 for all drugs and channels', default=False)
    parser.add_argument("--num-cores", type=int, help="processes reading the chain files", default=1)
    parser.add_argument("-np", "--no-plots", action='store_true', help="accepted for compatibility: plots are never made", default=False)
    parser.add_argument("-tu", "--top-up", action='store_true', help="accepted for compatibility", default=False)
    parser.add_argument("-sy", "--synthetic", action='store_true', help="accepted for compatibility", default=False)
    parser.add_argument("-Ne", "--num_expts", type=int, help="how many experiments to fit to", default=0)
    parser.add_argument("--data-file", type=str, required=True, help="csv file from which to read in data, in same format as provided crumb_data.csv")
    new = parser.add_argument_group('MI355X options')
    new.add_argument("--device", type=str, default="cuda:0")
    new.add_argument("--output-root", type=str, default="output")
    new.add_argument("--seed", type=int, default=1, help="numpy seed of the inverse-CDF draws (the reference seeds 1, :12-13)")
    new.add_argument("--write-workers", type=int, default=0)
    return parser


def read_post_burn_hyperparameters(chain_file):
    """:79-86 — columns (alpha, beta, mu, s) of the chain file, first quarter dropped; None when the file is missing"""
    try:
        mcmc = chainio.load_chain(chain_file, usecols=range(4))
    except (IOError, OSError):
        return None
    return mcmc[mcmc.shape[0] // 4:, :]


def main(argv=None):
    parser = build_parser()
    if argv is None and len(sys.argv) == 1:
        parser.print_help(); sys.exit(1)
    args = parser.parse_args(argv)
    dr.setup(args.data_file)                                                                      # :30
    dr.output_root = args.output_root
    drugs_to_run, channels_to_run = dr.list_drug_channel_options(args.all)                        # :32
    jobs = []
    for drug, channel in it.product(drugs_to_run, channels_to_run):                               # :151
        try:
            num_expts, _, _ = dr.load_crumb_data(drug, channel)                                   # :66
        except Exception as e:                                                                    # :160-164
            print(e); print("Failed to run {} + {}!".format(drug, channel)); continue
        if 0 < args.num_expts < num_expts:                                                        # :67-69
            num_expts, save_samples = args.num_expts, False
        else:
            save_samples = True
        d_clean, c_clean, _, _, _, chain_file = dr.hierarchical_output_dirs_and_chain_file(drug, channel, num_expts)
        jobs.append((d_clean, c_clean, num_expts, save_samples, chain_file))
    files = [j[4] for j in jobs]
    if args.num_cores > 1 and len(files) > 1:
        with ProcessPoolExecutor(min(args.num_cores, len(files)), mp_context=mp.get_context("spawn")) as pool:
            chains = list(pool.map(read_post_burn_hyperparameters, files))
    else:
        chains = [read_post_burn_hyperparameters(f) for f in files]
    have = [i for i, c in enumerate(chains) if c is not None]
    for i, c in enumerate(chains):
        if c is None:
            print("tried loading", files[i])                                                      # :81-83
            print("No MCMC file found for {} + {}\n".format(jobs[i][0], jobs[i][1]))
    from .predictive import curves_from_chains, save_cdfs_and_samples
    results = curves_from_chains([chains[i] for i in have], args.device)
    rng = np.random.RandomState(args.seed)
    done = []
    with chainio.WriterPool(args.write_workers) as writers:
        for i, res in zip(have, results):                                                         # pair order = the reference's
            d_clean, c_clean, num_expts, save_samples, _ = jobs[i]
            save_cdfs_and_samples(writers, d_clean, c_clean, num_expts, res, args.samples, rng, save_samples)
            print("\n{} + {} done!\n".format(d_clean, c_clean))                                   # :152
            done.append((d_clean, c_clean))
    return done


if __name__ == "__main__":
    main()
