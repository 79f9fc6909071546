"""Drop-in for the reference's python/compute_bayes_factors.py (thermodynamic integration -> Bayes factor B12).

    python -m pyhillfit_amd.compute_bayes_factors --data-file ../data/crumb_data.csv -d 0 -c 0 [--from-files]

The reference re-reads every rung's chain file and calls dr.log_data_likelihood(..., t=1) once per saved sample in a
Python loop (compute_bayes_factors.py:11-27; 41 rungs x 75 001 samples x 2 models per pair).  Here that expectation
is accumulated inside the tempered sampler kernel itself (the untempered log-likelihood is a by-product of the
tempered one), so normally this script only reads `thermodynamic_integration.json` written by PyHillTemp, applies
the trapezium rule over the ladder (:83, doseresponse.py:192-193) and writes BFs/<drug>_<channel>_B12.txt (:86-100).
With --from-files (or when the JSON is missing) it does what the reference does — sweeps the chain files — but
with the HIP batch evaluator (phf_single_level_log_target) instead of the Python loop."""
import argparse
import json
import os
import sys

import numpy as np

from . import doseresponse as dr
from .PyHillTemp import thermodynamic_integration_file


def log_py_from_chain_files(model, drug, channel, temperatures, concs, responses, device):
    """compute_log_py_approxn (:11-27) for every rung: mean of log L(theta; t=1) over the rows of the rung's chain file"""
    from .sampler import log_target_batch
    packed = dr.PackedPoints([(concs, responses)])
    out = []
    for temp in temperatures:
        _, _, chain_file, _ = dr.nonhierarchical_chain_file_and_figs_dir(model, drug, channel, temp)
        chain = np.loadtxt(chain_file, usecols=range(dr.num_params))                       # :14
        chain = np.atleast_2d(chain)
        lik, _ = log_target_batch(packed, model, np.zeros(len(chain), dtype=np.int32), np.ones(len(chain)), chain, device)
        out.append(float(np.sum(lik) / len(chain)))                                         # :16-24
    return out


def main(argv=None):
    parser = argparse.ArgumentParser(prog="compute_bayes_factors.py")
    parser.add_argument("-nc", "--num-cores", type=int, default=1, help="accepted for compatibility")
    req = parser.add_argument_group('required arguments')
    req.add_argument("-d", "--drug", type=int, help="drug index", required=True)
    req.add_argument("-c", "--channel", type=int, help="channel index", required=True)
    req.add_argument("--data-file", type=str, required=True)
    new = parser.add_argument_group('MI355X options')
    new.add_argument("--from-files", action="store_true", help="sweep the chain files (reference method) instead of the fused sums")
    new.add_argument("--rungs", type=int, default=None)
    new.add_argument("--device", type=str, default="cuda:0")
    new.add_argument("--output-root", type=str, default="output")
    new.add_argument("--bf-dir", type=str, default="BFs/")
    if argv is None and len(sys.argv) == 1:
        parser.print_help(); sys.exit(1)
    args = parser.parse_args(argv)
    dr.setup(args.data_file)
    dr.output_root = args.output_root
    top_drug, top_channel = dr.drugs[args.drug], dr.channels[args.channel]                 # :50-51
    num_expts, _, experiments = dr.load_crumb_data(top_drug, top_channel)
    concs, responses = dr.concatenate_experiments(num_expts, experiments)                  # :55-59
    temps = dr.temperature_ladder(args.rungs)                                              # :70
    expectations, sources = {}, {}
    for m in (1, 2):                                                                       # :67
        dr.define_model(m)
        ti_file = thermodynamic_integration_file(m, top_drug, top_channel)
        if os.path.exists(ti_file) and not args.from_files:
            with open(ti_file) as f:
                ti = json.load(f)
            if not np.allclose(ti["temperatures"], temps):
                raise SystemExit("ladder in %s does not match --rungs" % ti_file)
            log_p_ys, sources[m] = np.array(ti["log_py_pooled"]), "fused"
        else:
            log_p_ys, sources[m] = np.array(log_py_from_chain_files(m, top_drug, top_channel, temps, concs, responses, args.device)), "chain files"
        print(log_p_ys)
        expectations[m] = dr.trapezium_rule(temps, log_p_ys)                               # :83
    print(expectations)
    drug, channel, _, _ = dr.nonhierarchical_chain_file_and_figs_dir(1, top_drug, top_channel, 1)
    if not os.path.exists(args.bf_dir):
        os.makedirs(args.bf_dir)
    bf_file = args.bf_dir + "{}_{}_B12.txt".format(drug, channel)                          # :90
    B12 = np.exp(expectations[1] - expectations[2])                                        # :94
    np.savetxt(bf_file, [B12])                                                             # :100
    return {"B12": float(B12), "expectations": expectations, "sources": sources, "file": bf_file}


if __name__ == "__main__":
    main()
