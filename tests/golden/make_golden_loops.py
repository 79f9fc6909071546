"""G9: draw-by-draw traces of the reference's OTHER two Metropolis loops, executed from the reference's own statements.

Golden G3 pins the tempered loop (python/PyHillTemp.py:do_mcmc, a function).  The single-level loop of PyHillFit.py
(:748-751 initial covariance 0.05 diag|theta0|, :787-864 loop, thinning, burn-in: no mean reset, start = best fit) and the
hierarchical loop (:431-511: covariance 0.01 diag|theta0|, adaptation after 100 dim) are INLINED in run_single_level /
run_hierarchical between a CMA-ES search (`cma` is not installed) and matplotlib code, so the functions cannot be called —
but their statements can be lifted as they stand (tests/golden/_ref_loader.lift_statements) and executed in a namespace
that supplies what the skipped parts would have produced (theta_cur / first_iteration, the data, args).  npr is the
recording stand-in of make_golden.py, so every proposal and uniform is captured.

TEST INFRASTRUCTURE, generator side only (needs /root/reference; seconds).
    python tests/golden/make_golden_loops.py
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import numpy.random as npr

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as R  # noqa: E402
import make_golden as G  # noqa: E402


class Args(object):
    pass


def run_single_level_loop(dr, pair, model, theta0, iterations, thinning, burn_fraction):
    dr.define_model(model)
    args = Args()
    args.iterations = R.Py2Int(iterations); args.thinning = R.Py2Int(thinning); args.burn_in_fraction = R.Py2Int(burn_fraction)
    rec = G.Recorder()
    rec.seed = npr.seed                                  # the loop seeds the global generator itself (:824-825)
    glb = {"np": np, "npr": rec, "dr": dr, "time": time, "args": args, "num_params": dr.num_params,
           "responses": pair["responses"], "where_r_0": pair["w0"], "where_r_100": pair["w100"], "where_r_other": pair["wo"],
           "concs": pair["concs"], "pi_bit": pair["pi_bit"], "temperature": 1, "theta_cur": np.array(theta0, dtype=float)}
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        lines = R.lift_statements("PyHillFit.py", "run_single_level", [
            (lambda t: t.startswith("proposal_scale ="), lambda t: t.startswith("cov_estimate =")),
            (lambda t: t.startswith("when_to_adapt ="), lambda t: t.startswith("chain = chain[burn:")),
        ], glb)
    return glb, rec, lines


def run_hierarchical_loop(pair, first_iteration, iterations, thinning, seed):
    import scipy.stats as st
    dr_glb = {"np": np, "sys": sys, "st": st, "pic50_prior": [-2.]}
    dr = R.load_doseresponse()
    dr_glb["dr"] = dr
    R.lift_functions("PyHillFit.py", ["log_data_likelihood", "log_hill_i_log_logistic_likelihood", "log_pic50_i_logistic_likelihood",
                                      "log_target_distribution"], dr_glb)
    shapes, scales, locs, _ = G.elkins_prior_params()
    args = Args()
    args.iterations = R.Py2Int(iterations); args.thinning = R.Py2Int(thinning)
    rec = G.Recorder()
    glb = dict(dr_glb)
    glb.update({"npr": rec, "time": time, "args": args, "experiments": pair["experiments"], "shapes": shapes, "scales": scales,
                "locs": locs, "first_iteration": np.array(first_iteration, dtype=float)})
    npr.seed(seed)                                       # the reference does not seed this loop (seed = 1 at :225 is unused)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        lines = R.lift_statements("PyHillFit.py", "run_hierarchical", [
            (lambda t: t.startswith("first_cov ="), lambda t: t.startswith("while t <= total_iterations")),
        ], glb)
    return glb, rec, lines


SL_RUNS = [  # (name, drug, channel, model, theta0, T, thinning, burn_fraction)
    ("sl_amio_m2", "Amiodarone", "hERG", 2, [6.0, 0.8, 8.0], 6000, 1, 10 ** 9),
    ("sl_amio_m1", "Amiodarone", "hERG", 1, [6.0, 8.0], 5000, 1, 10 ** 9),
    ("sl_moxi_m2", "Moxifloxacin", "KvLQT1/mink", 2, [4.0, 1.0, 6.0], 6000, 1, 10 ** 9),
    ("sl_amio_m2_thin5_burn4", "Amiodarone", "hERG", 2, [6.0, 0.8, 8.0], 5000, 5, 4),
]
HIER_RUNS = [  # (name, drug, channel, first_iteration builder, T, thinning)
    ("hier_amio", "Amiodarone", "hERG", lambda ne: np.concatenate([[1., 5., 6., .3], np.tile([6.0, 0.8], ne), [8.0]]), 3000, 1),
    ("hier_amit", "Amitriptyline", "Kv4.3", lambda ne: np.concatenate([[1.2, 4., 5., .4], np.tile([4.5, 1.1], ne), [5.0]]), 3500, 1),
    ("hier_amio_thin5", "Amiodarone", "hERG", lambda ne: np.concatenate([[1., 5., 6., .3], np.tile([6.0, 0.8], ne), [8.0]]), 2000, 5),
]


def main():
    dr = R.load_doseresponse(); dr.setup(os.path.join(G.REF_DATA, "crumb_data.csv"))
    out, meta = {}, {"single_level": [], "hierarchical": []}
    for name, d, c, model, th0, T, thin, burn in SL_RUNS:
        pair = G.concat_pair(dr, d, c)
        glb, rec, lines = run_single_level_loop(dr, pair, model, th0, T, thin, burn)
        chain = np.array(glb["chain"])
        out[name + "_chain"] = chain; out[name + "_star"] = np.array(rec.stars); out[name + "_u"] = np.array(rec.us)
        covs = np.array(rec.covs)
        out[name + "_cov_every20"] = covs[::20]; out[name + "_cov_last"] = covs[-1]
        out[name + "_final"] = np.concatenate([np.ravel(glb["cov_estimate"]), np.ravel(glb["mean_estimate"]), [glb["loga"], glb["acceptance"]]])
        meta["single_level"].append({"name": name, "drug": d, "channel": c, "model": model, "theta0": th0, "iterations": T, "thinning": thin,
                                     "burn_in_fraction": burn, "seed": 25, "reference_lines": lines, "when_to_adapt": int(glb["when_to_adapt"]),
                                     "rows": int(chain.shape[0]), "accepted": int((np.diff(np.array(rec.stars)[:, 0]) != 0).sum())})
        print("  G9", name, chain.shape, "lines", lines, "acceptance %.3f" % glb["acceptance"], flush=True)
    for name, d, c, build, T, thin in HIER_RUNS:
        pair = G.concat_pair(dr, d, c)
        th0 = build(len(pair["experiments"]))
        glb, rec, lines = run_hierarchical_loop(pair, th0, T, thin, 1)
        chain = np.array(glb["chain"])
        out[name + "_chain"] = chain; out[name + "_star"] = np.array(rec.stars); out[name + "_u"] = np.array(rec.us)
        covs = np.array(rec.covs)
        out[name + "_cov_every50"] = covs[::50]; out[name + "_cov_last"] = covs[-1]
        out[name + "_final"] = np.concatenate([np.ravel(glb["cov_cur"]), np.ravel(glb["mean_estimate"]), [glb["loga"], glb["acceptance"]]])
        meta["hierarchical"].append({"name": name, "drug": d, "channel": c, "Ne": len(pair["experiments"]), "theta0": th0.tolist(), "iterations": T,
                                     "thinning": thin, "seed": 1, "reference_lines": lines, "when_to_adapt": int(glb["when_to_adapt"]),
                                     "burn": int(glb["burn"]), "saved_iterations": int(glb["saved_iterations"]), "rows": int(chain.shape[0])})
        print("  G9", name, chain.shape, "lines", lines, "acceptance %.3f" % glb["acceptance"], flush=True)
    np.savez_compressed(os.path.join(HERE, "g9_loop_traces.npz"), **out)
    with open(os.path.join(HERE, "g9_loop_traces_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
