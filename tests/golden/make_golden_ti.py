"""G6: thermodynamic integration with the reference's own pipeline — PyHillTemp.do_mcmc on every rung of the ladder
(python/PyHillTemp.py:57-125,151) followed by compute_bayes_factors.py's recipe (:11-27,67-83: per rung the mean over the
saved samples of log_data_likelihood(theta, t=1); trapezium rule over the temperatures; B12 = exp(E1 - E2)).

TEST INFRASTRUCTURE, generator side only (needs /root/reference).  2 pairs x 2 models x 41 rungs x 100 000 iterations of the
reference loop: about 10 minutes on 7 cores.
    python tests/golden/make_golden_ti.py [--iterations 100000] [--workers 7]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

PAIRS = [("Amiodarone", "hERG"), ("Quinidine", "Nav1.5-peak")]
_dr = None


def _worker(job):
    global _dr
    import _ref_loader as R
    import make_golden as G
    if _dr is None:
        _dr = R.load_doseresponse()
        _dr.setup(os.path.join(G.REF_DATA, "crumb_data.csv"))
    d, c, model, temperature, iterations = job
    pair = G.concat_pair(_dr, d, c)
    chain, _ = G.run_do_mcmc(_dr, pair, model, temperature, iterations, 5, 4, 1, False)     # leaves dr.define_model(model) set
    # compute_bayes_factors.py:14-24 on the rows do_mcmc returns (what PyHillTemp saves): likelihood at temperature 1
    ll = np.array([_dr.log_data_likelihood(pair["responses"], pair["w0"], pair["w100"], pair["wo"], pair["concs"],
                                           row[:_dr.num_params], 1, pair["pi_bit"]) for row in chain])
    nb = 20
    k = len(ll) // nb
    bm = ll[:nb * k].reshape(nb, k).mean(axis=1)
    return {"drug": d, "channel": c, "model": model, "temperature": float(temperature), "rows": int(len(ll)),
            "log_py": float(ll.sum() / len(ll)), "batch_means_se": float(bm.std(ddof=1) / np.sqrt(nb))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=100000)
    ap.add_argument("--workers", type=int, default=7)
    a = ap.parse_args()
    import _ref_loader as R
    import make_golden as G
    dr = R.load_doseresponse(); dr.setup(os.path.join(G.REF_DATA, "crumb_data.csv"))
    temps = (np.arange(dr.n + 1.) / dr.n) ** dr.c                                          # compute_bayes_factors.py:70
    jobs = [(d, c, m, t, a.iterations) for d, c in PAIRS for m in (1, 2) for t in temps]
    t0 = time.time()
    with mp.get_context("fork").Pool(a.workers) as pool:
        res = []
        for k, r in enumerate(pool.imap(_worker, jobs)):
            res.append(r)
            if k % 20 == 0:
                print(k, r, "%.0f s" % (time.time() - t0), flush=True)
    out = []
    for d, c in PAIRS:
        entry = {"drug": d, "channel": c, "temperatures": [float(t) for t in temps], "iterations": a.iterations, "models": {}}
        for m in (1, 2):
            rungs = [r for r in res if (r["drug"], r["channel"], r["model"]) == (d, c, m)]
            log_py = [r["log_py"] for r in rungs]
            entry["models"][str(m)] = {"log_py": log_py, "batch_means_se": [r["batch_means_se"] for r in rungs],
                                       "expectation": float(dr.trapezium_rule(temps, np.array(log_py)))}       # :83
        entry["B12"] = float(np.exp(entry["models"]["1"]["expectation"] - entry["models"]["2"]["expectation"]))  # :96
        out.append(entry)
        print(d, c, "E1 %.4f E2 %.4f B12 %.4f" % (entry["models"]["1"]["expectation"], entry["models"]["2"]["expectation"], entry["B12"]))
    with open(os.path.join(HERE, "g6_thermodynamic_integration.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("G6 written in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
