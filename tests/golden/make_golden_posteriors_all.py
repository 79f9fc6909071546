"""G5c: long-run posteriors of the reference's own sampler (PyHillTemp.do_mcmc, model 2, temperature 1) for EVERY Crumb
pair — the reference side of BASELINE's "posterior means within 1 % of the CPU reference" on the full data set.

TEST INFRASTRUCTURE, generator side only (needs /root/reference).  210 pairs x 200 000 iterations of the reference loop
(~40 s each) spread over worker processes: about 20 minutes on 7 cores.
    python tests/golden/make_golden_posteriors_all.py [--iterations 200000] [--workers 7]
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

_dr = None


def _worker(job):
    global _dr
    import _ref_loader as R
    import make_golden as G
    if _dr is None:
        _dr = R.load_doseresponse()
        _dr.setup(os.path.join(G.REF_DATA, "crumb_data.csv"))
    d, c, model, iterations = job
    pair = G.concat_pair(_dr, d, c)
    t0 = time.time()
    chain, _ = G.run_do_mcmc(_dr, pair, model, 1.0, iterations, 5, 4, 1, False)
    nb = 20
    k = chain.shape[0] // nb
    bm = chain[:nb * k].reshape(nb, k, -1).mean(axis=1)
    return {"drug": d, "channel": c, "model": model, "temperature": 1.0, "iterations": iterations, "thinning": 5,
            "burn_in_fraction": 4, "seed": 1, "rows": int(chain.shape[0]), "mean": chain.mean(axis=0).tolist(),
            "sd": chain.std(axis=0, ddof=1).tolist(), "batch_means_se": (bm.std(axis=0, ddof=1) / np.sqrt(nb)).tolist(),
            "q50": np.quantile(chain, 0.5, axis=0).tolist(), "seconds": round(time.time() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=200000)
    ap.add_argument("--workers", type=int, default=7)
    ap.add_argument("--model", type=int, default=2)
    a = ap.parse_args()
    import _ref_loader as R
    import make_golden as G
    dr = R.load_doseresponse(); dr.setup(os.path.join(G.REF_DATA, "crumb_data.csv"))
    jobs = [(d, c, a.model, a.iterations) for d in dr.drugs for c in dr.channels]
    t0 = time.time()
    res = []
    with mp.get_context("fork").Pool(a.workers) as pool:
        for k, r in enumerate(pool.imap(_worker, jobs)):
            res.append(r)
            if k % 10 == 0:
                print(k, r["drug"], r["channel"], [round(x, 3) for x in r["mean"]], "%.0f s elapsed" % (time.time() - t0), flush=True)
    with open(os.path.join(HERE, "g5c_posteriors_all_pairs_model_%d.json" % a.model), "w") as f:
        json.dump(res, f, indent=0)
    print("G5c written:", len(res), "pairs in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
