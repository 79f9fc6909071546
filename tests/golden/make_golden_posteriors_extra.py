"""G5b: long-run posteriors of the reference's own sampler (PyHillTemp.do_mcmc) for pairs of other shapes than G5's —
20 points with both kinds of censoring, the pair with the out-of-range response, a 6-point pair, an all-zero pair
(posterior = prior cut by the data), model 1 with a saturated point.

TEST INFRASTRUCTURE, generator side only (needs /root/reference; minutes of reference MCMC).
    python tests/golden/make_golden_posteriors_extra.py [--iterations 200000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as R  # noqa: E402
import make_golden as G  # noqa: E402

RUNS = [("Moxifloxacin", "KvLQT1/mink", 2, 1.0), ("Amitriptyline", "Kv4.3", 2, 1.0), ("Rufinamide", "hERG", 2, 1.0),
        ("Lidocaine", "KvLQT1/mink", 2, 1.0), ("Quinidine", "hERG", 1, 1.0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=200000)
    a = ap.parse_args()
    dr = R.load_doseresponse(); dr.setup(os.path.join(G.REF_DATA, "crumb_data.csv"))
    res = []
    for d, c, model, t in RUNS:
        pair = G.concat_pair(dr, d, c)
        t0 = time.time()
        chain, _ = G.run_do_mcmc(dr, pair, model, t, a.iterations, 5, 4, 1, False)
        # batch means (20 batches) give an honest standard error for a single autocorrelated chain
        nb = 20
        k = chain.shape[0] // nb
        bm = chain[:nb * k].reshape(nb, k, -1).mean(axis=1)
        res.append({"drug": d, "channel": c, "model": model, "temperature": t, "iterations": a.iterations, "thinning": 5,
                    "burn_in_fraction": 4, "seed": 1, "rows": int(chain.shape[0]), "mean": chain.mean(axis=0).tolist(),
                    "sd": chain.std(axis=0, ddof=1).tolist(), "batch_means_se": (bm.std(axis=0, ddof=1) / np.sqrt(nb)).tolist(),
                    "q05": np.quantile(chain, 0.05, axis=0).tolist(), "q50": np.quantile(chain, 0.5, axis=0).tolist(),
                    "q95": np.quantile(chain, 0.95, axis=0).tolist(), "seconds": round(time.time() - t0, 1)})
        print("  G5b", d, c, model, res[-1]["mean"], res[-1]["batch_means_se"], res[-1]["seconds"], "s", flush=True)
    with open(os.path.join(HERE, "g5b_posteriors.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
