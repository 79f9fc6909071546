"""G8: the reference's own least-squares objective on a dense (pIC50, Hill) grid, every Crumb pair, both models.

The reference picks the start point of a single-level chain by minimising `sum_of_square_diffs`
(python/PyHillFit.py:93-97) with CMA-ES (:699-735; `cma` is not installed here and is stochastic anyway).  What CAN be
pinned is the objective itself and a bound on its minimum: the reference function, lifted unmodified, is evaluated on a grid
of pIC50 = -3 .. 12 (step 0.1) x Hill = 0.02 .. 20 (120 log-spaced values; model 1: Hill = 1, pIC50 step 0.01) per pair.
The fixture holds, per pair and model, the grid minimum, where it is, and the reference's value at three probe points;
tests/test_host.py requires the product's batched fit to evaluate the same objective and to reach a sum of squares no
larger than the grid minimum.

TEST INFRASTRUCTURE, generator side only (needs /root/reference; ~2 minutes).
    python tests/golden/make_golden_bestfit.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as R  # noqa: E402
import make_golden as G  # noqa: E402

P_GRID = np.round(np.arange(-3.0, 12.0001, 0.1), 10)
H_GRID = np.exp(np.linspace(np.log(0.02), np.log(20.0), 120))
P_GRID_M1 = np.round(np.arange(-3.0, 12.00001, 0.01), 10)
PROBES = [(6.0, 1.0), (4.5, 0.6), (7.25, 2.5)]


def main():
    dr = R.load_doseresponse(); dr.setup(os.path.join(G.REF_DATA, "crumb_data.csv"))
    glb = {"np": np, "dr": dr}
    R.lift_functions("PyHillFit.py", ["sum_of_square_diffs", "initial_sigma"], glb)
    ssd, initial_sigma = glb["sum_of_square_diffs"], glb["initial_sigma"]
    out = []
    for drug in dr.drugs:
        for channel in dr.channels:
            pair = G.concat_pair(dr, drug, channel)
            concs, responses = pair["concs"], pair["responses"]
            with np.errstate(all="ignore"):
                ss2 = np.array([[ssd([p, h], concs, responses) for h in H_GRID] for p in P_GRID])
                ss1 = np.array([ssd([p, 1.], concs, responses) for p in P_GRID_M1])
            i2 = np.unravel_index(np.nanargmin(ss2), ss2.shape)
            i1 = int(np.nanargmin(ss1))
            out.append({"drug": drug, "channel": channel, "n": int(len(responses)),
                        "model_2": {"grid_min_ss": float(ss2[i2]), "pic50": float(P_GRID[i2[0]]), "hill": float(H_GRID[i2[1]]),
                                    "initial_sigma": float(initial_sigma(len(responses), ss2[i2]))},
                        "model_1": {"grid_min_ss": float(ss1[i1]), "pic50": float(P_GRID_M1[i1])},
                        "probes": [[p, h, float(ssd([p, h], concs, responses))] for p, h in PROBES]})
        print("  G8", drug, out[-1]["model_2"], flush=True)
    with open(os.path.join(HERE, "g8_least_squares_grid.json"), "w") as f:
        json.dump({"p_grid": [float(P_GRID[0]), float(P_GRID[-1]), len(P_GRID)], "h_grid": [float(H_GRID[0]), float(H_GRID[-1]), len(H_GRID)],
                   "p_grid_model_1": [float(P_GRID_M1[0]), float(P_GRID_M1[-1]), len(P_GRID_M1)], "pairs": out}, f, indent=1)


if __name__ == "__main__":
    main()
