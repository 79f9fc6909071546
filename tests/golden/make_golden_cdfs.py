"""G7: posterior-predictive CDFs/PDFs of Hill and pIC50 from the reference's own function.

TEST INFRASTRUCTURE, generator side only (needs /root/reference; never runs on the GPU box).  Executes
construct_posterior_predictive_cdfs (python/construct_hierarchical_cdfs.py:32-58), lifted unmodified from the
lib2to3-converted script, on seeded (alpha, beta, mu, s) samples and stores inputs + outputs in g7_predictive_cdfs.npz.

    python tests/golden/make_golden_cdfs.py
"""
import os
import sys

import numpy as np
import scipy
import scipy.stats as st

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader as R  # noqa: E402


def sample_sets():
    rng = np.random.RandomState(20240607)
    sets = {}
    # a: posterior-like cloud (Amiodarone-hERG scale: alpha ~0.7, mu ~6)
    n = 400
    sets["posterior_like"] = np.column_stack([np.abs(0.72 + 0.17 * rng.standard_normal(n)) + 0.05,
                                              2.0 + rng.gamma(2.5, 1.4, n) + 1e-3,
                                              6.0 + 0.25 * rng.standard_normal(n),
                                              0.01 + rng.gamma(2.5, 0.09, n)])
    # b: wide prior-like draws, incl. rows at the edges of the support (beta -> 2, s -> 0.01, tiny/large alpha)
    n = 250
    wide = np.column_stack([rng.gamma(5, 0.32, n) + 1e-3, 2.0 + rng.gamma(2.5, 1.5, n),
                            -4.0 + rng.gamma(7.5, 1.56, n), 0.01 + rng.gamma(2.5, 0.09, n)])
    edges = np.array([[0.02, 2.0001, -1.9, 0.0101], [3.9, 14.0, 11.9, 0.011], [1.0, 2.5, 5.0, 3.0],
                      [0.4, 40.0, 0.0, 0.02], [2.0, 3.0, 12.5, 0.5], [1e-3, 2.2, -3.0, 0.05]])
    sets["wide_and_edges"] = np.vstack([wide, edges])
    sets["single_sample"] = np.array([[1.3, 4.2, 6.15, 0.146]])
    return sets


def main():
    glb = {"np": np, "st": st}
    R.lift_functions("construct_hierarchical_cdfs.py", ["construct_posterior_predictive_cdfs"], glb)
    fn = glb["construct_posterior_predictive_cdfs"]
    out = {}
    for name, s in sample_sets().items():
        hx, hcdf, px, pcdf, hpdf, ppdf = fn(s[:, 0], s[:, 1], s[:, 2], s[:, 3])
        out[name + "_samples"] = s
        out[name + "_hill_x"], out[name + "_pic50_x"] = hx, px
        out[name + "_hill_cdf"], out[name + "_pic50_cdf"] = hcdf, pcdf
        out[name + "_hill_pdf"], out[name + "_pic50_pdf"] = hpdf, ppdf
        print(name, s.shape, hcdf[[0, 100, 500]], pcdf[[0, 250, 500]], hpdf[[0, 1]], flush=True)
    out["versions"] = np.array(["numpy " + np.__version__, "scipy " + scipy.__version__])
    np.savez_compressed(os.path.join(HERE, "g7_predictive_cdfs.npz"), **out)
    print("G7 written")


if __name__ == "__main__":
    main()
