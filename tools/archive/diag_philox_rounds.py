"""Diagnostic: are the posteriors sampled with Philox4x32-7 (shipped) and Philox4x32-10 (tools/_build/exp/libexp_philox10.so) the same?
All 210 pairs x 1 024 chains x 200 000 iterations (model 2), pooled posterior means per (pair, column) with their standard errors from
the scatter between chains; z = difference / sqrt(se7^2 + se10^2) over the 840 entries should be standard normal.
    python tools/diag_philox_rounds.py            (runs itself twice as a child, once per library, then compares)"""
import os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "gpurun_out")


def child(tag, lib):
    from pyhillfit_amd import _lib
    if lib != "default":
        _lib.LIB_PATH = os.path.abspath(lib)
    import torch
    from pyhillfit_amd import doseresponse as dr
    from pyhillfit_amd.sampler import SingleLevelSampler
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
    names = [(d, c) for d in dr.drugs for c in dr.channels]
    packed = dr.pack_single_level(names)
    C = 1024
    s = SingleLevelSampler(packed, 2, list(range(210)), [1.0] * 210, C, thinning=5, seed=77, reset_mean_at_adapt_start=True, device="cuda:0")
    s.init(np.ones(3), cov_identity=True, cov_scale=1.0)
    s.enable_moments(after_iteration=50000)
    s.advance(200000, save=False)
    mean, var, n = s.posterior_moments()
    m = mean.cpu().numpy()                                  # [4][210][C]
    np.savez(os.path.join(OUT, "philox_%s.npz" % tag), mean=m.mean(axis=2), se=m.std(axis=2, ddof=1) / np.sqrt(C),
             rounds=_lib.load().phf_philox_rounds())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) == 3:
        child(sys.argv[1], sys.argv[2]); sys.exit(0)
    for tag, lib in (("7", "default"), ("10", os.path.join(REPO, "tools", "_build", "exp", "libexp_philox10.so"))):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), tag, lib])
    a, b = np.load(os.path.join(OUT, "philox_7.npz")), np.load(os.path.join(OUT, "philox_10.npz"))
    assert int(a["rounds"]) == 7 and int(b["rounds"]) == 10
    z = (a["mean"] - b["mean"]) / np.sqrt(a["se"] ** 2 + b["se"] ** 2)
    lines = ["Philox4x32-7 against Philox4x32-10: 210 pairs x 1 024 chains x 200 000 iterations, model 2, pooled means of (pIC50, Hill, sigma, log-target)",
             "z over the %d entries: mean %.3f, sd %.3f, max |z| %.2f, fraction beyond 2: %.4f (standard normal: 0, 1, ~3.2 expected max, 0.0455)"
             % (z.size, z.mean(), z.std(), np.abs(z).max(), np.mean(np.abs(z) > 2)),
             "relative difference of the pooled means: median %.2e, max %.2e" % (np.median(np.abs(a["mean"] - b["mean"]) / np.abs(b["mean"])),
                                                                                np.max(np.abs(a["mean"] - b["mean"]) / np.abs(b["mean"])))]
    print("\n".join(lines))
    with open(os.path.join(OUT, "philox7_vs_10_posteriors.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
