"""Posterior-predictive curves on the GPU (-m gpu): phf_predictive_accumulate against the reference's own function
(golden G7), bit-identical to the CPU twin for every launch shape, and the construct_hierarchical_cdfs command line /
the fused --predictive-cdfs path writing the reference's files."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SETS = ["posterior_like", "wide_and_edges", "single_sample"]
CURVES = ["hill_cdf", "pic50_cdf", "hill_pdf", "pic50_pdf"]
RTOL, ATOL = 1e-11, 1e-70      # vs scipy: ~1e-14 observed; the +170 cap of the exponents moves values only below 1.5e-74


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


@pytest.fixture(scope="module")
def g7():
    return np.load(os.path.join(GOLDEN, "g7_predictive_cdfs.npz"))


@pytest.mark.parametrize("name", SETS)
def test_curves_match_reference_function(name, gpu, g7):
    from pyhillfit_amd.predictive import curves_from_chains, predictive_grids
    s = g7[name + "_samples"]
    hx, hc, px, pc, hp, pp = curves_from_chains([s], gpu)[0]
    gx, gp = predictive_grids()
    assert np.array_equal(hx, g7[name + "_hill_x"]) and np.array_equal(px, g7[name + "_pic50_x"])
    assert np.array_equal(hx, gx) and np.array_equal(px, gp)
    for got, k in ((hc, "hill_cdf"), (pc, "pic50_cdf"), (hp, "hill_pdf"), (pp, "pic50_pdf")):
        ref = g7[name + "_" + k]
        assert np.all(np.abs(got - ref) <= ATOL + RTOL * np.abs(ref)), (name, k, np.max(np.abs(got - ref)))
    assert hc[0] == 0.0 and hp[0] == 0.0                                  # x = 0 on the Hill axis: exactly scipy's zeros
    assert np.all(np.diff(hc) >= -4e-16) and np.all(np.diff(pc) >= -4e-16) and hc[-1] <= 1 and pc[-1] <= 1   # monotone to an ulp of 1


@pytest.mark.parametrize("shape", [
    dict(nr=37, Q=3, rs=12, C=5, used=5, G=501, chunk=64),       # sampler-shaped rows (dim+1 columns), ragged last chunk
    dict(nr=130, Q=2, rs=4, C=64, used=17, G=501, chunk=4096),   # a subset of the chains, one chunk
    dict(nr=9, Q=1, rs=16, C=7, used=7, G=1100, chunk=10),       # three grid tiles, many small chunks
    dict(nr=700, Q=2, rs=4, C=1, used=1, G=37, chunk=512),       # tile boundary inside a chunk (PHF_PRED_TILE 512)
])
def test_bit_identical_to_cpu_twin_for_any_launch_shape(shape, gpu):
    from oracle import c_oracle as co
    from pyhillfit_amd.predictive import PredictiveCurves
    rng = np.random.RandomState(7)
    nr, Q, rs, C, used, G, chunk = (shape[k] for k in ("nr", "Q", "rs", "C", "used", "G", "chunk"))
    rows = rng.standard_normal((nr, Q, rs, C))
    rows[:, :, 0] = 0.05 + rng.gamma(5, 0.3, (nr, Q, C))
    rows[:, :, 1] = 2.0 + rng.gamma(2.5, 1.5, (nr, Q, C))
    rows[:, :, 2] = -4.0 + rng.gamma(7.5, 1.5, (nr, Q, C))
    rows[:, :, 3] = 0.01 + rng.gamma(2.5, 0.09, (nr, Q, C))
    hill_x = np.linspace(0., 4., G); pic50_x = np.linspace(-2., 12., G)
    pc = PredictiveCurves(Q, gpu, hill_x, pic50_x, chunk=chunk)
    # two segments, as the sampler delivers them
    cut = nr // 3
    dev = torch.from_numpy(rows).to(gpu)
    pc.accumulate(dev[:cut].contiguous(), used)
    pc.accumulate(dev[cut:].contiguous(), used)
    twin = co.predictive_accumulate(rows[:cut], used, hill_x, pic50_x, chunk)
    twin = co.predictive_accumulate(rows[cut:], used, hill_x, pic50_x, chunk, sums=twin)
    got = pc.sums.cpu().numpy()
    assert pc.count == nr * used
    assert np.array_equal(got, twin)
    assert np.isfinite(got).all()


def test_argument_errors_are_reported(gpu):
    from pyhillfit_amd import _lib
    from pyhillfit_amd.predictive import PredictiveCurves
    pc = PredictiveCurves(2, gpu)
    with pytest.raises(ValueError):
        pc.accumulate(torch.zeros((4, 3, 4, 1), dtype=torch.float64, device=gpu))           # wrong number of problems
    with pytest.raises(_lib.PhfError):
        pc.accumulate(torch.ones((4, 2, 4, 2), dtype=torch.float64, device=gpu), chains_used=3)
    with pytest.raises(_lib.PhfError):
        pc.accumulate(torch.ones((4, 2, 3, 2), dtype=torch.float64, device=gpu))           # fewer than 4 columns
    with pytest.raises(ValueError):
        pc.means()


def test_cdf_command_line_and_fused_path_write_the_reference_files(gpu, tmp_path):
    """PyHillFit --hierarchical --predictive-cdfs --cdf-chains 1 accumulates during sampling what
    construct_hierarchical_cdfs.py computes afterwards from the chain file (chain 0, first quarter dropped)"""
    from oracle import pyhillfit_oracle as orc
    from pyhillfit_amd import PyHillFit, construct_hierarchical_cdfs
    from pyhillfit_amd import doseresponse as dr
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    csv = str(tmp_path / "crumb_data.csv")
    dr.table.to_csv(csv)
    out = str(tmp_path / "output")
    T = 8000
    PyHillFit.main(["--data-file", csv, "-m", "2", "--hierarchical", "-i", str(T), "-t", "5", "--drugs", "Amiodarone,Dofetilide",
                    "--channels", "hERG,Kv4.3", "--num-chains", "8", "--output-root", out, "--num-APs", "50", "--segment", "1500",
                    "--predictive-cdfs", "--cdf-chains", "1", "--write-workers", "2"])
    base = os.path.join(out, "crumb_data", "hierarchical")
    fused = {}
    for drug, ch in (("Amiodarone", "hERG"), ("Dofetilide", "hERG"), ("Amiodarone", "Kv4.3"), ("Dofetilide", "Kv4.3")):
        ne = dr.load_crumb_data(drug, ch)[0]
        cdir = os.path.join(base, drug, ch, "%d_expts" % ne, "cdfs")
        fused[(drug, ch)] = (np.loadtxt(os.path.join(cdir, "crumb_data_%s_%s_posterior_predictive_hill_cdf.txt" % (drug, ch))),
                             np.loadtxt(os.path.join(cdir, "crumb_data_%s_%s_posterior_predictive_pic50_cdf.txt" % (drug, ch))))
        smp = np.loadtxt(os.path.join(base, "posterior_predictive_hill_pic50_samples", "crumb_data_%s_%s_hill_pic50_samples.txt" % (drug, ch)))
        assert smp.shape == (50, 2) and np.all((smp[:, 0] >= 0) & (smp[:, 0] <= 4)) and np.all((smp[:, 1] >= -2) & (smp[:, 1] <= 12))
        os.remove(os.path.join(cdir, "crumb_data_%s_%s_posterior_predictive_hill_cdf.txt" % (drug, ch)))
    done = construct_hierarchical_cdfs.main(["--data-file", csv, "-a", "-s", "50", "--output-root", out, "--device", gpu, "--num-cores", "2"])
    assert set(fused) <= set(done) and len(done) == 4              # pairs without a chain file are reported and skipped
    for (drug, ch), (hill_f, pic50_f) in fused.items():
        ne = dr.load_crumb_data(drug, ch)[0]
        cdir = os.path.join(base, drug, ch, "%d_expts" % ne, "cdfs")
        hill = np.loadtxt(os.path.join(cdir, "crumb_data_%s_%s_posterior_predictive_hill_cdf.txt" % (drug, ch)))
        pic50 = np.loadtxt(os.path.join(cdir, "crumb_data_%s_%s_posterior_predictive_pic50_cdf.txt" % (drug, ch)))
        assert hill.shape == (501, 2) and pic50.shape == (501, 2)
        assert np.allclose(hill, hill_f, rtol=1e-12, atol=1e-70) and np.allclose(pic50, pic50_f, rtol=1e-12, atol=1e-70)
        # and both equal the reference's arithmetic (restated with scipy) on the rows of the chain file
        chain = np.loadtxt(os.path.join(base, drug, ch, "%d_expts" % ne, "chain", "crumb_data_%s_%s_hierarchical_chain.txt" % (drug, ch)),
                           usecols=range(4))
        chain = chain[chain.shape[0] // 4:]
        _, hc, _, pc, _, _ = orc.predictive_cdfs(chain[:, 0], chain[:, 1], chain[:, 2], chain[:, 3])
        assert np.allclose(hill[:, 1], hc, rtol=1e-11, atol=1e-70) and np.allclose(pic50[:, 1], pc, rtol=1e-11, atol=1e-70)
