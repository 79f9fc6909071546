"""The drop-in command lines on a GPU: same flags, same files in the same places as the reference
(python/PyHillFit.py, python/PyHillTemp.py), chain 0 bit-identical to the CPU twin through the text file."""
import json
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def csv_file(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pyhillfit_amd import doseresponse as dr
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    p = tmp_path_factory.mktemp("data") / "crumb_data.csv"
    dr.table.to_csv(str(p))
    return str(p)


def test_pyhillfit_single_level_cli(csv_file, tmp_path):
    from oracle import c_oracle as co
    from pyhillfit_amd import PyHillFit, bestfit
    from pyhillfit_amd import doseresponse as dr
    from pyhillfit_amd.sampler import gamma_table
    out = str(tmp_path / "output")
    T, thin = 4000, 5
    summ = PyHillFit.main(["--data-file", csv_file, "-m", "2", "-i", str(T), "-t", str(thin), "-b", "4", "--drugs", "Amiodarone,Bepridil",
                           "--channels", "hERG,KvLQT1/mink", "--num-chains", "64", "--output-root", out, "--segment", "1500",
                           "--save-all-chains"])
    assert len(summ) == 4
    all_pairs = [(a, b) for a in dr.drugs for b in dr.channels]
    for drug, channel in [("Amiodarone", "hERG"), ("Bepridil", "KvLQT1/mink")]:
        cc = channel.replace("/", "_")
        base = os.path.join(out, "crumb_data", "single-level", drug, cc, "model_2", "temperature_1")
        chain_file = os.path.join(base, "chain", "%s_%s_model_2_temp_1_chain_single-level.txt" % (drug, cc))
        with open(chain_file) as f:
            assert f.readline().startswith("# Nonhierarchical MCMC output for %s + %s" % (drug, cc))
        chain = np.loadtxt(chain_file)
        saved = T // thin + 1
        assert chain.shape == (saved - saved // 4, 4)                       # PyHillFit.py:810,861-864
        best = np.loadtxt(os.path.join(base, "figures", "%s_%s_best_fit_params.txt" % (drug, cc)))
        ne, _, ex = dr.load_crumb_data(drug, channel)
        concs, y = dr.concatenate_experiments(ne, ex)
        th0 = bestfit.best_fit_batch([(concs, y)], 2)[0][0]                # a pair's fit does not depend on its batch
        assert np.array_equal(best, th0)
        assert bestfit.best_fit(concs, y, 2)[0] == pytest.approx(th0, rel=1e-5)   # the scalar Nelder-Mead fit it replaces
        # chain 0 == CPU twin started at the same point, same Philox stream (problem id = index in drugs x channels)
        pk = co.PackedPair(concs, y, 2, 1.0)
        st = pk.init_state(th0, False, 0.05)
        rows = pk.advance(st, 0, T, thin, 3000, False, gamma_table(T), seed=25, chain_id=0, problem_id=all_pairs.index((drug, channel)))
        full = np.vstack([np.concatenate([th0, [pk.log_target(th0)]]), rows])
        assert np.array_equal(chain, full[saved // 4:])
        allc = np.load(chain_file[:-4] + "_all_chains.npy")
        assert allc.shape == (saved - saved // 4, 4, 64) and np.array_equal(allc[:, :, 0], chain)
        with open(chain_file[:-4] + "_summary.json") as f:
            s = json.load(f)
        assert s["chains"] == 64 and 0.05 < s["acceptance"] < 0.8
        post = allc[:, :, :].transpose(0, 2, 1).reshape(-1, 4)
        np.testing.assert_allclose(s["pooled_mean"], post.mean(axis=0), rtol=1e-9)   # device moments == file contents


def test_pyhilltemp_cli(csv_file, tmp_path):
    from pyhillfit_amd import PyHillTemp
    from pyhillfit_amd import doseresponse as dr
    out = str(tmp_path / "output")
    res = PyHillTemp.main(["--data-file", csv_file, "-m", "1", "-d", "0", "-c", "0", "-i", "3000", "-t", "5", "--rungs", "4",
                           "--num-chains", "64", "--output-root", out])
    assert len(res) == 5
    lad = dr.temperature_ladder(4)
    assert [dr.py2_str(t) for t in lad] == ["0.0", "0.015625", "0.125", "0.421875", "1.0"]
    for t in lad:
        f = os.path.join(out, "crumb_data", "single-level", "Amiodarone", "hERG", "model_1", "temperature_%s" % dr.py2_str(t), "chain",
                         "Amiodarone_hERG_model_1_temp_%s_chain_single-level.txt" % dr.py2_str(t))
        chain = np.loadtxt(f)                                                 # headerless, PyHillTemp.py:169
        assert chain.shape == (601 - 601 // 4, 3) and np.isfinite(chain).all()
    # the likelihood weight grows with temperature: hotter rungs sit closer to the data
    hot = np.loadtxt(f)[:, 0].mean()
    assert 5.0 < hot < 7.0


def test_bayes_factor_fused_equals_chain_file_sweep(csv_file, tmp_path):
    """thermodynamic integration (python/compute_bayes_factors.py): the expectation of log L(theta; t=1) per rung that the
    sampler accumulates on the device equals what the reference's method gets by re-reading the rung's chain file"""
    from pyhillfit_amd import PyHillTemp, compute_bayes_factors
    out = str(tmp_path / "output")
    common = ["--data-file", csv_file, "-d", "0", "-c", "0", "-i", "4000", "-t", "5", "--rungs", "5", "--num-chains", "64", "--output-root", out]
    for m in ("1", "2"):
        PyHillTemp.main(common + ["-m", m])
    bf_dir = str(tmp_path / "BFs") + "/"
    base = ["--data-file", csv_file, "-d", "0", "-c", "0", "--rungs", "5", "--output-root", out, "--bf-dir", bf_dir]
    fused = compute_bayes_factors.main(base)
    swept = compute_bayes_factors.main(base + ["--from-files"])
    assert fused["sources"] == {1: "fused", 2: "fused"} and swept["sources"] == {1: "chain files", 2: "chain files"}
    assert np.loadtxt(swept["file"]).shape == () and np.isfinite(fused["B12"]) and fused["B12"] > 0
    # chain 0's fused sums are the same samples the chain files hold: identical expectation up to summation order
    for m in (1, 2):
        with open(os.path.join(out, "crumb_data", "single-level", "Amiodarone", "hERG", "model_%d" % m, "thermodynamic_integration.json")) as f:
            ti = json.load(f)
        assert swept["expectations"][m] == pytest.approx(ti["expectation_chain0"], rel=1e-11)
        assert np.all(np.diff(ti["log_py_pooled"]) > 0)            # hotter rungs fit the data better
        # pooled 64 chains vs chain 0 alone (what the reference's method sees: 600 kept samples of ONE chain per rung).  The tolerance is the
        # run's own: 4 standard deviations of a single chain's estimate, measured over the 64 chains of every rung and carried through the
        # trapezium weights (PyHillTemp.assemble_thermodynamic_integration) — not a fixed number (G6 is the strict check)
        assert 0 < ti["expectation_chain_sd"] < 2.0
        assert abs(fused["expectations"][m] - swept["expectations"][m]) <= 4 * ti["expectation_chain_sd"], (m, fused["expectations"][m], swept["expectations"][m], ti["expectation_chain_sd"])
    assert 1e-3 < fused["B12"] < 1e3      # evidence ratio of two nested, similarly good models (better fit vs Occam factor)


def test_rccl_backend_collectives_of_the_multi_gpu_path(tmp_path):
    """The N > 1 path runs over RCCL (backend "nccl") and can only be launched by the driver on a multi-GPU node; the gloo
    world-2 tests cover its logic on CPU.  Here the SAME helper calls (one-reader data set-up, packed-points broadcast,
    ragged gather, MAX/SUM all-reduce, barrier, finalize) go through the RCCL backend itself in a world of one rank on the
    GPU — a fresh process, like a torchrun rank — so that API use the gloo tests cannot see (device placement of the
    collectives' tensors, gather support of the backend) is exercised before the scaling run."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = tmp_path / "rank.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from pyhillfit_amd import distributed as D, doseresponse as dr
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and D.collective_device(dev) == dev
box = [("a", 1)]
dist.broadcast_object_list(box, src=0)
D.setup_data_file(os.path.join(%r, "data", "crumb_dataset.json"))
packed = dr.pack_single_level([(d, c) for d in dr.drugs[:3] for c in dr.channels])
shape = torch.tensor([packed.num_pairs, packed.stride], dtype=torch.int64, device=dev)
dist.broadcast(shape, 0)
for a in (packed.ln_conc, packed.counts):
    t = torch.from_numpy(a).to(dev); dist.broadcast(t, 0); assert np.array_equal(t.cpu().numpy(), a)
local = torch.arange(12, dtype=torch.float64, device=dev).reshape(4, 3)
n = torch.tensor([4], dtype=torch.int64, device=dev); sizes = [torch.zeros_like(n)]
dist.all_gather(sizes, n)
outs = [torch.zeros_like(local)]
dist.gather(local, outs, dst=0)
assert torch.equal(outs[0], local) and int(sizes[0]) == 4
tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX); dist.all_reduce(tt, op=dist.ReduceOp.SUM)
assert float(tt) == 1.25
dist.barrier()
torch.cuda.synchronize()
D.finalize()
assert not dist.is_initialized()
print("RCCL_WORLD1_OK")
''' % (REPO, REPO))
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("workload", ["c3", "c4"])
def test_bench_line_keeps_the_contract(workload):
    """`python bench.py` prints ONE JSON line with the keys the driver reads (metric, value, unit, n_gpus, steps, warmup, ms_per_step,
    higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus `roofline` and — at N = 1 — `cpu_baseline`; run small
    (c3: the single-level queue path; c4: four kernels on four free-running streams) in a fresh process, as the driver does."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", workload,
           "--chains", "128", "--iters-per-step", "400", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and "workload" in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # whole-job throughput = chains x iterations x steps / time
    chains = d["config"]["chains_per_gpu"]
    assert d["value"] == pytest.approx(chains * 400 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)
    assert "other_workloads" not in d                       # only the default-size headline run carries the other configurations


def test_bench_default_line_carries_the_other_configurations():
    """the driver's command (N = 1, defaults): the headline keys are c3's, and `other_workloads` holds driver-timed short regions of
    c2, c4, c5 and of the kernels the command lines and the thermodynamic-integration path launch — c5 with the on-device moments and
    <log L> (`c5_moments`), c3 with model 1 (`c3_model1`) — value, ms_per_step, kernel_ms and the two roofline fractions each"""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "configs[2]" in d["config"]["workload"] and d["config"]["chains_per_gpu"] == 210 * 4096 and d["steps"] == 3
    assert d["value"] == pytest.approx(210 * 4096 * d["config"]["iterations_per_step"] * 3 / (d["ms_per_step"] * 3e-3), rel=1e-6)
    assert d["ms_per_step"] * d["steps"] > 300              # a timed region of the default size is no blink
    ow = d["other_workloads"]
    assert sorted(ow) == ["c2", "c3_model1", "c4", "c5", "c5_moments", "s3", "s3h"]
    assert "synthetic scaling set S3" in ow["s3"]["workload"] and ow["s3"]["chains"] == 1680 * 1024 and ow["s3"]["value"] > 1e10
    # the hierarchical model on the same generated set: every pair has the shape of the hand-allocated gfx950 kernel, one launch per step —
    # per block of 64 chains faster than the Crumb mix of C4, whose other Ne groups run through hipcc kernels beside it
    assert "HIERARCHICAL" in ow["s3h"]["workload"] and ow["s3h"]["chains"] == 1680 * 1024 and ow["s3h"]["kernel"].startswith("phf_hier3_advance")
    assert ow["s3h"]["value"] > 1.05 * ow["c4"]["value"]
    assert all(e["ms_per_step"] * e["steps"] >= 450 for e in ow.values()), {k: e["ms_per_step"] * e["steps"] for k, e in ow.items()}   # >= ~0.5 s each
    assert "model 1" in ow["c3_model1"]["workload"] and ow["c3_model1"]["kernel"].startswith("mh_advance_kernel<1")
    assert "moments" in ow["c5_moments"]["workload"] and ow["c5_moments"]["kernel"] == "mh_advance_kernel<2, moments>"
    assert ow["c5_moments"]["algorithmic_bytes_per_launch"] == ow["c5"]["algorithmic_bytes_per_launch"]      # the same rows are written
    assert ow["c3_model1"]["algorithmic_bytes_per_launch"] == pytest.approx(0.75 * d["roofline"]["algorithmic_bytes_per_launch"])   # 3 columns, not 4
    for w, chains in (("c2", 65536), ("c4", 210 * 1024), ("c5", 32 * 210 * 1024), ("c5_moments", 32 * 210 * 1024), ("c3_model1", 210 * 4096)):
        e = ow[w]
        assert e["chains"] == chains and e["value"] == pytest.approx(chains * e["iterations_per_step"] * e["steps"] / (e["ms_per_step"] * e["steps"] * 1e-3), rel=1e-6)
        assert 0 < e["kernel_ms"] <= e["ms_per_step"] * 1.05 and 0 < e["roofline_frac"] < 1 and 0.05 < e["mean_acceptance"] < 0.6
        assert e["value"] > 1e9


def test_bench_two_ranks_default_strong_region_is_the_synthetic_s3_batch():
    """the driver's N > 1 command (default workload): `value` is the weak-scaling C3 figure and `strong_value` the split of ONE batch
    of SURVEY 8(d)'s synthetic set S3 (1 680 generated pairs x 4 096 chains in the real run; here 210 x 256 so that two ranks fit
    the one GPU of the rehearsal) by (pair, 64-chain block) units"""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "210", "--strong-chains", "256",
           "--iters-per-step", "400", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, PHF_BENCH_BACKEND="gloo"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "configs[2]" in d["config"]["workload"]
    sr = d["strong_region"]
    assert "synthetic scaling set S3" in sr["workload"] and sr["chains_all_gpus"] == 210 * 256 and abs(sr["chains_rank0"] - 210 * 128) <= 64
    assert d["strong_value"] == pytest.approx(210 * 256 * 400 * 2 / (sr["ms_per_step"] * 2e-3), rel=1e-6) and d["strong_value"] > 0


def test_bench_two_ranks_split_the_hierarchical_synthetic_batch():
    """`bench.py --gpus 2 --workload s3h` rehearsed on the one GPU (gloo): the weak value = two full batches of the hierarchical model on the
    generated set, `strong_value` = ONE batch split by (pair, 64-chain block) units — every rank's share is one launch of the gfx950
    assembly kernel (all generated pairs have three experiments of four points)"""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "s3h", "--pairs", "96", "--chains", "256", "--steps", "2", "--warmup", "1",
           "--iters-per-step", "400", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, PHF_BENCH_BACKEND="gloo"))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and "HIERARCHICAL" in d["config"]["workload"] and d["config"]["chains_all_gpus"] == 2 * 96 * 256
    assert "synthetic dose-response rows" in d["data"] and d["roofline"]["kernel"].startswith("phf_hier3_advance")
    sr = d["strong_region"]
    assert sr["chains_all_gpus"] == 96 * 256 and abs(sr["chains_rank0"] - 96 * 128) <= 64
    assert d["strong_value"] == pytest.approx(96 * 256 * 400 * 2 / (sr["ms_per_step"] * 2e-3), rel=1e-6) and d["strong_value"] > 0


def test_bench_two_ranks_report_weak_and_strong_scaling():
    """`bench.py --gpus 2` rehearsed on the one GPU (gloo; RCCL refuses two ranks on a device): ONE JSON line with `value` (weak:
    every rank its own full batch) AND `strong_value` (one batch split by (pair, 64-chain block) units, distributed.shard_blocks)"""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--chains", "256",
           "--iters-per-step", "400", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, PHF_BENCH_BACKEND="gloo"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["chains_all_gpus"] == 2 * 210 * 256
    assert d["value"] == pytest.approx(2 * 210 * 256 * 400 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)
    sr = d["strong_region"]
    assert sr["chains_all_gpus"] == 210 * 256 and abs(sr["chains_rank0"] - 210 * 128) <= 64          # 840 blocks: 420 per rank
    assert d["strong_value"] == pytest.approx(210 * 256 * 400 * 2 / (sr["ms_per_step"] * 2e-3), rel=1e-6) and d["strong_value"] > 0
