"""Host-side logic (no GPU): data loading/packing mirror of python/doseresponse.py against the reference's
own loader output (tests/golden/g4_pairs.json), path helpers, gamma table."""
import os
import sys
import tempfile

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def dr():
    from pyhillfit_amd import doseresponse as d
    d.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    return d


def test_setup_and_load_match_reference_loader(dr, g4_pairs):
    assert len(dr.drugs) == 30 and len(dr.channels) == 7 and dr.dir_name == "crumb_dataset"
    assert dr.drugs[0] == "Amiodarone" and dr.channels[0] == "hERG"
    for (d, c), g in g4_pairs.items():
        ne, nums, ex = dr.load_crumb_data(d, c)
        assert ne == g["num_expts"] and [len(e) for e in ex] == g["expt_sizes"] and nums[0] == 0
        concs, y = dr.concatenate_experiments(ne, ex)
        assert np.array_equal(concs, g["concs"]) and np.array_equal(y, g["responses"])
        w0, w100, wo = dr.response_masks(y)
        assert (w0.sum(), w100.sum(), wo.sum()) == (g["n0"], g["n100"], g["n_other"])
        assert dr.compute_pi_bit_of_log_likelihood(wo) == g["pi_bit"]
    with pytest.raises(ValueError):
        dr.load_crumb_data("NoSuchDrug", "hERG")


def test_csv_round_trip(dr, tmp_path):
    """the CLI's --data-file is the reference CSV format (data/readme.md): same table through a CSV"""
    p = tmp_path / "crumb_data.csv"
    dr.table.to_csv(str(p))
    with open(p) as f:
        assert f.readline().strip() == "Compound,Channel,Experiment,Dose,Response"
    t = dr.Table.from_csv(str(p))
    assert np.array_equal(t.dose, dr.table.dose) and np.array_equal(t.response, dr.table.response)
    assert list(t.drug) == list(dr.table.drug) and np.array_equal(t.experiment, dr.table.experiment)


def test_packing_groups_points_like_the_masks(dr, g4_pairs):
    names = list(g4_pairs)
    pk = dr.pack_single_level(names, merge=False)          # one entry per point
    assert pk.num_pairs == 210 and pk.stride == 20 and np.all(pk.weight[pk.weight != 0] == 1) and np.all(pk.extra[:, 1] == 0)
    for p, key in enumerate(names):
        g = g4_pairs[key]
        n_other, n0, n100, ntot = pk.counts[p]
        assert (n_other, n0, n100, ntot) == (g["n_other"], g["n0"], g["n100"], g["n_total"])
        y = np.array(g["responses"]); c = np.array(g["concs"])
        assert np.array_equal(pk.response[p, :n_other], y[(y > 0) & (y < 100)])
        assert np.array_equal(pk.ln_conc[p, :n_other], np.log(c[(y > 0) & (y < 100)]))
        assert np.all(pk.response[p, n_other:n_other + n0] == 0) and np.all(pk.response[p, n_other + n0:n_other + n0 + n100] == 100)
        assert pk.pi_bit[p] == g["pi_bit"]
    # the -2.6 response (crumb_data.csv:155) is in no mask: dropped from the points, still counted in pi_bit
    p = names.index(("Amitriptyline", "Kv4.3"))
    assert pk.counts[p].tolist() == [17, 1, 0, 19]
    assert pk.extra[p, 0] == 17


def test_merged_entries_carry_the_same_likelihood_sums(dr, g4_pairs):
    """replicates at one concentration become one weighted entry: counts, weighted residual sums and the censored
    multiplicities of every Crumb pair are those of the point-by-point sums (doseresponse.py:244-247)"""
    names = list(g4_pairs)
    pk = dr.pack_single_level(names)
    assert pk.stride < 20 and pk.counts[:, :3].sum() < 0.5 * sum(g4_pairs[k]["n_other"] + g4_pairs[k]["n0"] + g4_pairs[k]["n100"] for k in names)
    rng = np.random.RandomState(0)
    for p, key in enumerate(names):
        g = g4_pairs[key]
        k_other, k0, k100, ntot = pk.counts[p]
        y = np.array(g["responses"]); c = np.array(g["concs"])
        other = (y > 0) & (y < 100)
        assert pk.extra[p, 0] == g["n_other"] == pk.weight[p, :k_other].sum()
        assert pk.weight[p, k_other:k_other + k0].sum() == g["n0"] and pk.weight[p, k_other + k0:k_other + k0 + k100].sum() == g["n100"]
        assert np.all(pk.response[p, k_other:k_other + k0] == 0) and np.all(pk.response[p, k_other + k0:k_other + k0 + k100] == 100)
        lc = pk.ln_conc[p, :k_other]
        assert len(set(lc)) == k_other == len(set(c[other]))
        # any curve that depends on the point through its concentration only
        f = lambda lnc: 50 + 30 * np.sin(lnc * rng_a + rng_b)
        rng_a, rng_b = rng.uniform(0.5, 2), rng.uniform(0, 6)
        want = np.sum((y[other] - f(np.log(c[other]))) ** 2)
        got = pk.extra[p, 1] + np.sum(pk.weight[p, :k_other] * (pk.response[p, :k_other] - f(lc)) ** 2)
        assert got == pytest.approx(want, rel=1e-13, abs=1e-10)
    p = names.index(("Amiodarone", "hERG"))                  # 3 doses x 3 uncensored + 1 dose x 3 zeros
    assert pk.counts[p].tolist() == [3, 1, 0, 12] and pk.weight[p, :4].tolist() == [3, 3, 3, 3]


def test_zero_dose_and_empty_groups():
    from pyhillfit_amd.doseresponse import PackedPoints
    pk = PackedPoints([(np.array([0.0, 1.0, 10.0]), np.array([0.0, 50.0, 100.0])), (np.array([1.0]), np.array([0.0]))])
    assert pk.ln_conc[0, 1] == -np.inf and pk.counts.tolist() == [[1, 1, 1, 3], [0, 1, 0, 1]]


def test_output_paths_follow_the_reference_contract(dr, tmp_path, monkeypatch):
    monkeypatch.setattr(dr, "output_root", str(tmp_path / "output"))
    drug, channel, chain_file, images_dir = dr.nonhierarchical_chain_file_and_figs_dir(2, "Moxifloxacin", "KvLQT1/mink", 1)
    assert channel == "KvLQT1_mink"
    assert chain_file.endswith("output/crumb_dataset/single-level/Moxifloxacin/KvLQT1_mink/model_2/temperature_1/chain/"
                               "Moxifloxacin_KvLQT1_mink_model_2_temp_1_chain_single-level.txt")
    assert os.path.isdir(images_dir)
    _, _, f2, _ = dr.nonhierarchical_chain_file_and_figs_dir(1, "Amiodarone", "hERG", dr.temperature_ladder()[1])
    # PyHillTemp formats the float into the path with Python-2 str(): 12 significant digits
    assert "temperature_1.5625e-05/" in f2 and f2.endswith("_temp_1.5625e-05_chain_single-level.txt")
    assert [dr.py2_str(v) for v in (1, 1.0, 0.0, 0.125, (3 / 40.) ** 3, (39 / 40.) ** 3)] == ["1", "1.0", "0.0", "0.125", "0.000421875", "0.926859375"]
    d, c, out, chain_dir, figs, cf = dr.hierarchical_output_dirs_and_chain_file("Amiodarone", "hERG", 3)
    assert cf.endswith("output/crumb_dataset/hierarchical/Amiodarone/hERG/3_expts/chain/crumb_dataset_Amiodarone_hERG_hierarchical_chain.txt")
    assert dr.alpha_mu_downsampling("Amiodarone", "hERG").endswith("hierarchical/alpha_mu_samples/Amiodarone_hERG_hill_pic50_samples.txt")


def test_gamma_table_is_the_reference_expression():
    from pyhillfit_amd.sampler import gamma_table
    g = gamma_table(5000)
    for s in (1, 2, 17, 4999):
        assert g[s] == 1 / (s + 1) ** 0.6            # PyHillFit.py:842
    assert len(gamma_table(10)) == 11


def test_queue_quantum_is_bounded_however_long_the_advance():
    """a queued launch's quantum: ~n / quanta, whole thinning periods, >= 100, and never above MAX_QUANTUM_ITERATIONS — the kernel's
    give-up limit for a hand-over wait is a fixed number of polls, so a quantum must not grow with the call"""
    from pyhillfit_amd.sampler import MAX_QUANTUM_ITERATIONS, queue_quantum
    assert queue_quantum(8000, 4, 5) == 2000 and queue_quantum(2000, 4, 5) == 500
    assert queue_quantum(399, 4, 5) == 0 and queue_quantum(8000, 0, 5) == 0 and queue_quantum(8000, 1, 5) == 0     # plain launch
    assert queue_quantum(401, 4, 5) == 105                                       # rounded up to a thinning period
    for n in (24000, 500000, 50_000_000):
        for thin in (1, 5, 7):
            q = queue_quantum(n, 4, thin)
            assert 100 <= q <= MAX_QUANTUM_ITERATIONS and q % thin == 0
    assert queue_quantum(24000, 4, 5) == 4000                                    # six quanta instead of four of 6 000


def test_ladder_and_models(dr):
    lad = dr.temperature_ladder()
    assert len(lad) == 41 and lad[1] == (1 / 40.) ** 3 and len(dr.temperature_ladder(31)) == 32
    dr.define_model(1); assert dr.num_params == 2 and dr.file_labels == ['pIC50', 'sigma']
    dr.define_model(2); assert dr.num_params == 3
    with pytest.raises(ValueError):
        dr.define_model(3)


def test_cli_skips_pairs_without_data_or_with_missing_responses(tmp_path, capsys):
    """python/PyHillFit.py:653-669"""
    from pyhillfit_amd import PyHillFit
    from pyhillfit_amd import doseresponse as d
    p = tmp_path / "tiny.csv"
    p.write_text("Compound,Channel,Experiment,Dose,Response\nA,X,1,0.1,10\nA,X,1,1,50\nA,X,2,0.1,12\nA,X,2,1,\nB,X,1,0.1,5\nB,X,1,1,40\n")
    d.setup(str(p))
    loaded = PyHillFit.load_single_level_pairs([("A", "X"), ("B", "X"), ("B", "Y")])
    assert [(x[0], x[1]) for x in loaded] == [("B", "X")]
    out = capsys.readouterr().out
    assert "Skipping ('A', 'X')" in out and "no entries for B + Y" in out
    d.setup(os.path.join(REPO, "data", "crumb_dataset.json"))


def test_best_fit_start_points(dr):
    from pyhillfit_amd import bestfit
    ne, _, ex = dr.load_crumb_data("Amiodarone", "hERG")
    concs, y = dr.concatenate_experiments(ne, ex)
    th2, ss2 = bestfit.best_fit(concs, y, 2)
    th1, ss1 = bestfit.best_fit(concs, y, 1)
    assert ss2 < ss1 and abs(th2[0] - 6.03) < 0.05 and abs(th2[1] - 0.58) < 0.05 and th2[2] == pytest.approx(np.sqrt(ss2 / 12))
    ne, _, ex = dr.load_crumb_data("Lidocaine", "KvLQT1/mink")          # all responses 0: SS = 0
    th, ss = bestfit.best_fit(*dr.concatenate_experiments(ne, ex), 2)
    assert ss < 1e-9 and th[2] == 1.0                                   # not the reference's absorbing sigma0 = 0


def test_chain_start_moves_only_a_hill_above_the_priors_bound():
    """bestfit.chain_start: the fit is the start, except a model-2 Hill coefficient above hill_uniform_upper (doseresponse.py:18), which is
    moved ONTO the bound (inside the support: doseresponse.py:181-182 rejects hill > 10 only); model 1 has no Hill column"""
    from pyhillfit_amd import bestfit
    th = np.array([[6.0, 0.7, 8.0], [5.0, 24.98, 3.0], [4.0, 10.0, 2.0], [3.0, 10.0000001, 1.0]])
    out = bestfit.chain_start(th, 2)
    assert np.array_equal(out, [[6.0, 0.7, 8.0], [5.0, 10.0, 3.0], [4.0, 10.0, 2.0], [3.0, 10.0, 1.0]]) and out is not th and th[1, 1] == 24.98
    assert np.array_equal(bestfit.chain_start(th[1], 2), [5.0, 10.0, 3.0])             # one row
    assert np.array_equal(bestfit.chain_start(np.array([[6.0, 25.0]]), 1), [[6.0, 25.0]])   # model 1: (pIC50, sigma)


def test_batched_least_squares_against_the_reference_objective_grid(dr):
    """golden G8 (tests/golden/make_golden_bestfit.py): the reference's own sum_of_square_diffs (python/PyHillFit.py:93-97),
    lifted and evaluated on a dense (pIC50, Hill) grid for every Crumb pair and both models.  The product's batched fit
    (a) evaluates the same objective (probe points equal the reference's values), (b) reaches, for all 210 pairs, a sum of
    squares no larger than the reference grid's minimum, with the reference's initial_sigma (:101-102) at it, (c) agrees with
    the scalar Nelder-Mead fit it replaces, and (d) its start point is written in the form assemble_BFs.py:62-63 reads."""
    import json
    from conftest import GOLDEN
    from pyhillfit_amd import bestfit, chainio
    with open(os.path.join(GOLDEN, "g8_least_squares_grid.json")) as f:
        g8 = json.load(f)
    assert len(g8["pairs"]) == 210
    pairs = []
    for w in g8["pairs"]:
        ne, _, ex = dr.load_crumb_data(w["drug"], w["channel"])
        pairs.append(dr.concatenate_experiments(ne, ex))
        for p_, h_, v in w["probes"]:
            assert bestfit.sum_of_square_diffs([p_, h_], *pairs[-1]) == pytest.approx(v, rel=1e-13, abs=1e-300)
    for model in (2, 1):
        theta, ss = bestfit.best_fit_batch(pairs, model)
        grid_min = np.array([w["model_%d" % model]["grid_min_ss"] for w in g8["pairs"]])
        assert theta.shape == (210, model + 1)
        assert np.all(ss <= grid_min * (1 + 1e-12) + 1e-20), np.max(ss - grid_min)
        for k in range(210):                                          # the SS reported is the objective at the point reported
            assert bestfit.sum_of_square_diffs([theta[k, 0], theta[k, 1] if model == 2 else 1.0], *pairs[k]) == pytest.approx(ss[k], rel=1e-9, abs=1e-12)
        n = np.array([w["n"] for w in g8["pairs"]])
        sig = np.sqrt(ss / n)
        assert np.all(theta[:, -1] == np.where(sig > dr.sigma_loc, sig, 1.0))       # initial_sigma; 1 where SS = 0
        assert np.all(theta[:, 0] >= dr.pic50_exp_lower) and (model == 1 or np.all(theta[:, 1] > 0))
        sub = list(range(3, 97, 5))                                   # a pair's fit does not depend on its batch (the multi-GPU
        th_sub, ss_sub = bestfit.best_fit_batch([pairs[k] for k in sub], model)   # partitions start their chains from identical points)
        assert np.array_equal(th_sub, theta[sub]) and np.array_equal(ss_sub, ss[sub])
        for k in range(0, 210, 23):                                   # the scalar fit it replaces
            th_nm, ss_nm = bestfit.best_fit(*pairs[k], model)
            assert ss[k] <= ss_nm + 1e-6 * (1 + ss_nm)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "Amiodarone_hERG_best_fit_params.txt")
        chainio.save_best_fit_params(path, theta[0], 1)
        assert np.array_equal(np.loadtxt(path), theta[0])            # assemble_BFs.py:62-63
        with open(path) as f:
            assert f.readline().startswith("#") and f.readline().startswith("# pIC50, sigma")


def test_hierarchical_start_points_batched(dr):
    """per-experiment (pIC50, Hill) fits of all pairs in one batch == the scalar fits they replace (PyHillFit.py:243-257)"""
    from pyhillfit_amd import bestfit
    from pyhillfit_amd import hierarchical as H
    shapes, scales, locs = H.prior_params()
    names = [("Amiodarone", "hERG"), ("Amitriptyline", "Kv4.3"), ("Lidocaine", "KvLQT1/mink"), ("Verapamil", "Cav1.2")]
    exs = [dr.load_crumb_data(d, c)[2] for d, c in names]
    out = bestfit.hierarchical_first_iteration_batch(exs, locs)
    for e, th in zip(exs, out):
        assert len(th) == 5 + 2 * len(e) and np.all(th[:4] > locs[:4]) and th[-1] > locs[4]
        assert np.array_equal(th, bestfit.hierarchical_first_iteration(e, locs))
        # the two distribution fits (PyHillFit.py:310-334) against scipy's scalar optimisers
        m = len(e)
        table = np.column_stack([th[4:-1:2], th[5:-1:2], np.full(m, th[-1])])
        a_nm, b_nm, mu_nm, s_nm, _ = bestfit._hyper_start(table, np.asarray(locs))
        assert th[2] == pytest.approx(mu_nm, rel=1e-6) and th[3] == pytest.approx(s_nm, rel=1e-5)
        assert th[1] == pytest.approx(b_nm, rel=1e-5) and th[0] == pytest.approx(a_nm, rel=(1e-5 if b_nm < 19.99 else 0.02))
        for i, ex in enumerate(e):
            th_nm, ss_nm = bestfit._fit_pic50_hill(ex[:, 0], ex[:, 1])
            assert bestfit.sum_of_square_diffs([th[4 + 2 * i], th[5 + 2 * i]], ex[:, 0], ex[:, 1]) <= ss_nm + 1e-7 * (1 + ss_nm)
            assert th[4 + 2 * i] >= -2.0 and th[5 + 2 * i] > 0
    with pytest.raises(ValueError, match="without points"):              # not a NaN start sigma
        bestfit.hierarchical_first_iteration_batch([exs[0], [exs[1][0], np.zeros((0, 2))]], locs)
    with pytest.raises(ValueError, match="no experiments"):
        bestfit.hierarchical_first_iteration_batch([[]], locs)


def test_bench_profile_facts_only_apply_to_the_measured_launch_shape():
    import bench
    f = bench.profile_facts("c2", 65536, 2000, 5)
    assert 300 < f["flop_per_iteration"] < 2000 and f["traffic_bytes_per_launch"] > 8e8
    assert "traffic_bytes_per_launch" not in bench.profile_facts("c2", 65536, 1000, 5)
    assert bench.profile_facts("zz", 1, 1, 1) == {}


def test_bench_gpus_n_starts_n_ranks_as_a_child(monkeypatch):
    """`bench.py --gpus N` outside torchrun hands over to `torch.distributed.run --nproc-per-node N` as a CHILD process
    (before any GPU call, never an exec) and passes its exit code through; with fewer GPUs than ranks it refuses."""
    import bench
    import torch
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--workload", "c3"])
    a = bench.parse_args(sys.argv[1:])
    assert a.workload == "c3" and bench.parse_args([]).workload == "c3"      # the metric's own config is the default
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    from pyhillfit_amd import distributed as D
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(D, "visible_gpu_count", lambda: 4)
    assert bench.spawn_ranks(a) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd   # the launcher picks the port
    assert cmd[-6:] == sys.argv[1:]
    assert os.path.basename(cmd[-7]) == "bench.py" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(D, "visible_gpu_count", lambda: 1)
    seen.clear()
    assert bench.spawn_ranks(a) == 2 and not seen                            # one GPU, four RCCL ranks: refused, nothing started
    monkeypatch.setenv("PHF_BENCH_BACKEND", "gloo")                          # rehearsal: ranks share the one GPU
    assert bench.spawn_ranks(a) == 7 and seen


def test_visible_gpu_count_asks_a_child_then_the_driver_topology_never_this_process_first(monkeypatch, tmp_path):
    """distributed.visible_gpu_count: the runtime's answer from a short-lived CHILD interpreter; if that fails, GPUs = nodes with SIMDs
    in /sys/class/kfd cut by ROCR_/HIP_/CUDA_VISIBLE_DEVICES; torch in this process (which may initialise the HIP runtime) only last"""
    import glob as _glob
    import subprocess
    import torch
    from pyhillfit_amd import distributed as D

    class Done(object):
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out
    monkeypatch.setattr(torch.cuda, "device_count", lambda: pytest.fail("this process's runtime must not be asked"))
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Done(0, "some warning\n6\n"))
    assert D.visible_gpu_count() == 6                                # the child's answer (last line of its output)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Done(1, ""))   # the child fails: the driver's topology
    nodes = []
    for k, simds in enumerate([0, 0, 1024, 1024, 1024]):            # two CPU nodes, three GPUs
        d = tmp_path / str(k); d.mkdir()
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count %d\nmem_banks_count 1\n" % simds)
        nodes.append(str(d / "properties"))
    monkeypatch.setattr(_glob, "glob", lambda pattern: nodes if "kfd" in pattern else [])
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert D.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert D.visible_gpu_count() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert D.visible_gpu_count() == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert D.visible_gpu_count() == 0

    def boom(*a, **k):
        raise OSError("no interpreter")
    monkeypatch.setattr(subprocess, "run", boom)
    monkeypatch.setattr(_glob, "glob", lambda pattern: [])          # no child, no topology (not an AMD host): torch, last
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 5)
    assert D.visible_gpu_count() == 5


def test_writer_pool_files_equal_the_synchronous_ones(tmp_path):
    """chain files written by spawned worker processes are byte-identical to np.savetxt in the caller, in the reference's
    format (header lines, '%.18e'), and readable the way the reference's downstream scripts read them"""
    from pyhillfit_amd import chainio
    rng = np.random.RandomState(3)
    chains = [rng.standard_normal((200 + 10 * i, 4)) for i in range(5)]
    hier = rng.standard_normal((300, 12))
    for workers, processes, sub in ((0, None, "sync"), (2, None, "pool"), (2, True, "procs")):   # caller / threads (native formatter) / processes
        os.makedirs(str(tmp_path / sub))
        with chainio.WriterPool(workers, processes=processes) as w:
            for i, c in enumerate(chains):
                w.submit(chainio.save_single_level_chain, str(tmp_path / sub / ("c%d.txt" % i)), c, "Drug", "hERG", 2)
            w.submit(chainio.save_hierarchical_chain, str(tmp_path / sub / "h.txt"), hier)
            w.submit(chainio.save_alpha_mu_samples, str(tmp_path / sub / "am.txt"),
                     chainio.pick_alpha_mu_rows(hier, 50, 75, np.random.RandomState(1)), "Drug", "hERG")
    for name in ["c%d.txt" % i for i in range(5)] + ["h.txt", "am.txt"]:
        a = open(str(tmp_path / "sync" / name), "rb").read()
        assert a == open(str(tmp_path / "pool" / name), "rb").read(), name
        assert a == open(str(tmp_path / "procs" / name), "rb").read(), name
    assert np.array_equal(chainio.load_chain(str(tmp_path / "pool" / "c2.txt")), chains[2])
    assert np.array_equal(chainio.load_chain(str(tmp_path / "pool" / "h.txt"), usecols=range(4)), hier[:, :4])
    assert chainio.load_chain(str(tmp_path / "pool" / "am.txt")).shape == (50, 2)
    first = open(str(tmp_path / "pool" / "h.txt")).readline()
    assert first.startswith("# Hill ~ log-logistic")


def test_writer_pool_reports_a_failed_write(tmp_path):
    from pyhillfit_amd import chainio
    w = chainio.WriterPool(1)
    w.submit(chainio.save_tempered_chain, str(tmp_path / "no_such_dir" / "x.txt"), np.zeros((2, 2)))
    with pytest.raises(OSError):
        w.close()


def test_stream_writers_produce_the_file_of_one_savetxt(tmp_path):
    """chain files appended segment by segment (by ordered single-process lanes, or in the caller) are byte-identical to
    writing the whole chain at once"""
    from pyhillfit_amd import chainio
    rng = np.random.RandomState(11)
    chains = {str(tmp_path / ("h%d.txt" % i)): rng.standard_normal((130 + i, 9)) for i in range(7)}
    for path, chain in chains.items():
        chainio.save_hierarchical_chain(path + ".whole", chain)
    for workers, processes in ((0, None), (3, None), (2, True)):      # caller / threads around the native formatter / spawned processes
        sw = chainio.StreamWriters(workers, processes=processes)
        for path, chain in chains.items():
            sw.create(path, chainio.HIERARCHICAL_HEADER, chain[0:1])
        for lo in range(1, 140, 17):
            for path, chain in chains.items():
                if lo < len(chain):
                    sw.append(path, chain[lo:lo + 17])
        sw.close()
        for path in chains:
            assert open(path, "rb").read() == open(path + ".whole", "rb").read(), (workers, path)
            os.remove(path)


def test_stream_writer_survives_a_dead_worker(tmp_path):
    """a writer process killed in the middle of a run: the files of its lane are rewritten from the job history by the
    caller, the other lanes keep working, and every file ends up byte-identical to the one-call np.savetxt text"""
    import signal
    import time
    from pyhillfit_amd import chainio
    rng = np.random.RandomState(3)
    chains = {str(tmp_path / ("chain_%d.txt" % k)): rng.standard_normal((90 + k, 5)) for k in range(6)}
    for path, chain in chains.items():
        chainio.save_hierarchical_chain(path + ".whole", chain)
    sw = chainio.StreamWriters(2, processes=True)
    for path, chain in chains.items():
        sw.create(path, chainio.HIERARCHICAL_HEADER, chain[0:1])
    for path, chain in chains.items():
        sw.append(path, chain[1:40])
    victim = sw.lanes[0]
    for f, lane in list(sw.pending):                                     # let the queued jobs finish, then kill lane 0's process
        f.result()
    for proc in list(victim._processes.values()):
        os.kill(proc.pid, signal.SIGKILL)
    time.sleep(0.3)
    for path, chain in chains.items():
        sw.append(path, chain[40:])
    sw.close()
    assert sw.dead == {0} or sw.dead == set()                            # (set() only if no file hashed to lane 0)
    for path in chains:
        assert open(path, "rb").read() == open(path + ".whole", "rb").read(), path


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke()/build() and bench.py's cpu_baseline may touch it"""
    import ast
    import glob
    offenders = []
    for path in glob.glob(os.path.join(REPO, "pyhillfit_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(REPO, "python", "*.py")) \
            + glob.glob(os.path.join(REPO, "tools", "*.py")):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""] if isinstance(node, ast.ImportFrom) else []
            if any(m.split(".")[0] == "oracle" for m in mods):
                offenders.append(os.path.relpath(path, REPO))
    assert not offenders, offenders
    bench = ast.parse(open(os.path.join(REPO, "bench.py")).read())
    for fn in [n for n in bench.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert uses == (fn.name == "cpu_baseline"), fn.name


def test_native_text_formatter_is_numpy_savetxt_byte_for_byte(tmp_path):
    """csrc/phf_textio.cpp (std::to_chars, scientific, 18 digits) against np.savetxt's default '%.18e': random bit patterns over
    the whole exponent range, subnormals, exact decimal ties (2^-28 has 20 significant digits ending in 5), signed zeros,
    infinities and NaN, headers, appending, one column, no rows"""
    import io
    from pyhillfit_amd import chainio
    assert chainio._textio(), "libphf_textio.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 64, 200000, dtype=np.uint64).view(np.float64)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, -np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                        2.0 ** -28, 2.0 ** -29, 3 * 2.0 ** -30, 9.999999999999999e22, 1e23, 0.1, 1 / 3., 123456789.123456789, 2.5e-9, 1e100])
    vals = np.concatenate([bits, special, rng.standard_normal(20000) * 10.0 ** rng.integers(-30, 30, 20000)])
    vals = vals[:len(vals) // 4 * 4]
    for cols in (4, 1, 12):
        table = vals[:len(vals) // cols * cols].reshape(-1, cols)
        want = io.BytesIO()
        with np.errstate(all="ignore"):
            np.savetxt(want, table)
        path = str(tmp_path / ("t%d.txt" % cols))
        chainio.write_text(path, False, ("# header one\n", "# two\n"), table[:1000])
        chainio.write_text(path, True, (), table[1000:])
        chainio.write_text(path, True, (), table[:0])
        got = open(path, "rb").read()
        assert got == b"# header one\n# two\n" + want.getvalue(), cols
    view = vals[:4000].reshape(1000, 4)[:, ::2]                      # a non-contiguous view is copied, not misread
    chainio.write_text(str(tmp_path / "v.txt"), False, (), view)
    want = io.BytesIO(); np.savetxt(want, view)
    assert open(str(tmp_path / "v.txt"), "rb").read() == want.getvalue()
    with pytest.raises(OSError):
        chainio.write_text(str(tmp_path / "no_such_dir" / "x.txt"), False, (), table[:2])
