"""BASELINE.json's configurations at their FULL sizes on the GPU (-m gpu), and the sampler kernels' log-target column
re-evaluated by the INDEPENDENT numpy oracle.

  C3  all 210 Crumb pairs x 4 096 chains, single level   — shard invariance (the multi-GPU partition) and checkpoint identity
  C4  all 210 pairs x 1 024 chains, hierarchical          — the same two properties
  C5  32 rungs (i/31)^3 x 210 pairs x 1 024 chains        — prior-only rung against the analytic prior, t = 1 rung against the
                                                            reference sampler's posteriors (golden G5c), per-rung <log L> monotone
and, for every pair, rows sampled by mh_advance_kernel (both register builds) and hier_advance_kernel whose log-target column
is recomputed with oracle/pyhillfit_oracle.py (numpy/scipy, shares no source with the kernels) to |diff| <= 1e-12 (|value| + 1): the
twin that the bit-identity tests use includes the kernels' own target headers, so THIS is what pins the arithmetic of the
specialised loop bodies to the reference (python/doseresponse.py:187-189,229-248; python/PyHillFit.py:173-193)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pyhillfit_amd import _lib
    _lib.load()
    return "cuda:0"


@pytest.fixture(scope="module")
def dr(gpu):
    from pyhillfit_amd import doseresponse as d
    d.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    return d


def _all_names(dr):
    return [(d, c) for d in dr.drugs for c in dr.channels]


# ------------------------------------------------------------------------------------- independent re-evaluation
@pytest.mark.parametrize("model,chains,temps", [(2, 64, (1.0,)),          # 210 wavefronts: the one-wavefront-per-SIMD build (512 registers)
                                                (2, 640, (1.0, 0.125)),   # 4 200 wavefronts: the two-per-SIMD build (256 registers)
                                                (1, 640, (1.0,))])
def test_sampled_rows_log_target_recomputed_by_the_numpy_oracle(model, chains, temps, gpu, dr, oracle_pair):
    from oracle import pyhillfit_oracle as orc
    from pyhillfit_amd.sampler import SingleLevelSampler
    names = _all_names(dr)
    packed = dr.pack_single_level(names)
    pair_index = [p for p in range(len(names)) for _ in temps]
    tt = [t for _ in names for t in temps]
    s = SingleLevelSampler(packed, model, pair_index, tt, chains, thinning=5, seed=77, adapt_start=300, device=gpu)
    d = s.d
    s.init([6.0, 0.8, 8.0] if model == 2 else [6.0, 8.0], cov_identity=False, cov_scale=0.05)
    rows = s.advance(1500).cpu().numpy()                         # [300][Q][d+1][C], adaptation on from iteration 300
    assert np.isfinite(rows).all()
    rng = np.random.default_rng(1)
    worst = 0.0
    for q in range(len(pair_index)):
        pair = oracle_pair(*names[pair_index[q]])
        for r, c in zip(rng.integers(0, rows.shape[0], 6), rng.integers(0, chains, 6)):
            th, got = rows[r, q, :d, c], rows[r, q, d, c]
            want = orc.log_target(model, pair, th, tt[q])
            worst = max(worst, abs(got - want) / (abs(want) + 1.0))
    assert worst <= 1e-12, worst
    # the chains moved: a column that never changed would make the check vacuous
    assert (np.abs(np.diff(rows[:, :, 0, :], axis=0)) > 0).mean() > 0.1


def test_hierarchical_sampled_rows_recomputed_by_the_numpy_oracle(gpu, dr, oracle_pair):
    """every Crumb pair (Ne = 3..6: all four compiled kernels, straight-line and run-time-loop bodies)"""
    from oracle import pyhillfit_oracle as orc
    from pyhillfit_amd import bestfit
    from pyhillfit_amd import hierarchical as H
    shapes, scales, locs = H.prior_params()
    groups = {}
    for dname, c in _all_names(dr):
        ne, _, ex = dr.load_crumb_data(dname, c)
        groups.setdefault(len(ex), []).append((dname, c, ex))
    assert sorted(groups) == [3, 4, 5, 6]
    rng = np.random.default_rng(2)
    worst, checked = 0.0, 0
    for ne, members in sorted(groups.items()):
        packed = H.PackedHierPoints([m[2] for m in members])
        theta0 = np.array([bestfit.hierarchical_first_iteration(m[2], locs) for m in members])
        s = H.HierarchicalSampler(packed, list(range(len(members))), 64, thinning=5, seed=9, adapt_start=200, device=gpu)
        s.init(theta0, cov_scale=0.01)
        rows = s.advance(600).cpu().numpy()                      # [120][Q][dim+1][64]
        dim = 5 + 2 * ne
        assert np.isfinite(rows).all()
        for q, m in enumerate(members):
            expts = oracle_pair(m[0], m[1]).experiments
            for r, c in zip(rng.integers(0, rows.shape[0], 3), rng.integers(0, 64, 3)):
                th = rows[r, q, :dim, c]
                want = orc.hier_log_target(expts, th, shapes, scales, locs)
                # conditioning: SSE/(2 sigma^2) turns a relative error e in a prediction into |y - pred| pred e / sigma^2; chains do visit
                # sigma ~ 0.02 (Toremifene-Nav1.5-peak: predictions within 0.05 of the data), where 4 ulp on pred ~ 100 is 1e-11 of
                # the target.  There the kernel's value is the closer of the two to a 50-digit evaluation.
                cond = sum(np.sum(np.abs(e[:, 1] - orc.hill_curve(e[:, 0], th[5 + 2 * i], orc.ic50_of(th[4 + 2 * i]))) * 100.0) for i, e in enumerate(expts)) / th[-1] ** 2
                worst = max(worst, abs(rows[r, q, dim, c] - want) / (1e-12 * (abs(want) + 1.0) + 4 * 2.0 ** -53 * cond))
                checked += 1
        assert (np.abs(np.diff(rows[:, :, 0, :], axis=0)) > 0).mean() > 0.02
    assert checked == 630 and worst <= 1.0, (checked, worst)        # in units of the tolerance above


# ------------------------------------------------------------------------------------- work-queue launch
@pytest.mark.parametrize("model,moments", [(2, False), (2, True), (1, False)])
def test_queued_launch_is_bit_identical_to_the_plain_launch(model, moments, gpu, dr):
    """phf_single_level_advance_queued (quanta of a block chained through HBM state with agent-scope release/acquire, tasks pulled
    from a counter) against phf_single_level_advance: same rows, state and moments bit for bit — 210 pairs x 1 024 chains
    (3 360 blocks over 2 048 resident wavefronts), 2 launches of uneven length, tempered problems mixed in"""
    from pyhillfit_amd.sampler import SingleLevelSampler
    names = _all_names(dr)
    packed = dr.pack_single_level(names)
    Q = len(names)
    temps = [1.0 if q % 4 else 0.3 for q in range(Q)]
    got = []
    for quanta in (0, 4):
        s = SingleLevelSampler(packed, model, list(range(Q)), temps, 1024, thinning=5, seed=41, adapt_start=600, device=gpu, queue_quanta=quanta)
        s.init([6.0, 0.8, 8.0] if model == 2 else [6.0, 8.0])
        if moments:
            s.enable_moments(after_iteration=500)
        rows = torch.cat([s.advance(1700), s.advance(2305)])
        assert (s._queue is not None) == (quanta > 0)
        if quanta:
            q_ = s._queue.cpu().numpy()
            nq = -(-2305 // max(100, -(-(-(-2305 // quanta)) // 5) * 5))
            assert q_[0] >= 3360 * nq and q_[0] < 0x40000000 and np.all(q_[1:-1] == nq) and q_[-1] == 0   # every task pulled, every block through all its quanta, no fault
        got.append((rows, s.state.clone(), None if s.moments is None else s.moments.clone()))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    if moments:
        assert torch.equal(got[0][2], got[1][2])
    assert torch.isfinite(got[1][0]).all()


def test_a_drained_queued_launch_is_an_error_not_stale_chains(gpu, dr):
    """the sticky fault word of the queue workspace (set by a wavefront that gave up waiting, ABI 4) reaches the host as an error at
    every point that hands results on: the word is written here by hand, as such a launch would leave it"""
    import ctypes as C
    from pyhillfit_amd import _lib
    from pyhillfit_amd.sampler import SingleLevelSampler
    names = _all_names(dr)
    packed = dr.pack_single_level(names)
    Q = len(names)
    s = SingleLevelSampler(packed, 2, list(range(Q)), [1.0] * Q, 1024, thinning=5, seed=41, adapt_start=600, device=gpu, queue_quanta=4)
    s.init([6.0, 0.8, 8.0])
    s.enable_moments(after_iteration=0)
    s.advance(800, save=False)
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream)
    assert lib.phf_single_level_queue_status(C.c_void_p(s._queue.data_ptr()), s.nblocks, stream) == 0
    s.acceptance(); s.posterior_moments(); s.state_dict()                  # a healthy queue: no raise
    s._queue[-1] = 1
    assert lib.phf_single_level_queue_status(C.c_void_p(s._queue.data_ptr()), s.nblocks, stream) == -4 and b"drained" in lib.phf_last_error()
    s.advance(400, save=False)                                             # the library never clears the word ...
    assert int(s._queue[-1].item()) == 1 and int(s._queue[0].item()) > 0   # ... while it does reset counter and progress words
    for call in (s.acceptance, s.posterior_moments, s.mean_log_likelihood_t1, s.state_dict, lambda: s.run(100)):
        with pytest.raises(_lib.PhfError, match="drained"):
            call()


def test_state_and_moments_must_be_device_memory(gpu, dr):
    """the ABI's memory-kind rule (include/pyhillfit_amd.h): a host-pinned state or moments buffer is refused with an error code —
    fp64 atomics and the queued launch's release/acquire hand-overs are not guaranteed there — instead of sampling into it"""
    import ctypes as C
    from pyhillfit_amd import _lib
    from pyhillfit_amd import hierarchical as H
    from pyhillfit_amd.sampler import SingleLevelSampler
    packed = dr.pack_single_level([("Amiodarone", "hERG")])
    s = SingleLevelSampler(packed, 2, [0], [1.0], 64, thinning=5, seed=1, device=gpu)
    s.init([6.0, 0.8, 8.0])
    good = s.state
    s.state = torch.zeros(good.shape, dtype=torch.float64).pin_memory()
    with pytest.raises(_lib.PhfError, match="device memory"):
        s.advance(10, save=False)
    s.state = good
    s.enable_moments(0)
    s.moments = torch.zeros(s.moments.shape, dtype=torch.float64).pin_memory()
    with pytest.raises(_lib.PhfError, match="moments"):
        s.advance(10, save=False)
    s.moments = None
    assert s.advance(10).shape == (2, 1, 4, 64)                    # and the sampler is still usable
    ex = dr.load_crumb_data("Amiodarone", "hERG")[2]
    hs = H.HierarchicalSampler(H.PackedHierPoints([ex]), [0], 64, device=gpu)
    hs.init(np.concatenate([[1., 5., 6., .3], np.tile([6.0, 0.8], 3), [8.0]])[None])
    hs.state = torch.zeros(hs.state.shape, dtype=torch.float64).pin_memory()
    with pytest.raises(_lib.PhfError, match="device memory"):
        hs.advance(10, save=False)


def test_fine_grained_moments_buffer_is_refused(gpu, dr):
    """the memory-kind check also refuses FINE-GRAINED device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)),
    on which the hardware fp64 atomics of the hierarchical moments are not guaranteed — allocated here through the HIP runtime the
    process already has loaded, handed to the C ABI as the moments buffer; and the verdict cache (the last 8 (address, device) pairs)
    is emptied by an *_init, so that an address once accepted is asked about again for the next sampler"""
    import ctypes as C
    from pyhillfit_amd import _lib
    from pyhillfit_amd import hierarchical as H
    hip_path = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                hip_path = line.split()[-1]
                break
    assert hip_path, "the HIP runtime is not mapped?"
    hip = C.CDLL(hip_path)
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipFree.argtypes = [C.c_void_p]
    ex = dr.load_crumb_data("Amiodarone", "hERG")[2]
    hs = H.HierarchicalSampler(H.PackedHierPoints([ex]), [0], 64, device=gpu)
    hs.init(np.concatenate([[1., 5., 6., .3], np.tile([6.0, 0.8], 3), [8.0]])[None])
    hs.enable_moments(0)
    ptr = C.c_void_p()
    nbytes = hs.moments.numel() * 8
    assert hip.hipExtMallocWithFlags(C.byref(ptr), nbytes, 1) == 0 and ptr.value           # hipDeviceMallocFinegrained = 0x1
    try:
        cfg = hs._config(10)
        rc = hs.lib.phf_hierarchical_advance(C.byref(hs.points.struct), C.byref(hs.prob), C.byref(hs.prior), C.byref(cfg), 0, 10,
                                             hs.state.data_ptr(), None, ptr.value, 0, None)
        assert rc == -1 and b"moments" in hs.lib.phf_last_error() and b"fine-grained" in hs.lib.phf_last_error(), (rc, hs.lib.phf_last_error())
    finally:
        hip.hipFree(ptr)
    assert hs.advance(10).shape == (2, 1, 12, 64)                  # the sampler is still usable with its own (coarse-grained) buffers


def test_queued_launch_at_the_bench_shape(gpu, dr):
    """the C3 launch exactly as bench.py runs it — 210 pairs x 4 096 chains = 13 440 blocks over 2 048 persistent wavefronts, 2 000
    iterations in 4 quanta, three launches back to back — ends in the same state and moments as the plain launches"""
    from pyhillfit_amd.sampler import SingleLevelSampler
    names = _all_names(dr)
    packed = dr.pack_single_level(names)
    Q = len(names)
    got = []
    for quanta in (0, 4):
        s = SingleLevelSampler(packed, 2, list(range(Q)), [1.0] * Q, 4096, thinning=5, seed=25, device=gpu, queue_quanta=quanta)
        s.init([6.0, 0.8, 8.0], cov_identity=False, cov_scale=0.05)
        s.enable_moments(after_iteration=2000)
        for _ in range(3):
            s.advance(2000, save=False)
        if quanta:
            q_ = s._queue.cpu().numpy()
            assert q_[0] >= 13440 * 4 and q_[0] < 0x40000000 and np.all(q_[1:-1] == 4) and q_[-1] == 0
        got.append((s.state.clone(), s.moments.clone()))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    acc = got[1][0][14].view(Q, 4096) / 6000.0                     # accepted count / iterations (state row 2d + 2 + d(d+1)/2)
    assert 0.15 < float(acc.mean()) < 0.45


def test_one_long_queued_advance_is_cut_into_bounded_quanta(gpu, dr):
    """ONE advance() of 120 000 iterations on a queued launch (210 pairs x 1 024 chains = 3 360 blocks over 2 048 persistent wavefronts):
    the host cuts it into quanta of at most MAX_QUANTUM_ITERATIONS (30 here, not the 4 of `queue_quanta`), so that a wavefront's wait
    for its block's previous quantum stays far below the kernel's give-up limit however long the call — no fault word,
    every block's progress word at the quantum count, and the same final state, bit for bit, as twelve calls of 10 000."""
    from pyhillfit_amd.sampler import MAX_QUANTUM_ITERATIONS, SingleLevelSampler, queue_quantum
    names = _all_names(dr)
    packed = dr.pack_single_level(names)
    Q, T = len(names), 120000
    assert queue_quantum(T, 4, 5) == MAX_QUANTUM_ITERATIONS == 4000
    got = []
    for calls in (1, 12):
        s = SingleLevelSampler(packed, 2, list(range(Q)), [1.0] * Q, 1024, thinning=5, seed=25, device=gpu, queue_quanta=4)
        s.init([6.0, 0.8, 8.0], cov_identity=False, cov_scale=0.05)
        s.enable_moments(after_iteration=20000)
        s.reserve(T)
        for _ in range(calls):
            s.advance(T // calls, save=False)
        s.check_queue()                                               # raises if any wavefront gave up waiting
        q_ = s._queue.cpu().numpy()
        quanta_last_call = -(-(T // calls) // queue_quantum(T // calls, 4, 5))
        assert q_[-1] == 0 and np.all(q_[1:-1] == quanta_last_call) and q_[0] < 0x40000000, (calls, q_[0], q_[-1])
        got.append((s.state.clone(), s.moments.clone()))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])


# ------------------------------------------------------------------------------------- C3 / C4 at full width
def test_c3_full_width_shard_invariance_and_checkpoint(gpu, dr):
    """210 pairs x 4 096 chains (BASELINE configs[2]): one launch == the pairs split over 3 'ranks' by cost
    (distributed.shard_problems, global problem ids) == the chains split in two (chain_id_base) == stop / checkpoint / resume."""
    from pyhillfit_amd import distributed as D
    from pyhillfit_amd.sampler import SingleLevelSampler
    names = _all_names(dr)
    packed = dr.pack_single_level(names)
    Q, C, T = len(names), 4096, 40
    kw = dict(thinning=5, seed=25, adapt_start=15, device=gpu)
    full = SingleLevelSampler(packed, 2, list(range(Q)), [1.0] * Q, C, **kw)
    full.init([6.0, 0.8, 8.0])
    a = full.run(T)                                                # [9][210][4][4096]
    assert torch.isfinite(a).all()
    costs = packed.counts[:, 3]
    parts = D.shard_problems(costs, 3)
    assert sorted(np.concatenate(parts).tolist()) == list(range(Q))
    for mine in parts:
        sub = dr.pack_single_level([names[i] for i in mine])
        s = SingleLevelSampler(sub, 2, list(range(len(mine))), [1.0] * len(mine), C, problem_ids=mine, **kw)
        s.init([6.0, 0.8, 8.0])
        assert torch.equal(s.run(T), a[:, torch.as_tensor(mine, device=a.device)])
    for h in range(2):
        s = SingleLevelSampler(packed, 2, list(range(Q)), [1.0] * Q, C // 2, chain_id_base=h * (C // 2), **kw)
        s.init([6.0, 0.8, 8.0])
        assert torch.equal(s.run(T), a[..., h * (C // 2):(h + 1) * (C // 2)])
    # strong scaling's partition: (pair, 64-chain block) units by LPT over 8 'ranks' (distributed.shard_blocks: 1 680 blocks each);
    # a rank lists its blocks as problems of 64 chains (problem id = the pair, chain offset = 64 x block): same chains, bit for bit
    parts = D.shard_blocks(525.0 + 28.0 * packed.counts[:, 0] + 115.0 * (packed.counts[:, 1] + packed.counts[:, 2]), C // 64, 8)
    assert [len(u) for u in parts] == [1680] * 8
    blocks = a.view(a.shape[0], Q, 4, C // 64, 64)                  # [9][210][4][64 blocks][64 chains]
    for units in (parts[0], parts[5]):
        s = SingleLevelSampler(packed, 2, units[:, 0].tolist(), [1.0] * len(units), 64, problem_ids=units[:, 0], chain_offsets=64 * units[:, 1], **kw)
        s.init([6.0, 0.8, 8.0])
        u = torch.as_tensor(units, device=a.device)
        assert torch.equal(s.run(T), blocks[:, u[:, 0], :, u[:, 1], :].permute(1, 0, 2, 3).contiguous())
    r = SingleLevelSampler(packed, 2, list(range(Q)), [1.0] * Q, C, **kw)
    r.init([6.0, 0.8, 8.0])
    first = r.advance(25)
    r2 = SingleLevelSampler(packed, 2, list(range(Q)), [1.0] * Q, C, **kw)
    r2.load_state_dict(r.state_dict())
    assert torch.equal(torch.cat([first, r2.advance(15)]), a[1:])


def test_c4_full_width_shard_invariance_and_checkpoint(gpu, dr):
    """hierarchical, 210 pairs x 1 024 chains (BASELINE configs[3]), every Ne group: chain shards (8 'ranks' of 128 chains)
    and a checkpointed run reproduce the one-launch run bit for bit"""
    from pyhillfit_amd import bestfit
    from pyhillfit_amd import hierarchical as H
    shapes, scales, locs = H.prior_params()
    groups = {}
    for dname, c in _all_names(dr):
        ne, _, ex = dr.load_crumb_data(dname, c)
        groups.setdefault(len(ex), []).append(ex)
    C, T = 1024, 40
    total = 0
    for ne, exs in sorted(groups.items()):
        packed = H.PackedHierPoints(exs)
        theta0 = np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs])
        kw = dict(thinning=5, seed=25, adapt_start=15, device=gpu)
        full = H.HierarchicalSampler(packed, list(range(len(exs))), C, **kw)
        full.init(theta0, cov_scale=0.01)
        a = torch.cat([full.advance(25), full.advance(15)])     # [8][Q][dim+1][1024]
        assert torch.isfinite(a).all()
        for r in (0, 3, 7):
            s = H.HierarchicalSampler(packed, list(range(len(exs))), C // 8, chain_id_base=r * (C // 8), **kw)
            s.init(theta0, cov_scale=0.01)
            assert torch.equal(s.advance(T), a[..., r * (C // 8):(r + 1) * (C // 8)])
        # a strong-scaling rank's share: (pair, 64-chain block) units as problems of 64 chains with chain offsets
        from pyhillfit_amd import distributed as D
        units = D.shard_blocks(packed.expt_start[:, -1].astype(float), C // 64, 8)[3]
        s = H.HierarchicalSampler(packed, units[:, 0].tolist(), 64, problem_ids=units[:, 0], chain_offsets=64 * units[:, 1], **kw)
        s.init(theta0[units[:, 0]], cov_scale=0.01)
        u = torch.as_tensor(units, device=a.device)
        blocks = a.view(a.shape[0], len(exs), a.shape[2], C // 64, 64)
        assert torch.equal(s.advance(T), blocks[:, u[:, 0], :, u[:, 1], :].permute(1, 0, 2, 3).contiguous())
        one = H.HierarchicalSampler(packed, list(range(len(exs))), C, **kw)
        one.init(theta0, cov_scale=0.01)
        assert torch.equal(one.advance(T), a) and torch.equal(one.state, full.state)
        total += len(exs)
    assert total == 210


# ------------------------------------------------------------------------------------- C5 at full size
def test_c5_full_ladder_all_pairs(gpu, dr):
    """BASELINE configs[4]: 32 rungs t_i = (i/31)^3 (python/PyHillTemp.py:151 with n = 31) x 210 pairs x 1 024 chains =
    6.9 M chains, PyHillTemp's start (ones, identity covariance, mean reset), 200 000 iterations like the reference's runs
    behind golden G5c, moments on the device:
      * rung t = 0 samples the prior alone (doseresponse.py:230-231): pooled over its 210 x 1 024 chains the means are those of
        the reference's own t = 0 runs of this length (G5d) — for Hill and sigma also the analytic (5, 7.49975) — and the sds the
        analytic (5, 2.887, 3.354);
      * rung t = 1 reproduces the reference sampler's posterior means of EVERY pair and column (G5c, G5d: PyHillTemp.do_mcmc);
      * per pair, <log L(theta; t=1)> under the tempered posteriors rises along the ladder (its t-derivative is a variance):
        what python/compute_bayes_factors.py integrates."""
    from pyhillfit_amd.sampler import SingleLevelSampler
    with open(os.path.join(GOLDEN, "g5c_posteriors_all_pairs_model_2.json")) as f:
        g5c = json.load(f)
    names = [(w["drug"], w["channel"]) for w in g5c]
    assert len(names) == 210
    packed = dr.pack_single_level(names)
    ladder = dr.temperature_ladder(31)
    R, P, C = len(ladder), len(names), 1024
    assert R == 32 and ladder[0] == 0.0 and ladder[-1] == 1.0
    pair_index = [p for p in range(P) for _ in range(R)]
    temps = [float(t) for _ in range(P) for t in ladder]
    s = SingleLevelSampler(packed, 2, pair_index, temps, C, thinning=5, seed=25, reset_mean_at_adapt_start=True, device=gpu)
    s.init(np.ones(3), cov_identity=True, cov_scale=1.0)
    s.enable_moments(after_iteration=50000)
    for _ in range(8):                                             # 8 launches of 25 000 iterations
        s.advance(25000, save=False)
    mean, var, n = s.posterior_moments()                           # [4][P*R][C]
    assert n == 30000
    mean = mean.view(4, P, R, C); var = var.view(4, P, R, C)
    # t = 0: the prior
    m0 = mean[:3, :, 0, :].reshape(3, -1)
    prior_mean = m0.mean(dim=1).cpu().numpy()
    prior_sd = torch.sqrt(var[:3, :, 0, :].reshape(3, -1).mean(dim=1) + m0.var(dim=1)).cpu().numpy()
    print("C5 t=0 rung: pooled mean", prior_mean, "sd", prior_sd)
    # the exponential pIC50 tail (mean 5 beyond -3) is explored slowly by a random walk started at 1: at THIS run length the
    # reference's own sampler averages pIC50 to 1.893 +- 0.013 (golden G5d: eight seeds of PyHillTemp.do_mcmc at t = 0), not to the
    # analytic 2 — so the rung is held to the reference run (1 % + 4 of its standard errors); Hill and sigma also to their analytic
    # values.  (test_prior_only_rung_known_answer runs twice as long and pins all three to the analytic prior.)
    with open(os.path.join(GOLDEN, "g5d_posteriors_reseeded.json")) as f:
        ref0 = [e for e in json.load(f) if e["temperature"] == 0.0 and e["model"] == 2][0]
    r_mean, r_se = np.array(ref0["mean"][:3]), np.maximum(ref0["se_between_seeds"][:3], np.array(ref0["se_single_chain_batch_means"][:3]) / np.sqrt(len(ref0["seeds"])))
    assert np.all(np.abs(prior_mean - r_mean) <= 0.01 * np.abs(r_mean) + 4 * r_se), (prior_mean, r_mean, r_se)
    assert np.all(np.abs(prior_mean[1:] - [5.0, 7.49975]) <= [0.03, 0.05]), prior_mean
    assert np.all(np.abs(prior_sd - [5.0, 10 / np.sqrt(12), np.sqrt(5) * 1.49975]) <= [0.35, 0.03, 0.06]), prior_sd
    # t = 1: the reference sampler's posteriors, all pairs, every (pair, column) within 1 % + 4 standard errors
    from conftest import reference_posteriors
    names_ref, want, se, _, _ = reference_posteriors(2)
    assert names_ref == names
    pooled = mean[:, :, R - 1, :].mean(dim=2).cpu().numpy()        # [4][P]
    ratio = np.abs(pooled - want) / (0.01 * np.abs(want) + 4 * se)
    worst = np.unravel_index(np.argmax(ratio), ratio.shape)
    print("C5 t=1 rung vs G5c+G5d: worst %.2f at %s column %d" % (ratio.max(), names[worst[1]], worst[0]))
    assert ratio.max() < 1.0, (ratio.max(), names[worst[1]], worst[0], pooled[worst], want[worst], se[worst])
    # <log L> along the ladder
    ll = s.mean_log_likelihood_t1().view(P, R, C)
    e = ll.mean(dim=2).cpu().numpy()                               # [P][R]
    e_se = (ll.std(dim=2) / np.sqrt(C)).cpu().numpy()
    assert np.isfinite(e).all()
    step = np.diff(e, axis=1)
    slack = 4 * np.sqrt(e_se[:, 1:] ** 2 + e_se[:, :-1] ** 2) + 1e-9 * np.abs(e[:, 1:])
    bad = step < -slack
    print("C5 ladder: %d of %d steps fall by more than 4 standard errors; largest fall %.3g" % (bad.sum(), bad.size, (-step).max()))
    assert bad.mean() <= 0.002, (bad.sum(), np.argwhere(bad)[:5])
    acc = s.acceptance().view(P, R, C).mean(dim=2).cpu().numpy()
    assert 0.1 < acc.min() and acc.max() < 0.6, (acc.min(), acc.max())


def test_synthetic_s3_sampled_rows_recomputed_by_the_numpy_oracle(gpu):
    """SURVEY 8(d)'s synthetic scaling set (pyhillfit_amd/synthetic.py; bench.py --workload s3): 1 680 generated pairs x 64 chains,
    model 2: sampled rows' log-targets recomputed by the independent numpy oracle, the censoring rate of the generated responses, and
    chains that move"""
    from oracle import pyhillfit_oracle as orc
    from pyhillfit_amd import doseresponse as d
    from pyhillfit_amd import synthetic as S
    from pyhillfit_amd.sampler import SingleLevelSampler
    ex, truth = S.generate(1680)
    assert 0.21 < S.censoring_rate(ex) < 0.26
    pairs = S.single_level_pairs(ex)
    packed = d.PackedPoints(pairs)
    s = SingleLevelSampler(packed, 2, list(range(1680)), [1.0] * 1680, 64, thinning=5, seed=12, adapt_start=300, device=gpu)
    s.init([6.0, 0.8, 8.0], cov_identity=False, cov_scale=0.05)
    rows = s.advance(1500).cpu().numpy()
    assert np.isfinite(rows).all()
    rng = np.random.default_rng(2)
    worst = 0.0
    for q in rng.choice(1680, 300, replace=False):
        concs, y = pairs[q]
        pair = orc.PairData(concs, y, ex[q])
        for r, c in zip(rng.integers(0, rows.shape[0], 3), rng.integers(0, 64, 3)):
            th, got = rows[r, q, :3, c], rows[r, q, 3, c]
            worst = max(worst, abs(got - orc.log_target(2, pair, th, 1.0)) / (abs(orc.log_target(2, pair, th, 1.0)) + 1.0))
    assert worst <= 1e-12, worst
    assert (np.abs(np.diff(rows[:, :, 0, :], axis=0)) > 0).mean() > 0.1
    # the chains find the generating parameters: pooled pIC50 of the last rows within a few posterior widths of the truth for most pairs
    est = rows[-100:, :, 0, :].mean(axis=(0, 2))
    assert np.median(np.abs(est - truth["pic50"])) < 0.15
