"""The hand-allocated gfx950 code object (pyhillfit_amd/csrc/generated/phf_hier3_gfx950.s, emitted by tools/gen_hier_isa.py): every
elementary function of tools/isa/phf_isa_math.py, evaluated by a unit kernel of that code object, must give the bits of its C
namesake in pyhillfit_amd/csrc/phf_math.h — compared with the hipcc build of the same function on the device (phf_debug_math), which
tests/test_gpu_parity.py in turn holds to the host build and the twin."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return "cuda:0"


def _bits_equal(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_isa_elementary_functions_bit_identical_to_the_hipcc_build(gpu):
    from pyhillfit_amd.sampler import debug_isa, debug_math
    rng = np.random.default_rng(7)
    n = 200000
    # exp: the whole clamped range and beyond, zeros, infinities
    x = np.concatenate([rng.uniform(-760, 720, n), rng.normal(0, 3, n), [0.0, -0.0, 709.9, 710.0, 711.0, -745.9, -746.0, -800.0, np.inf, -np.inf, 1e-300]])
    assert _bits_equal(debug_isa(0, x, gpu), debug_math(9, x, gpu)), "phf_exp_fast_k"
    xc = x[x <= 709.0]
    assert _bits_equal(debug_isa(1, xc, gpu), debug_math(9, xc, gpu)), "phf_exp_capped_k"
    # log: positive normal numbers over the whole exponent range, near 1, powers of two
    xp = np.concatenate([np.exp(rng.uniform(-700, 700, n)), 1.0 + rng.normal(0, 1e-3, n), 2.0 ** rng.integers(-1000, 1000, 1000), [1.0, 2.0 ** -1022, 1.7976931348623157e308]])
    assert _bits_equal(debug_isa(2, xp, gpu), debug_math(10, xp, gpu)), "phf_log_pos_k"
    xl = np.concatenate([xp, [0.0, -0.0, -1.0, 1e-310, 2.0 ** -1023, -np.inf]])
    assert _bits_equal(debug_isa(3, xl, gpu), debug_math(10, xl, gpu)), "phf_log_fast_k"
    # erfc table: the table's range, the cut, beyond it, negatives (clamped index: any value is safe)
    y = np.concatenate([rng.uniform(0, 6.5, n), rng.uniform(-1, 40, n), [0.0, 5.999999, 6.0, 6.000001, 1e5, 1e300]])
    assert _bits_equal(debug_isa(4, y, gpu), debug_math(19, y, gpu)), "phf_erfc_tab"
    # reciprocal, square root
    xr = np.concatenate([np.exp(rng.uniform(-400, 400, n)) * rng.choice([-1.0, 1.0], n), [1.0, -1.0, 3.0]])
    assert _bits_equal(debug_isa(5, xr, gpu), debug_math(12, xr, gpu)), "phf_rcp"
    xs = np.concatenate([np.exp(rng.uniform(-400, 400, n)), [0.0, -0.0, -1.0, 4.0]])
    assert _bits_equal(debug_isa(6, xs, gpu), debug_math(15, xs, gpu)), "phf_sqrt_nonneg"
    # normals and the accept uniform's logarithm from 32-bit words
    w = np.concatenate([rng.integers(0, 2 ** 32, n, dtype=np.uint64), [0, 1, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1]]).astype(np.uint32)
    assert _bits_equal(debug_isa(7, w, gpu), debug_math(17, w.astype(np.float64), gpu)), "phf_normal_u32"
    u = (w.astype(np.float64) + 0.5) * 2.0 ** -32                               # phf_unit_open32: exact in fp64
    assert _bits_equal(debug_isa(8, w, gpu), debug_math(10, u, gpu)), "phf_log_pos_k(phf_unit_open32(w))"


def test_isa_philox_known_answers(gpu):
    from pyhillfit_amd.sampler import debug_isa, debug_philox
    rng = np.random.default_rng(11)
    for key in ((0, 0), (0xffffffff, 0xffffffff), (0xa4093822, 0x299f31d0), (25, 0)):
        ck = rng.integers(0, 2 ** 32, (5000, 6), dtype=np.uint64).astype(np.uint32)
        ck[:8, :4] = [[0, 0, 0, 0], [0xffffffff] * 4, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [1, 2, 3, 4],
                      [5, 0, 77, 2], [0, 7, 0, 1], [9, 9, 9, 0], [64, 209, 24000, 2]]
        ck[:, 4], ck[:, 5] = key
        assert np.array_equal(debug_isa(9, ck, gpu), debug_philox(ck, gpu, rounds=7)), key


# ---------------------------------------------------------------------------------------------------------------------------------
# phf_hier3_advance: the hand-allocated Ne = 3 / 4 + 4 + 4 iteration against the hipcc one-lane kernel and the scalar twin
def _setup(gpu, names, oracle_pair, C, thin, adapt, seed, isa, chain_offsets=None, launch_order=None):
    from pyhillfit_amd import hierarchical as H
    pairs = [oracle_pair(d, c) for d, c in names]
    packed = H.PackedHierPoints([p.experiments for p in pairs])
    assert (packed.n_expts, packed.points_per_expt) in H.ISA_SHAPES
    Q = len(names)
    s = H.HierarchicalSampler(packed, list(range(Q)), C, thinning=thin, seed=seed, adapt_start=adapt, problem_ids=[7 + 3 * q for q in range(Q)],
                              chain_id_base=9, chain_offsets=chain_offsets, device=gpu)
    if launch_order is not None:
        import torch
        s.launch_order = torch.tensor(np.asarray(launch_order, dtype=np.int32), device=gpu)
        s.prob.launch_order = s.launch_order.data_ptr()
    s.set_kernel_hint(lanes=1, isa=isa)
    return s, pairs


THETA0 = [np.concatenate([[1., 5., 6., .3], np.tile([6.0, 0.8], 3), [8.0]]), np.concatenate([[1.2, 4., 5., .4], np.tile([4.5, 1.1], 3), [5.0]]),
          np.concatenate([[0.9, 3., 5.5, .25], np.tile([5.2, 1.4], 3), [6.5]])]
UNIFORM4 = [("Amiodarone", "hERG"), ("Amitriptyline", "Cav1.2"), ("Azithromycin", "Nav1.5-late")]


@pytest.mark.parametrize("C,thin,cuts", [(256, 5, (137, 9, 454)), (70, 1, (100, 201)), (1024, 5, (600,)), (64, 5, (150, 100))])
def test_isa_advance_bit_identical_to_the_hipcc_kernel(C, thin, cuts, gpu, oracle_pair):
    """same launches through kernel_hint bit 4 (hipcc one-lane kernel) and without it (phf_hier3_advance): rows, final state and
    moments must agree bit for bit — launches cut before, at and after the start of the adaptation, a ragged last wavefront
    (70 chains), thinning 1 and 5, a permuted launch order, per-problem chain offsets"""
    adapt = 140
    got = {}
    for isa in (False, True):
        s, _ = _setup(gpu, UNIFORM4, oracle_pair, C, thin, adapt, 424242, isa, chain_offsets=[0, 64, 640], launch_order=[2, 0, 1])
        s.init(np.array(THETA0), cov_scale=0.01)
        s.enable_moments(after_iteration=adapt + 10)
        chain = np.concatenate([s.advance(k).cpu().numpy() for k in cuts])
        from pyhillfit_amd.hierarchical import last_kernel
        assert last_kernel() == (4 if isa else 1), last_kernel()          # the kernel meant is the kernel that ran
        got[isa] = (chain, s.state.cpu().numpy(), s.moments.cpu().numpy())
    for name, a, b in zip(("rows", "state", "moments"), got[False], got[True]):
        same = a.view(np.uint64) == b.view(np.uint64)
        assert same.all(), (name, int((~same).sum()), np.argwhere(~same)[:5].tolist())
    acc = got[True][1][-1].mean() / sum(cuts)
    assert 0.01 < acc < 0.95, acc


OTHER_SHAPES = {"2 + 2 + 2": [("Rufinamide", "hERG"), ("Rufinamide", "Cav1.2"), ("Rufinamide", "Kir2.1")],
                "5 + 5 + 4": [("Verapamil", "hERG")] * 3}


@pytest.mark.parametrize("shape", sorted(OTHER_SHAPES))
def test_isa_other_point_shapes_bit_identical_to_the_hipcc_kernel_and_the_twin(shape, gpu, oracle_pair):
    """the code object's kernels for the other point shapes of the Crumb set's three-experiment pairs (2 + 2 + 2: a half without points;
    5 + 5 + 4: a half with a pair AND an odd point): rows, state and moments against the hipcc one-lane kernel (run-time point loops)
    bit for bit, launches cut around the start of the adaptation, a ragged last wavefront; four chains against the scalar twin"""
    from oracle import c_oracle as co
    from pyhillfit_amd import hierarchical as H
    from pyhillfit_amd.sampler import gamma_table
    names = OTHER_SHAPES[shape]
    C, thin, adapt, cuts = 200, 5, 140, (137, 9, 354)
    got = {}
    for isa in (False, True):
        s, pairs = _setup(gpu, names, oracle_pair, C, thin, adapt, 20240229, isa)
        s.init(np.array(THETA0), cov_scale=0.01)
        s.enable_moments(after_iteration=adapt + 10)
        chain = np.concatenate([s.advance(k).cpu().numpy() for k in cuts])
        assert H.last_kernel() == (4 if isa else 1), H.last_kernel()
        got[isa] = (chain, s.state.cpu().numpy(), s.moments.cpu().numpy())
    for name, a, b in zip(("rows", "state", "moments"), got[False], got[True]):
        same = a.view(np.uint64) == b.view(np.uint64)
        assert same.all(), (shape, name, int((~same).sum()), np.argwhere(~same)[:5].tolist())
    acc = got[True][1][-1].mean() / sum(cuts)
    assert 0.01 < acc < 0.95, acc
    shapes, scales, locs = H.prior_params()
    gam = gamma_table(sum(cuts))
    state = got[True][1].reshape(-1, 3, C)
    for q in (0, 2):
        pk = co.PackedHierPair(pairs[q].experiments, shapes, scales, locs)
        for c in (0, C - 1):
            st = pk.init_state(THETA0[q], 0.01)
            rows = pk.advance(st, 0, sum(cuts), thin, adapt, gam, seed=20240229, chain_id=9 + c, problem_id=7 + 3 * q)
            assert np.array_equal(got[True][0][:, q, :, c], rows), (shape, q, c)
            assert np.array_equal(state[:, q, c], st), (shape, q, c)


NE4 = {"4 + 4 + 4 + 1": [("Amiodarone", "Nav1.5-peak"), ("Amitriptyline", "KvLQT1/mink"), ("Azithromycin", "KvLQT1/mink")],
       "4 + 4 + 4 + 2": [("Bepridil", "Nav1.5-peak"), ("Dofetilide", "Cav1.2"), ("Propafenone", "hERG")],
       "4 + 4 + 4 + 3": [("Amiodarone", "Kv4.3"), ("Saquinavir", "hERG"), ("Amiodarone", "Kv4.3")],
       "2 + 2 + 2 + 1": [("Rufinamide", "KvLQT1/mink")] * 3, "5 + 5 + 5 + 1": [("Verapamil", "Cav1.2")] * 3}
THETA0_4 = [np.concatenate([t[:4], np.tile(t[4:6], 4), t[-1:]]) for t in THETA0]
NE5 = {"4 + 4 + 4 + 1 + 1": [("Lopinavir", "KvLQT1/mink"), ("Mibefradil", "KvLQT1/mink"), ("Mibefradil", "Kv4.3")],
       "4 + 4 + 4 + 2 + 1": [("Azithromycin", "Cav1.2"), ("Lidocaine", "Cav1.2"), ("Quinine", "Kv4.3")],
       "4 + 4 + 4 + 4 + 4": [("Moxifloxacin", "KvLQT1/mink")] * 3, "5 + 5 + 4 + 2 + 2": [("Dofetilide", "hERG")] * 3}
THETA0_5 = [np.concatenate([t[:4], np.tile(t[4:6], 5), t[-1:]]) for t in THETA0]
NE6 = {"4 + 4 + 4 + 1 + 1 + 1": [("Chloroquine", "Kv4.3"), ("Cibenzoline", "Kv4.3"), ("Chloroquine", "Kv4.3")], "4 + 4 + 4 + 4 + 2 + 1": [("Amitriptyline", "Kv4.3")] * 3}
THETA0_6 = [np.concatenate([t[:4], np.tile(t[4:6], 6), t[-1:]]) for t in THETA0]


@pytest.mark.parametrize("shape", ["4 + 4 + 4 + 1", "4 + 4 + 4 + 2", "4 + 4 + 4 + 3"])         # (the shapes with a kernel of their own)
def test_isa_four_experiments_bit_identical_to_the_hipcc_kernel_and_the_twin(shape, gpu, oracle_pair):
    """phf_hier4_advance_*: the Ne = 4 iteration (13 parameters, 135 doubles of state per chain) at two wavefronts per SIMD — 27 elements
    of L in registers, 9 + mean + d in LDS, rows 9..12 of L in a device-memory scratch tier inside the queue workspace (ABI 7:
    phf_hierarchical_queue_words, kernel_hint bit 6).  Rows, state and moments against the hipcc one-lane kernel bit for bit: plain launches
    cut around the start of the adaptation, a ragged last wavefront; four chains against the scalar twin"""
    from oracle import c_oracle as co
    from pyhillfit_amd import hierarchical as H
    from pyhillfit_amd.sampler import gamma_table
    names = NE4[shape]
    C, thin, adapt, cuts = 200, 5, 140, (137, 9, 354)
    got = {}
    for isa in (False, True):
        s, pairs = _setup(gpu, names, oracle_pair, C, thin, adapt, 20240301, isa)
        assert s.queue.numel() > 2 + s.nblocks + 42 * 128          # the scratch tier is in the workspace
        s.init(np.array(THETA0_4), cov_scale=0.01)
        s.enable_moments(after_iteration=adapt + 10)
        chain = np.concatenate([s.advance(k).cpu().numpy() for k in cuts])
        assert H.last_kernel() == (4 if isa else 1), H.last_kernel()
        got[isa] = (chain, s.state.cpu().numpy(), s.moments.cpu().numpy())
    for name, a, b in zip(("rows", "state", "moments"), got[False], got[True]):
        same = a.view(np.uint64) == b.view(np.uint64)
        assert same.all(), (shape, name, int((~same).sum()), np.argwhere(~same)[:5].tolist())
    acc = got[True][1][-1].mean() / sum(cuts)
    assert 0.01 < acc < 0.95, acc
    shapes, scales, locs = H.prior_params()
    gam = gamma_table(sum(cuts))
    state = got[True][1].reshape(-1, 3, C)
    for q in (0, 1):
        pk = co.PackedHierPair(pairs[q].experiments, shapes, scales, locs)
        for c in (0, C - 1):
            st = pk.init_state(THETA0_4[q], 0.01)
            rows = pk.advance(st, 0, sum(cuts), thin, adapt, gam, seed=20240301, chain_id=9 + c, problem_id=7 + 3 * q)
            assert np.array_equal(got[True][0][:, q, :, c], rows), (shape, q, c)
            assert np.array_equal(state[:, q, c], st), (shape, q, c)


def test_isa_four_experiments_work_queue_at_full_width(gpu):
    """all 32 Crumb pairs with 4 + 4 + 4 + 1 points x 4 160 chains = 2 080 blocks on 2 048 wavefront slots: the launch runs as a work queue
    (a block's quanta chain through its state in HBM, the scratch tier belongs to the resident wavefront, not to the block) — rows, state
    and moments against the hipcc kernel's plain launch, bit for bit"""
    import os
    import torch
    from conftest import REPO
    from pyhillfit_amd import bestfit, doseresponse as dr, hierarchical as H
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    shapes, scales, locs = H.prior_params()
    exs = []
    for d in dr.drugs:
        for c in dr.channels:
            ne, _, ex = dr.load_crumb_data(d, c)
            if H.group_key(ex, H.ISA_SHAPES) == (4, H.shape_code((4, 4, 4, 1))):
                exs.append(ex)
    assert len(exs) == 32
    packed = H.PackedHierPoints(exs)
    theta0 = np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs])
    got = {}
    for isa in (False, True):
        s = H.HierarchicalSampler(packed, list(range(len(exs))), 4160, thinning=5, seed=78, adapt_start=500, device=gpu)
        s.set_kernel_hint(lanes=1, isa=isa)
        s.init(theta0, cov_scale=0.01)
        s.enable_moments(after_iteration=600)
        s.advance(400, save=False)
        assert H.last_kernel() == (5 if isa else 1), H.last_kernel()
        rows = torch.cat([s.advance(k) for k in (300, 200)])
        torch.cuda.synchronize()
        s.check_queue()
        got[isa] = (rows, s.state.clone(), s.moments.clone())
    for name, a, b in zip(("rows", "state", "moments"), got[False], got[True]):
        same = a.view(torch.int64) == b.view(torch.int64)
        assert bool(same.all()), (name, int((~same).sum()))


def test_fused_launch_of_fourteen_groups_bit_identical_to_separate_launches(gpu, oracle_pair):
    """phf_hierarchical_advance_fused: the fourteen kinds of launch groups (Ne = 3: 4 + 4 + 4, 2 + 2 + 2, 5 + 5 + 4; Ne = 4: 4 + 4 + 4 + 1 / 2 / 3, 2 + 2 + 2 + 1,
    5 + 5 + 5 + 1; Ne = 5: 4 + 4 + 4 + 1 + 1, 4 + 4 + 4 + 2 + 1, 4 + 4 + 4 + 4 + 4, 5 + 5 + 4 + 2 + 2; Ne = 6: 4 + 4 + 4 + 1 + 1 + 1, 4 + 4 + 4 + 4 + 2 + 1 — every one of the 210 Crumb pairs has one of these fourteen shapes) in ONE
    persistent grid pulling from one queue — rows, state and moments of every group against its own launches of the hipcc kernel, bit for bit;
    quanta of 50 iterations (a block's quanta chain through its state, tasks of all groups interleave), a ragged last wavefront, a subset of
    the groups (bodies without a group), moments, and a fused launch continued by separate ones"""
    import torch
    from pyhillfit_amd import hierarchical as H
    groups = [(UNIFORM4, THETA0), (OTHER_SHAPES["2 + 2 + 2"], THETA0), (OTHER_SHAPES["5 + 5 + 4"], THETA0),
              (NE4["4 + 4 + 4 + 1"], THETA0_4), (NE4["4 + 4 + 4 + 2"], THETA0_4), (NE4["4 + 4 + 4 + 3"], THETA0_4),
              (NE4["2 + 2 + 2 + 1"], THETA0_4), (NE4["5 + 5 + 5 + 1"], THETA0_4)] + [(NE5[k_], THETA0_5) for k_ in sorted(NE5)] + [(NE6[k_], THETA0_6) for k_ in sorted(NE6)]
    C, thin, adapt, cuts = 200, 5, 140, (135, 10, 355)

    def make(isa, which):
        out = []
        for j in which:
            names, th0 = groups[j]
            s, _ = _setup(gpu, names, oracle_pair, C, thin, adapt, 777, isa)
            s.problem_ids = torch.tensor([100 * j + q for q in range(3)], dtype=torch.int32, device=gpu)
            s.prob.problem_id = s.problem_ids.data_ptr()
            s.init(np.array(th0), cov_scale=0.01)
            s.enable_moments(after_iteration=adapt + 10)
            out.append(s)
        return out

    for which in (range(14), (1, 4, 9), (0, 12, 13)):
        ref = make(False, which)
        want = [torch.cat([s.advance(k) for k in cuts]) for s in ref]
        assert H.last_kernel() == 1
        fus = make(True, which)
        f = H.FusedSamplers(fus)
        f.quantum = 50
        parts = [f.advance(k) for k in cuts[:2]]
        assert H.last_kernel() == 6, H.last_kernel()
        last = [s.advance(cuts[2]) for s in fus]                       # ... continued by the samplers' own launches
        if len(fus) == 3:                                              # a launch the one grid does not take (it starts between two saved rows) falls
            f.advance(3, save=False)                                   # back to the samplers' own launches: the same numbers
            assert H.last_kernel() == 6
            f.advance(2, save=False)
            assert H.last_kernel() != 6
            [r.advance(5, save=False) for r in ref]
        torch.cuda.synchronize()
        f.check_queue()
        for j, (s, r) in enumerate(zip(fus, ref)):
            got = torch.cat([parts[0][j], parts[1][j], last[j]])
            for name, a, b in (("rows", got, want[j]), ("state", s.state, r.state), ("moments", s.moments, r.moments)):
                same = a.view(torch.int64) == b.view(torch.int64)
                assert bool(same.all()), (list(which), j, name, int((~same).sum()))


def test_isa_advance_bit_identical_to_the_twin(gpu, oracle_pair):
    from oracle import c_oracle as co
    from pyhillfit_amd import hierarchical as H
    from pyhillfit_amd.sampler import gamma_table
    shapes, scales, locs = H.prior_params()
    C, T, thin, adapt = 130, 500, 5, 150
    s, pairs = _setup(gpu, UNIFORM4[:2], oracle_pair, C, thin, adapt, 987654321, True)
    s.init(np.array(THETA0[:2]), cov_scale=0.01)
    chain = np.concatenate([s.advance(k).cpu().numpy() for k in (adapt - 3, 4, T - adapt - 1)])
    assert H.last_kernel() == 4
    state = s.state.cpu().numpy().reshape(s.S, 2, C)
    gam = gamma_table(T)
    for q in range(2):
        pk = co.PackedHierPair(pairs[q].experiments, shapes, scales, locs)
        for c in (0, 63, 64, C - 1):
            st = pk.init_state(THETA0[q], 0.01)
            rows = pk.advance(st, 0, T, thin, adapt, gam, seed=987654321, chain_id=9 + c, problem_id=7 + 3 * q)
            assert np.array_equal(chain[:, q, :, c], rows), (q, c)
            assert np.array_equal(state[:, q, c], st), (q, c)


def test_isa_on_the_synthetic_set_bit_identical_to_the_hipcc_kernel_and_the_twin(gpu):
    """bench.py --workload s3h in small: 48 pairs of SURVEY 8(d)'s generated set S3 (three experiments of four points, ~23 % of the
    responses exactly 0 or 100: the censored ends of the hierarchical likelihood's truncated Gaussian on every pair) x 192 chains through
    phf_hier3_advance and through the hipcc kernel — rows, state and moments bit for bit — and chains of three pairs against the twin"""
    from oracle import c_oracle as co
    from pyhillfit_amd import hierarchical as H, synthetic
    from pyhillfit_amd.sampler import gamma_table
    exs = synthetic.generate(48)[0]
    assert 0.15 < synthetic.censoring_rate(exs) < 0.35
    shapes, scales, locs = H.prior_params()
    start = np.array([H.first_iteration(e, locs) for e in exs])
    C, T, thin, adapt = 192, 400, 5, 120
    got = {}
    for isa in (False, True):
        s = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(len(exs))), C, thinning=thin, seed=31337, adapt_start=adapt, chain_id_base=5, device=gpu)
        s.set_kernel_hint(lanes=1, isa=isa)
        s.init(start, cov_scale=0.01)
        s.enable_moments(after_iteration=adapt)
        chain = np.concatenate([s.advance(k).cpu().numpy() for k in (adapt + 7, T - adapt - 7)])
        assert H.last_kernel() == (4 if isa else 1)
        got[isa] = (chain, s.state.cpu().numpy(), s.moments.cpu().numpy())
    for name, a, b in zip(("rows", "state", "moments"), got[False], got[True]):
        same = a.view(np.uint64) == b.view(np.uint64)
        assert same.all(), (name, int((~same).sum()), np.argwhere(~same)[:5].tolist())
    chain = got[True][0]
    assert np.isfinite(chain).all()
    gam = gamma_table(T)
    for q in (0, 17, 47):
        pk = co.PackedHierPair(exs[q], shapes, scales, locs)
        for c in (0, C - 1):
            st = pk.init_state(start[q], 0.01)
            rows = pk.advance(st, 0, T, thin, adapt, gam, seed=31337, chain_id=5 + c, problem_id=q)
            assert np.array_equal(chain[:, q, :, c], rows), (q, c)


def test_isa_randomized_launch_shapes_bit_identical_to_the_hipcc_kernels(gpu):
    """twelve seeded random launch shapes of the generated set — pairs, chains per pair (ragged last wavefronts, more blocks than the chip
    has wavefront slots so that the work queue runs, fewer so that it does not), thinning, start of the adaptation, where the launches are
    cut, start of the moments, per-problem chain offsets, a permuted launch order, tempered problems — through the hierarchical assembly
    kernel and the hipcc one-lane kernel, and (model 2, no moments) through the single-level assembly kernel and the hipcc kernel: rows,
    states, moments bit for bit, and the kernel meant is the kernel that ran"""
    import torch
    from pyhillfit_amd import doseresponse as dr, hierarchical as H, synthetic
    from pyhillfit_amd.sampler import SingleLevelSampler
    rng = np.random.default_rng(20261005)
    shapes, scales, locs = H.prior_params()
    for case in range(12):
        P = int(rng.integers(1, 40))
        exs = synthetic.generate(P, seed=1000 + case)[0]
        C = int(rng.choice([1, 63, 64, 65, 130, 200, 256, 1000, 4096]))
        thin = int(rng.choice([1, 2, 5, 7]))
        adapt = int(rng.integers(0, 90))
        cuts = [int(x) for x in rng.integers(1, 160, size=int(rng.integers(1, 4)))]
        order = rng.permutation(P).astype(np.int32)
        offsets = [int(x) * 64 for x in rng.integers(0, 5, size=P)]
        seed = int(rng.integers(1, 2 ** 40))
        mom_after = int(rng.integers(0, sum(cuts) + 1))
        quanta = int(rng.choice([0, 3]))
        temps = [float(t) for t in rng.choice([1.0, 0.5, 0.0], size=P)]
        what = (case, P, C, thin, adapt, cuts, mom_after, quanta)
        # ---- hierarchical
        start = np.array([H.first_iteration(e, locs) for e in exs])
        got = {}
        for isa in (False, True):
            s = H.HierarchicalSampler(H.PackedHierPoints(exs), list(range(P)), C, thinning=thin, seed=seed, adapt_start=adapt, chain_id_base=3,
                                      chain_offsets=offsets, device=gpu)
            s.launch_order = torch.tensor(order, device=gpu)
            s.prob.launch_order = s.launch_order.data_ptr()
            s.set_kernel_hint(lanes=1, isa=isa)
            s.init(start, cov_scale=0.01)
            s.enable_moments(after_iteration=mom_after)
            parts, kernels = [], []
            for k_ in cuts:
                parts.append(s.advance(k_).cpu().numpy())
                kernels.append(H.last_kernel())
            got[isa] = (np.concatenate(parts), s.state.cpu().numpy(), s.moments.cpu().numpy(), kernels)
        assert set(got[False][3]) == {1} and set(got[True][3]) <= {4, 5}, (what, got[False][3], got[True][3])
        for name, a, b in zip(("rows", "state", "moments"), got[False][:3], got[True][:3]):
            same = a.view(np.uint64) == b.view(np.uint64)
            assert same.all(), ("hierarchical", what, name, int((~same).sum()))
        # ---- single level, model 2
        packed = dr.PackedPoints(synthetic.single_level_pairs(exs))
        got = {}
        for isa in (False, True):
            s = SingleLevelSampler(packed, 2, list(range(P)), temps, C, thinning=thin, seed=seed, adapt_start=adapt, device=gpu, chain_id_base=3,
                                   chain_offsets=offsets, queue_quanta=quanta, reset_mean_at_adapt_start=bool(case % 2))
            s.set_kernel_hint(isa=isa)
            s.init(np.array([6.0, 0.8, 8.0]), cov_identity=bool(case % 3 == 0), cov_scale=0.05)
            parts, kernels = [], []
            for k_ in cuts:
                parts.append(s.advance(k_).cpu().numpy())
                kernels.append(s.last_kernel())
            got[isa] = (np.concatenate(parts), s.state.cpu().numpy(), kernels)
        lone = P * ((C + 63) // 64) <= 1024                       # one wavefront per SIMD or fewer: the assembly build is not used (by design)
        assert set(got[False][2]) == {1}, (what, got[False][2])
        assert set(got[True][2]) == {1} if lone else bool({2, 3} & set(got[True][2])), (what, got[True][2])
        for name, a, b in zip(("rows", "state"), got[False][:2], got[True][:2]):
            same = a.view(np.uint64) == b.view(np.uint64)
            assert same.all(), ("single level", what, name, int((~same).sum()))


def test_isa_work_queue_bit_identical_at_full_width(gpu):
    """all 147 Crumb pairs with 3 x 4 points x 1 024 chains = 2 352 blocks on 2 048 wavefront slots: the launch runs as a work queue
    (quanta of 125 iterations, blocks chaining through their state in HBM) — rows, state and moments against the hipcc kernel's plain
    launch, bit for bit; launches of 300 (125 + 125 + 50) and 500 iterations, moments from iteration 1 200 on"""
    import os
    import torch
    from conftest import REPO
    from pyhillfit_amd import bestfit, doseresponse as dr, hierarchical as H
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
    shapes, scales, locs = H.prior_params()
    exs = []
    for d in dr.drugs:
        for c in dr.channels:
            ne, _, ex = dr.load_crumb_data(d, c)
            if H.group_key(ex) == (3, 4):
                exs.append(ex)
    assert len(exs) == 147
    packed = H.PackedHierPoints(exs)
    theta0 = np.array([bestfit.hierarchical_first_iteration(e, locs) for e in exs])
    got = {}
    for isa in (False, True):
        s = H.HierarchicalSampler(packed, list(range(len(exs))), 1024, thinning=5, seed=77, adapt_start=1100, device=gpu)
        s.set_kernel_hint(lanes=1, isa=isa)
        s.init(theta0, cov_scale=0.01)
        s.enable_moments(after_iteration=1200)
        s.advance(1000, save=False)
        assert H.last_kernel() == (5 if isa else 1), H.last_kernel()
        rows = torch.cat([s.advance(k) for k in (300, 500)])
        torch.cuda.synchronize()
        assert int(s.queue[1 + s.nblocks]) == 0                                  # the queue's sticky fault word
        got[isa] = (rows, s.state.clone(), s.moments.clone())
    for name, a, b in zip(("rows", "state", "moments"), got[False], got[True]):
        same = a.view(torch.int64) == b.view(torch.int64)
        assert bool(same.all()), (name, int((~same).sum()))


def test_hierarchical_drained_queue_raises_and_checkpoint_continues_bit_identically(gpu, oracle_pair):
    """(i) the queue workspace's sticky fault word — which a wavefront of the assembly kernel raises when its wait for a block's previous
    quantum does not end — reaches the host as PhfError at every point that hands results on (written here by hand, as such a launch would
    leave it; the library never clears it); (ii) state_dict / load_state_dict: a second sampler continues a checkpoint bit for bit, and a
    checkpoint of another generator / ABI is refused"""
    import torch
    from pyhillfit_amd import _lib
    adapt = 140
    s, _ = _setup(gpu, UNIFORM4, oracle_pair, 256, 5, adapt, 31337, True)
    s.init(np.array(THETA0), cov_scale=0.01)
    s.enable_moments(after_iteration=0)
    s.advance(200, save=False)
    s.acceptance(); s.posterior_moments()                                   # a healthy queue: no raise
    sd = s.state_dict()
    rows = s.advance(300).cpu().numpy()
    s2, _ = _setup(gpu, UNIFORM4, oracle_pair, 256, 5, adapt, 31337, False)   # the continuation runs the hipcc kernel: same bits
    s2.load_state_dict(sd)
    rows2 = s2.advance(300).cpu().numpy()
    assert _bits_equal(rows, rows2) and _bits_equal(s.state.cpu().numpy(), s2.state.cpu().numpy())
    bad = dict(sd); bad["philox_rounds"] = 10 if sd["philox_rounds"] != 10 else 7
    with pytest.raises(_lib.PhfError, match="bit-identically"):
        s2.load_state_dict(bad)
    s.queue[1 + s.nblocks] = 1
    s.advance(100, save=False)
    torch.cuda.synchronize()
    assert int(s.queue[1 + s.nblocks].item()) == 1                                      # sticky
    for call in (s.acceptance, s.posterior_moments, s.state_dict):
        with pytest.raises(_lib.PhfError, match="drained"):
            call()


# ---------------------------------------------------------------------------------------------------------------------------------
# phf_sl3_advance: the hand-allocated single-level model-2 iteration against the hipcc kernel and the scalar twin
def _sl_runs(gpu, packed, Q, temps, C, thin, adapt, cuts, reset_mean=False, cov_identity=False, theta0=(6.0, 0.8, 8.0), queue_quanta=4, **kw):
    from pyhillfit_amd.sampler import SingleLevelSampler
    got = {}
    for isa in (False, True):
        s = SingleLevelSampler(packed, 2, list(range(Q)) if "pair_index" not in kw else kw["pair_index"], temps, C, thinning=thin, seed=2024,
                               adapt_start=adapt, reset_mean_at_adapt_start=reset_mean, device=gpu, queue_quanta=queue_quanta,
                               chain_id_base=3, chain_offsets=kw.get("chain_offsets"), launch_order=kw.get("launch_order", "cost"))
        s.set_kernel_hint(isa=isa)
        s.init(np.asarray(theta0), cov_identity=cov_identity, cov_scale=1.0 if cov_identity else 0.05)
        parts, kernels = [], []
        for k in cuts:
            parts.append(s.advance(k).cpu().numpy())
            kernels.append(s.last_kernel())
        got[isa] = (np.concatenate(parts), s.state.cpu().numpy(), kernels)
    return got


@pytest.mark.parametrize("C,thin,cuts,quanta", [(1024, 5, (300, 7, 493), 4), (1000, 1, (150, 250), 0)])
def test_isa_single_level_bit_identical_to_the_hipcc_kernel(C, thin, cuts, quanta, gpu):
    """every entry-count shape of the Crumb set (all 210 pairs: 0..5 uncensored x 0..4 censored entries — the run-time loops of the
    assembly target against hipcc's twenty straight-line bodies and its generic one), tempered problems among them (t = 1, 0.125, 0),
    a ragged last wavefront (1 000 chains), thinning 1 and 5, launches cut before / at / after the start of the adaptation"""
    import os
    from conftest import REPO
    from pyhillfit_amd import doseresponse as dr
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
    names = [(d, c) for d in dr.drugs for c in dr.channels]
    packed = dr.pack_single_level(names)
    temps = [(1.0, 0.125, 0.0, 1.0)[q % 4] for q in range(len(names))]
    got = _sl_runs(gpu, packed, len(names), temps, C, thin, 200, cuts, reset_mean=True, cov_identity=True, theta0=(1.0, 1.0, 1.0), queue_quanta=quanta)
    # the kernel meant is the kernel that ran: assembly (queued where the launch is queued and starts on a multiple of the thinning;
    # a queued launch from a ragged start falls back to the hipcc queue) against hipcc
    assert set(got[False][2]) == {1} and got[True][2][0] in (2, 3) and got[True][2][1] == 2, (got[False][2], got[True][2])
    for name, a, b in zip(("rows", "state"), got[False][:2], got[True][:2]):
        same = a.view(np.uint64) == b.view(np.uint64)
        assert same.all(), (name, int((~same).sum()), np.argwhere(~same)[:5].tolist())


def test_isa_single_level_queue_and_twin(gpu):
    """the C3 shape in small: 210 pairs x 4 096 chains = 13 440 blocks through the work queue (quanta of a quarter launch) — rows and
    state against the hipcc kernel's queued launch bit for bit, and sampled chains against the scalar twin"""
    import os
    from conftest import REPO
    from oracle import c_oracle as co
    from pyhillfit_amd import doseresponse as dr
    from pyhillfit_amd.sampler import gamma_table
    dr.setup(os.path.join(REPO, "data", "crumb_dataset.json")); dr.define_model(2)
    names = [(d, c) for d in dr.drugs for c in dr.channels]
    packed = dr.pack_single_level(names)
    C, thin, adapt = 4096, 5, 120
    got = _sl_runs(gpu, packed, len(names), [1.0] * len(names), C, thin, adapt, (200, 400))
    assert got[True][2][0] in (2, 3) and got[True][2][1] == 3 and got[False][2] == [1, 1], (got[True][2], got[False][2])
    for name, a, b in zip(("rows", "state"), got[False][:2], got[True][:2]):
        assert (a.view(np.uint64) == b.view(np.uint64)).all(), name
    rows = got[True][0]
    gam = gamma_table(600)
    for q in (0, 57, 209):
        ne, _, ex = dr.load_crumb_data(*names[q])
        concs, y = dr.concatenate_experiments(ne, ex)
        pk = co.PackedPair(concs, y, 2, 1.0)
        for c in (0, 4095):
            st = pk.init_state([6.0, 0.8, 8.0], False, 0.05)
            want = pk.advance(st, 0, 600, thin, adapt, False, gam, seed=2024, chain_id=3 + c, problem_id=q)
            assert np.array_equal(rows[:, q, :, c], want), (q, c)
