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
