"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
function include/pyhillfit_amd.h declares; argument validation works without touching a GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from pyhillfit_amd import _lib
    return _lib.load()


def _declared():
    with open(os.path.join(REPO, "include", "pyhillfit_amd.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(phf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from pyhillfit_amd import _lib
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names


def test_version_and_state_sizes(lib):
    assert lib.phf_version() == 7
    assert lib.phf_hierarchical_last_kernel() == 0       # nothing launched yet in this thread
    assert lib.phf_philox_rounds() == 7                  # phf_philox.h: PHF_PHILOX_ROUNDS
    assert lib.phf_debug_philox_rounds(8, 1, 1, 1, None) == -1 and b"rounds" in lib.phf_last_error()
    assert lib.phf_simd_count() >= 4                      # without a GPU: the MI355X figure (1 024)
    assert lib.phf_single_level_state_size(1) == 11      # 2+1+2+3+1+1+1
    assert lib.phf_single_level_state_size(2) == 16      # 3+1+3+6+1+1+1
    assert lib.phf_single_level_state_size(3) < 0
    assert b"model" in lib.phf_last_error()
    assert lib.phf_hierarchical_state_size(3) == 2 * 11 + 66 + 3 and lib.phf_hierarchical_state_size(6) == 2 * 17 + 153 + 3
    assert lib.phf_hierarchical_state_size(50) == 2 * 105 + 105 * 106 // 2 + 3
    assert lib.phf_hierarchical_state_size(65) == -3 and lib.phf_hierarchical_state_size(0) == -3


def test_argument_validation_without_gpu(lib):
    from pyhillfit_amd import _lib
    rc = lib.phf_single_level_advance(None, None, None, 0, 10, None, None, None, 0, None)
    assert rc == -1 and b"null" in lib.phf_last_error()
    pts = _lib.Points(1, 16, 1, 1, 1, 1, 1, 1)
    prob = _lib.Problems(1, 64, 1, 1, 1, 0, 0, None, None)
    cfg = _lib.MhConfig(5, 5, 3000, 0, 0, 25, None)
    assert lib.phf_single_level_advance(C.byref(pts), C.byref(prob), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1
    cfg.model = 2; cfg.thinning = 0
    assert lib.phf_single_level_advance(C.byref(pts), C.byref(prob), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1
    cfg.thinning = 5
    assert lib.phf_single_level_advance(C.byref(pts), C.byref(prob), C.byref(cfg), 0, 4000, 1, None, None, 0, None) == -1
    assert b"gamma" in lib.phf_last_error()
    assert lib.phf_single_level_advance(C.byref(pts), C.byref(prob), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1
    assert b"gamma" in lib.phf_last_error()               # the table is read (times zero) before the adaptation starts too
    pts.stride = 20000                                    # 480 KB of entries cannot be staged in a CU's LDS
    assert lib.phf_single_level_advance(C.byref(pts), C.byref(prob), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -3
    with pytest.raises(_lib.PhfError):
        _lib.check(-1, "x")
    assert lib.phf_single_level_queue_status(None, 10, None) == -1 and lib.phf_single_level_queue_status(1, 0, None) == -1
    assert lib.phf_hierarchical_set_kernel_policy(3, 0) == -1 and lib.phf_hierarchical_set_kernel_policy(0, -1) == -1
    assert lib.phf_hierarchical_set_kernel_policy(2, 1) == 0 and lib.phf_hierarchical_set_kernel_policy(0, 0) == 0
    # ABI 6: the hierarchical launch's own checks — kernel_hint fields, points_per_expt, the queued entry's workspace
    from pyhillfit_amd import hierarchical as H
    H._bind(lib)
    hp = H.HierPoints(1, 16, 3, 0, 1, 1, 1)
    pr = H.make_prior()
    prob.kernel_hint = 3
    assert lib.phf_hierarchical_advance(C.byref(hp), C.byref(prob), C.byref(pr), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1
    assert b"kernel_hint" in lib.phf_last_error()
    prob.kernel_hint = 32
    assert lib.phf_hierarchical_advance(C.byref(hp), C.byref(prob), C.byref(pr), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1
    prob.kernel_hint = 64                                  # a single-level launch knows bits 4 and 5 only
    pts.stride = 16
    assert lib.phf_single_level_advance(C.byref(pts), C.byref(prob), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1
    assert b"kernel_hint of a single-level launch" in lib.phf_last_error()
    prob.kernel_hint = 16 | 1
    hp.points_per_expt = 7                                 # 3 experiments x 7 points do not fit a stride of 16
    assert lib.phf_hierarchical_advance(C.byref(hp), C.byref(prob), C.byref(pr), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1
    assert b"points_per_expt" in lib.phf_last_error()
    for bad in (-4, 16, 5 | (12 << 4), 1 << 8):             # negative; per = 0; 5 + 5 + 12 points beyond the stride; last > 15
        hp.points_per_expt = bad
        assert lib.phf_hierarchical_advance(C.byref(hp), C.byref(prob), C.byref(pr), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1, bad
        assert b"points_per_expt" in lib.phf_last_error()
    hp.points_per_expt = 4
    assert lib.phf_hierarchical_advance_queued(C.byref(hp), C.byref(prob), C.byref(pr), C.byref(cfg), 0, 10, 1, None, None, 0, 0, None, None) == -1
    assert b"queue" in lib.phf_last_error()
    assert lib.phf_debug_isa(99, 1, 1, 1, None) == -1 and lib.phf_debug_isa(0, 0, None, None, None) == 0
    # ABI 7: workspace sizes (no device call), the fused launch's checks
    lib.phf_hierarchical_queue_words.restype = C.c_int64
    prob.kernel_hint = 0
    hp.points_per_expt = 4
    blocks = prob.num_problems * -(-prob.chains_per_problem // 64)
    assert lib.phf_hierarchical_queue_words(C.byref(hp), C.byref(prob)) == 2 + blocks           # Ne = 3: no scratch tier
    assert lib.phf_hierarchical_queue_words(None, C.byref(prob)) == -1
    grp = (H.HierGroup * 2)()
    lib.phf_hierarchical_fused_queue_words.argtypes = [C.c_int32, C.POINTER(H.HierGroup)]
    lib.phf_hierarchical_fused_queue_words.restype = C.c_int64
    lib.phf_hierarchical_advance_fused.argtypes = [C.c_int32, C.POINTER(H.HierGroup), C.POINTER(H.HierPrior), C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                                   C.c_void_p, C.c_void_p]
    for g in grp:
        g.pts = C.pointer(hp); g.prob = C.pointer(prob); g.cfg = C.pointer(cfg); g.state = 1
    assert lib.phf_hierarchical_fused_queue_words(1, grp) == 2 + blocks
    assert lib.phf_hierarchical_fused_queue_words(2, grp) == -3 and b"two groups of one" in lib.phf_last_error()
    assert lib.phf_hierarchical_fused_queue_words(0, grp) == -1 and lib.phf_hierarchical_fused_queue_words(15, grp) == -1      # 1..14 groups
    hp.points_per_expt = (1 << 30) | 0x124                     # the list form: 4 + 2 + 1 points — a valid code, no gfx950 kernel for that shape
    assert lib.phf_hierarchical_fused_queue_words(1, grp) == -3 and b"no gfx950 kernel" in lib.phf_last_error()
    for bad in ((1 << 30) | 0x1124, (1 << 30) | 0x104):        # a nibble beyond the third experiment; an experiment without points
        hp.points_per_expt = bad
        assert lib.phf_hierarchical_advance(C.byref(hp), C.byref(prob), C.byref(pr), C.byref(cfg), 0, 10, 1, None, None, 0, None) == -1, hex(bad)
        assert b"points_per_expt" in lib.phf_last_error()
    hp.points_per_expt = 7 | (1 << 4)                          # 7 + 7 + 1 points: no gfx950 kernel for that shape
    assert lib.phf_hierarchical_fused_queue_words(1, grp) == -3 and b"no gfx950 kernel" in lib.phf_last_error()
    hp.points_per_expt = 4
    assert lib.phf_hierarchical_advance_fused(1, grp, C.byref(pr), 0, 10, 0, 0, None, None) == -1      # no workspace
    assert lib.phf_hierarchical_advance_fused(1, grp, C.byref(pr), 3, 10, 0, 0, 1, None) == -3 and b"multiple of the thinning" in lib.phf_last_error()


def test_drained_queue_raises_on_the_host():
    """sampler.raise_if_drained reads the workspace's last word: what SingleLevelSampler calls at its synchronisation points"""
    import torch
    from pyhillfit_amd import _lib
    from pyhillfit_amd.sampler import raise_if_drained
    q = torch.zeros(2 + 7, dtype=torch.int32)
    q[0] = 0x40000000                                     # a poisoned task counter alone is history the next launch resets ...
    raise_if_drained(None); raise_if_drained(q)
    q[-1] = 1                                             # ... the sticky fault word is what says "stale results"
    with pytest.raises(_lib.PhfError, match="drained"):
        raise_if_drained(q)


def test_missing_library_fails_loudly(monkeypatch):
    from pyhillfit_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpyhillfit_amd.so")
    with pytest.raises(_lib.PhfError, match="no CPU fallback"):
        _lib.load()


def test_sampler_refuses_cpu_device(lib):
    import numpy as np
    from pyhillfit_amd.doseresponse import PackedPoints
    from pyhillfit_amd.sampler import SingleLevelSampler
    from pyhillfit_amd._lib import PhfError
    packed = PackedPoints([(np.array([0.1, 1.0]), np.array([10.0, 60.0]))])
    with pytest.raises(PhfError, match="HIP device only"):
        SingleLevelSampler(packed, 2, [0], [1.0], 64, device="cpu")


def test_textio_library_exports_its_header():
    """include/pyhillfit_textio.h <-> pyhillfit_amd/lib/libphf_textio.so (host-only chain-file text formatter)"""
    import __graft_entry__ as g
    g.build()
    with open(os.path.join(REPO, "include", "pyhillfit_textio.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(phf_[a-z0-9_]+)\s*\(", src)))
    assert names == ["phf_format_rows", "phf_savetxt"]
    lib = C.CDLL(os.path.join(REPO, "pyhillfit_amd", "lib", "libphf_textio.so"))
    for n in names:
        assert hasattr(lib, n), n
    lib.phf_format_rows.restype = C.c_int64
    lib.phf_format_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int64]
    import numpy as np
    a = np.array([[1.0, -2.5], [np.inf, np.nan]])
    buf = C.create_string_buffer(4 * 28)
    n = lib.phf_format_rows(a.ctypes.data, 2, 2, 2, buf, len(buf))
    assert buf.raw[:n] == b"1.000000000000000000e+00 -2.500000000000000000e+00\ninf nan\n"
    assert lib.phf_format_rows(a.ctypes.data, 2, 2, 2, buf, 10) == -4 * 28
