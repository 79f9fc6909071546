"""ctypes binding of the C ABI in include/pyhillfit_amd.h (pyhillfit_amd/lib/libpyhillfit_amd.so).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpyhillfit_amd.so")
ABI_VERSION = 7

# every symbol include/pyhillfit_amd.h declares
EXPORTS = ["phf_version", "phf_last_error", "phf_simd_count", "phf_single_level_state_size", "phf_single_level_init",
           "phf_single_level_advance", "phf_single_level_advance_queued", "phf_single_level_queue_status", "phf_single_level_last_kernel", "phf_single_level_log_target",
           "phf_debug_math", "phf_debug_isa", "phf_philox_rounds", "phf_debug_philox", "phf_debug_philox_rounds", "phf_hierarchical_state_size", "phf_hierarchical_init", "phf_hierarchical_advance", "phf_hierarchical_advance_queued", "phf_hierarchical_queue_words", "phf_hierarchical_advance_fused", "phf_hierarchical_fused_queue_words",
           "phf_hierarchical_set_kernel_policy", "phf_hierarchical_last_kernel", "phf_hierarchical_log_target", "phf_predictive_scratch_bytes", "phf_predictive_accumulate"]


class PhfError(RuntimeError):
    pass


class Points(C.Structure):
    _fields_ = [("num_pairs", C.c_int32), ("stride", C.c_int32), ("ln_conc", C.c_void_p), ("response", C.c_void_p),
                ("weight", C.c_void_p), ("counts", C.c_void_p), ("pi_bit", C.c_void_p), ("extra", C.c_void_p)]


class Problems(C.Structure):
    _fields_ = [("num_problems", C.c_int32), ("chains_per_problem", C.c_int32), ("pair_index", C.c_void_p),
                ("temperature", C.c_void_p), ("problem_id", C.c_void_p), ("chain_id_base", C.c_uint32),
                ("kernel_hint", C.c_uint32), ("launch_order", C.c_void_p), ("chain_offset", C.c_void_p)]


class MhConfig(C.Structure):
    _fields_ = [("model", C.c_int32), ("thinning", C.c_int32), ("adapt_start", C.c_int64),
                ("reset_mean_at_adapt_start", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64),
                ("gamma", C.c_void_p)]


_lib = None


def load():
    """dlopen the HIP library; raises PhfError (never falls back) if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7); it must be mapped BEFORE our library so that the
    # loader binds us to that same runtime instance — two HIP runtimes in one process do not share a device context
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PhfError("HIP library %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(pyhillfit_amd has no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise PhfError("%s does not export %s" % (LIB_PATH, name))
    lib.phf_last_error.restype = C.c_char_p
    lib.phf_version.restype = C.c_int
    if lib.phf_version() != ABI_VERSION:
        raise PhfError("ABI version mismatch: library %d, binding %d" % (lib.phf_version(), ABI_VERSION))
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.phf_single_level_state_size.argtypes = [i32]
    lib.phf_single_level_init.argtypes = [C.POINTER(Points), C.POINTER(Problems), i32, i32, f64, vp, vp, vp, vp]
    lib.phf_single_level_advance.argtypes = [C.POINTER(Points), C.POINTER(Problems), C.POINTER(MhConfig), i64, i64, vp,
                                             vp, vp, i64, vp]
    lib.phf_single_level_advance_queued.argtypes = [C.POINTER(Points), C.POINTER(Problems), C.POINTER(MhConfig), i64, i64, vp,
                                                    vp, vp, i64, i32, vp, vp]
    lib.phf_single_level_queue_status.argtypes = [vp, i64, vp]
    lib.phf_hierarchical_set_kernel_policy.argtypes = [i32, i32]
    lib.phf_simd_count.restype = C.c_int
    lib.phf_single_level_log_target.argtypes = [C.POINTER(Points), i32, i64, vp, vp, vp, vp, vp, vp]
    lib.phf_debug_math.argtypes = [i32, i64, vp, vp, vp]
    lib.phf_debug_isa.argtypes = [i32, i64, vp, vp, vp]
    lib.phf_debug_philox.argtypes = [i64, vp, vp, vp]
    lib.phf_debug_philox_rounds.argtypes = [i32, i64, vp, vp, vp]
    lib.phf_philox_rounds.restype = C.c_int
    lib.phf_predictive_scratch_bytes.argtypes = [i32, i64, i32, i32]
    lib.phf_predictive_scratch_bytes.restype = C.c_size_t
    lib.phf_predictive_accumulate.argtypes = [i32, vp, i64, i32, i32, i32, i32, vp, vp, i32, vp, vp, C.c_size_t, vp]
    _lib = lib
    return lib


PHF_ERR_UNSUPPORTED = -3          # include/pyhillfit_amd.h


def check(rc, what=""):
    if rc != 0:
        raise PhfError("%s failed (%d): %s" % (what or "pyhillfit_amd call", rc, load().phf_last_error().decode()))
