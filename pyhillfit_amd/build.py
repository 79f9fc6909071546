"""Compile the HIP kernels + C ABI into pyhillfit_amd/lib/libpyhillfit_amd.so for gfx950 (in-tree).

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the kernels spell every fma explicitly so
that the device evaluates the same fp64 operation sequence as the host twin used by the parity tests."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpyhillfit_amd.so")
SOURCES = ["phf_capi.hip", "phf_single_level.hip", "phf_hierarchical.hip", "phf_predictive.hip"]
HEADERS = ["phf_common.h", "phf_math.h", "phf_philox.h", "phf_model.h", "phf_hier_model.h", os.path.join("..", "..", "include", "pyhillfit_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc()] + [f for f in FLAGS if f] + list(extra_flags) + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
