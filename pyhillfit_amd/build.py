"""Compile the HIP kernels + C ABI into pyhillfit_amd/lib/libpyhillfit_amd.so for gfx950 (in-tree).

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the kernels spell every fma explicitly so
that the device evaluates the same fp64 operation sequence as the host twin used by the parity tests."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpyhillfit_amd.so")
TEXTIO_SRC = os.path.join(CSRC, "phf_textio.cpp")                 # host-only C++ (chain-file text), built with g++
TEXTIO_LIB = os.path.join(LIB_DIR, "libphf_textio.so")
SOURCES = ["phf_capi.hip", "phf_single_level.hip", "phf_hierarchical.hip", "phf_predictive.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_textio(force=False, verbose=False):
    """the chain-file text formatter: no GPU code, g++ alone"""
    if not force and os.path.exists(TEXTIO_LIB) and os.path.getmtime(TEXTIO_LIB) >= os.path.getmtime(TEXTIO_SRC):
        return TEXTIO_LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", TEXTIO_LIB, TEXTIO_SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TEXTIO_LIB


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # every header and source of csrc/ (globbed, not a hand-kept list: the oracle's Makefile wildcards the same headers,
    # and a stale library next to a rebuilt twin would make the two silently diverge) + the public header
    deps = (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip"))
            + [os.path.join(HERE, "..", "include", "pyhillfit_amd.h"), os.path.abspath(__file__)])
    return any(os.path.getmtime(p) > t for p in deps)


def _compile_one(src, obj, extra_flags, verbose):
    cmd = [_hipcc()] + [f for f in FLAGS if f and f != "-shared"] + list(extra_flags) + ["-c", "-o", obj, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False, extra_flags=()):
    """one object per source (compiled side by side: the sampler kernels take a minute each), then one link"""
    build_textio(force, verbose)
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "pyhillfit_amd.h"), os.path.abspath(__file__)]
    newest_header = max(os.path.getmtime(h) for h in headers)
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or extra_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_header, os.path.getmtime(src)):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for f in [ex.submit(_compile_one, src, obj, extra_flags, verbose) for src, obj in jobs]:
            f.result()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
