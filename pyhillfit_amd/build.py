"""Compile the HIP kernels + C ABI into pyhillfit_amd/lib/libpyhillfit_amd.so for gfx950 (in-tree).

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the kernels spell every fma explicitly so
that the device evaluates the same fp64 operation sequence as the host twin used by the parity tests."""
import glob
import hashlib
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
SOURCES = ["phf_capi.hip", "phf_single_level.hip", "phf_hierarchical.hip", "phf_predictive.hip", "phf_hier3_isa.hip"]
# the hand-allocated gfx950 code object: generated assembly (tools/gen_hier_isa.py; committed) -> .o -> .co, embedded by phf_hier3_isa.hip
ISA_SRC = os.path.join(CSRC, "generated", "phf_hier3_gfx950.s")
ISA_CO = os.path.join(LIB_DIR, "obj", "phf_hier3_gfx950.co")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _digest(paths, extra=()):
    """sha256 over the CONTENT of the files (sorted by name) and the extra strings: a stamp that a fresh checkout, a copied tree
    or a touched file cannot fool the way modification times can"""
    h = hashlib.sha256()
    for p in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(p).encode()); h.update(b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode()); h.update(b"\0")
    return h.hexdigest()


def _stamp_path(artefact):
    return artefact + ".stamp"


def _is_current(artefact, digest):
    try:
        with open(_stamp_path(artefact)) as f:
            return os.path.exists(artefact) and f.read().strip() == digest
    except OSError:
        return False


def _write_stamp(artefact, digest):
    with open(_stamp_path(artefact), "w") as f:
        f.write(digest + "\n")


def _headers():
    return (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "generated", "*.h"))
            + [os.path.join(HERE, "..", "include", "pyhillfit_amd.h")])


def build_isa(verbose=False):
    """assemble and link the generated gfx950 assembly into a code object (clang as the assembler, ld.lld; no compiler involved)"""
    os.makedirs(os.path.dirname(ISA_CO), exist_ok=True)
    digest = _digest([ISA_SRC], ["isa-1"])
    if _is_current(ISA_CO, digest):
        return ISA_CO
    obj = ISA_CO[:-3] + ".o"
    for cmd in ([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", ISA_SRC, "-o", obj],
                [os.path.join(LLVM_BIN, "ld.lld"), "-shared", obj, "-o", ISA_CO]):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    _write_stamp(ISA_CO, digest)
    return ISA_CO


def _object_digest(src, extra_flags=()):
    # every header of csrc/ (globbed, not a hand-kept list: the oracle's Makefile wildcards the same headers, and a stale library
    # next to a rebuilt twin would make the two silently diverge) + the public header + the flags
    # (the generated assembly counts as a source of the object that embeds its code object)
    isa = [ISA_SRC] if os.path.basename(src) == "phf_hier3_isa.hip" else []
    return _digest([src] + isa + _headers(), [f for f in FLAGS if f] + list(extra_flags))


def _library_digest(extra_flags=()):
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + [ISA_SRC] + _headers(), [f for f in FLAGS if f] + list(extra_flags))


def build_textio(force=False, verbose=False):
    """the chain-file text formatter: no GPU code, g++ alone"""
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"]
    digest = _digest([TEXTIO_SRC, os.path.join(HERE, "..", "include", "pyhillfit_textio.h")], flags)
    if not force and _is_current(TEXTIO_LIB, digest):
        return TEXTIO_LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [shutil.which("g++") or "g++"] + flags + ["-o", TEXTIO_LIB, TEXTIO_SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _write_stamp(TEXTIO_LIB, digest)
    return TEXTIO_LIB


def needs_build():
    """True unless libpyhillfit_amd.so carries the stamp of exactly these sources, headers and flags (content hash, not mtimes:
    a prebuilt library shipped next to different sources must not pass as current)"""
    return not _is_current(LIB_PATH, _library_digest())


def _compile_one(src, obj, extra_flags, verbose):
    cmd = [_hipcc()] + [f for f in FLAGS if f and f != "-shared"] + list(extra_flags) + ["-c", "-o", obj, src]
    if os.path.basename(src) == "phf_hier3_isa.hip":
        cmd.insert(1, '-DPHF_ISA_CO_PATH="%s"' % os.path.abspath(ISA_CO))
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _write_stamp(obj, _object_digest(src, extra_flags))


def build(force=False, verbose=False, extra_flags=()):
    """one object per source (compiled side by side: the sampler kernels take a minute each), then one link"""
    build_textio(force, verbose)
    build_isa(verbose)
    digest = _library_digest(extra_flags)
    if not force and _is_current(LIB_PATH, digest):
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not _is_current(obj, _object_digest(src, extra_flags)):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for f in [ex.submit(_compile_one, src, obj, extra_flags, verbose) for src, obj in jobs]:
            f.result()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _write_stamp(LIB_PATH, digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
