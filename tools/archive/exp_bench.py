"""Timing experiments only: run bench.py against an experimental build of the library (tools/_build/exp/*.so, made by
hand from a patched copy of pyhillfit_amd/csrc).  The results of such builds are NOT valid samples (e.g. entry counts
forced at compile time); only the kernel time is of interest.

    python tools/exp_bench.py tools/_build/exp/libexp_31.so [bench.py arguments]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    lib = os.path.abspath(sys.argv[1])
    sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[2:]
    from pyhillfit_amd import _lib
    _lib.LIB_PATH = lib
    import bench
    bench.main()


if __name__ == "__main__":
    main()
