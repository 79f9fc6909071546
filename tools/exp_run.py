"""Timing experiments only: run any script of tools/ (or bench.py) against an experimental build of the library
(tools/_build/exp/*.so, made by hand from a patched copy of pyhillfit_amd/csrc).  Results of such builds are NOT valid samples.

    python tools/exp_run.py tools/_build/exp/libexp_X.so tools/diag_hier_lanes.py [arguments]
"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    lib, script = sys.argv[1], os.path.abspath(sys.argv[2])
    sys.argv = [script] + sys.argv[3:]
    from pyhillfit_amd import _lib
    if lib != "default":
        _lib.LIB_PATH = os.path.abspath(lib)
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
