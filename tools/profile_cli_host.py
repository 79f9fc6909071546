#!/usr/bin/env python3
"""Where the wall time of the drop-in command line goes on the HOST side: `PyHillFit.py -m 2 -a` at the
reference's defaults spends under a second sampling, so imports, the data file, the start-point fits, the device set-up and the
file writers are most of what a user waits for.  Prints (and writes to gpurun_out/profile_cli_host.txt): import times, the wall
time of the whole main(), the command line's own phase line, and the top entries of a cProfile of the call by cumulative time.

    python tools/profile_cli_host.py [--hierarchical] [extra PyHillFit flags]
"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out = io.StringIO()

    def say(*a):
        print(*a); print(*a, file=out)
    t0 = time.perf_counter()
    import numpy  # noqa: F401
    t1 = time.perf_counter()
    import torch
    t2 = time.perf_counter()
    import csv  # noqa: F401  (the data reader is the stdlib csv module: nothing to time)
    t3 = time.perf_counter()
    from pyhillfit_amd import PyHillFit
    from pyhillfit_amd import hierarchical  # noqa: F401
    t4 = time.perf_counter()
    say("imports: numpy %.2f s, torch %.2f s, pyhillfit_amd %.2f s (scipy is imported only by the scalar fits)" % (t1 - t0, t2 - t1, t4 - t3))
    t5 = time.perf_counter()
    torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
    say("first GPU call (HIP runtime + context): %.2f s" % (time.perf_counter() - t5))
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        from pyhillfit_amd import doseresponse as dr
        dr.setup(os.path.join(REPO, "data", "crumb_dataset.json"))
        csv = os.path.join(tmp, "crumb_data.csv")
        dr.table.to_csv(csv)                                # the reference's data file format (data/crumb_data.csv there)
        argv = ["--data-file", csv, "-m", "2", "-a", "--num-chains", "64", "--output-root", os.path.join(tmp, "output")] + extra
        say("command: PyHillFit.py " + " ".join(argv))
        pr = cProfile.Profile()
        t6 = time.perf_counter()
        pr.enable()
        try:
            PyHillFit.main(argv)
        finally:
            pr.disable()
        wall = time.perf_counter() - t6
        say("main(): %.2f s wall" % wall)
        st = pstats.Stats(pr, stream=out)
        st.sort_stats("cumulative").print_stats(35)
    text = out.getvalue()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    name = "profile_cli_host%s.txt" % ("_hier" if "--hierarchical" in extra else "")
    with open(os.path.join(REPO, "gpurun_out", name), "w") as f:
        f.write(text)
    print("\n".join(text.splitlines()[-45:]))


if __name__ == "__main__":
    main()
