#!/usr/bin/env python3
"""Per-kernel timeline of a C4 run from a rocprofv3 --kernel-trace directory (tools/gpu_r05_c4_timeline.sh): for every dispatch of the
advance kernels its start and end relative to the first one, grouped by kernel — which groups overlap, which wait.
Produced profiles/r05/c4_timeline.txt.    python tools/c4_timeline.py DIR"""
import csv
import glob
import os
import re
import sys


def main():
    rows = []
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                if "advance" not in name:
                    continue
                m = re.search(r"(hier_\w+<[^>]*>|phf_\w+)", name)
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else name[:60], r.get("Queue_Id", "?"),
                             int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
    rows.sort()
    if not rows:
        print("no advance kernels in the trace")
        return
    # the timed region: the last 6 dispatches of every kernel name (warm-up and set-up launches come first)
    t0 = rows[0][0]
    names = sorted({r[2] for r in rows})
    print("dispatches of the advance kernels, ms from the first one (start -> end, duration); grid in threads")
    for n in names:
        mine = [r for r in rows if r[2] == n]
        print("\n%s   [%d dispatches, workgroup %d, grid %d, queue %s]" % (n, len(mine), mine[-1][4], mine[-1][5], mine[-1][3]))
        for s, e, _, _, _, _ in mine[-8:]:
            print("    %9.2f -> %9.2f   (%7.2f ms)" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
    print("\nall dispatches in start order (the timed region is the tail):")
    for s, e, n, q, _, g in rows:
        print("    %9.2f -> %9.2f   (%7.2f ms)  %-28s grid %6d  queue %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n, g, q))
    last = rows[-6 * len(names):]
    span = (max(r[1] for r in last) - min(r[0] for r in last)) / 1e6
    busy = sum(r[1] - r[0] for r in last) / 1e6
    print("\nlast %d dispatches: span %.1f ms, sum of durations %.1f ms" % (len(last), span, busy))


if __name__ == "__main__":
    main()
