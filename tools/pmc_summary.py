#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch of each kernel."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                k = "mh_advance_kernel" + k.split("mh_advance_kernel")[1][:3] if "mh_advance_kernel" in k else k[:60]
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "mh_advance" not in k and "hier" not in k and "pred_" not in k and "phf_" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 2:]      # later dispatches = timed (adaptive) steps
        print("  %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
