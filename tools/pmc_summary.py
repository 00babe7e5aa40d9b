#!/usr/bin/env python3
"""Averages rocprofv3 --pmc CSV output per kernel (one row per dispatch and counter) over the dispatches of each kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if not any(s in k for s in ("k_scratch", "k_frames", "k_finish")):
        continue
    print("== %s" % k[:110])
    for c in sorted(acc[k]):
        v = acc[k][c]
        # skip warm-up dispatches: use the last half
        h = v[len(v) // 2:]
        print("  %-28s n=%3d  mean=%16.1f" % (c, len(v), sum(h) / len(h)))
