#!/usr/bin/env python3
"""bytes moved / (counter x 1024) per kernel of tools/ubench/fetch_calib from the rocprofv3 counter CSVs"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
GiB = float(1 << 30)
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    names = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (d, c), v in per.items():
        acc[(names[d], c)].append(v)
print("# known bytes per launch: 1 GiB for every kernel; counters are KiB")
for (k, c), v in sorted(acc.items()):
    m = sum(v) / len(v)
    print("%-44s %-11s n=%d mean=%.6g KiB  -> true bytes / (counter x 1024) = %.3f" % (k[:44], c, len(v), m, GiB / (m * 1024.0) if m else float("nan")))
