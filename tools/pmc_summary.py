#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch for kernels whose name contains argv[2]."""
import csv, glob, os, sys
from collections import defaultdict
root, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row["Kernel_Name"]:
                per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    for d, cs in per.items():
        for c, v in cs.items():
            acc[c].append(v)
for c in sorted(acc):
    v = acc[c]
    print("%-28s n=%-3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
