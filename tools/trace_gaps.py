#!/usr/bin/env python3
"""Per-step GPU timeline from a rocprofv3 --kernel-trace CSV: for the last N occurrences of the anchor kernel, the
kernels between consecutive anchors with duration and the idle gap before each.
  python tools/trace_gaps.py <dir-or-csv> [anchor-substring] [N]"""
import csv, glob, os, sys
from collections import defaultdict
path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_welch_carry"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
idx = idx[-(N + 1):]
agg = defaultdict(lambda: [0, 0.0, 0.0])
spans = []
for a, b in zip(idx[:-1], idx[1:]):
    spans.append((rows[b][0] - rows[a][0]) / 1e3)
    for j in range(a, b):
        s, e, n = rows[j]
        gap = (s - rows[j - 1][1]) / 1e3 if j > 0 else 0.0
        k = (j - a, n.split("(")[0][-60:])
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
        agg[k][2] += gap
print("step span (anchor start to next anchor start): mean %.1f us over %d steps" % (sum(spans) / len(spans), len(spans)))
tot_d = tot_g = 0.0
for k in sorted(agg):
    c, d, g = agg[k]
    print("  #%d %-62s dur %8.1f us   gap before %6.1f us" % (k[0], k[1], d / c, g / c))
    tot_d += d / c
    tot_g += g / c
print("  sum of kernels %.1f us, sum of gaps %.1f us" % (tot_d, tot_g))
