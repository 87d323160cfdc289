#!/usr/bin/env python3
"""Do the epilogue kernels run beside the next main kernel?  From a rocprofv3 --kernel-trace CSV: for the last N main kernels
(anchor substring), start / end of each and of the kernels between two anchors, relative to the anchor's start (us).
  python tools/overlap_trace.py <dir-or-csv> [anchor] [N]"""
import csv, glob, os, sys
path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_welch_pipe"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
idx = idx[-(N + 1):]
for a, b in zip(idx[:-1], idx[1:]):
    t0 = rows[a][0]
    print("main kernel: 0.0 .. %.1f  (queue %s); next main starts at %.1f" % ((rows[a][1] - t0) / 1e3, rows[a][3], (rows[b][0] - t0) / 1e3))
    for j in range(a + 1, b):
        print("      %-40s %8.1f .. %8.1f  (queue %s)" % (rows[j][2], (rows[j][0] - t0) / 1e3, (rows[j][1] - t0) / 1e3, rows[j][3]))
