#!/bin/bash
# every kernel of tools/cfgbench.py (all configs) with its average time: rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/allprof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 tools/cfgbench.py > $OUT/p.log 2>&1
python3 - $(ls $OUT/p/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 8000 and "at::native" not in r["Name"] and "vectorized" not in r["Name"]:
        print("%-96s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
