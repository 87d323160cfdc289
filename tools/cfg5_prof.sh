#!/bin/bash
# per-kernel times of cfg5 (64 ch x 2^24, nfft 4096): rocprofv3 --kernel-trace --stats -- python3 tools/cfgbench.py --only cfg5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cfg5prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 tools/cfgbench.py --only cfg5 > $OUT/p.log 2>&1
python3 - $(ls $OUT/p/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 20000 and "at::native" not in r["Name"]:
        print("%-100s calls %4s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
