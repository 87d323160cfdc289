#!/bin/bash
# per-kernel times of the cog call and the reference-against-63-channels CSD (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/auxprof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cog -- python3 tools/cfgbench.py --only cog > $OUT/cog.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -- python3 tools/cfgbench.py --only cfg5 > $OUT/c5.log 2>&1
for t in cog c5; do
  f=$(ls $OUT/$t/*/*kernel_stats.csv | head -1)
  echo "== $t"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"]
    if "at::native" in n or "elementwise" in n: continue
    print("%-100s calls %5s  avg %9.1f us  total %8.2f ms" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
