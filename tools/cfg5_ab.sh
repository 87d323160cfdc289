#!/bin/bash
# cfg5 per-kernel times, one-pass means (default) against SP_CSDM_TWOPASS=1, on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cfg5ab
rm -rf $OUT; mkdir -p $OUT
for v in onepass twopass; do
  if [ $v = twopass ]; then export SP_CSDM_TWOPASS=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 tools/cfgbench.py --only cfg5 > $OUT/$v.log 2>&1
  echo "== $v"; grep "cfg5 csd matrix" $OUT/$v.log
  python3 - $(ls $OUT/$v/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 15000 and "at::native" not in r["Name"] and "csd_pair" not in r["Name"]:
        print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
