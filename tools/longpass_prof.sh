#!/bin/bash
# per-pass kernel times of the long whole-signal transforms (Hilbert 2^24, ccf 2^24): rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/longpass
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h -- python3 tools/cfgbench.py --only hilbert > $OUT/h.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x -- python3 tools/cfgbench.py --only xcorr > $OUT/x.log 2>&1
for t in h x; do
  f=$(ls $OUT/$t/*/*kernel_stats.csv | head -1)
  echo "== $t: $f"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_fft_cols", "k_fft_rows_rev", "k_hilbert_mid", "k_xc_mid", "k_moments", "k_hilbert<", "rowsmid", "k_xcorr_norm")):
        print("%-90s calls %5s  avg %9.1f us" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
