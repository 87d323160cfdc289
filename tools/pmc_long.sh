#!/bin/bash
# HBM-side traffic of the long whole-signal transforms' passes (Hilbert 2^24, ccf 2^24): FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 passes (kernel-trace only), mean per dispatch; bytes = FETCH_SIZE [KiB] x 2 x 1024 for reads (the gfx950 correction
# calibrated in profiles/r02_fetch_size_calibration.txt), WRITE_SIZE [KiB] x 1024 for writes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_long
rm -rf $OUT; mkdir -p $OUT
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  SP_CFGBENCH_ISOLATED=1 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 tools/cfgbench.py --only hilbert,xcorr --reps 3 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for d, cs in per.items():
        for c, v in cs.items():
            acc[names[d]][c].append(v)
print("%-74s %6s %12s %12s" % ("kernel", "calls", "read MB", "written MB"))
for k in sorted(acc):
    if not any(t in k for t in ("k_fft_cols", "rowsmid", "k_moments", "k_hilbert<")):
        continue
    f = acc[k].get("FETCH_SIZE", [])
    w = acc[k].get("WRITE_SIZE", [])
    rd = (sum(f) / len(f)) * 2 * 1024 / 1e6 if f else float("nan")
    wr = (sum(w) / len(w)) * 1024 / 1e6 if w else float("nan")
    print("%-74s %6d %12.1f %12.1f" % (k[:74], max(len(f), len(w)), rd, wr))
PY
