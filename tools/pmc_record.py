#!/usr/bin/env python3
"""tools/pmc_record.py <dir> <kernel substring> <fetch correction>: mean FETCH_SIZE / WRITE_SIZE per dispatch of the metric
kernel -> profiles/pmc_metric_kernel_current.json, tied to the kernel sources by their sha256 (bench.py reads it back and
reports roofline.traffic only when the digest matches the build it runs)."""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
root, pat, corr = sys.argv[1], sys.argv[2], float(sys.argv[3])
vals = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row["Kernel_Name"]:
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (d, c), v in per.items():
        vals[c].append(v)
if not vals.get("FETCH_SIZE") or not vals.get("WRITE_SIZE"):
    raise SystemExit("no FETCH_SIZE / WRITE_SIZE rows for %r under %s" % (pat, root))
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
# the digest function only (bench imports torch at module level: read the function's source instead)
import hashlib
h = hashlib.sha256()
d = os.path.join(ROOT, "pyfft_amd", "csrc")
for name in sorted(os.listdir(d)):
    if name.endswith((".h", ".hip")):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
rec = {"kernel": pat, "log2n": 28, "nfft": 4096, "fetch_size_kib": fetch, "write_size_kib": write,
       "fetch_correction": corr,
       "traffic_bytes_per_launch": fetch * 1024.0 * corr + write * 1024.0,
       "algorithmic_bytes_per_launch": 8.0 * ((131071 - 1) * 2048 + 4096),
       "dispatches": len(vals["FETCH_SIZE"]), "source_sha256": h.hexdigest(),
       "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 tools/kbench.py; FETCH_SIZE x "
              "correction for 8 B/lane coalesced loads as calibrated by tools/pmc_calib.sh (profiles/r02_fetch_size_calibration.txt)"}
with open(os.path.join(ROOT, "profiles", "pmc_metric_kernel_current.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec, indent=1))
