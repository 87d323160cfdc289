#!/bin/bash
# average shader clock during the metric kernel for library variants: GRBM_GUI_ACTIVE (summed over 8 XCDs) / duration
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ $v = main ]; then lib=pyfft_amd/lib/libspectral.so; else lib=build/variants/$v/libspectral.so; fi
  out=gpurun_out/clk_$v
  rm -rf $out
  SP_LIB_PATH=$lib rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 tools/kbench.py --reps 10 > $out.log 2>&1
  python3 - "$out" "$v" <<'PY'
import csv, glob, sys
root, name = sys.argv[1], sys.argv[2]
cc = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if "k_welch_carry" in r["Kernel_Name"] and "false>" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
val = {}
for r in csv.DictReader(open(cc)):
    if r["Dispatch_Id"] in dur and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        val[r["Dispatch_Id"]] = val.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
ids = sorted(val, key=int)[2:]
ghz = [val[i] / 8.0 / dur[i] for i in ids]
print("%-8s kernel %.3f ms  clock %.2f GHz  (n=%d)" % (name, sum(dur[i] for i in ids) / len(ids) / 1e6, sum(ghz) / len(ghz), len(ids)))
PY
done
