#!/bin/bash
# cfg4 (FIR 513 taps, 2^28 samples): two workgroups per CU with prefetch (main) against three without (build/variants/fir3)
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  echo "-- round $r main"; python3 tools/cfgbench.py --only cfg4 2>&1 | grep "nfft=4096"
  for cap in 3 6 12; do
    echo "-- round $r fir3 cap $cap"; SP_STRIDED_CAP=$cap SP_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/fir3/libspectral.so python3 tools/cfgbench.py --only cfg4 2>&1 | grep "nfft=4096"
  done
done
