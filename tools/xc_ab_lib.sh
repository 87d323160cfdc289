#!/bin/bash
# A/B of library builds on the long ccf / Hilbert, interleaved on one box: tools/xc_ab_lib.sh <variant>... (build/variants/<name>; "main")
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  for n in "$@"; do
    lib=$GRAFT_REPO_ROOT/build/variants/$n/libspectral.so
    [ $n = main ] && lib=$GRAFT_REPO_ROOT/pyfft_amd/lib/libspectral.so
    echo "-- round $r $n"
    SP_LIB_PATH=$lib python3 tools/cfgbench.py --only xcorr 2>&1 | grep -i "ccf 2"
    SP_LIB_PATH=$lib python3 tools/cfgbench.py --only hilbert 2>&1 | grep -i "one row"
    SP_LIB_PATH=$lib python3 tools/cfgbench.py --only fft 2>&1 | grep -i "2^2[0-9]\|long" | head -3
  done
done
