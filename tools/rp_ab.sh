#!/bin/bash
# real-input Welch PSD (k_welch_rp, 2^26 samples, nfft 2048, 75 %): 2-wave hint + 4 groups per CU (main) against the plain form at 3 waves
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  echo "-- round $r main"; python3 tools/cfgbench.py --only cfg3 2>&1 | grep "power only"
  for g in 3 6 4; do
    echo "-- round $r nohint groups/CU $g"; SP_GROUPS_PER_CU=$g SP_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/rpnohint/libspectral.so python3 tools/cfgbench.py --only cfg3 2>&1 | grep "power only"
  done
  echo "-- round $r main groups/CU 6"; SP_GROUPS_PER_CU=6 python3 tools/cfgbench.py --only cfg3 2>&1 | grep "power only"
done
