#!/bin/bash
# grid cap (workgroups per CU) of the row kernels against their resident workgroups per CU: tools/cap_sweep.sh
cd "$GRAFT_REPO_ROOT"
for cap in 0 2 3 4 6 9 12; do
  echo "-- SP_STRIDED_CAP=$cap"
  SP_STRIDED_CAP=$cap python3 tools/cfgbench.py --only cfg2,cfg4,hilbert 2>&1 | grep -E "cfg2|nfft=4096|4096 rows"
done
