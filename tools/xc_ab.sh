#!/bin/bash
# A/B of the long ccf / Hilbert on one box: default against the switches given as arguments (e.g. SP_COLS_NOXPAIR=1), interleaved
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  echo "-- round $r default"; python3 tools/cfgbench.py --only xcorr 2>&1 | grep -i "ccf 2"; python3 tools/cfgbench.py --only hilbert 2>&1 | grep -i "one row"
  for sw in "$@"; do
    echo "-- round $r $sw"; env $sw python3 tools/cfgbench.py --only xcorr 2>&1 | grep -i "ccf 2"; env $sw python3 tools/cfgbench.py --only hilbert 2>&1 | grep -i "one row"
  done
done
