#!/bin/bash
# does the 8.6 GB spectra round trip of cfg5 get cheaper when the chunk fits the memory-side cache?  frames per chunk swept
cd "$GRAFT_REPO_ROOT"
for c in 0 4096 2048 1024 512 256; do
  echo "-- chunk $c frames ($((c/2*2)) MiB of packed spectra)"
  if [ $c = 0 ]; then python3 tools/cfgbench.py --only cfg5 --cfg5-detrend 0 2>&1 | grep "csd matrix"
  else SP_CSDM_CHUNK=$c SP_CSDM_MINPAIRS=8 python3 tools/cfgbench.py --only cfg5 --cfg5-detrend 0 2>&1 | grep "csd matrix"; fi
done
