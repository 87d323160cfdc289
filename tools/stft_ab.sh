#!/bin/bash
# cfg3 (STFT 2^26 f32, nfft 2048, 75 %, one-sided complex out): generic k_stft_rp (SP_STFT_NOFAST=1) against the compile-time one-sided
# form at two workgroups per CU (main) and at three (build/variants/stftfast3 = tools/build_file_variant.sh stftfast3
# pyfft_amd/csrc/k_stft.hip -DSP_STFT_FAST_EU=3: 13 spilled registers), with matching groups per CU
cd "$GRAFT_REPO_ROOT"
M=$GRAFT_REPO_ROOT/pyfft_amd/lib/libspectral.so; V=$GRAFT_REPO_ROOT/build/variants/stftfast3/libspectral.so
run() { env "$@" python3 tools/cfgbench.py --only cfg3 2>&1 | grep "cfg3 stft"; }
for r in 1 2; do
  echo "-- generic (2 resident, 4 groups/CU)"; run SP_STFT_NOFAST=1 SP_LIB_PATH=$M
  echo "-- fast, 2 resident, 4 groups/CU"; run SP_LIB_PATH=$M
  echo "-- fast, 3 resident (spills), 3 groups/CU"; run SP_LIB_PATH=$V SP_GROUPS_PER_CU=3
  echo "-- fast, 3 resident (spills), 6 groups/CU"; run SP_LIB_PATH=$V SP_GROUPS_PER_CU=6
done
