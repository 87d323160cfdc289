#!/bin/bash
# cfg3: FAST form with the window in registers (main: 178 VGPRs, two workgroups per CU) against the window in LDS (tools/build_file_variant.sh stftw0 pyfft_amd/csrc/k_stft.hip -DSP_STFT_WLDS=0 gives the register form; this script compared main-with-registers against build/variants/stftw:
# 158 VGPRs, three per CU) at 3 / 6 groups per CU; sustained ms per call
cd "$GRAFT_REPO_ROOT"
M=$GRAFT_REPO_ROOT/pyfft_amd/lib/libspectral.so; V=$GRAFT_REPO_ROOT/build/variants/stftw/libspectral.so
run() { env "$@" python3 tools/cfgbench.py --only cfg3 2>&1 | grep "cfg3 stft"; }
for r in 1 2; do
  echo "-- window in registers, 2 resident, 4 groups/CU"; run SP_LIB_PATH=$M
  echo "-- window in LDS, 3 resident, 3 groups/CU"; run SP_LIB_PATH=$V SP_GROUPS_PER_CU=3
  echo "-- window in LDS, 3 resident, 6 groups/CU"; run SP_LIB_PATH=$V SP_GROUPS_PER_CU=6
  echo "-- window in LDS, 3 resident, 12 groups/CU"; run SP_LIB_PATH=$V SP_GROUPS_PER_CU=12
done
