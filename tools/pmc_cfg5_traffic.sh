#!/bin/bash
# HBM traffic of cfg5's two big kernels (GPU box): FETCH_SIZE and WRITE_SIZE in separate passes, per dispatch
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_cfg5_traffic
rm -rf $OUT; mkdir -p $OUT
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 tools/cfgbench.py --only cfg5 --reps 2 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
for k in "k_csdm_bf16" "k_welch_pipe<false, 8, 7>" "k_csdm_fold"; do
  echo "== $k"; python3 tools/pmc_summary.py $OUT "$k"
done
