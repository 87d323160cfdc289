#!/bin/bash
# PMC passes for the cfg5 contraction kernel (GPU box).  usage: tools/pmc_cfg5.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}
OUT=gpurun_out/pmc_cfg5_$TAG
mkdir -p $OUT
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES"
P2="FETCH_SIZE"
P3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 tools/cfgbench.py --only cfg5 --reps 2 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT "${2:-k_csdm_bf16}" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
