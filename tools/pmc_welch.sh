#!/bin/bash
# PMC passes for the Welch kernel (GPU box). Each pass is its own rocprofv3 run (no trace domains besides kernel-trace).
# usage: tools/pmc_welch.sh <tag>   -> gpurun_out/pmc_<tag>/passN/...
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 tools/kbench.py --reps 2 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT "sp::k_welch_carry<4096, true, 8, true>" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
