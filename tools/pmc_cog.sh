#!/bin/bash
# PMC passes for the per-frame centre-of-gravity kernels (GPU box).  usage: tools/pmc_cog.sh <tag>
# streaming form (k_welch_carry<..., COG>) and, with SP_COG_GENERIC=1 exported by the caller, the generic k_stft form
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}
OUT=gpurun_out/pmc_cog_$TAG
mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="FETCH_SIZE"
P3="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 tools/cfgbench.py --only cog --reps 2 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT "${2:-k_welch_carry_cog<4096, true, 8>}" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
