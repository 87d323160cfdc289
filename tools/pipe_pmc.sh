#!/bin/bash
# PMC passes for k_welch_pipe (GPU box): tools/pipe_pmc.sh <variant>...  -> gpurun_out/pipe_pmc_<variant>.txt
# (variant = main | directory under build/variants).  One rocprofv3 run per counter group, kernel-trace only.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
P3="SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH"
for v in "$@"; do
  if [ $v = main ]; then lib=pyfft_amd/lib/libspectral.so; else lib=build/variants/$v/libspectral.so; fi
  OUT=gpurun_out/pipe_pmc_$v
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    SP_LIB_PATH=$lib SP_WELCH_PIPE=1 SP_PIPE_GPC=1 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pass$i -- python3 tools/kbench.py --reps 4 > $OUT/pass$i.log 2>&1 || echo "pass $i failed ($v)"
  done
  python3 tools/pmc_summary.py $OUT "k_welch_pipe" > gpurun_out/pipe_pmc_$v.txt 2>&1
  python3 - $OUT >> gpurun_out/pipe_pmc_$v.txt <<'PY'
import csv, glob, sys
kt = glob.glob(sys.argv[1] + "/pass1/**/*kernel_trace.csv", recursive=True)[0]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "k_welch_pipe" in r["Kernel_Name"]]
print("kernel duration under PMC pass 1: n=%d mean=%.4f ms" % (len(d), sum(d) / len(d) / 1e6))
PY
  echo "== $v"; cat gpurun_out/pipe_pmc_$v.txt
done
