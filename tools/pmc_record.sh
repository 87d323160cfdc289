#!/bin/bash
# HBM traffic record of the metric kernel for bench.py's roofline.traffic (GPU box):
#   tools/pmc_record.sh   -> profiles/pmc_metric_kernel_current.json (+ gpurun_out/pmc_record/*)
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (they do not fit one pass), kernel-trace only.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_record
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pass1 -- python3 tools/kbench.py --reps 3 > $OUT/pass1.log 2>&1 || echo "pass 1 failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pass2 -- python3 tools/kbench.py --reps 3 > $OUT/pass2.log 2>&1 || echo "pass 2 failed"
python3 tools/pmc_record.py $OUT "k_welch_pipe<true, 8, 9>" ${1:-2.0} > $OUT/record.log 2>&1
cat $OUT/record.log
cp profiles/pmc_metric_kernel_current.json gpurun_out/pmc_record/ 2>/dev/null
