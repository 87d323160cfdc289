#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (GPU box): tools/pmc_calib.sh  -> gpurun_out/pmc_calib/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_calib
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pass1 -- tools/ubench/fetch_calib > $OUT/pass1.log 2>&1 || echo "pass 1 failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pass2 -- tools/ubench/fetch_calib > $OUT/pass2.log 2>&1 || echo "pass 2 failed"
python3 tools/pmc_calib.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
