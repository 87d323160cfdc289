#!/bin/bash
# usage: tools/overlap_trace.sh <log2n> : kernel trace of the streamed steps, are epilogue and next main kernel concurrent?
L=${1:-25}
OUT=gpurun_out/overlap$L
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 bench.py --steps 60 --warmup 5 --settle-steps 10 --log2n $L --cpu-log2n 0 > $OUT/run.log 2>&1
python3 tools/overlap_trace.py $OUT/tr k_welch_pipe 6 > $OUT/overlap.txt 2>&1
rm -rf $OUT/tr
cat $OUT/overlap.txt
