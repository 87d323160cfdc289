#!/bin/bash
# per-step GPU timeline (kernels + gaps) of bench.py's step at a shard size: single-GPU path and sharded path (RCCL group of one)
# usage: tools/steptrace_r3.sh <log2n> [outdir]
L=${1:-25}
OUT=${2:-gpurun_out/steptrace}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/single -- python3 bench.py --steps 60 --warmup 5 --settle-steps 10 --log2n $L --cpu-log2n 0 > $OUT/single.log 2>&1
python3 tools/trace_gaps.py $OUT/single "k_welch_pipe" 40 > $OUT/single_gaps.txt 2>&1
SP_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/dist -- python3 bench.py --steps 60 --warmup 5 --settle-steps 10 --log2n $L --cpu-log2n 0 --gate-log2n 0 > $OUT/dist.log 2>&1
python3 tools/trace_gaps.py $OUT/dist "k_welch_pipe" 40 > $OUT/dist_gaps.txt 2>&1
cat $OUT/single_gaps.txt $OUT/dist_gaps.txt
