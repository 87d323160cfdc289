#!/bin/bash
# A/B of the k_op_fused hand-off variants (build/variants/opf{1,2,3}): kernel timeline of a 2^25-sample step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/opf_ab
rm -rf $OUT && mkdir -p $OUT
for v in 0 1 2 3; do
  if [ $v != 0 ]; then export SP_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/opf$v/libspectral.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/v$v -- python3 bench.py --steps 100 --warmup 5 --settle-steps 10 --log2n 25 --cpu-log2n 0 > $OUT/v$v.log 2>&1
  echo "== variant $v" >> $OUT/summary.txt
  python3 tools/trace_gaps.py $OUT/v$v "k_welch_pipe" 80 >> $OUT/summary.txt 2>&1
  rm -rf $OUT/v$v
done
cat $OUT/summary.txt
