#!/bin/bash
# smallest shard that takes the pipeline kernel (frames per CU threshold SP_PIPE_MINFPC): pipeline against the symmetric kernel
cd "$GRAFT_REPO_ROOT"
run() { L=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 400 --warmup 20 --cpu-log2n 0 --gate-log2n 0 --log2n $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   step %.4f ms  kernel %.4f ms  %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'][:40]))"; }
for L in ${SIZES:-24 23 22 21 20 19}; do
  for m in 4096 1; do echo "-- 2^$L SP_PIPE_MINFPC=$m"; run $L SP_PIPE_MINFPC=$m; done
done
