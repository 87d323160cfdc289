#!/bin/bash
# metric kernel: frames per workgroup 512 (8 MiB between the workgroups' streams) against skewed strides; bench.py warm, interleaved
cd "$GRAFT_REPO_ROOT"
run() { env "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 20 --cpu-log2n 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   step %.4f ms  kernel %.4f ms  value %.0f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"; }
for r in 1 2; do
  for f in 0 513 515 520 529 544; do
    echo "-- round $r SP_FPG1=$f"; if [ $f = 0 ]; then run SP_X=0; else run SP_FPG1=$f; fi
  done
done
