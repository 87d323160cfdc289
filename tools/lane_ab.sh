#!/bin/bash
# streaming engine: main kernels on two lanes (SP_DIST_TWO_LANES=1) against the launch stream only (default), bench.py warm, interleaved
cd "$GRAFT_REPO_ROOT"
run() { L=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 20 --cpu-log2n 0 --log2n $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   step %.4f ms  kernel %.4f ms  value %.0f  parity %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d.get('parity',{}).get('ok')))"; }
for r in 1 2 3; do
  echo "-- round $r two lanes"; run 28 SP_DIST_TWO_LANES=1
  echo "-- round $r one lane";  run 28 SP_X=0
done
echo "-- 2^25 shard (forced collective path off): two lanes / one lane"
for r in 1 2; do
  run 25 SP_DIST_TWO_LANES=1; run 25 SP_X=0
done
