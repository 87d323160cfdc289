#!/bin/bash
# cost of leaving CUs to the collective: bench.py through the RCCL group of one rank (SP_BENCH_FORCE_DIST=1), main kernel over all
# CUs against ncu - 4 / ncu - 8, at 2^28 and 2^25 samples
cd "$GRAFT_REPO_ROOT"
run() { L=$1; shift; env SP_BENCH_FORCE_DIST=1 "$@" timeout -k 10 300 python bench.py --steps 300 --warmup 20 --cpu-log2n 0 --gate-log2n 0 --log2n $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   step %.4f ms  kernel %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for L in 28 25; do for r in 0 4 8; do echo "-- 2^$L reserve $r"; run $L SP_DIST_RESERVE_CUS=$r; done; done
