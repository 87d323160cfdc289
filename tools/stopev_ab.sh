#!/bin/bash
# streaming engine: the main kernel's event as its launch's completion signal (default) against a hipEventRecord behind it
# (SP_DIST_RECORD_EVENT=1); bench.py warm, interleaved, 2^28 and 2^25 samples, also through the RCCL group of one rank
cd "$GRAFT_REPO_ROOT"
run() { L=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 400 --warmup 20 --cpu-log2n 0 --gate-log2n 0 --log2n $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   step %.4f ms  kernel %.4f ms  host %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d.get('host_enqueue_ms_per_step', 0)))"; }
for r in 1 2; do
  for L in 28 25; do
    echo "-- 2^$L stop event";   run $L SP_X=0
    echo "-- 2^$L record";       run $L SP_DIST_RECORD_EVENT=1
  done
  echo "-- 2^25 RCCL group of one, reserve 8, stop event"; run 25 SP_BENCH_FORCE_DIST=1 SP_DIST_RESERVE_CUS=8
  echo "-- 2^25 RCCL group of one, reserve 8, record";     run 25 SP_BENCH_FORCE_DIST=1 SP_DIST_RESERVE_CUS=8 SP_DIST_RECORD_EVENT=1
done
