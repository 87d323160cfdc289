#!/bin/bash
# warm A/B of library builds with bench.py itself (settle + warm-up + 300 timed steps, kernel time by HIP events), interleaved rounds
# on one box:  tools/bench_ab.sh <rounds> <variant>...   (variant = directory under build/variants; "main" = the in-tree library)
rounds=$1; shift
for r in $(seq 1 $rounds); do
for n in "$@"; do
 lib=$GRAFT_REPO_ROOT/build/variants/$n/libspectral.so
 [ $n = main ] && lib=$GRAFT_REPO_ROOT/pyfft_amd/lib/libspectral.so
 SP_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 300 --warmup 20 --cpu-log2n 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$r] %-12s step %.4f ms  kernel %.4f ms  frac %.3f  %s' % ('$n', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'][:34]))"
done
done
