#!/bin/bash
# tools/abk.sh <rounds> <variant>...   -- A/B of library variants (build/variants/<name>/libspectral.so; "main" = the
# in-tree library): a parity check per variant first, then kbench for each variant, <rounds> times, interleaved
rounds=$1; shift
lib() { if [ "$1" = main ]; then echo pyfft_amd/lib/libspectral.so; else echo build/variants/$1/libspectral.so; fi; }
for v in "$@"; do
  echo "== parity $v"
  SP_LIB_PATH=$(lib $v) timeout -k 10 120 python tools/kbench.py --check --reps 3 --log2n 24 2>&1 | grep -E "parity|Error|error" || exit 1
done
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    echo "[$r] $v"
    SP_LIB_PATH=$(lib $v) timeout -k 10 120 python tools/kbench.py ${KBENCH_ARGS} 2>&1 | grep welch || exit 1
  done
done
