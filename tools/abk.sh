#!/bin/bash
# tools/abk.sh <rounds> <variant>...   -- A/B of library variants.  variant = name[:ENV=V[,ENV=V...]] with name = "main" (the
# in-tree library) or build/variants/<name>/libspectral.so: a parity check per variant first, then kbench for each
# variant, <rounds> times, interleaved
rounds=$1; shift
lib() { n=${1%%:*}; if [ "$n" = main ]; then echo pyfft_amd/lib/libspectral.so; else echo build/variants/$n/libspectral.so; fi; }
envs() { case "$1" in *:*) echo "${1#*:}" | tr ',' ' ';; *) echo "";; esac; }
for v in "$@"; do
  echo "== parity $v"
  env $(envs $v) SP_LIB_PATH=$(lib $v) timeout -k 10 120 python tools/kbench.py --check --reps 3 --log2n 24 2>&1 | grep -E "parity|Error|error" || exit 1
done
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    echo "[$r] $v"
    env $(envs $v) SP_LIB_PATH=$(lib $v) timeout -k 10 120 python tools/kbench.py ${KBENCH_ARGS} 2>&1 | grep welch || exit 1
  done
done
