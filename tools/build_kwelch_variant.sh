#!/bin/bash
# tools/build_kwelch_variant.sh <name> <flags...>: variant library that differs from the main build only in k_welch.hip
# (the carry kernel) -> build/variants/<name>/libspectral.so; needs `make` done first (reuses build/obj/*.o)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=build/variants/$NAME
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Iinclude -Ipyfft_amd/csrc -Wno-unused-function "$@" -c pyfft_amd/csrc/k_welch.hip -o $OUT/k_welch.o
OBJS=$(ls build/obj/*.o | grep -v k_welch.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $OUT/k_welch.o -o $OUT/libspectral.so
ls -la $OUT/libspectral.so
