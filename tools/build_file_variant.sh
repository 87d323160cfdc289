#!/bin/bash
# tools/build_file_variant.sh <name> <source.hip> <flags...>: variant library that differs from the main build only in one
# translation unit -> build/variants/<name>/libspectral.so; needs `make` done first (reuses build/obj/*.o)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift; shift
OUT=build/variants/$NAME
mkdir -p $OUT
B=$(basename $SRC .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Iinclude -Ipyfft_amd/csrc -Wno-unused-function "$@" -c pyfft_amd/csrc/$B.hip -o $OUT/$B.o
OBJS=$(ls build/obj/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $OUT/$B.o -o $OUT/libspectral.so
ls -la $OUT/libspectral.so
