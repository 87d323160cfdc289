#!/bin/bash
# Diagnostic library variants: tools/build_variant.sh <name> <extra hipcc flags...>  ->  build/variants/<name>/libspectral.so
# (select with SP_LIB_PATH=build/variants/<name>/libspectral.so; used for the ablation tables in DESIGN.md)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=build/variants/$NAME
mkdir -p $OUT
for f in pyfft_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Iinclude -Ipyfft_amd/csrc -Wno-unused-function "$@" -c $f -o $OUT/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/*.o -o $OUT/libspectral.so
ls -la $OUT/libspectral.so
