# kernel time of the metric kernel for ablation builds: tools/ablate_sweep.sh <prefix> (variants <prefix>8/9/10 = no loads /
# + no butterflies / + no LDS exchanges; build with tools/build_variant.sh <name> -DSP_ABLATE=<bits> [-DSP_PACKED=0])
P=${1:-pa}
for v in main ${P}8 ${P}9 ${P}10; do
  if [ $v = main ]; then lib=pyfft_amd/lib/libspectral.so; else lib=build/variants/$v/libspectral.so; fi
  echo "== $v"
  SP_LIB_PATH=$lib timeout -k 10 100 python tools/kbench.py 2>&1 | grep "detrend=0"
done
