export SP_WELCH_NOPIPE=1
for v in main a1 a2 a3 a8 a9 a10 a11; do
  if [ $v = main ]; then lib=pyfft_amd/lib/libspectral.so; else lib=build/variants/$v/libspectral.so; fi
  echo "== $v"
  SP_LIB_PATH=$lib timeout -k 10 100 python tools/kbench.py 2>&1 | grep "detrend=0"
done
