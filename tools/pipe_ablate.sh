# ablation builds of k_welch_pipe (results wrong by construction): 1 = no butterflies, 2 = no LDS exchanges, 8 = no global loads in the loop
for r in 1 2; do
for n in main a8 a2 a1 a10 a9 a11; do
 lib=build/variants/pipe_$n/libspectral.so; [ $n = main ] && lib=pyfft_amd/lib/libspectral.so
 echo "[$r] $n"
 SP_LIB_PATH=$lib SP_WELCH_PIPE=1 SP_PIPE_GPC=1 timeout -k 10 200 python tools/kbench.py --reps 20 2>&1 | grep -E "detrend=1|rror"
done
done
