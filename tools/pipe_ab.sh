# A/B of the wave-specialised Welch kernel (k_welch_pipe.hip) against k_welch_carry and of its build variants
# usage: tools/pipe_ab.sh <rounds> <variant>...   (variant = directory under build/variants; "main" = in-tree library; "carry" = k_welch_carry)
rounds=$1; shift
for r in $(seq 1 $rounds); do
for n in "$@"; do
 lib=build/variants/$n/libspectral.so; pipe=1
 [ $n = main ] && lib=pyfft_amd/lib/libspectral.so
 [ $n = carry ] && lib=pyfft_amd/lib/libspectral.so && pipe=0
 if [ $r = 1 ]; then SP_LIB_PATH=$lib SP_WELCH_PIPE=$pipe SP_PIPE_GPC=1 timeout -k 10 200 python tools/kbench.py --check --check-log2n 23 --reps 1 --log2n 23 2>&1 | grep -E "parity detrend=1|rror"; fi
 echo "[$r] $n"
 SP_LIB_PATH=$lib SP_WELCH_PIPE=$pipe SP_PIPE_GPC=1 timeout -k 10 200 python tools/kbench.py --reps 20 2>&1 | grep -E "detrend=1|rror"
done
done
