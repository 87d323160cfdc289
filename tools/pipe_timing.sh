# per-role busy / barrier-wait cycles of k_welch_pipe (diagnostic builds with s_memtime stamps around every barrier)
for n in timing timing_a8 timing_a2 timing_a1; do
 echo "== $n"
 SP_LIB_PATH=build/variants/pipe_$n/libspectral.so SP_WELCH_PIPE=1 SP_PIPE_GPC=1 timeout -k 10 200 python tools/kbench.py --reps 1 2>&1 | grep -E "^block|detrend=1|rror" | sort | uniq -c | sort -k2,5 | tail -16
done
