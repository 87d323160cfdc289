# per-role busy / barrier-wait cycles of k_welch_pipe (diagnostic builds with s_memtime stamps around every barrier)
for n in "$@"; do
 echo "== $n"
 SP_LIB_PATH=build/variants/pipe_$n/libspectral.so SP_WELCH_PIPE=1 SP_PIPE_GPC=1 timeout -k 10 200 python tools/kbench.py --reps 1 2>&1 | grep -E "^block 3|rror" | sort | awk '{k=$4; b[k]+=$6; w[k]+=$8; q[k]+=$10; n[k]++} END {for (k in b) printf "role %s busy/period %.0f wait/period %.0f load-issue/period %.0f\n", k, b[k]/n[k]/516, w[k]/n[k]/516, q[k]/n[k]/516}' | sort
done
