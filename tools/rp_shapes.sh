#!/bin/bash
# real-input Welch PSD (k_welch_rp) over transform lengths: partition by the kernel's residency (default) against the old 4 groups per CU
cd "$GRAFT_REPO_ROOT"
for nfft in 256 512 1024 2048 8192; do
  for ov in 0.5 0.75; do
    a=$(python3 tools/kbench.py --real --nfft $nfft --ov $ov --log2n 27 --reps 20 2>&1 | grep "k_welch" | sed 's/.*k_welch \([0-9.]*\) ms.*/\1/')
    b=$(SP_GROUPS_PER_CU=4 python3 tools/kbench.py --real --nfft $nfft --ov $ov --log2n 27 --reps 20 2>&1 | grep "k_welch" | sed 's/.*k_welch \([0-9.]*\) ms.*/\1/')
    echo "nfft $nfft overlap $ov: residency rule $a ms   4 per CU $b ms"
  done
done
