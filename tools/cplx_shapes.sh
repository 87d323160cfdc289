#!/bin/bash
# complex Welch PSD (k_welch_carry / k_welch) over transform lengths: groups per CU 4 (default rule) against 3 / 6 / 8
cd "$GRAFT_REPO_ROOT"
for nfft in 256 512 1024 2048 8192; do
  line="nfft $nfft:"
  for g in 4 3 6 8; do
    a=$(SP_GROUPS_PER_CU=$g python3 tools/kbench.py --nfft $nfft --ov 0.5 --log2n 27 --reps 20 2>&1 | grep "k_welch" | head -1 | sed 's/.*k_welch \([0-9.]*\) ms.*/\1/')
    line="$line  gpc$g $a"
  done
  echo "$line"
done
