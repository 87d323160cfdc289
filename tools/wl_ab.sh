#!/bin/bash
# k_welch_carry at <= 1024 points: exchanges without workgroup barriers (main) against with (build/variants/nowl); complex Welch PSD, 2^27
cd "$GRAFT_REPO_ROOT"
M=$GRAFT_REPO_ROOT/pyfft_amd/lib/libspectral.so; V=$GRAFT_REPO_ROOT/build/variants/nowl/libspectral.so
for nfft in 256 512 1024; do
  for ov in 0.5 0.75; do
    a=$(SP_LIB_PATH=$M python3 tools/kbench.py --nfft $nfft --ov $ov --log2n 27 --reps 20 2>&1 | grep "k_welch" | head -1 | sed 's/.*k_welch \([0-9.]*\) ms.*/\1/')
    b=$(SP_LIB_PATH=$V python3 tools/kbench.py --nfft $nfft --ov $ov --log2n 27 --reps 20 2>&1 | grep "k_welch" | head -1 | sed 's/.*k_welch \([0-9.]*\) ms.*/\1/')
    echo "nfft $nfft overlap $ov: wave-local $a ms   workgroup barriers $b ms"
  done
done
