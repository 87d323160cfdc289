#!/usr/bin/env python3
"""Wall time of consecutive blocks of 5 bench steps (sync only between blocks): shows warm-up drift."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyfft_amd import engine as E
from pyfft_amd.windows import windows
dev = torch.device("cuda", 0)
nfft, hop = 4096, 2048
x = bench.synth_stream(0, 1 << 28, dev, 1)
win = windows("Hanning", nwins=nfft, verbose=False)
M = ((1 << 28) - nfft) // hop + 1
torch.cuda.synchronize()
out = []
for b in range(12):
    t0 = time.perf_counter()
    for _ in range(5):
        E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    torch.cuda.synchronize()
    out.append(1e3 * (time.perf_counter() - t0) / 5)
print(" ".join("%.3f" % v for v in out))
