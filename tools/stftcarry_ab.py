#!/usr/bin/env python3
"""A/B of the register-carried real-pair STFT (k_stft_rp<N, false, SHIFT>) against the generic fetch: run with and without
SP_STFT_NOCARRY=1.  2^26 float32 samples, one-sided complex output."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E
from pyfft_amd.windows import windows
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(2)
x = torch.randn(1 << 26, generator=g, device=dev, dtype=torch.float32)
for nfft in (1024, 2048, 4096, 8192):
    for ov in (0.75, 0.5):
        hop = int(nfft * (1 - ov)); M = (x.numel() - nfft) // hop + 1
        win = windows("Hanning", nwins=nfft, verbose=False)
        ts = []
        for i in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            X = E.stft_frames(x, win, hop, M, detrend=False, sided=E.SIDED_ONE)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        del X
        print("nfft %5d overlap %.2f: %.3f ms" % (nfft, ov, 1e3 * float(np.mean(ts[3:]))))
