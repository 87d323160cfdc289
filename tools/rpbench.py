#!/usr/bin/env python3
"""Real-input Welch PSD (k_welch_rp) timing over transform lengths and overlaps, 2^26 float32 samples on one MI355X."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E
from pyfft_amd.windows import windows

n = 1 << 26
x = torch.randn(n, device="cuda", dtype=torch.float32)
for nfft in (512, 1024, 2048, 4096, 8192):
    for ov in (2, 4):
        hop = nfft // ov
        M = (n - nfft) // hop + 1
        win = windows("Hanning", nwins=nfft, verbose=False)
        f = lambda: E.welch_psd(x, win, hop, M, detrend=False, sided=E.SIDED_ONE, scale=1.0)
        f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("welch_rp nfft=%5d hop=nfft/%d  %.3f ms" % (nfft, ov, float(np.median(ts))), flush=True)
