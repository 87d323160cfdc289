#!/usr/bin/env python3
"""Host-side cost of one E.welch_psd call: tiny device-resident input, so the GPU is never the bottleneck."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E
from pyfft_amd.windows import windows
dev = torch.device("cuda", 0)
nfft = 4096
x = torch.view_as_complex(torch.randn((1 << 16, 2), device=dev))
win = windows("Hanning", nwins=nfft, verbose=False)
M = (x.numel() - nfft) // (nfft // 2) + 1
for prof in (False, True):
    E.profile_enable(prof)
    for _ in range(10):
        E.welch_psd(x, win, nfft // 2, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 500
    for _ in range(n):
        E.welch_psd(x, win, nfft // 2, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("profile hook %s: host %.1f us per call (enqueue only), %.1f us per call incl. drain" % (prof, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
