#!/usr/bin/env python3
"""time of consecutive sp_stft_cog calls (events around each call), metric shape: does it drift?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E
from pyfft_amd.windows import windows
n, nfft, hop = 1 << 28, 4096, 2048
x = torch.view_as_complex(torch.randn((n, 2), device="cuda", dtype=torch.float32))
M = (n - nfft) // hop + 1
for name, win, det in (("rect, no detrend (mode 2)", np.ones(nfft), False), ("Hann, mean detrend (mode 8)", windows("Hanning", nwins=nfft, verbose=False), True)):
    ks = []
    for i in range(24):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        E.stft_cog(x, win, hop, M, 1.0e6, detrend=det)
        b.record(); torch.cuda.synchronize()
        ks.append(a.elapsed_time(b))
    print(name, " ".join("%.3f" % k for k in ks))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        E.stft_cog(x, win, hop, M, 1.0e6, detrend=det)
    e1.record(); torch.cuda.synchronize()
    print("   20 calls back to back: %.3f ms per call" % (e0.elapsed_time(e1) / 20))
# the PSD kernel the same way
win = windows("Hanning", nwins=nfft, verbose=False)
E.profile_enable(True)
ks = []
for i in range(24):
    E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    ks.append(E.profile_last_ms())
E.profile_enable(False)
print("welch psd", " ".join("%.3f" % k for k in ks))
