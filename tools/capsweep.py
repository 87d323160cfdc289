#!/usr/bin/env python3
"""Row-kernel grid cap sweep: SP_STRIDED_CAP=<blocks per CU> python tools/capsweep.py  (one line per shape)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


dev = torch.device("cuda", 0)
out = []
for L in (64, 256, 1024, 2048, 4096, 8192):
    for rows in (4096, 1 << 16):
        if rows * L > (1 << 28):
            continue
        x = torch.view_as_complex(torch.randn((rows, L, 2), device=dev))
        out.append("fft %5d x %6d: %.4f" % (L, rows, timed(lambda: E.fft(x))))
        xr = torch.randn((rows, L), device=dev)
        out.append("hil %5d x %6d: %.4f" % (L, rows, timed(lambda: E.hilbert_rows(xr, L))))
print("cap %s | " % os.environ.get("SP_STRIDED_CAP", "16") + " | ".join(out))
