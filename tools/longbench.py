#!/usr/bin/env python3
"""Time the reference's own test_fftanal shape (N = 2^19, Navr = 8 -> 116 508-point segments, fft_analysis.py:2950-2993) through
the drop-in (GPU long-segment path, host staging and epilogue included) and through the CPU oracle.
  python tools/longbench.py [--reps 5]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs_long                      # noqa: E402
import pyfft_amd as P                   # noqa: E402
from oracle import cpu_ref as O         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    t, x, y = inputs_long.long_signals()
    kw = dict(Navr=8, windowfunction="hamming", detrend_style=1, onesided=True)
    P.fft_pwelch(t, x, y, [t[0], t[-1]], **kw)
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        r = P.fft_pwelch(t, x, y, [t[0], t[-1]], **kw)
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    ro = O.fft_pwelch(t, x, y, tbounds=[t[0], t[-1]], **kw)
    tc = time.perf_counter() - t0
    err = float(np.max(np.abs(r[2] - ro[2])) / np.max(np.abs(ro[2])))
    print("fft_pwelch N=2^19 Navr=8 (nwins=%d): drop-in %.1f ms per call (min %.1f; numpy in, numpy out, epilogue included), "
          "CPU oracle %.0f ms; max |dPxx| / max Pxx = %.1e" % (r[6].nwins, 1e3 * np.median(ts), 1e3 * min(ts), 1e3 * tc, err))
    import torch
    E = P.engine
    xs = torch.from_numpy(x.astype(np.float32)).cuda()
    ys = torch.from_numpy(y.astype(np.float32)[None, :]).cuda()
    nw = int(r[6].nwins)
    win = np.asarray(r[6].win)
    hop = nw - int(r[6].noverlap)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps + 1):
        t0 = time.perf_counter()
        E.welch_csd(xs, ys, win, hop, 8, detrend=True, sided=E.SIDED_ONE, scale=1.0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("   device-resident sp_welch_csd alone (8 frames x 2 signals): %.2f ms" % (1e3 * np.median(ts[1:])))


if __name__ == "__main__":
    main()
