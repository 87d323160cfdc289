#!/usr/bin/env python3
"""Default dispatch against its alternatives away from the headline shapes (one GPU): cog at 0 / 75 % overlap, the CSD matrix and
the reference-against-channels CSD at small channel counts.  Prints ms per call (median of 5 isolated calls)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E
from pyfft_amd.windows import windows

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return timed(fn)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
nfft = 4096
win = windows("Hanning", nwins=nfft, verbose=False)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "csdm"):
    n = 1 << 24
    for nch in (4, 8, 16, 32):
        x = torch.randn((nch, n), generator=g, device=dev, dtype=torch.float32) + 0.5
        M = (n - nfft) // 2048 + 1
        f = lambda: E.csd_matrix(x, win, 2048, M, detrend=True, scale=1.0)
        print("csd_matrix %2d ch x 2^24: default %.3f  SP_CSDM_TWOPASS %.3f  SP_CSDM_NOPIPESPEC %.3f  SP_CSDM_SPLIT3 %.3f ms" %
              (nch, timed(f), with_env({"SP_CSDM_TWOPASS": "1"}, f), with_env({"SP_CSDM_NOPIPESPEC": "1"}, f),
               with_env({"SP_CSDM_SPLIT3": "1"}, f)), flush=True)
        del x
if which in ("all", "pair"):
    n = 1 << 24
    for nch in (2, 4, 8, 16):
        x = torch.randn((nch + 1, n), generator=g, device=dev, dtype=torch.float32) + 0.5
        M = (n - nfft) // 2048 + 1
        f = lambda: E.welch_csd(x[0], x[1:], win, 2048, M, detrend=True, sided=E.SIDED_ONE, scale=1.0)
        print("welch_csd ref x %2d ch x 2^24: default %.3f  SP_CSD_TWOPASS %.3f ms" % (nch, timed(f), with_env({"SP_CSD_TWOPASS": "1"}, f)), flush=True)
        del x
if which in ("all", "cog"):
    n = 1 << 27
    z = torch.view_as_complex(torch.randn((n, 2), generator=g, device=dev, dtype=torch.float32))
    for hop in (4096, 2048, 1024):
        M = (n - nfft) // hop + 1
        f = lambda: E.stft_cog(z, win, hop, M, fs=1.0, detrend=True)
        print("cog c64 2^27 hop %4d (SP_WELCH_PIPE=%s): %.3f ms" % (hop, os.environ.get("SP_WELCH_PIPE", "default"), timed(f)), flush=True)
