#!/usr/bin/env python3
"""SP_CSDM_SPLIT2=1 (two bf16 pieces per operand) against the default three-piece contraction: cfg5 time and errors."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyfft_amd import engine as E
from oracle import cpu_ref as O

def run(x, win, hop, M, split2):
    if split2:
        os.environ["SP_CSDM_SPLIT2"] = "1"
    try:
        return E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    finally:
        os.environ.pop("SP_CSDM_SPLIT2", None)

nfft, hop = 4096, 2048
win = O.windows("Hanning", nwins=nfft)
for label, nch, M, kind in (("noise, 303 frames", 64, 303, 0), ("noise, 2049 frames", 64, 2049, 0), ("hop-synchronous line, 303 frames", 64, 303, 1)):
    rng = np.random.default_rng(5)
    nsig = (M - 1) * hop + nfft
    if kind == 0:
        x = (rng.standard_normal((nch, nsig)) + 0.7 * rng.standard_normal(nsig)[None, :] + 0.3).astype(np.float32)
    else:
        t = np.arange(nsig)
        x = np.stack([np.sin(2 * np.pi * 200 * t / nfft + 0.1 * c) * (1 + 0.01 * c) for c in range(nch)]).astype(np.float32)
        x += (1e-3 * rng.standard_normal((nch, nsig))).astype(np.float32)
    G3 = run(x, win, hop, M, 0)
    G2 = run(x, win, hop, M, 1)
    ref = O.csd_matrix(x[:3].astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    mx = np.abs(ref).max()
    print("%-36s err vs float64 oracle / max|G|: three pieces %.2e   two pieces %.2e   (two vs three: %.2e)" %
          (label, np.abs(G3[:, :3, :3] - ref).max() / mx, np.abs(G2[:, :3, :3] - ref).max() / mx,
           np.abs(G2 - G3).max() / np.abs(G3).max()), flush=True)

dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
nch, n = 64, 1 << 24
x = torch.randn((nch, n), generator=g, device=dev, dtype=torch.float32)
M = (n - nfft) // hop + 1
for split2 in (0, 1, 0, 1):
    if split2:
        os.environ["SP_CSDM_SPLIT2"] = "1"
    else:
        os.environ.pop("SP_CSDM_SPLIT2", None)
    for _ in range(2):
        E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    torch.cuda.synchronize()
    print("cfg5 64 ch x 2^24  %s pieces  %.3f ms" % ("two" if split2 else "three", (time.perf_counter() - t0) / 5 * 1e3), flush=True)
