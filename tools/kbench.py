#!/usr/bin/env python3
"""Kernel micro-benchmark (GPU box): times the dominant kernels with the library's HIP-event hook.
  python tools/kbench.py [--log2n 28] [--nfft 4096] [--reps 20]
Prints per-kernel ms and algorithmic GB/s.  Random data (never zeros: DVFS reads high on zeros)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E
from pyfft_amd.windows import windows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--nfft", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--ov", type=float, default=0.5)
    ap.add_argument("--check", action="store_true", help="parity of the first 2^check_log2n samples against the CPU oracle first")
    ap.add_argument("--check-log2n", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = 1 << a.log2n
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    if a.real:
        x = torch.randn(n, generator=g, device=dev, dtype=torch.float32) + 0.1
    else:
        x = torch.view_as_complex(torch.randn((n, 2), generator=g, device=dev, dtype=torch.float32)) + (0.1 - 0.05j)
    nfft = a.nfft
    hop = int(round(nfft * (1 - a.ov)))
    M = (n - nfft) // hop + 1
    win = windows("Hanning", nwins=nfft, verbose=False)
    if a.check:
        from oracle import cpu_ref as O
        nc = min(n, 1 << a.check_log2n)
        Mc = (nc - nfft) // hop + 1
        xc = x[:nc].cpu().numpy()
        for detrend in (True, False):
            got = E.welch_psd(x[:nc], win, hop, Mc, detrend=detrend, sided=E.SIDED_TWO, scale=1.0 / float(np.sum(win ** 2))).cpu().numpy()
            ref = O.welch_psd_stream(xc, win, nfft, hop, Mc, 1.0, detrend_style=1 if detrend else 0)
            err = float(np.max(np.abs(got - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))
            print("parity detrend=%d: worst/tolerance %.4f %s" % (detrend, err, "ok" if err <= 1 else "FAIL"))
    E.profile_enable(True)
    for detrend in (True, False):
        ms = []
        wall = []
        for i in range(a.reps + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p = E.welch_psd(x, win, hop, M, detrend=detrend, sided=E.SIDED_TWO, scale=1.0)
            torch.cuda.synchronize()
            wall.append(time.perf_counter() - t0)
            ms.append(E.profile_last_ms())
        ms = np.array(ms[3:]); wall = 1e3 * np.array(wall[3:])
        nbytes = x.element_size() * ((M - 1) * hop + nfft)
        print("welch nfft=%d hop=%d frames=%d detrend=%d: k_welch %.4f ms (min %.4f) -> %.0f GB/s alg, %.1f%% of 8 TB/s | call wall %.4f ms"
              % (nfft, hop, M, detrend, ms.mean(), ms.min(), nbytes / ms.mean() / 1e6, 100 * nbytes / ms.mean() / 1e6 / 8000, wall.mean()))
    print("psd checksum", float(p.sum()))


if __name__ == "__main__":
    main()
