#!/usr/bin/env python3
"""Times the BASELINE.json configs 2-5 at full size on one MI355X (device-resident data, torch events on the
current stream = the library's launch stream) and prints algorithmic GB/s per SURVEY.md section 8d.  Round 3: the headline
figure of a line is the SUSTAINED time per call (settled clocks, calls back to back); the isolated median of the earlier rounds
follows in brackets (SP_CFGBENCH_ISOLATED=1: only that).
  python tools/cfgbench.py [--only cfg2,cfg3,...] [--reps 5]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E
from pyfft_amd.windows import windows


ISOLATED = {}      # name of the last timed() call's isolated median, printed by report()


def timed(fn, reps):
    """ms per call SUSTAINED: after ~30 ms of the same work (the clocks of this part settle only then: isolated calls with a host
    wait in between read up to 30 % longer for the first dozen calls, tools/cog_series.py), `n` calls back to back between two
    events, n chosen to fill ~20 ms.  The median of `reps` isolated calls (the figure of the earlier rounds) is kept in ISOLATED."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    iso = float(np.median(ts))
    ISOLATED["last"] = iso
    if os.environ.get("SP_CFGBENCH_ISOLATED", "0") == "1":
        return iso, out
    n = max(3, min(200, int(20.0 / max(iso, 1e-3))))
    for _ in range(max(3, int(30.0 / max(iso, 1e-3)))):       # settle
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


def report(name, ms, alg_bytes, units, unit_name):
    print("%-34s %9.3f ms  %8.0f GB/s alg (%4.1f%% of 8 TB/s)  %10.1f M%s/s   [isolated median %.3f ms]" %
          (name, ms, alg_bytes / ms / 1e6, 100 * alg_bytes / ms / 1e6 / 8000, units / ms / 1e3, unit_name, ISOLATED.get("last", float("nan"))),
          flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="cfg2,cfg3,cfg4,cfg5,hilbert,xcorr,cog")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cfg5-detrend", type=int, default=1, help="0: cfg5 without the mean detrend (chunking experiments)")
    a = ap.parse_args()
    only = set(a.only.split(","))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(2)

    if "cfg2" in only:      # 65536 x 4096 complex64 forward + inverse
        x = torch.view_as_complex(torch.randn((65536, 4096, 2), generator=g, device=dev, dtype=torch.float32))
        ms, X = timed(lambda: E.fft(x), a.reps)
        report("cfg2 fft 65536x4096 c64 fwd", ms, 16.0 * x.numel(), x.numel(), "pts")
        ms, xr = timed(lambda: E.ifft(X), a.reps)
        report("cfg2 fft 65536x4096 c64 inv", ms, 16.0 * x.numel(), x.numel(), "pts")
        err = float((xr - x).abs().max() / x.abs().max())
        print("     round-trip max|ifft(fft(x))-x|/max|x| = %.2e (tolerance 5e-6)" % err)
        del x, X, xr

    if "cfg3" in only:      # STFT 2^26 float32, 2048-pt Hann, 75 % overlap, one-sided complex64 out
        n, nfft, hop = 1 << 26, 2048, 512
        x = torch.randn(n, generator=g, device=dev, dtype=torch.float32)
        M = (n - nfft) // hop + 1
        win = windows("Hanning", nwins=nfft, verbose=False)
        ms, (Xs, _) = timed(lambda: E.stft_frames(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, amp_scale=1.0), a.reps)
        report("cfg3 stft 2^26 f32 n2048 ov75", ms, 4.0 * n + 8.0 * Xs.numel(), n, "samples")
        ms, P = timed(lambda: E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0), a.reps)
        report("cfg3-shape welch psd (power only)", ms, 4.0 * n, n, "samples")
        del x, Xs

    if "cfg4" in only:      # FIR 513 taps over 2^28 float32
        import scipy.signal as ss
        n = 1 << 28
        x = torch.randn(n, generator=g, device=dev, dtype=torch.float32)
        h = ss.firwin(513, 0.12)
        for nfft in (2048, 4096, 8192):
            ms, y = timed(lambda: E.fir_filter(h, x, nfft=nfft), a.reps)
            report("cfg4 fir 513 taps 2^28 f32 nfft=%d" % nfft, ms, 8.0 * n, n, "samples")
        del x, y

    if "cfg5" in only:      # 64 channels x 2^24 float32, full CSD matrix, nfft 4096, 50 %
        nch, n, nfft, hop = 64, 1 << 24, 4096, 2048
        x = torch.randn((nch, n), generator=g, device=dev, dtype=torch.float32)
        M = (n - nfft) // hop + 1
        win = windows("Hanning", nwins=nfft, verbose=False)
        dt = bool(a.cfg5_detrend)
        ms, G = timed(lambda: E.csd_matrix(x, win, hop, M, detrend=dt, scale=1.0), max(2, a.reps // 2))
        flops = (nfft // 2 + 1) * nch * nch * 8.0 * M
        print("%-34s %9.3f ms  input %.0f GB/s, contraction %.1f TFLOP/s (of 157 fp32)  %8.1f Msamples/s   [isolated median %.3f ms]" %
              ("cfg5 csd matrix 64ch x 2^24", ms, 4.0 * nch * n / ms / 1e6, flops / ms / 1e9, nch * n / ms / 1e3, ISOLATED.get("last", float("nan"))), flush=True)
        # the same call 5 times back to back (no idle gap between the calls: clocks stay up), mean per call
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            E.csd_matrix(x, win, hop, M, detrend=dt, scale=1.0)
        torch.cuda.synchronize()
        print("%-34s %9.3f ms  per call" % ("cfg5 csd matrix, 5 back to back", (time.perf_counter() - t0) / 5 * 1e3), flush=True)
        ms, out = timed(lambda: E.welch_csd(x[0], x[1:], win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0), 2)
        report("cfg5 ref x 63 channels csd", ms, 4.0 * nch * n, nch * n, "samples")
        del x, G

    if "biquad" in only:    # cfg4's "+ notch_filter": exact biquad over 2^28 float32
        from pyfft_amd.notch_filter import iirnotch
        n = 1 << 28
        x = torch.randn(n, device=dev, dtype=torch.float32)
        b, a_ = iirnotch(0.01, 30.0)
        ms, _ = timed(lambda: E.biquad_filter(b, a_, x), a.reps)
        report("cfg4 notch (exact biquad) 2^28 f32", ms, 8.0 * n, n, "samples")
        del x
    if "hilbert" in only:
        x = torch.randn((4096, 4096), generator=g, device=dev, dtype=torch.float32)
        ms, z = timed(lambda: E.hilbert_rows(x, 4096), a.reps)
        report("hilbert 4096 rows x 4096", ms, 12.0 * x.numel(), x.numel(), "samples")
        x1 = torch.randn((1, 1 << 24), generator=g, device=dev, dtype=torch.float32)
        ms, z = timed(lambda: E.hilbert_rows(x1, 1 << 24), 3)
        report("hilbert one row of 2^24", ms, 12.0 * x1.numel(), x1.numel(), "samples")
        del x, x1, z

    if "xcorr" in only:
        n = 1 << 24
        x1 = torch.randn(n, generator=g, device=dev, dtype=torch.float32)
        x2 = torch.roll(x1, 100) + 0.1 * torch.randn(n, generator=g, device=dev, dtype=torch.float32)
        ms, co = timed(lambda: E.xcorr_normalised(x1, x2), 3)
        report("ccf 2^24 samples (2^25-1 lags)", ms, 16.0 * n, n, "samples")
        print("     argmax lag = %d (expected -100)" % (int(torch.argmax(co)) - (n - 1)))

    if "cog" in only:       # N3: centre of gravity per frame, metric shape (2^28 complex64, 4096-point windows, 50 % overlap)
        n, nfft, hop = 1 << 28, 4096, 2048
        x = torch.view_as_complex(torch.randn((n, 2), generator=g, device=dev, dtype=torch.float32))
        M = (n - nfft) // hop + 1
        ms, cg = timed(lambda: E.stft_cog(x, np.ones(nfft), hop, M, 1.0e6), a.reps)
        report("cog per frame 2^28 c64 n4096 ov50", ms, 8.0 * n + 8.0 * M, n, "samples")
        del x


if __name__ == "__main__":
    main()
