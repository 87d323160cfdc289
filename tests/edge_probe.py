#!/usr/bin/env python3
"""Edge-shape probe of the round-1 additions (cog_frames, frame_sum, spectral_filter_rows, fft_deriv, cog) against the
oracle / numpy on one MI355X: tiny, odd, Bluestein, multi-wave and long shapes.  Prints one OK/FAIL line per case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfft_amd as P
from oracle import cpu_ref as O


def run():
    fails = []
    rng = np.random.default_rng(0)
    fs = 1e3
    def chk(name, a, b, tol):
        e = float(np.max(np.abs(np.asarray(a) - np.asarray(b))))
        print("%-40s err %.3e tol %.3e %s" % (name, e, tol, "OK" if e <= tol else "FAIL"), flush=True)
        if not e <= tol:
            fails.append(name)
    z = (rng.standard_normal(70000) + 1j * rng.standard_normal(70000)).astype(np.complex64) * np.exp(2j*np.pi*0.07*np.arange(70000)).astype(np.complex64)
    r = rng.standard_normal(70000).astype(np.float32) + np.cos(2*np.pi*0.11*np.arange(70000)).astype(np.float32)
    t = np.arange(70000) / fs
    for win, ov in ((2, 0.5), (4, 0.5), (8, 0.5), (16, 0.5), (32, 0.75), (64, 0.0), (100, 0.5), (8192, 0.5), (8192, 0.75), (4096, 0.0), (3000, 0.3)):
        for x, nm in ((z, "c64"), (r, "f32")):
            try:
                _, a = P.cog_frames(t, x, fs, win=win, ov=ov)
                _, b = O.cog_frames(t, x, fs, win=win, ov=ov)
                chk("cog win=%d ov=%.2f %s n=%d" % (win, ov, nm, len(a)), a, b, 5e-6 * fs * max(1, 16 / win))
            except Exception as e:
                print("cog win=%d %s EXC %r" % (win, nm, e))
                fails.append("cog win=%d %s" % (win, nm))
    # frame_sum with many channels (per-channel trend fallback) and one frame
    y = rng.standard_normal((600, 300)).astype(np.float32) + 0.5
    for det in (False, True, "linear"):
        got = P.engine.frame_sum(y, 64, 16, (300 - 64) // 16 + 1, detrend=det)
        y64 = y.astype(np.float64)
        if det is True: y64 = y64 - y64.mean(axis=1, keepdims=True)
        if det == "linear":
            import scipy.signal; y64 = scipy.signal.detrend(y64, axis=1)
        idx = (np.arange((300 - 64) // 16 + 1) * 16)[:, None] + np.arange(64)[None, :]
        ref = np.stack([y64[c][idx].sum(axis=0) for c in range(600)])
        chk("frame_sum 600ch det=%s" % det, got, ref, 2e-4)
    got = P.engine.frame_sum(y[:2, :64], 64, 64, 1, detrend=False)
    chk("frame_sum single frame", got, y[:2, :64].astype(np.float64), 1e-7)
    # spectral filter tiny / odd sizes
    for n in (2, 3, 5, 16, 17, 8191, 8193, 12289):
        x = rng.standard_normal((2, n)).astype(np.float32)
        H = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.ifft(H * np.fft.fft(x.astype(np.float64), axis=-1), axis=-1)
        try:
            got = P.engine.spectral_filter_rows(x, H)
            chk("spectral_filter n=%d" % n, got, ref, 1e-5 * np.max(np.abs(ref)))
        except Exception as e:
            print("spectral_filter n=%d EXC %r" % (n, e))
            fails.append("spectral_filter n=%d" % n)
    # fft_deriv short
    for n in (8, 33, 100):
        xx = np.linspace(0, 1, n); yy = np.sin(3 * xx)
        d, _ = P.fft_deriv(yy, xx); dr, _ = O.fft_deriv(yy, xx)
        chk("fft_deriv n=%d" % n, d, dr, 1e-4 * np.max(np.abs(dr)) * max(1, n / 50))
    # whole-vector cog for odd and large lengths
    for n in (7, 4097, 65536, 70000):
        chk("cog n=%d" % n, P.cog(z[:n], fs), O.cog(z[:n], fs), 5e-6 * fs)
    return fails


if __name__ == "__main__":
    sys.exit(1 if run() else 0)
