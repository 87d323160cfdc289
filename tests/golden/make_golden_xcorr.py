#!/usr/bin/env python3
"""Golden vectors for the class-path correlation epilogues: fftanal.crosscorr_stft (fft_analysis.py:1880-1920) and
fftanal.crosscorr (:1840-1878) -- per-segment and averaged auto-/cross-correlations by inverse FFT of the spectra.

TEST INFRASTRUCTURE, build container only (see make_golden.py for the shims).  The reference's last line of both methods
(`corrcoef[_seg] = Rxy / (ones((nch, 1)) * sqrt(Ex Ey))`) needs `self.nch`, which the class never sets for 1-D signals
(AttributeError), so everything the methods assign BEFORE that line is captured and the exception type is recorded.

Usage:  python tests/golden/make_golden_xcorr.py        (writes tests/golden/crosscorr_class.npz)
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from make_golden import _install_shims, _load, save, c


def main():
    _install_shims()
    _load("windows")
    fa = _load("fft_analysis")
    rng = np.random.default_rng(4)
    n, fs = 6000, 1.0e3
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 50 * t) + 0.3 * rng.standard_normal(n)
    y = np.sin(2 * np.pi * 50 * t + 0.7) + 0.3 * rng.standard_normal(n)
    out = dict(n=np.int64(n), fs=np.float64(fs), seed=np.int64(4))     # inputs are rebuilt from the seed by the tests
    for onesided in (True, False):
        tag = "one" if onesided else "two"
        ft = fa.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=8, windowoverlap=0.5, windowfunction="hanning",
                        onesided=onesided, plotit=False, verbose=False)
        ft.Xstft()
        ft.Ystft()
        ft.Pstft()
        err = "none"
        try:
            ft.crosscorr_stft()
        except Exception as e:                                  # the corrcoef_seg line
            err = type(e).__name__
        out["err_stft_" + tag] = np.array(err)
        for k in ("Rxx_seg", "Ryy_seg", "Rxy_seg", "Ex_seg", "Ey_seg"):
            out[k + "_" + tag] = c(np.asarray(getattr(ft, k)))
        # averaged spectra the way averagewins forms them (:1976-1988), then crosscorr
        ft.Pxx = np.mean(ft.Pxx_seg, axis=0)
        ft.Pyy = np.mean(ft.Pyy_seg, axis=0)
        ft.Pxy = np.mean(ft.Pxy_seg, axis=0)
        err = "none"
        try:
            ft.crosscorr()
        except Exception as e:
            err = type(e).__name__
        out["err_avg_" + tag] = np.array(err)
        for k in ("Rxx", "Ryy", "Rxy", "Ex", "Ey"):
            out[k + "_" + tag] = c(np.asarray(getattr(ft, k)))
        out["nwins_" + tag] = np.int64(ft.nwins)
        out["Nnyquist_" + tag] = np.int64(ft.Nnyquist)
    save("crosscorr_class", **out)


if __name__ == "__main__":
    main()
