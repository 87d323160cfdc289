#!/usr/bin/env python3
"""Golden vectors for fft_pwelch(..., useMLAB=True) (fft_analysis.py:254-330: matplotlib.mlab.csd per channel with the
reference's window array and per-segment detrend).  TEST INFRASTRUCTURE, build container only.
Usage:  python tests/golden/make_golden_usemlab.py   -> tests/golden/pwelch_usemlab_*.npz"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
from make_golden import _install_shims, _load, gauss, save, c


def main():
    _install_shims()
    _load("windows")
    fa = _load("fft_analysis")
    n, fs = 16384, 1.0e4
    t = np.arange(n) / fs
    common = np.sin(2 * np.pi * 433.0 * t) + 0.4 * gauss(21, n)
    x = common + 0.5 * gauss(22, n) + 0.7
    y = np.stack([0.6 * np.roll(common, 4) + 0.8 * gauss(23, n) - 0.3, 0.3 * common + gauss(24, n) + 2e-4 * np.arange(n)], axis=1)

    def run(tag, **kw):
        freq, Pxy, Pxx, Pyy, Cxy, phi, info = fa.fft_pwelch(t, x, y, plotit=False, verbose=False, useMLAB=True, **kw)
        out = dict(t=t, x=x, y=y, freq=c(freq), Pxy=c(Pxy), Pxx=c(Pxx), Pyy=c(Pyy), Cxy=c(Cxy), phi_xy=c(phi))
        for a in ("nwins", "noverlap", "Navr", "Fs", "ENBW", "S1", "S2"):
            out["info_" + a] = np.asarray(getattr(info, a))
        for a in ("Lxx", "Lxy", "Rxy", "corrcoef", "lags", "Cxy2"):
            out["info_" + a] = c(np.asarray(getattr(info, a)))
        save(tag, **out)

    run("pwelch_usemlab_onesided", tbounds=[t[0], t[-2]], Navr=15, windowoverlap=0.5, windowfunction="Hamming")
    run("pwelch_usemlab_twosided_linear", tbounds=[t[0], t[-2]], Navr=9, windowoverlap=0.5, windowfunction="Hanning",
        onesided=False, detrend_style=-1)


if __name__ == "__main__":
    main()
