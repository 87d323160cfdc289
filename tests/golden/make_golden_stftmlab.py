#!/usr/bin/env python3
"""Golden vectors for fftanal.stft() with useMLAB=True -- the scipy.signal.stft branch (fft_analysis.py:1805-1824):
zero-extended boundaries (nperseg // 2 on both sides), zero padding to a whole number of hops, window = the object's
window table, scaling 1 / sum(window), and `detrend=self.detrend`, a callable that scipy applies to the [segment, sample]
array with the callable's own default axis (0: ACROSS segments -- a quirk of the reference, reproduced).

TEST INFRASTRUCTURE, build container only (see make_golden.py for the shims; detrend_* are stand-ins, parity unpinned
there).  The branch ends in Pstft() / averagewins(), which fail on the [frequency, segment] layout scipy returns for 1-D
signals (Cxy_Cxy2, :1669); everything assigned before the failure is captured together with the exception type.

Usage:  python tests/golden/make_golden_stftmlab.py        (writes tests/golden/stft_usemlab.npz)
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
    rng = np.random.default_rng(12)
    n = 5000
    t = np.arange(n) / 2.0e3
    x = np.sin(2 * np.pi * 120.0 * t) + 0.2 * rng.standard_normal(n) + 0.7
    y = np.cos(2 * np.pi * 120.0 * t + 0.4) + 0.2 * rng.standard_normal(n) - 0.3 + 0.1 * t
    out = dict(n=np.int64(n), seed=np.int64(12))
    for tag, kw in (("one_mean", dict(onesided=True, detrend=1)), ("two_none", dict(onesided=False, detrend=0)),
                    ("one_linear", dict(onesided=True, detrend=-1))):
        ft = fa.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=12, windowoverlap=0.5, windowfunction="hamming", useMLAB=True,
                        plotit=False, verbose=False, **kw)
        err = "none"
        try:
            ft.stft()
        except Exception as e:
            err = type(e).__name__
        out["err_" + tag] = np.array(err)
        out["nwins_" + tag] = np.int64(ft.nwins)
        out["noverlap_" + tag] = np.int64(ft.noverlap)
        for k in ("freq", "tseg", "Xseg", "Yseg", "Pxx", "Pxy", "varPxx"):
            if hasattr(ft, k):
                out[k + "_" + tag] = c(np.asarray(getattr(ft, k)))
    save("stft_usemlab", **out)


if __name__ == "__main__":
    main()
