#!/usr/bin/env python3
"""Golden vectors for the reference's matplotlib.mlab wrappers psd / csd / coh / coh2 (fft_analysis.py:1060-1155).

TEST INFRASTRUCTURE, build container only (needs /root/reference and matplotlib).  Runs the reference functions
unmodified through the shims of make_golden.py and stores inputs + outputs in tests/golden/mlab_wrappers.npz.
Usage:  python tests/golden/make_golden_mlab.py
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
from make_golden import _install_shims, _load, gauss, save


def main():
    _install_shims()
    _load("windows")
    fa = _load("fft_analysis")
    n, fs = 30000, 1.0e3
    t = np.arange(n) / fs
    common = np.sin(2 * np.pi * 61.7 * t) + 0.5 * gauss(11, n)
    x = common + 0.8 * gauss(12, n) + 0.3
    y = 0.7 * np.roll(common, 3) + 0.9 * gauss(13, n) - 0.2 + 1e-4 * np.arange(n)
    d = {"x": x, "y": y, "fs": np.array(fs)}
    # psd: defaults (nfft 2048, detrend 'none', ov 0.67) and a band-limited, mean-detrended, short-segment variant
    p, f = fa.psd(x, fs)
    d["psd_p"], d["psd_f"] = p, f
    p, f = fa.psd(x, fs, nfft=500, fmin=20.0, fmax=300.0, detrend="mean", ov=0.5)
    d["psd2_p"], d["psd2_f"] = p, f
    p, f = fa.csd(x, y, fs)
    d["csd_p"], d["csd_f"] = p, f
    p, f = fa.csd(x, y, fs, nfft=1024, fmin=None, fmax=None, detrend="mean", ov=0.75)
    d["csd2_p"], d["csd2_f"] = p, f
    p, f = fa.psd(x, fs, nfft=1024, detrend="linear", ov=0.5)
    d["psd3_p"], d["psd3_f"] = p, f
    p, f = fa.csd(x, y, fs, nfft=600, fmin=None, fmax=None, detrend="linear", ov=0.25)
    d["csd3_p"], d["csd3_f"] = p, f
    c, f = fa.coh(x, y, fs)
    d["coh_c"], d["coh_f"] = c, f
    c, f = fa.coh(x, y, fs, nfft=512, fmin=10.0, fmax=400.0, detrend="none", ov=0.5)
    d["coh2_c"], d["coh2_f"] = c, f
    try:
        r = fa.coh2(x, y, fs)
        d["cohb_coh"], d["cohb_f"], d["cohb_PS"], d["cohb_pha"] = r["coh"], r["f"], r["PS"], r["pha"]
        d["cohb_ok"] = np.array(1)
    except Exception as e:          # noqa: BLE001  (matplotlib >= 3.? rejects the float noverlap = nfft/2 the reference passes)
        print("coh2 raised in the reference:", type(e).__name__, e)
        d["cohb_ok"] = np.array(0)
    save("mlab_wrappers", **d)


if __name__ == "__main__":
    main()
