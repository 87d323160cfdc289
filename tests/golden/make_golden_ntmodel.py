#!/usr/bin/env python3
"""Golden vectors for the nT-model branch of the reference's fft_pwelch (fft_analysis.py:169-176, :346-393): sigx is a
one-window model signal, correlated with every window of the longer sigy.  The branch runs in the reference only with
Navr=None and tbounds inside the record (Navr given: UnboundLocalError at :172; full record: the model is reflected too and
win*xtemp fails at :374) -- the error cases are recorded as exception names.

TEST INFRASTRUCTURE, build container only (needs /root/reference).
Usage:  python tests/golden/make_golden_ntmodel.py
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np


def gauss(seed, n):
    return np.random.default_rng(seed).standard_normal(n)


def inputs():
    """deterministic inputs, rebuilt by the tests (imported from here; nothing of the reference is read at import)"""
    fs, n, nw = 1.0e4, 20000, 1024
    t = np.arange(n) / fs
    xm = np.sin(2 * np.pi * 500.0 * t[:nw]) + 0.3 * np.sin(2 * np.pi * 1230.0 * t[:nw]) + 0.05
    y1 = np.sin(2 * np.pi * 500.0 * t + 0.4) + 0.5 * gauss(31, n) + 0.2 + 1e-5 * np.arange(n)
    y2 = 0.7 * np.sin(2 * np.pi * 1230.0 * t - 1.1) + 0.8 * gauss(32, n) - 0.1
    return fs, t, xm, np.stack([y1, y2], axis=1)


def main():
    from make_golden import _install_shims, _load, save
    _install_shims()
    _load("windows")
    fa = _load("fft_analysis")
    fs, t, xm, y = inputs()
    tb = [t[10], t[-10]]
    d = {"tb": np.array(tb)}
    keys = ("S1", "S2", "ENBW", "NENBW", "Navr", "nwins", "noverlap", "Lxx", "Lyy", "Lxy", "Rxy", "corrcoef", "lags")
    for tag, kw in (("one_mean", dict(detrend_style=1)), ("one_linear_hamming", dict(detrend_style=-1, windowfunction="hamming")),
                    ("two_none", dict(detrend_style=0, onesided=False))):
        for ych, ytag in ((y[:, 0], "1ch"), (y, "2ch")):
            r = fa.fft_pwelch(t, xm.copy(), ych.copy(), tb, plotit=False, verbose=False, **kw)
            p = "%s_%s_" % (tag, ytag)
            for nm, v in zip(("freq", "Pxy", "Pxx", "Pyy", "Cxy", "phi_xy"), r[:6]):
                d[p + nm] = np.asarray(v)
            for k in keys:
                d[p + k] = np.asarray(getattr(r[6], k))
    errs = []
    for kw in (dict(tbounds=tb, Navr=37), dict(tbounds=None)):
        try:
            fa.fft_pwelch(t, xm.copy(), y[:, 0].copy(), kw.pop("tbounds"), plotit=False, verbose=False, **kw)
            errs.append("none")
        except Exception as e:                                  # noqa: BLE001
            errs.append(type(e).__name__)
    d["errors"] = np.array(errs)
    save("pwelch_ntmodel", **d)
    print(errs)


if __name__ == "__main__":
    main()
