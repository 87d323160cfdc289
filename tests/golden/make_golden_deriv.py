#!/usr/bin/env python3
"""Golden vectors for the reference's fft_deriv (fft_analysis.py:1453-1587) on the inputs of its own test_fft_deriv
(:1590-1655: Gaussian, line, aperiodic and periodic sine; the box case needs pybaseutils.rect/delta and is replaced by a
smooth bump) plus a power-of-two and a long (multi-pass transform) case.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  The inputs are analytic, so only the outputs are
stored; tests rebuild the inputs with `cases()` below (imported from here -- this script reads nothing from the
reference at import time).
Usage:  python tests/golden/make_golden_deriv.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    """name -> (yy, xx, kwargs)"""
    N, L = 2000, 13.0
    dx = L / N
    xx = dx * np.arange(N)
    out = {}
    out["gauss"] = (np.exp(-0.5 * (xx / L) ** 2 / 0.25 ** 2), xx, {})
    out["line"] = (np.linspace(-1.2, 11.3, num=N, endpoint=True), xx, {})
    out["sine_aperiodic"] = (np.sin(xx), xx, {})
    xp = (6.0 * np.pi * xx / L)[:-1]
    out["sine_periodic"] = (np.sin(xp), xp, {})
    out["sine_periodic_unmodified"] = (np.sin(xp), xp, {"modified": False})
    out["gauss_hamming"] = (out["gauss"][0], xx, {"window": np.hamming})
    x2 = np.linspace(-4.0, 4.0, 4096)
    out["bump_pow2"] = (np.exp(-x2 ** 2) * np.cos(3 * x2), x2, {})
    out["bump_default_axis"] = (np.exp(-x2 ** 2) * np.cos(3 * x2), None, {})
    x3 = np.linspace(0.0, 40.0, 20000)
    out["long_row"] = (np.sin(x3) * np.exp(-0.02 * x3), x3, {})
    return out


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, HERE)
    from make_golden import _install_shims, _load, save
    _install_shims()
    _load("windows")
    fa = _load("fft_analysis")
    d = {}
    for name, (yy, xx, kw) in cases().items():
        ds, xo = fa.fft_deriv(yy.copy(), None if xx is None else xx.copy(), **kw)
        d[name + "_d"] = ds
        if name in ("gauss", "bump_default_axis"):          # the returned axis is the input axis; two samples suffice
            d[name + "_x"] = xo
    # with a detrend handle (the shim's detrend_mean)
    from pybaseutils.utils import detrend_mean
    yy, xx, _ = cases()["sine_aperiodic"]
    ds, xo = fa.fft_deriv(yy.copy(), xx.copy(), detrend=detrend_mean)
    d["sine_aperiodic_detrend_d"] = ds
    save("fft_deriv", **d)


if __name__ == "__main__":
    main()
