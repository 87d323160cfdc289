#!/usr/bin/env python3
"""Golden vectors for the uncertainty / band-integration epilogues of fft_analysis.py: varcoh (:1218-1262), varphi
(:1300-1330), mean_angle (:1334-1376) and integratespectra (:835-937; the harmonic-band integration that
HeatPulse_Funcs.py:498-530 runs per channel).

TEST INFRASTRUCTURE, build container only (see make_golden.py for the shims).  varcoh / varphi / mean_angle are pure numpy
in the reference and run as they are.  integratespectra calls pybaseutils.utils.reshapech and .trapz_var, which are absent
from the reference checkout: this script supplies stand-ins with the semantics their call sites imply (column-vector
reshape; trapezoidal integral along `dim` with the variance propagated through the trapezoid weights, sum_i w_i^2 var_i)
-- PARITY UNPINNED at that boundary, like the detrend_* stand-ins.  Everything else in the captured outputs (band
selection, varcoh, varphi, the cross-power weighted frequency) is the reference's own arithmetic.

Usage:  python tests/golden/make_golden_integrate.py        (writes tests/golden/integrate.npz)
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from make_golden import _install_shims, _load, save, c


def reshapech(x):
    x = np.asarray(x)
    return x.reshape(-1, 1) if x.ndim == 1 else x


def trapz_var(x, y, vx=None, vy=None, dim=0):
    """[integral, variance of the integral, None, None]; trapezoid weights w_i = (x_{i+1} - x_{i-1}) / 2 (one-sided at the ends)"""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y)
    w = np.zeros_like(x)
    if x.size > 1:
        d = np.diff(x)
        w[:-1] += 0.5 * d
        w[1:] += 0.5 * d
    shape = [1] * y.ndim
    shape[dim] = x.size
    ww = w.reshape(shape)
    integ = np.sum(ww * y, axis=dim)
    var = None if vy is None else np.sum(ww ** 2 * np.asarray(vy), axis=dim)
    return [integ, var, None, None]


def main():
    _install_shims()
    ut = sys.modules["pybaseutils.utils"]
    ut.reshapech = reshapech
    ut.trapz_var = trapz_var
    _load("windows")
    fa = _load("fft_analysis")
    rng = np.random.default_rng(8)
    nb, nch = 300, 3
    freq = np.linspace(0.0, 500.0, nb)
    Pxx = 1.0 + rng.random(nb)
    Pyy = 0.5 + rng.random((nb, nch))
    Pxy = (rng.standard_normal((nb, nch)) + 1j * rng.standard_normal((nb, nch))) * 0.3
    Pxy[100:140] += 0.9 * np.exp(0.6j)
    Navr = 16
    vPxx = (Pxx / np.sqrt(Navr)) ** 2
    vPyy = (Pyy / np.sqrt(Navr)) ** 2
    vPxy = (Pxy.real / np.sqrt(Navr)) ** 2 + 1j * (Pxy.imag / np.sqrt(Navr)) ** 2
    out = dict(freq=freq, Pxx=Pxx, Pyy=Pyy, Pxy=Pxy, vPxx=vPxx, vPyy=vPyy, vPxy=vPxy)
    # varcoh both modes, varphi both ranges, mean_angle
    Pxxb = Pxx[:, None] * np.ones((1, nch))
    for ms in (True, False):
        Coh, vCoh = fa.varcoh(Pxy, vPxy, Pxxb, vPxx[:, None] * np.ones((1, nch)), Pyy, vPyy, meansquared=ms)
        out["coh_ms%d" % ms], out["vcoh_ms%d" % ms] = c(Coh), c(vCoh)
    for ar, tag in ((np.pi, "pi"), (0.25 * np.pi, "qpi")):
        ph, vph = fa.varphi(Pxy.real, Pxy.imag, vPxy.real, vPxy.imag, angle_range=ar)
        out["ph_" + tag], out["vph_" + tag] = c(ph), c(vph)
    phi = np.angle(Pxy)
    mph, vmph = fa.mean_angle(phi, vphi=0.01 * np.ones_like(phi), dim=0, angle_range=np.pi, vsyst=0.001 * np.ones_like(phi))
    out["mean_phi"], out["var_mean_phi"] = c(mph), c(vmph)
    # integratespectra over a band that holds the coherent peak
    frange = [freq[95], freq[145]]
    Pxy_i, Pxx_i, Pyy_i, Cxy_i, ph_i, info = fa.integratespectra(freq, Pxy, Pxx, Pyy, frange, vPxy, vPxx, vPyy)
    out.update(frange=np.asarray(frange), Pxy_i=c(Pxy_i), Pxx_i=c(Pxx_i), Pyy_i=c(Pyy_i), Cxy_i=c(Cxy_i), ph_i=c(ph_i),
               ifrange=c(info.ifrange), varPxy_i=c(info.varPxy_i), varPxx_i=c(info.varPxx_i), varPyy_i=c(info.varPyy_i),
               varCxy_i=c(info.varCxy_i), varph_i=c(info.varph_i), fweighted=c(info.fweighted))
    save("integrate", **out)


if __name__ == "__main__":
    main()
