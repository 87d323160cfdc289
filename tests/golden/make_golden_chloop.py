#!/usr/bin/env python3
"""Golden vectors for the multi-channel driver shape of the reference's heat-pulse analysis:
`HeatPulse_Funcs._PWELCH_chloop` (HeatPulse_Funcs.py:576-583) = for every channel one `fft_pwelch(tt, ref, sig[:, ii], ...)`
(:537-540) followed by `integratespectra` over the band around each harmonic of the modulation frequency (:498-530, :452-496).

TEST INFRASTRUCTURE, build container only (see make_golden.py for the shims).  HeatPulse_Funcs.py itself cannot be imported
(h5py, IO, FIT are missing, and its `numpy.asscalar` calls are gone from numpy >= 1.23), so this script LOOPS THE
REFERENCE'S OWN `fft_pwelch` + `integratespectra` PER CHANNEL exactly as `_PWELCH_ch` does and applies the class's
bookkeeping as text: the harmonic search `_getharmindex_` (:412-441) on channel 0's Pxx, the band edges (:500-502), the
noise floor beside the band (:509-512) and the closing conversions (:585-602).  integratespectra needs
pybaseutils.utils.reshapech / trapz_var: stand-ins as in make_golden_integrate.py, PARITY UNPINNED at that boundary.

Two shapes: Navr = 8 (the reference's default regime: 7281-point segments of a 2^15-sample record -- the long-segment path
of the GPU build) and Navr = 64 (1008-point segments: one fused kernel).  Inputs are rebuilt from the seed by the test.

Usage:  python tests/golden/make_golden_chloop.py        (writes tests/golden/chloop.npz)
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from make_golden import _install_shims, _load, save, c
from make_golden_integrate import reshapech, trapz_var

from inputs_chloop import SEED, NT, FS, NCH, FMOD, chloop_inputs


def main():
    _install_shims()
    ut = sys.modules["pybaseutils.utils"]
    ut.reshapech = reshapech
    ut.trapz_var = trapz_var
    _load("windows")
    fa = _load("fft_analysis")
    tt, ref, sig = chloop_inputs()
    harms, fwid = [1, 2, 3], 12.0
    out = dict(seed=SEED, nt=NT, fs=FS, nch=NCH, fmod=FMOD, harms=np.asarray(harms), fwid=fwid)
    for Navr in (8, 64):
        tag = "n%d" % Navr
        nh = len(harms)
        Txy = np.zeros((NCH, nh), dtype=np.complex128)
        Vxy = np.zeros((NCH, nh), dtype=np.complex128)
        Amp, varA, Coh = (np.zeros((NCH, nh)) for _ in range(3))
        Tnn = np.zeros((NCH, nh), dtype=np.complex128)                           # :401 zeros_like(Txy)
        Txx, Vxx = np.zeros(nh), np.zeros(nh)
        ifk = ifw = freq = None
        Pxy_all, Pyy_all = [], []
        for ii in range(NCH):                                                     # HeatPulse_Funcs.py:581-583
            freq, _, _, _, _, _, info = fa.fft_pwelch(tt, ref, sig[:, ii], None, Navr=Navr, windowoverlap=0.5,
                                                      windowfunction="hanning", useMLAB=False, plotit=False, verbose=False)
            nf = len(freq)
            if ii == 0:                                                           # :542-546 -> _getharmindex_ :412-441
                dT = nf / (freq[-1] - freq[0])
                ifw = int(1 + np.floor(dT * (0.5 * fwid)))
                P = np.abs(info.Pxx.reshape((nf,), order="C").copy())
                ifk = np.zeros(nh, dtype=np.int64)
                for jj, kk in enumerate(harms):
                    itemp = np.where(freq > kk * FMOD)[0][0]
                    isl = np.arange(itemp - 2 * ifw, itemp + 2 * ifw, dtype=int)
                    ifk[jj] = np.argmax(P[isl]) + isl[0]
                out["Pxx_" + tag] = c(info.Pxx.reshape((nf,), order="C"))
                out["nwins_" + tag], out["Navr_" + tag], out["ENBW_" + tag] = info.nwins, info.Navr, info.ENBW
            Pxy_all.append(info.Pxy.reshape((nf,), order="C").copy())             # :556-557
            Pyy_all.append(info.Pyy.reshape((nf,), order="C").copy())
            for jj in range(nh):                                                  # _integrate_spectra :498-530
                frange = np.asarray([freq[ifk[jj] - ifw], freq[ifk[jj] + ifw]])
                isl = np.arange(ifk[jj] - ifw, ifk[jj] + ifw, 1, dtype=int)
                Pxy_i, Pxx_i, Pyy_i, Cxy_i, ph_i, ii_ = fa.integratespectra(                       # integrate_spectra :484-486
                    info.freq, info.Pxy, info.Pxx, info.Pyy, frange, varPxy=info.varPxy, varPxx=info.varPxx, varPyy=info.varPyy)
                Tnn[ii, jj] = (0.5 * info.ENBW * (info.Pyy[isl[0] - 1] + info.Pyy[isl[-1] + 1]).T).item()
                Txy[ii, jj] = np.asarray(Pxy_i).item()
                Vxy[ii, jj] = np.asarray(ii_.varPxy_i).item()
                Amp[ii, jj] = np.real(np.asarray(Pyy_i)).item()
                varA[ii, jj] = np.real(np.asarray(ii_.varPyy_i)).item()
                Coh[ii, jj] = np.real(np.asarray(Cxy_i)).item()
                if ii == 0:
                    Txx[jj] = np.real(np.asarray(Pxx_i)).item()
                    Vxx[jj] = np.real(np.asarray(ii_.varPxx_i)).item()
        # closing block of _PWELCH_chloop, :585-602 (Navr as the fftinfo reports it)
        NA = out["Navr_" + tag]
        Coh = np.sqrt(Coh)
        varC = ((1.0 - Coh ** 2.0) / np.sqrt(2 * NA)) ** 2.0
        varP = (np.sqrt(1.0 - Coh ** 2) / np.sqrt(2.0 * NA * Coh)) ** 2.0
        Phase = -1 * np.angle(Txy)
        sub = slice(None, None, 1 if len(freq) <= 1024 else 4)                    # spectra stored sub-sampled when long
        out.update({"freq_" + tag: c(freq), "ifk_" + tag: ifk, "ifw_" + tag: ifw, "Txy_" + tag: Txy, "Vxy_" + tag: Vxy,
                    "Amp_" + tag: Amp, "varA_" + tag: varA, "Coh_" + tag: Coh, "varC_" + tag: varC, "varP_" + tag: varP,
                    "Phase_" + tag: Phase, "Tnn_" + tag: Tnn, "Txx_" + tag: Txx, "Vxx_" + tag: Vxx,
                    "Pxy_" + tag: np.stack(Pxy_all, axis=1)[sub], "Pyy_" + tag: np.stack(Pyy_all, axis=1)[sub],
                    "sub_" + tag: sub.step})
    save("chloop", **out)


if __name__ == "__main__":
    main()
