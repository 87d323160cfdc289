"""The multi-channel driver shape of the reference's heat-pulse analysis: ONE call that takes a
reference signal and all channels, computes the reference-vs-channels cross-spectral density on the GPU with the reference
signal transformed ONCE (k_pairspec / k_welch_csd_pair behind fft_pwelch -> engine.welch_csd; the reference repeats the
whole fft_pwelch per channel, HeatPulse_Funcs.py:576-583 `_PWELCH_chloop` -> :532-546 `_PWELCH_ch`), and integrates the
spectra over the modulation frequency and its harmonics per channel (HeatPulse_Funcs.py:498-530 `_integrate_spectra` ->
:452-496 `integrate_spectra` -> fft_analysis.py:835-937 `integratespectra`), with the closing conversions of
`_PWELCH_chloop` (:585-602: linear coherence, its variance, the phase variance, the sign of the cross-phase).

Pinned by tests/golden/chloop.npz, which loops the reference's OWN fft_pwelch + integratespectra per channel
(tests/golden/make_golden_chloop.py).  The band integration rests on pybaseutils.utils.trapz_var / reshapech, which are
absent from the reference checkout: stated stand-ins, PARITY UNPINNED at that boundary (as for integratespectra itself).
HeatPulse_Funcs.py cannot be imported here at all (h5py, IO, FIT missing; `numpy.asscalar`, :515-517, is gone from numpy
>= 1.23), so the harmonic search (:412-441) and the per-harmonic bookkeeping are followed as text.
"""
import numpy as np

from . import fft_analysis as _fa
from .fft_analysis import Struct


def harmonic_indices(freq, Pxx, fmod, harms, fwid):
    """HeatPulse_Funcs.py:412-441 `_getharmindex_`: half-width of the integration band in bins, ifw = 1 + floor(dT fwid/2)
    with dT = nf / (freq[-1] - freq[0]); for every harmonic kk the bin of the largest |Pxx| within +-2 ifw of the first
    frequency above kk*fmod.  -> (ifk[nharms], ifw)"""
    freq = np.asarray(freq)
    nf = len(freq)
    dT = nf / (freq[-1] - freq[0])
    ifw = int(1 + np.floor(dT * (0.5 * fwid)))
    P = np.abs(np.asarray(Pxx).reshape((nf,), order="C"))
    ifk = np.zeros(len(harms), dtype=np.int64)
    for jj, kk in enumerate(harms):
        itemp = int(np.where(freq > kk * fmod)[0][0])
        isl = np.arange(itemp - 2 * ifw, itemp + 2 * ifw, dtype=int)
        ifk[jj] = int(np.argmax(P[isl])) + isl[0]
    return ifk, ifw


def pwelch_chloop(tt, refsig, sig, fmod, harms=(1,), fwid=None, tbounds=None, Navr=8, windowoverlap=0.5,
                  windowfunction="hanning", useMLAB=False, verbose=False):
    """Reference signal `refsig[nt]` against every channel of `sig[nt, nch]`: averaged spectra, then band integration at the
    harmonics `harms` of the modulation frequency `fmod` (band half-width from `fwid`, default fmod/2 as a Hz width), per
    channel.  One fft_pwelch call for all channels (the GPU transforms the reference once) instead of the reference's loop.

    Returns a Struct with the reference's attribute names (HeatPulse_Funcs.py:387-409, :444-450): freq[nf], Pxy / Pyy /
    vPxy / vPyy [nf, nch], Pxx[nf], Txy / Amp / Coh / Phase / Tnn / Vxy / varA / varC / varP [nch, nharms], Txx / Vxx
    [nharms], fmods, _ifk, _ifw, Navr and the fftinfo of the call."""
    sig = np.asarray(sig)
    if sig.ndim == 1:
        sig = sig[:, None]
    nch = sig.shape[1]
    harms = list(harms)
    if fwid is None:
        fwid = 0.5 * fmod
    freq, Pxy, Pxx, Pyy, Cxy, phi, info = _fa.fft_pwelch(tt, refsig, sig, tbounds, Navr=Navr, windowoverlap=windowoverlap,
                                                        windowfunction=windowfunction, useMLAB=useMLAB, plotit=False,
                                                        verbose=False)
    nf = len(freq)
    hp = Struct()
    hp.freq, hp.nf, hp.nch, hp.harms, hp.fmod, hp.fwid, hp.fftinfo = freq, nf, nch, harms, fmod, fwid, info
    hp.Navr = info.Navr
    two = lambda a: np.asarray(a).reshape(nf, -1)                                      # noqa: E731
    hp.Pxx = np.asarray(info.Pxx).reshape(nf, -1)[:, 0].copy()
    hp.Pxy, hp.Pyy = two(info.Pxy).copy(), two(info.Pyy).copy()
    hp.vPxy, hp.vPyy = two(info.varPxy).copy(), two(info.varPyy).copy()
    varPxx = np.asarray(info.varPxx).reshape(nf, -1)[:, 0]
    hp._ifk, hp._ifw = harmonic_indices(freq, hp.Pxx, fmod, harms, fwid)
    hp.fmods = freq[hp._ifk]
    nh = len(harms)
    hp.Txx, hp.Vxx = np.zeros(nh), np.zeros(nh)
    hp.Txy, hp.Vxy = np.zeros((nch, nh), dtype=np.complex128), np.zeros((nch, nh), dtype=np.complex128)
    hp.Tnn = np.zeros((nch, nh), dtype=np.complex128)                                 # (:401: zeros_like(Txy))
    for k in ("Amp", "Coh", "Phase", "varA", "varC", "varP"):
        setattr(hp, k, np.zeros((nch, nh)))
    for jj in range(nh):
        lo, hi = hp._ifk[jj] - hp._ifw, hp._ifk[jj] + hp._ifw
        frange = np.asarray([freq[lo], freq[hi]])                                      # HeatPulse_Funcs.py:500-502
        isl = np.arange(lo, hi, 1, dtype=int)
        # all channels in one integratespectra call (it is column-wise: fft_analysis.py:881-915)
        Pxy_i, Pxx_i, Pyy_i, Cxy_i, ph_i, ii = _fa.integratespectra(freq, hp.Pxy, hp.Pxx, hp.Pyy, frange, varPxy=hp.vPxy,
                                                                   varPxx=varPxx, varPyy=hp.vPyy)
        hp.Txy[:, jj] = np.asarray(Pxy_i).ravel()
        hp.Vxy[:, jj] = np.asarray(ii.varPxy_i).ravel()
        hp.Amp[:, jj] = np.real(np.asarray(Pyy_i).ravel())
        hp.varA[:, jj] = np.real(np.asarray(ii.varPyy_i).ravel())
        hp.Coh[:, jj] = np.real(np.asarray(Cxy_i).ravel())
        hp.varC[:, jj] = np.real(np.asarray(ii.varCxy_i).ravel())
        hp.Phase[:, jj] = np.real(np.asarray(ph_i).ravel())
        hp.varP[:, jj] = np.real(np.asarray(ii.varph_i).ravel())
        hp.Txx[jj] = float(np.real(np.asarray(Pxx_i).ravel()[0]))
        hp.Vxx[jj] = float(np.real(np.asarray(ii.varPxx_i).ravel()[0]))
        hp.Tnn[:, jj] = 0.5 * info.ENBW * (hp.Pyy[isl[0] - 1, :] + hp.Pyy[isl[-1] + 1, :])     # :509-512: noise floor beside the band
    # the closing block of _PWELCH_chloop (HeatPulse_Funcs.py:585-602)
    hp.Coh = np.sqrt(hp.Coh)                                                           # linear coherence
    hp.varC = ((1.0 - hp.Coh ** 2.0) / np.sqrt(2 * hp.Navr)) ** 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        hp.varP = (np.sqrt(1.0 - hp.Coh ** 2) / np.sqrt(2.0 * hp.Navr * hp.Coh)) ** 2.0
    hp.Phase = np.angle(hp.Txy)
    if not useMLAB:
        hp.Phase = hp.Phase * -1
    return hp
