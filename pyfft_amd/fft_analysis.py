"""Drop-in for the hot path of the reference's fft_analysis.py: `fft_pwelch(...)` and the `fftanal` class.

Same names, argument meaning and return layout as the reference (fft_analysis.py:36-38,790; :1695-2048); the
numeric core -- detrend, window, overlapped FFTs, |X|^2 / Y.conj(X), segment average, and the inverse FFTs of
the correlation epilogue -- runs on the MI355X through libspectral.so in float32/complex64.  Segment geometry,
window tables and the O(nfft) epilogue algebra stay on the host in float64, as in the reference.

Deliberately not carried over (outside the hot-path scope): plotting (`plotit` is accepted
and ignored), the Monte-Carlo uncertainty helpers (monticoh / montiphi), getNpeaks.
"""
import numpy as np

from . import engine as _E
from .windows import windows


class Struct(object):
    """Attribute bag (stand-in for pybaseutils.Struct used by the reference, fft_analysis.py:22)."""

    def __init__(self, d=None):
        if d is not None:
            if not isinstance(d, dict):
                d = d.dict_from_class()
            for k, v in d.items():
                setattr(self, k, v)

    def dict_from_class(self):
        return dict(self.__dict__)


class fftinfosc(Struct):
    """Result container of fft_pwelch (reference: fft_analysis.py:796)."""
    pass


# ------------------------------------------------------------------------------------------
# segment geometry and normalisation (host, exact integer/float64 arithmetic of the reference)
# ------------------------------------------------------------------------------------------
def _nwins(nsig, Navr, ov):                     # fft_analysis.py:2412-2418
    n = int(np.floor(nsig * 1.0 / (Navr - Navr * ov + ov)))
    return nsig if n >= nsig else n


def _noverlap(nwins, ov):                       # :2421-2422
    return int(np.ceil(ov * nwins))


def _navr(nsig, nwins, noverlap):               # :2425-2429
    return 1 if nwins >= nsig else (nsig - noverlap) // (nwins - noverlap)


def _nnyquist(nfft):                            # :2471-2484
    return (nfft + 1) // 2 if nfft % 2 else nfft // 2


def _norms(win, Nnyquist, Fs):                  # :2487-2510 (NENBW uses Nnyquist: reference quirk Q3)
    S1 = np.sum(win)
    S2 = np.sum(win ** 2.0)
    return S1, S2, Nnyquist * 1.0 * S2 / (S1 ** 2), Fs * S2 / (S1 ** 2)


def _fs(tvec):                                  # :2376-2377
    return (len(tvec) - 1) / (tvec[-1] - tvec[0])


def _sided(onesided):
    return _E.SIDED_ONE if onesided else _E.SIDED_TWO


def _check_detrend(style):
    """reference convention (fft_analysis.py:2539-2549): >0 mean, 0/None none, <0 linear -> device mode 1/0/2"""
    if style is None or style == 0:
        return 0
    return 2 if style < 0 else 1


def Cxy_Cxy2(Pxx, Pyy, Pxy, ibg=None):
    """Complex and mean-squared coherence (fft_analysis.py:1662-1688)."""
    Pxx = np.atleast_2d(np.array(Pxx, copy=True))
    Pyy = np.array(Pyy, copy=True)
    Pxy = np.array(Pxy, copy=True)
    if np.size(Pxx, axis=1) != np.size(Pyy, axis=1):
        Pxx = Pxx.T * np.ones((1, np.size(Pyy, axis=1)), dtype=Pxx.dtype)
    Cxy2 = Pxy * np.conj(Pxy) / (np.abs(Pxx) * np.abs(Pyy))
    Cxy = Pxy / np.sqrt(np.abs(Pxx) * np.abs(Pyy))
    if ibg is None:
        return Cxy, Cxy2
    iCxy = np.imag(Cxy) / (1.0 - np.real(Cxy))
    Cp = np.real(Cxy - np.mean(Cxy[:, ibg], axis=-1))
    return iCxy, Cp / (1.0 - Cp)


def _ifft_cols(P, nfft, hermitian_half):
    """Inverse length-nfft FFT along axis 0 on the GPU.  hermitian_half: P holds bins [0, len) of a one-sided
    spectrum and np.fft.irfft(P, n=nfft, axis=0) semantics apply (missing bins are zero, imaginary parts of the
    DC / Nyquist bins are dropped); returns the real result.  Otherwise a plain complex ifft."""
    P = np.asarray(P)
    vec = P.ndim == 1
    P2 = P[:, None] if vec else P
    if hermitian_half:
        nh = nfft // 2 + 1
        H = np.zeros((nh, P2.shape[1]), dtype=np.complex128)
        m = min(nh, P2.shape[0])
        H[:m] = P2[:m]
        H[0] = H[0].real
        if nfft % 2 == 0:
            H[-1] = H[-1].real
        full = np.zeros((nfft, P2.shape[1]), dtype=np.complex128)
        full[:nh] = H
        full[nh:] = np.conj(H[1:nfft - nh + 1][::-1])
        out = _E.ifft(full.T).T.real.astype(np.float64)
    else:
        out = _E.ifft(np.ascontiguousarray(P2.T)).T.astype(np.complex128)
    return out[:, 0] if vec else out


# ------------------------------------------------------------------------------------------
# uncertainty propagation and band integration (fft_analysis.py:835-937, :1218-1376): O(nfft) host algebra on the
# averaged spectra, like the rest of the epilogue
# ------------------------------------------------------------------------------------------
def varcoh(Pxy, varPxy, Pxx, varPxx, Pyy, varPyy, meansquared=True):
    """Coherence and its propagated variance (fft_analysis.py:1218-1262).  varPxy carries the variances of Re Pxy and
    Im Pxy in its real and imaginary parts."""
    ms, mc = np.imag(Pxy), np.real(Pxy)
    vs, vc = np.imag(varPxy), np.real(varPxy)
    prop = (vc * (2 * mc / (mc ** 2 + ms ** 2)) ** 2 + vs * (2 * ms / (mc ** 2 + ms ** 2)) ** 2
            + varPxx * (1 / Pxx) ** 2 + varPyy * (1 / Pyy) ** 2)
    if meansquared:
        Coh = np.abs(Pxy * np.conj(Pxy)) / (np.abs(Pxx) * np.abs(Pyy))
        return Coh, Coh ** 2 * prop
    Coh = Pxy / np.sqrt(np.abs(Pxx) * np.abs(Pyy))              # complex coherence; the reference then takes its
    varCoh = 0.25 * (Coh ** 2 * prop) / Coh                      # complex square root (:1253-1258)
    return np.sqrt(Coh), varCoh


def varphi(Pxy_real, Pxy_imag, varPxy_real, varPxy_imag, angle_range=np.pi):
    """Cross-phase and its propagated variance (fft_analysis.py:1300-1330)."""
    if angle_range > 0.5 * np.pi:
        ph = np.arctan2(Pxy_imag, Pxy_real)
    else:
        ph = np.arctan(Pxy_imag / Pxy_real)
    tangent = Pxy_imag / Pxy_real
    vartang = (varPxy_imag + varPxy_real * tangent ** 2) / (Pxy_real ** 2)
    return ph, vartang / (1 + tangent ** 2) ** 2


def mean_angle(phi, vphi=None, dim=0, angle_range=0.5 * np.pi, vsyst=None):
    """Average of phase angles through their cartesian components (fft_analysis.py:1334-1376)."""
    if vphi is None:
        vphi = np.zeros_like(phi)
    if vsyst is None:
        vsyst = np.zeros_like(phi)
    nphi = np.size(phi, dim)
    cp = np.exp(1.0j * phi)
    cvar = vphi * np.abs(cp) ** 2
    cvsy = vsyst * np.abs(cp) ** 2
    ca, sa = np.real(cp), np.imag(cp)
    mca, msa = np.nanmean(ca, axis=dim), np.nanmean(sa, axis=dim)
    vca = np.nanvar(ca, axis=dim) + np.nansum(cvar, axis=dim) / nphi ** 2
    vsa = np.nanvar(sa, axis=dim) + np.nansum(cvar, axis=dim) / nphi ** 2
    vca += (np.nansum(np.sqrt(cvsy), axis=dim) / nphi) ** 2.0
    vsa += (np.nansum(np.sqrt(cvsy), axis=dim) / nphi) ** 2.0
    return varphi(Pxy_real=mca, Pxy_imag=msa, varPxy_real=vca, varPxy_imag=vsa, angle_range=angle_range)


def reshapech(x):
    """1-D spectra as one column (stand-in for pybaseutils.utils.reshapech, absent from the reference checkout)."""
    x = np.asarray(x)
    return x.reshape(-1, 1) if x.ndim == 1 else x


def trapz_var(x, y, vx=None, vy=None, dim=0):
    """Trapezoidal integral of y over x along `dim` and the variance it inherits from vy: sum_i w_i^2 vy_i with the
    trapezoid weights w.  Stand-in for pybaseutils.utils.trapz_var, which the reference imports but does not contain:
    PARITY UNPINNED at this boundary (the semantics its call sites imply, fft_analysis.py:891-902).
    Returns [integral, variance, None, None] like the four-value call sites expect."""
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


def integratespectra(freq, Pxy, Pxx, Pyy, frange, varPxy=None, varPxx=None, varPyy=None):
    """Integrate auto- and cross-power over the band frange = [f_lo, f_hi] and derive the band's coherence and cross-phase
    with propagated variances (fft_analysis.py:835-937; HeatPulse_Funcs.py:498-530 runs it per harmonic and channel on
    the output of fft_pwelch).  -> (Pxy_i, Pxx_i, Pyy_i, Cxy_i, ph_i, info).
    The reference's defaults for the variances call the non-existent numpy.size_like (:876-878); zeros are used."""
    freq = np.asarray(freq)
    Pxy, Pxx, Pyy = reshapech(Pxy), reshapech(Pxx), reshapech(Pyy)
    varPxy = np.zeros_like(Pxy) if varPxy is None else reshapech(varPxy)
    varPxx = np.zeros_like(Pxx) if varPxx is None else reshapech(varPxx)
    varPyy = np.zeros_like(Pyy) if varPyy is None else reshapech(varPyy)
    inds = np.where((freq >= frange[0]) * (freq <= frange[1]))[0]
    f = freq[inds]
    Pxy_real, varPxy_real, _, _ = trapz_var(f, np.real(Pxy[inds, :]), None, np.real(varPxy[inds, :]), dim=0)
    Pxy_imag, varPxy_imag, _, _ = trapz_var(f, np.imag(Pxy[inds, :]), None, np.imag(varPxy[inds, :]), dim=0)
    Pxy_i = Pxy_real + 1j * Pxy_imag
    varPxy_i = varPxy_real + 1j * varPxy_imag
    Pxx_i, varPxx_i, _, _ = trapz_var(f, Pxx[inds, :], None, varPxx[inds, :], dim=0)
    Pyy_i, varPyy_i, _, _ = trapz_var(f, Pyy[inds, :], None, varPyy[inds, :], dim=0)
    meansquared = 0
    Cxy_i, varCxy_i = varcoh(Pxy_i, varPxy_i, Pxx_i, varPxx_i, Pyy_i, varPyy_i, meansquared)
    angle_range = np.pi
    ph_i, varph_i = varphi(Pxy_real, Pxy_imag, varPxy_real, varPxy_imag, angle_range)
    info = Struct()
    info.frange = np.asarray([frange[0], frange[1]])
    info.ifrange = inds
    info.Pxy_i, info.varPxy_i = Pxy_i, varPxy_i
    info.Pxx_i, info.varPxx_i = Pxx_i, varPxx_i
    info.Pyy_i, info.varPyy_i = Pyy_i, varPyy_i
    info.angle_range, info.ph_i, info.varph_i = angle_range, ph_i, varph_i
    info.meansquared, info.Cxy_i, info.varCxy_i = meansquared, Cxy_i, varCxy_i
    fw = np.dot(f.reshape(len(inds), 1), np.ones((1, np.size(Pxy, axis=1)), dtype=float))     # :931-934 (unit spacing)
    _trapz = getattr(np, "trapezoid", None) or np.trapz
    info.fweighted = _trapz(fw * np.abs(Pxy[inds, :])) / _trapz(np.abs(Pxy[inds, :]))
    return Pxy_i, Pxx_i, Pyy_i, Cxy_i, ph_i, info


# ------------------------------------------------------------------------------------------
# fft_pwelch
# ------------------------------------------------------------------------------------------
def fft_pwelch(tvec, sigx, sigy, tbounds=None, Navr=None, windowoverlap=None, windowfunction=None, useMLAB=None,
               plotit=None, verbose=None, detrend_style=None, onesided=None, **kwargs):
    """(freq, Pxy, Pxx, Pyy, Cxy, phi_xy, fftinfo) -- see the reference docstring (fft_analysis.py:39-100).

    Extra keyword: segments=True also returns the per-segment arrays (fftinfo.Xfft_seg, Yfft_seg, Pxx_seg,
    Pyy_seg, Pxy_seg, phixy_seg) the reference always materialises (:475-481); default False."""
    calcNavr = Navr is None
    if windowfunction is None:
        windowfunction = "Hanning"
    if windowoverlap is None:
        windowoverlap = windows(windowfunction, verbose=False)
    if verbose is None:
        verbose = False
    if detrend_style is None:
        detrend_style = 1
    if tbounds is None:
        tbounds = [tvec[0], tvec[-1]]
    tvec = np.asarray(tvec)
    sigx = np.asarray(sigx)
    sigy = np.asarray(sigy)
    if onesided is None:
        onesided = not (np.iscomplexobj(sigx) or np.iscomplexobj(sigy))
    want_segments = bool(kwargs.pop("segments", False))
    dflag = _check_detrend(detrend_style)

    Fs = (len(tvec) - 1) / (tvec[-1] - tvec[0])                       # :136
    i0 = int(np.floor(Fs * (tbounds[0] - tvec[0])))                  # :143-144
    i1 = int(np.floor(1 + Fs * (tbounds[1] - tvec[0])))
    nsig = np.size(tvec[i0:i1])

    sigy = np.atleast_2d(sigy)                                        # :163-167
    if np.shape(sigy)[1] == len(tvec):
        sigy = sigy.T
    nch = np.size(sigy, axis=1)
    nTmodel = np.size(sigx, axis=0) != np.size(sigy, axis=0)          # :169-176 sigx is ONE window long: a model signal
    if nTmodel:                                                       #          correlated with every window of sigy
        if not calcNavr:
            # the reference only defines calcNavr when Navr is None (:120-131) and fails here (:172) otherwise
            raise UnboundLocalError("local variable 'calcNavr' referenced before assignment (reference behaviour: the "
                                    "nT-model branch of fft_pwelch takes Navr=None only, fft_analysis.py:172)")
        nwins = np.size(sigx, axis=0)
        if i0 == 0 and i1 == len(tvec):
            # the reference reflects the one-window model as well (:202) and then fails at win*xtemp (:374)
            raise ValueError("operands could not be broadcast together: the nT-model branch needs tbounds inside the "
                             "record (no end-point reflection), as in the reference (fft_analysis.py:197-205, :374)")
        if useMLAB:
            raise NotImplementedError("nT-model with useMLAB=True (periodic wrapping of the model, fft_analysis.py:268-281)")
    elif "minFreq" in kwargs or "tper" in kwargs:                     # :180-190
        if "minFreq" in kwargs:
            kwargs["tper"] = 2.0 / kwargs["minFreq"]
        nwins = int(Fs * kwargs["tper"])
    else:
        if Navr is None:
            Navr = 8
        calcNavr = False
        nwins = _nwins(nsig, Navr, windowoverlap)
    noverlap = _noverlap(nwins, windowoverlap)

    reflecting = False
    if i0 == 0 and i1 == len(tvec):                                   # :197-205 end-point reflection (Q2)
        reflecting = True
        sigx = np.concatenate((sigx[nwins - 1:0:-1, ...], sigx, sigx[-1:-nwins:-1, ...]), axis=0)
        sigy = np.concatenate((sigy[nwins - 1:0:-1, ...], sigy, sigy[-1:-nwins:-1, ...]), axis=0)
        nsig = sigx.shape[0]
    if calcNavr:
        Navr = _navr(nsig, nwins, noverlap)
    if nwins >= nsig:
        if nTmodel:
            raise ValueError("nT-model: the model signal must be shorter than the analysed record (fft_analysis.py:215-217, :374)")
        Navr = 1
        nwins = nsig
    nfft = nwins
    Nnyquist = _nnyquist(nfft)

    win, winparams = windows(windowfunction, nwins=nwins, verbose=verbose, msgout=True)
    info = fftinfosc()
    info.win = win
    info.winparams = winparams
    info.windowoverlap = windowoverlap
    info.ibnds = [i0, i1]
    info.S1, info.S2, info.NENBW, info.ENBW = _norms(win, Nnyquist, Fs)

    # ---- device: detrend + window + FFT + products + segment mean  (fft_analysis.py:339-446)
    x_in = sigx if nTmodel else sigx[i0:i1]                           # :346-354
    y_in = np.ascontiguousarray(sigy[i0:i1, :].T)                    # channel-major for coalesced frame loads
    hop = nwins - noverlap
    scale = 1.0 / (info.S1 ** 2) / info.ENBW                          # :432-440
    sided = _sided(onesided)
    freq = np.fft.fftfreq(nfft, 1.0 / Fs)
    if useMLAB:
        # the matplotlib.mlab.csd branch (:254-330) on the device: the same window array, detrend PER SEGMENT,
        # (len - noverlap) // step segments, conj(X) Y / Fs / sum(w^2), mlab's one-sided doubling (every bin but DC
        # and the even-length Nyquist), then the first Nnyquist bins (:317-326)
        nseg = (len(x_in) - noverlap) // hop
        seg_d = {0: False, 1: "segmean", 2: "seglinear"}[dflag]
        sc = 1.0 / (Fs * np.sum(np.asarray(win, dtype=np.float64) ** 2))
        pxx, pyy, pxy = _E.welch_csd(x_in, y_in, win, hop, nseg, detrend=seg_d, sided=_E.SIDED_RAW, scale=sc)
        if onesided:
            nb = nfft // 2 + 1
            dbl = np.full(nb, 2.0)
            dbl[0] = 1.0
            if nfft % 2 == 0:
                dbl[-1] = 1.0
            cut = lambda P: (P[..., :nb] * dbl)[..., :Nnyquist]           # noqa: E731
            freq = freq[:Nnyquist]
        else:
            cut = lambda P: np.fft.fftshift(P, axes=-1)                   # noqa: E731
            freq = np.fft.fftshift(freq)
        Pxx = cut(np.asarray(pxx)).astype(np.complex128)
        Pyy = cut(np.asarray(pyy)).T.astype(np.float64)                # the reference allocates Pyy as float64 (:294)
        Pxy = np.ascontiguousarray(cut(np.asarray(pxy)).T)
        want_segments = False                                             # mlab returns no per-segment arrays
    elif nTmodel:
        # every segment pairs the SAME windowed model spectrum X with the segment's Y_g (:366-393):
        #   Pxx = |X|^2, Pyy = mean_g |Y_g|^2 (fused Welch PSD per channel), Pxy = mean_g(Y_g) conj(X), and
        #   sum_g Y_g = FFT(win * sum_g y_g) by linearity -- the frames are summed in the time domain on the device
        #   (sp_frame_sum), one length-nfft transform per channel follows.
        Xm = _E.stft_frames(x_in, win, nfft, 1, detrend=dflag, sided=_E.SIDED_RAW, amp_scale=1.0)[0][0].astype(np.complex128)
        csum = _E.frame_sum(y_in, nfft, hop, Navr, detrend=dflag)
        Ybar = np.atleast_2d(_E.fft((np.asarray(win) * csum).astype(np.complex64))).astype(np.complex128) / Navr

        def cut_raw(P):
            if onesided:
                P = P[..., :Nnyquist].copy()
                P[..., 1:-1] *= 2
                if nfft % 2:
                    P[..., -1] *= 2
                return P
            return np.fft.fftshift(P, axes=-1)
        pxx = cut_raw((Xm * np.conj(Xm)).real * scale)
        pxy = cut_raw(Ybar * np.conj(Xm)[None, :] * scale)
        pyy = np.stack([_E.welch_psd(y_in[c], win, hop, Navr, detrend=dflag, sided=sided, scale=scale) for c in range(nch)])
        freq = freq[:Nnyquist] if onesided else np.fft.fftshift(freq)
        Pxx = pxx.astype(np.complex128)
        Pyy = pyy.T.astype(np.complex128)
        Pxy = np.ascontiguousarray(pxy.T)
    else:
        pxx, pyy, pxy = _E.welch_csd(x_in, y_in, win, hop, Navr, detrend=dflag, sided=sided, scale=scale)
        freq = freq[:Nnyquist] if onesided else np.fft.fftshift(freq)
        Pxx = pxx.astype(np.complex128)                               # reference dtype: complex128 with zero imag
        Pyy = pyy.T.astype(np.complex128)                             # [nfreq, nch]
        Pxy = np.ascontiguousarray(pxy.T)

    if want_segments:
        amp = 1.0
        if nTmodel:
            Xs = np.repeat(_E.stft_frames(x_in, win, nfft, 1, detrend=dflag, sided=_E.SIDED_RAW, amp_scale=amp)[0], Navr, axis=0)
        else:
            Xs, _ = _E.stft_frames(x_in, win, hop, Navr, detrend=dflag, sided=_E.SIDED_RAW, amp_scale=amp)
        Ys = np.stack([_E.stft_frames(y_in[c], win, hop, Navr, detrend=dflag, sided=_E.SIDED_RAW,
                                      amp_scale=amp)[0] for c in range(nch)])
        info.Xfft_seg = Xs.astype(np.complex128)
        info.Yfft_seg = Ys.astype(np.complex128)

        def cut(P):
            if onesided:
                P = P[..., :Nnyquist].copy()
                P[..., 1:-1] *= 2
                if nfft % 2:
                    P[..., -1] *= 2
                return P
            return np.fft.fftshift(P, axes=-1)
        info.Pxx_seg = cut(info.Xfft_seg * np.conj(info.Xfft_seg)) * scale
        info.Pyy_seg = cut(info.Yfft_seg * np.conj(info.Yfft_seg)) * scale
        info.Pxy_seg = cut(info.Yfft_seg * np.conj(info.Xfft_seg)[None]) * scale
        info.phixy_seg = np.angle(info.Pxy_seg)
        info.varphi_seg = np.zeros_like(info.phixy_seg)

    # ---- epilogue (fft_analysis.py:489-648) on the device (sp_csd_epilogue: coherence, phase, amplitude spectra, the
    # correlations by length-nfft inverse FFTs); the variance formulas are O(nfft) host lines on its results
    ep = _E.csd_epilogue(np.real(Pxx), np.ascontiguousarray(np.real(Pyy).T), np.ascontiguousarray(Pxy.T), nfft, onesided,
                         info.ENBW)
    Cxy = np.ascontiguousarray(ep["Cxy"].T)
    Cxy2 = np.ascontiguousarray(ep["Cxy2"].T).astype(np.complex128)
    info.varCxy = ((1.0 - Cxy * np.conjugate(Cxy)) / np.sqrt(2 * Navr)) ** 2.0
    info.varCxy2 = 4.0 * Cxy2 * info.varCxy
    info.varPxx = (Pxx / np.sqrt(Navr)) ** 2.0
    info.varPyy = (Pyy / np.sqrt(Navr)) ** 2.0
    info.varPxy = (Pxy / np.sqrt(Navr)) ** 2.0
    info.varPhxy = (np.sqrt(1.0 - np.abs(Cxy2))) / np.sqrt(2 * Navr * np.sqrt(np.abs(Cxy2))) ** 2.0
    phi_xy = np.ascontiguousarray(ep["phi"].T)
    info.Lxx = ep["Lxx"].copy()
    info.Lyy = np.ascontiguousarray(ep["Lyy"].T)
    info.Lxy = np.ascontiguousarray(ep["Lxy"].T)
    info.Rxx = ep["Rxx"].copy()
    info.Ryy = np.ascontiguousarray(ep["Ryy"].T)
    info.Rxy = np.ascontiguousarray(ep["Rxy"].T)
    info.iCxy = np.ascontiguousarray(ep["iCxy"].T)
    info.corrcoef = np.ascontiguousarray(ep["corrcoef"].T)
    info.Ex = np.asarray(ep["Ex"]).copy()
    info.Ey = np.asarray(ep["Ey"]).copy()
    info.lags = (np.asarray(range(1, nfft + 1), dtype=int) - Nnyquist) / Fs
    info.varLxx = (info.Lxx ** 2) * (info.varPxx / np.abs(Pxx) ** 2)
    info.varLyy = (info.Lyy ** 2) * (info.varPyy / np.abs(Pyy) ** 2)
    info.varLxy = (info.Lxy ** 2) * (info.varPxy / np.abs(Pxy) ** 2)

    if nch == 1:                                                      # :605-631
        Pyy, Pxy, Cxy, Cxy2, phi_xy = Pyy.flatten(), Pxy.flatten(), Cxy.flatten(), Cxy2.flatten(), phi_xy.flatten()
        for k in ("lags", "Rxx", "Ryy", "Rxy", "corrcoef", "iCxy", "Lxx", "Lyy", "Lxy", "varLxx", "varLyy", "varLxy",
                  "varCxy", "varCxy2", "varPxx", "varPyy", "varPxy", "varPhxy"):
            setattr(info, k, getattr(info, k).flatten())

    info.nch, info.Fs, info.Navr, info.nwins, info.noverlap = nch, Fs, Navr, nwins, noverlap
    info.overlap, info.window, info.minFreq = windowoverlap, windowfunction, 2.0 * Fs / nwins
    info.reflecting = reflecting
    info.freq, info.Pxx, info.Pyy, info.Pxy = freq.copy(), Pxx.copy(), Pyy.copy(), Pxy.copy()
    info.Cxy, info.Cxy2, info.phi_xy = Cxy.copy(), Cxy2.copy(), phi_xy.copy()
    return freq, Pxy, Pxx, Pyy, Cxy, phi_xy, info


# ------------------------------------------------------------------------------------------
# detrend handles (reference: pybaseutils.utils.detrend_*, imported at fft_analysis.py:23 and re-exported by the
# package; that source is absent, the semantics -- along axis 0, like matplotlib.mlab.detrend_* -- are SURVEY 8c's
# stated assumption) and the derivative by FFT (fft_analysis.py:1399-1587)
# ------------------------------------------------------------------------------------------
def detrend_none(x, axis=0):
    return x


def detrend_mean(x, axis=0):
    x = np.asarray(x)
    return x - x.mean(axis=axis, keepdims=True)


def detrend_linear(x, axis=0):
    x = np.asarray(x, dtype=np.float64 if not np.iscomplexobj(x) else np.complex128)
    n = x.shape[axis]
    k = np.arange(n, dtype=np.float64) - 0.5 * (n - 1)
    shape = [1] * x.ndim
    shape[axis] = n
    k = k.reshape(shape)
    slope = (x * k).sum(axis=axis, keepdims=True) / (k * k).sum()
    return x - x.mean(axis=axis, keepdims=True) - slope * k


def unwrap_tol(data, scal=np.pi, atol=None, rtol=None, itol=None):
    """Phase unwrapping with a tolerance on the jump size (fft_analysis.py:1399-1408); modifies and returns `data`."""
    if atol is None and rtol is None:
        atol = 0.2
    if atol is None and rtol is not None:
        atol = rtol * scal
    if itol is None:
        itol = 1
    tt = np.arange(len(data))
    ti = tt[::itol]
    jumps = np.diff(data[::itol]) / scal
    jumps = np.sign(jumps) * np.floor(np.abs(jumps) + atol)
    data[1:] = data[1:] - np.interp(tt[1:], ti[1:], scal * np.cumsum(jumps))
    return data


def rescale(xx, yy, scaley=True, scalex=True):
    """Map y to [0, 1] and x to unit span (fft_analysis.py:1419-1438). -> xx, yy, (slope, offset, xslope, xoffset)"""
    slope, offset, xslope, xoffset = 1.0, 0.0, 1.0, 0.0
    if scaley:
        slope = np.nanmax(yy) - np.nanmin(yy)
        offset = np.nanmin(yy)
        if slope == 0:
            slope = 1.0
        yy = (np.array(yy, copy=True) - offset) / slope
    if scalex:
        xslope = np.nanmax(xx) - np.nanmin(xx)
        xoffset = -1e-4
        if xslope == 0:
            xslope = 1.0
        xx = (np.array(xx, copy=True) - xoffset) / xslope
    return xx, yy, (slope, offset, xslope, xoffset)


def unscale(xx, yy, scl, dydx=None):
    """Inverse of rescale (fft_analysis.py:1440-1451)."""
    slope, offset, xslope, xoffset = scl
    xx = xx * xslope + xoffset
    yy = slope * yy + offset
    if dydx is not None:
        return xx, yy, dydx * slope / xslope
    return xx, yy


def fft_deriv(sig, xx=None, lowpass=True, Fs_new=None, modified=True, detrend=detrend_none, window=None):
    """dsig/dx by FFT: real(ifft(wavenumber * fft(sig))) with the modified wavenumber j sin(k dx)/dx (or j k), end points
    replaced by one-sided differences (fft_analysis.py:1453-1587).  -> (dsdx, xx)

    The transform pair runs on the device in one fused kernel (sp_spectral_filter: forward FFT, times the wavenumber table,
    inverse FFT); scaling, detrend handle, window table and the two end points are O(n) host work in float64 as in the
    reference.  The pre-filter/downsample branch (Fs_new below the sampling rate) needs filters.downsample_efficient,
    which depends on the absent pybaseutils.interp, and raises."""
    sig = np.asarray(sig, dtype=np.float64)
    if xx is None:
        xx = 1.0 * np.arange(len(sig))
    xx = np.asarray(xx, dtype=np.float64)
    if lowpass:
        dxo = xx[1] - xx[0]
        if lowpass is True:
            lowpass = 0.5 * 1.0 / dxo
        Fs = 1.0 / dxo
        if Fs_new is None:
            Fs_new = min(5.0 * lowpass, Fs)
        if Fs_new < Fs:
            raise NotImplementedError("fft_deriv: the downsampling pre-filter (filters.downsample_efficient) needs "
                                      "pybaseutils.interp, absent from the reference")
    xx, sig, scl = rescale(xx, sig, scaley=True, scalex=True)
    sig = detrend(sig)
    N = len(xx)
    dx = xx[1] - xx[0]
    L = N * dx
    k = 2.0 * np.pi * np.fft.fftfreq(N, d=dx / L)
    wavenumber = (1.0j * np.sin(k * dx) / dx if modified else 1.0j * k) / L
    win = np.ones_like(sig) if window is None else window(N)
    sig = win * sig
    ds0 = (sig[1] - sig[0]) / (xx[1] - xx[0])
    ds1 = (sig[-1] - sig[-2]) / (xx[-1] - xx[-2])
    d = _E.spectral_filter_rows(sig[None, :], wavenumber)[0].real.astype(np.float64)
    d /= win
    d[0] = ds0
    d[-1] = ds1
    xx, _, d = unscale(xx, d.copy(), scl=scl, dydx=d)
    return d, xx


# ------------------------------------------------------------------------------------------------------------------
# psd / csd / coh / coh2 (fft_analysis.py:1060-1155): the reference's thin wrappers over matplotlib.mlab.psd / csd.
# mlab's estimator (symmetric Hann = mlab.window_hanning, step NFFT - noverlap, detrend PER SEGMENT, conj(X) Y / Fs /
# sum(w^2), one-sided doubling except DC and Nyquist, mean over segments) runs on the device through welch_psd /
# welch_csd; only the band selection stays on the host.  Real input (the reference's use).
# ------------------------------------------------------------------------------------------------------------------
def _mlab_detrend_code(detrend):
    if detrend in (None, "none", False):
        return False
    if detrend == "mean":
        return "segmean"
    if detrend == "linear":
        return "seglinear"
    raise ValueError("detrend must be 'none', 'mean' or 'linear' (got %r)" % (detrend,))


def _mlab_onesided(nfft, fs):
    nb = nfft // 2 + 1
    f = np.fft.fftfreq(nfft, 1.0 / fs)[:nb].copy()
    dbl = np.full(nb, 2.0)
    dbl[0] = 1.0
    if nfft % 2 == 0:
        f[-1] *= -1.0
        dbl[-1] = 1.0
    return f, dbl, nb


def _mlab_spectra(x, y, fs, nfft, noverlap, detrend):
    """(Pxx, Pyy, Pxy, F) as matplotlib.mlab.psd / csd return them (y None: Pxx only)."""
    if np.iscomplexobj(x) or (y is not None and np.iscomplexobj(y)):
        # matplotlib.mlab switches to two-sided spectra for complex input; these wrappers band-select one-sided output
        # and the reference only feeds them real signals -- refuse instead of silently dropping the imaginary part
        raise TypeError("psd/csd/coh/coh2 take real signals (one-sided matplotlib.mlab spectra, fft_analysis.py:1060-1155)")
    x = np.ascontiguousarray(x, dtype=np.float32)
    step = nfft - int(noverlap)
    nseg = (x.size - int(noverlap)) // step
    if nseg < 1:
        raise ValueError("signal shorter than one segment")
    w = np.hanning(nfft)
    f, dbl, nb = _mlab_onesided(nfft, fs)
    scale = 1.0 / (fs * np.sum(w ** 2))
    d = _mlab_detrend_code(detrend)
    if y is None:
        pxx = _E.welch_psd(x, w, step, nseg, detrend=d, sided=_E.SIDED_RAW, scale=scale)
        return np.asarray(pxx)[:nb] * dbl, None, None, f
    y = np.ascontiguousarray(y, dtype=np.float32)
    pxx, pyy, pxy = _E.welch_csd(x, y[None, :], w, step, nseg, detrend=d, sided=_E.SIDED_RAW, scale=scale)
    return (np.asarray(pxx)[:nb] * dbl, np.asarray(pyy)[0, :nb] * dbl, np.asarray(pxy)[0, :nb] * dbl, f)


def _band(P, F, fmin, fmax, peak_threshold):
    keep = np.ones(P.shape, dtype=bool)
    if fmin is not None:
        keep &= F >= fmin
    if fmax is not None:
        keep &= F <= fmax
    if peak_threshold is not None:
        keep &= P > peak_threshold
    return P[keep], F[keep]


def psd(x, fs, nfft=2048, fmin=None, fmax=None, detrend='none', peak_threshold=None, ov=0.67):
    """Power spectral density within a frequency range (fft_analysis.py:1113-1131): (pso, fo)."""
    P, _, _, F = _mlab_spectra(x, None, fs, nfft, int(np.floor(ov * nfft)), detrend)
    return _band(P, F, fmin, fmax, peak_threshold)


def csd(x, y, fs, nfft=2048, fmin=0, fmax=500e3, detrend='none', peak_threshold=None, ov=0.67):
    """Cross power spectral density conj(X) Y within a frequency range (fft_analysis.py:1134-1155): (pso, fo)."""
    _, _, P, F = _mlab_spectra(x, y, fs, nfft, int(np.floor(ov * nfft)), detrend)
    return _band(P, F, fmin, fmax, peak_threshold)


def coh(x, y, fs, nfft=2048, fmin=0.0, fmax=500e3, detrend='mean', ov=0.67):
    """Coherence (square root of the magnitude-squared coherence) below a maximum frequency (fft_analysis.py:1060-1088)."""
    Pxx, Pyy, Pxy, F = _mlab_spectra(x, y, fs, nfft, int(ov * nfft), detrend)
    c2 = np.abs(Pxy) ** 2 / (Pxx * Pyy)
    keep = (F <= fmax) & (F >= fmin)
    return np.sqrt(c2[keep]), F[keep]


def coh2(x, y, fs, nfft=4096, fmin=0, fmax=500e3, detrend='none', peak_treshold=None):
    """Magnitude-squared coherence, cross-phase and auto-power w.r.t. x (fft_analysis.py:1090-1110).  The reference passes
    noverlap = nfft/2 as a float, which current matplotlib rejects; nfft // 2 is used (parity unpinned)."""
    Pxx, Pyy, Pxy, F = _mlab_spectra(x, y, fs, nfft, nfft // 2, 'none')
    keep = np.abs(F) <= fmax
    return {'coh': (np.abs(Pxy) ** 2 / (Pxx * Pyy))[keep], 'f': F[keep], 'PS': np.abs(Pxx)[keep],
            'pha': np.arctan2(Pxy.imag, Pxy.real)[keep]}


# ------------------------------------------------------------------------------------------
# fftanal
# ------------------------------------------------------------------------------------------
class fftanal(Struct):
    """Welch / STFT analysis object (reference: fft_analysis.py:1695-2048).

    fftanal(tvec, sigx, sigy=None, tbounds=, Navr=, windowfunction=, windowoverlap=, onesided=, detrend=,
            tper=|minFreq=, verbose=, ...).  Extra keywords: nwins= (set the segment length directly; the
    reference's tper path truncates int(Fs*tper), quirk Q5), segments= (default True: keep Xseg/Pxx_seg...
    like the reference; False: averaged spectra only -- the fused, memory-light path)."""

    def __init__(self, tvec=None, sigx=None, sigy=None, **kwargs):
        self.verbose = kwargs.get("verbose", True)
        if tvec is None or sigx is None:
            if self.verbose:
                print("Please give at least a time-vector [s] and a signal vector [a.u.]")
            return
        self.init(tvec, sigx, sigy, **kwargs)

    def init(self, tvec=None, sigx=None, sigy=None, **kwargs):
        self.nosigy = sigy is None or sigx is sigy
        self.tvec = np.asarray(tvec)
        self.sigx = np.asarray(sigx)
        self.sigy = None if sigy is None else np.asarray(sigy)
        self.tbounds = kwargs.get("tbounds", [self.tvec.min(), self.tvec.max()])
        self.useMLAB = kwargs.get("useMLAB", False)
        self.plotit = kwargs.get("plotit", False)
        self.verbose = kwargs.get("verbose", True)
        self.Navr = kwargs.get("Navr", None)
        self.window = kwargs.get("windowfunction", "Hanning")
        if self.window is None:
            self.window = "Hanning"
        self.overlap = kwargs.get("windowoverlap", windows(self.window, verbose=False))
        self.tvecy = kwargs.get("tvecy", None)
        self.onesided = kwargs.get("onesided", None)
        self.detrendstyle = kwargs.get("detrend", 1)
        self.frange = kwargs.get("frange", None)
        self.axes = kwargs.get("axes", -1)
        self.segments = kwargs.get("segments", True)
        if self.tvecy is not None:
            raise NotImplementedError("tvecy resampling needs pybaseutils.utils.interp (absent from the reference)")
        if self.onesided is None:
            self.onesided = not (np.iscomplexobj(self.sigx) or (sigy is not None and np.iscomplexobj(self.sigy)))
        self.Fs = _fs(self.tvec)
        self.ibounds = self.__ibounds__(self.tvec, self.tbounds)
        self.nsig = np.size(self.tvec[self.ibounds[0]:self.ibounds[1]])
        calcNavr = False
        if self.Navr is None:
            calcNavr = True
            self.Navr = 8
        if "minFreq" in kwargs:
            kwargs["tper"] = 2.0 / kwargs["minFreq"]
        if "nwins" in kwargs:
            self.nwins = int(kwargs["nwins"])
            calcNavr = True
        elif "tper" in kwargs:
            self.tper = kwargs["tper"]
            self.nwins = int(self.Fs * self.tper)                      # :1770 (Q5)
        else:
            calcNavr = False
            self.nwins = _nwins(self.nsig, self.Navr, self.overlap)
        self.noverlap = _noverlap(self.nwins, self.overlap)
        if calcNavr:
            self.Navr = _navr(self.nsig, self.nwins, self.noverlap)
        self.win, self.winparams = windows(self.window, nwins=self.nwins, verbose=self.verbose, msgout=True)
        self.nfft = self.nwins
        self.Nnyquist = _nnyquist(self.nwins)
        self.S1, self.S2, self.NENBW, self.ENBW = _norms(self.win, self.Nnyquist, self.Fs)

    # -- reference statics kept for callers that use them
    _getNwins = staticmethod(_nwins)
    _getNoverlap = staticmethod(_noverlap)
    _getNavr = staticmethod(_navr)
    _getNnyquist = staticmethod(_nnyquist)
    _getNorms = staticmethod(_norms)
    __Fs__ = staticmethod(_fs)

    @staticmethod
    def _getS1(win):
        return np.sum(win)

    @staticmethod
    def _getS2(win):
        return np.sum(win ** 2.0)

    @staticmethod
    def _getNENBW(Nnyquist, S1, S2):
        return Nnyquist * 1.0 * S2 / (S1 ** 2)

    @staticmethod
    def _getENBW(Fs, S1, S2):
        return Fs * S2 / (S1 ** 2)

    @staticmethod
    def _checkCOLA(nsig, nwins, noverlap):
        return (nsig - nwins) % (nwins - noverlap) == 0

    @staticmethod
    def __ibounds__(tvec, tbounds):
        Fs = _fs(tvec)
        return [int(np.floor((tbounds[0] - tvec[0]) * Fs)), int(np.floor(1 + (tbounds[1] - tvec[0]) * Fs))]

    @staticmethod
    def __trimsig__(sigt, ibounds):
        return sigt[ibounds[0]:ibounds[1]]

    @staticmethod
    def makewindowfn(windowfunction, nwins, verbose=True):
        return windows(windowfunction, nwins=nwins, verbose=verbose, msgout=True)

    def update(self, d=None):
        if d is not None:
            super(fftanal, self).__init__(d)

    # -- transforms (fft_analysis.py:2096-2124) on the GPU
    def fft(self, sig, nfft=None, axes=None):
        return _E.fft(sig, n=self.nfft if nfft is None else nfft, axis=self.axes if axes is None else axes)

    def ifft(self, sig, nfft=None, axes=None):
        return _E.ifft(sig, n=self.nfft if nfft is None else nfft, axis=self.axes if axes is None else axes)

    def fftshift(self, sig, axes=None):
        return np.fft.fftshift(sig, axes=self.axes if axes is None else axes)

    def ifftshift(self, sig, axes=None):
        return np.fft.ifftshift(sig, axes=self.axes if axes is None else axes)

    # -- fft_win (fft_analysis.py:2126-2203)
    def fft_win(self, sig, tvec=None, detrendwin=False):
        sig = np.asarray(sig)
        if tvec is None:
            tvec = np.linspace(0.0, 1.0, len(sig))
        Fs = _fs(tvec)
        nwins, Navr, hop = self.nwins, self.Navr, self.nwins - self.noverlap
        dflag = _check_detrend(self.detrendstyle)
        if detrendwin:
            # per-window detrend instead of the global one (:2148 / :2171): the mean style runs in the kernel
            if dflag == 1:
                dflag = "segmean"
            elif dflag == 2:
                dflag = "seglinear"
        amp = 1.0 / (self.S1 * np.sqrt(self.ENBW))                     # :2197, :2202
        Xseg, pseg = _E.stft_frames(sig, self.win, hop, Navr, detrend=dflag, sided=_sided(self.onesided),
                                    amp_scale=amp, want_pseg=True)
        # mean frame time (:2163) from a running sum, and the first frame's span (:2161)
        csum = np.concatenate(([0.0], np.cumsum(np.asarray(tvec, dtype=np.float64))))
        st = np.arange(Navr) * hop
        tt = (csum[st + nwins] - csum[st]) / nwins
        if nwins < len(tvec):
            self.tper = tvec[nwins] - tvec[0]
        freq = np.fft.fftfreq(nwins, 1.0 / Fs)
        freq = freq[:self.Nnyquist] if self.onesided else np.fft.fftshift(freq)
        dt = (tvec[-1] - tvec[0]) / (len(tvec) - 1)
        return tt, freq, Xseg.astype(np.complex128), pseg * dt / self.S2

    @staticmethod
    def _fft_win(sig, **kwargs):
        """Static multi-channel twin of fft_win (fft_analysis.py:2554-2640): sig [nt] or [nt, nch]; returns
        (tt, freq, Xfft [nch, Navr, nbins] squeezed, pseg [nch, Navr] squeezed)."""
        x = np.asarray(sig)
        tvec = kwargs.get('tvec', None)
        onesided = kwargs.get('onesided', False)
        win, nwins, Navr, noverlap = kwargs['win'], kwargs['nwins'], kwargs['Navr'], kwargs['noverlap']
        Nnyquist, S1, S2, ENBW = kwargs['Nnyquist'], kwargs['S1'], kwargs['S2'], kwargs['ENBW']
        dflag = _check_detrend(kwargs['detrend_style'])
        if kwargs.get('detrendwin', False):
            dflag = {0: 0, 1: "segmean", 2: "seglinear"}[dflag]
        if tvec is None:
            tvec = np.linspace(0.0, 1.0, x.shape[0])
        Fs = kwargs.get('Fs', _fs(tvec))
        hop = nwins - noverlap
        cols = x[:, None] if x.ndim == 1 else x
        amp = 1.0 / (S1 * np.sqrt(ENBW))
        Xs, ps = [], []
        for c in range(cols.shape[1]):
            Xc, pc = _E.stft_frames(np.ascontiguousarray(cols[:, c]), win, hop, Navr, detrend=dflag, sided=_sided(onesided),
                                    amp_scale=amp, want_pseg=True)
            Xs.append(np.asarray(Xc).astype(np.complex128))
            ps.append(np.asarray(pc))
        csum = np.concatenate(([0.0], np.cumsum(np.asarray(tvec, dtype=np.float64))))
        st = np.arange(Navr) * hop
        tt = (csum[st + nwins] - csum[st]) / nwins
        freq = np.fft.fftfreq(nwins, 1.0 / Fs)
        freq = freq[:Nnyquist] if onesided else np.fft.fftshift(freq)
        dt = (tvec[-1] - tvec[0]) / (len(tvec) - 1)
        return tt, freq, np.stack(Xs).squeeze(), (np.stack(ps) * dt / S2).squeeze()

    # -- Welch (fft_analysis.py:1831-1836, :1924-2018)
    def Xstft(self):
        sig = self.__trimsig__(self.sigx, self.ibounds)
        t = self.__trimsig__(self.tvec, self.ibounds)
        self.tseg, self.freq, self.Xseg, self.Xpow = self.fft_win(sig, t)
        self.Xfft = np.mean(self.Xseg, axis=0)
        return self.freq, self.Xseg

    def Ystft(self):
        sig = self.__trimsig__(self.sigy, self.ibounds)
        t = self.__trimsig__(self.tvec, self.ibounds)
        self.tseg, self.freq, self.Yseg, self.Ypow = self.fft_win(sig, t)
        self.Yfft = np.mean(self.Yseg, axis=0)
        return self.freq, self.Yseg

    def Pstft(self):
        amp = np.sqrt(2) if self.onesided else 1.0
        if hasattr(self, "Xseg"):
            self.Pxx_seg = self.Xseg * np.conj(self.Xseg)
            self.Lxx_seg = amp * np.sqrt(np.abs(self.ENBW * self.Pxx_seg))
        if hasattr(self, "Yseg"):
            self.Pyy_seg = self.Yseg * np.conj(self.Yseg)
            self.Lyy_seg = amp * np.sqrt(np.abs(self.ENBW * self.Pyy_seg))
        if hasattr(self, "Xseg") and hasattr(self, "Yseg"):
            self.Pxy_seg = self.Xseg * np.conj(self.Yseg)              # class-path conjugation (Q4)
            self.Lxy_seg = amp * np.sqrt(np.abs(self.ENBW * self.Pxy_seg))
            self.phixy_seg = np.angle(self.Pxy_seg)
            self.Cxy_seg, self.Cxy2_seg = Cxy_Cxy2(self.Pxx_seg, self.Pyy_seg, self.Pxy_seg)

    def averagewins(self):
        """Averaged spectra straight from the fused device kernels (no [Navr, nfft] intermediates)."""
        i0, i1 = self.ibounds
        x = self.sigx[i0:i1]
        hop = self.nwins - self.noverlap
        dflag = _check_detrend(self.detrendstyle)
        scale = 1.0 / (self.S1 ** 2) / self.ENBW
        sided = _sided(self.onesided)
        if self.nosigy:
            self.Pxx = _E.welch_psd(x, self.win, hop, self.Navr, detrend=dflag, sided=sided,
                                    scale=scale).astype(np.complex128)
        else:
            y = self.sigy[i0:i1]
            pxx, pyy, pxy = _E.welch_csd(x, y, self.win, hop, self.Navr, detrend=dflag, sided=sided, scale=scale)
            self.Pxx = pxx.astype(np.complex128)
            self.Pyy = pyy[0].astype(np.complex128)
            self.Pxy = np.conj(pxy[0])                                 # X conj(Y)
        for p in ("Pxx", "Pyy", "Pxy"):
            if hasattr(self, p):
                setattr(self, "var" + p, (getattr(self, p) / np.sqrt(self.Navr)) ** 2.0)
        if hasattr(self, "Pxy"):
            self.phi_xy = np.angle(self.Pxy)
            Cxy, Cxy2 = Cxy_Cxy2(self.Pxx, self.Pyy[None, :], self.Pxy[None, :])
            self.Cxy, self.Cxy2 = Cxy[0], Cxy2[0]
            self.varPhxy = (np.sqrt(1.0 - self.Cxy2) / np.sqrt(2.0 * self.Navr * self.Cxy)) ** 2.0
            self.varCxy = ((1 - self.Cxy2) / np.sqrt(2 * self.Navr)) ** 2.0
            self.varCxy2 = 4.0 * self.Cxy2 * self.varCxy

    def pwelch(self):
        if self.segments:
            self.Xstft()
            if not self.nosigy:
                self.Ystft()
            self.Pstft()
        else:
            Fs = self.Fs
            freq = np.fft.fftfreq(self.nwins, 1.0 / Fs)
            self.freq = freq[:self.Nnyquist] if self.onesided else np.fft.fftshift(freq)
        self.averagewins()

    def _scipy_stft(self, sig):
        """scipy.signal.stft(sig, fs, window=self.win, nperseg=nwins, noverlap, nfft=nwins, detrend=self.detrend,
        return_onesided, boundary='zeros', padded=True) as the reference calls it (fft_analysis.py:1814-1822), frames on
        the GPU: zero extension by nperseg // 2 on both sides, zero padding to a whole number of hops, scaling
        1 / sum(window), bins 0..nfft/2 (one-sided) or all bins in fftfreq order.  `self.detrend` is a callable whose
        default axis is 0, so scipy removes the mean / line ACROSS segments at every sample position (reference quirk);
        that is linear in the segments, hence applied to the spectra.  -> (freq, t, Zxx[nfreq, nseg])"""
        nper, hop = self.nwins, self.nwins - self.noverlap
        x = np.asarray(sig)
        dt = np.complex128 if np.iscomplexobj(x) else np.float64
        x = np.concatenate([np.zeros(nper // 2, dtype=dt), x, np.zeros(nper // 2, dtype=dt)])
        nadd = (-(x.shape[-1] - nper) % hop) % nper
        x = np.concatenate([x, np.zeros(nadd, dtype=dt)])
        nseg = (x.shape[-1] - nper) // hop + 1
        win = np.asarray(self.win, dtype=np.float64)
        Z, _ = _E.stft_frames(x, win, hop, nseg, detrend=False, sided=_E.SIDED_HALF if self.onesided else _E.SIDED_RAW,
                              amp_scale=1.0 / np.sum(win))
        Z = np.asarray(Z).astype(np.complex128)
        d = _check_detrend(self.detrendstyle)
        if d == 1:
            Z = Z - Z.mean(axis=0, keepdims=True)
        elif d == 2 and nseg > 1:
            g = np.arange(nseg, dtype=np.float64) - 0.5 * (nseg - 1)
            slope = (g[:, None] * Z).sum(axis=0) / np.sum(g * g)
            Z = Z - Z.mean(axis=0, keepdims=True) - g[:, None] * slope[None, :]
        freq = np.fft.rfftfreq(nper, 1.0 / self.Fs) if self.onesided else np.fft.fftfreq(nper, 1.0 / self.Fs)
        t = np.arange(nper / 2.0, x.shape[-1] - nper / 2.0 + 1, hop) / float(self.Fs) - (nper / 2.0) / self.Fs
        return freq, t, np.ascontiguousarray(Z.T)

    def stft(self):
        if self.useMLAB:
            # the scipy.signal.stft branch (fft_analysis.py:1806-1824).  Its spectra are [frequency, segment]; the reference
            # then runs Pstft() and averagewins() on that layout: the segment means are taken along axis 0 (= frequency)
            # and Cxy_Cxy2 fails on the resulting 1-D arrays (IndexError at :1669) -- the same attributes are set in the
            # same order and the same exception type is raised.
            self.freq, self.tseg, self.Xseg = self._scipy_stft(self.sigx)
            _, _, self.Yseg = self._scipy_stft(self.sigy)
            self.Pstft()
            for p in ("Pxx", "Pyy", "Pxy"):
                setattr(self, p, np.mean(getattr(self, p + "_seg"), axis=0))
                setattr(self, "var" + p, (getattr(self, p) / np.sqrt(self.Navr)) ** 2.0)
            self.phi_xy = np.angle(self.Pxy)
            # the reference's next statement (averagewins -> Cxy_Cxy2, fft_analysis.py:1669) takes np.size(Pyy, axis=1) of the
            # 1-D means: numpy raises its own exception there (numpy >= 2: AxisError, a subclass of IndexError AND ValueError;
            # older: IndexError) -- the same call is made here so that `except` clauses behave as with the reference.  The
            # working half of this branch is public as `scipy_stft()`.
            self.Cxy, self.Cxy2 = Cxy_Cxy2(self.Pxx, self.Pyy, self.Pxy)
            raise IndexError("tuple index out of range (reference behaviour: Cxy_Cxy2 on the 1-D means of the "
                             "[frequency, segment] spectra, fft_analysis.py:1669)")      # (not reached with numpy's size())
        self.pwelch()

    def scipy_stft(self, sig=None):
        """(freq, t, Zxx[nfreq, nseg]) of scipy.signal.stft as the reference's useMLAB branch calls it (fft_analysis.py:
        1814-1822), frames on the GPU -- the part of `stft()` with useMLAB=True that works; `stft()` itself ends in the
        reference's exception.  sig: default self.sigx."""
        return self._scipy_stft(self.sigx if sig is None else sig)

    def fftpwelch(self):
        self.freq, self.Pxy, self.Pxx, self.Pyy, self.Cxy, self.phi_xy, self.fftinfo = fft_pwelch(
            self.tvec, self.sigx, self.sigy, self.tbounds, Navr=self.Navr, windowoverlap=self.overlap,
            windowfunction=self.window, useMLAB=self.useMLAB, plotit=self.plotit, verbose=self.verbose,
            detrend_style=self.detrendstyle, onesided=self.onesided)
        self.update(self.fftinfo)

    def convert2amplitudes(self):
        for p in ("Pxx", "Pyy", "Pxy"):
            if hasattr(self, p):
                tmp = np.sqrt(np.abs(self.ENBW * getattr(self, p)))
                if self.onesided:
                    tmp[1:-1] = np.sqrt(2) * tmp[1:-1]
                    if self.nfft % 2:
                        tmp[-1] = np.sqrt(2) * tmp[-1]
                setattr(self, "L" + p[1:], tmp)
                setattr(self, "varL" + p[1:], (tmp ** 2) * (getattr(self, "var" + p) / np.abs(getattr(self, p)) ** 2))

    # ---- geometry / normalisation getters under the reference's names (fft_analysis.py:2054-2076, :2412-2510)
    @staticmethod
    def _getNwins(nsig, Navr, windowoverlap):
        return _nwins(nsig, Navr, windowoverlap)

    @staticmethod
    def _getNoverlap(nwins, windowoverlap):
        return _noverlap(nwins, windowoverlap)

    @staticmethod
    def _getNavr(nsig, nwins, noverlap):
        return _navr(nsig, nwins, noverlap)

    @staticmethod
    def _getNnyquist(nfft):
        return _nnyquist(nfft)

    @staticmethod
    def _getNorms(win, Nnyquist, Fs):
        return _norms(win, Nnyquist, Fs)

    @staticmethod
    def _getMINoverlap(nsig, nwins, Navr):
        noverlap = 1
        while fftanal._checkCOLA(nsig, nwins, noverlap) is False and noverlap < 1e4:
            noverlap += 1
        return noverlap

    @staticmethod
    def _getMAXoverlap(nsig, nwins, Navr):
        noverlap = int(nwins) - 1
        while fftanal._checkCOLA(nsig, nwins, noverlap) is False and noverlap > 0:
            noverlap -= 1
        return noverlap

    def getNavr(self):
        self.Navr = fftanal._getNavr(self.nsig, self.nwins, self.noverlap)
        return self.Navr

    def getNwins(self):
        self.nwins = fftanal._getNwins(self.nsig, self.Navr, self.overlap)
        return self.nwins

    def getNoverlap(self):
        self.noverlap = fftanal._getNoverlap(self.nwins, self.overlap)
        return self.noverlap

    def getNnyquist(self):
        self.Nnyquist = self._getNnyquist(self.nwins)
        return self.Nnyquist

    def getNorms(self):
        self.S1, self.S2, self.NENBW, self.ENBW = fftanal._getNorms(self.win, self.Nnyquist, self.Fs)

    def crosscorr_stft(self):
        """Per-segment correlations from the per-segment spectra (fft_analysis.py:1880-1920); inverse FFTs on the GPU."""
        nfft = self.nwins
        for p in ("Pxx_seg", "Pyy_seg", "Pxy_seg"):
            if hasattr(self, p):
                tmp = np.array(getattr(self, p), dtype=np.complex128)
                if self.onesided:
                    tmp[..., 1:-1] *= 0.5
                    if nfft % 2:
                        tmp[..., -1] *= 0.5
                    tmp = np.sqrt(nfft) * _ifft_cols(tmp.T, nfft, True).T
                else:
                    tmp = np.sqrt(nfft) * _ifft_cols(np.fft.ifftshift(tmp, axes=-1).T, nfft, False).T
                if p.startswith("Pxx"):
                    self.Ex_seg = tmp[..., 0].copy()
                if p.startswith("Pyy"):
                    self.Ey_seg = tmp[..., 0].copy()
                setattr(self, "R" + p[1:], np.fft.fftshift(tmp, axes=-1))
        if hasattr(self, "Rxy_seg"):
            self.corrcoef_seg = self.Rxy_seg.copy() / np.sqrt(self.Ex_seg * self.Ey_seg)[..., None]
        self.lags = (np.asarray(range(1, nfft + 1), dtype=int) - self.Nnyquist) / self.Fs

    def crosscorr(self):
        """Correlations from the averaged spectra (fft_analysis.py:1840-1878); inverse FFTs on the GPU."""
        nfft = self.nwins
        for p in ("Pxx", "Pyy", "Pxy"):
            if hasattr(self, p):
                tmp = getattr(self, p).copy()
                if self.onesided:
                    tmp[..., 1:-1] *= 0.5
                    if nfft % 2:
                        tmp[..., -1] *= 0.5
                    tmp = np.sqrt(nfft) * _ifft_cols(tmp, nfft, True)
                else:
                    tmp = np.sqrt(nfft) * _ifft_cols(np.fft.ifftshift(tmp, axes=-1), nfft, False)
                if p == "Pxx":
                    self.Ex = tmp[..., 0].copy()
                if p == "Pyy":
                    self.Ey = tmp[..., 0].copy()
                setattr(self, "R" + p[1:], np.fft.fftshift(tmp, axes=-1))
        if hasattr(self, "Rxy"):
            self.corrcoef = self.Rxy.copy() / np.sqrt(self.Ex * self.Ey)
        self.lags = (np.asarray(range(1, nfft + 1), dtype=int) - self.Nnyquist) / self.Fs
