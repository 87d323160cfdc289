"""CPU oracle: a numpy restatement of gmweir/PYFFT's spectral hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  pyfft_amd/ never imports it.

Every function cites the reference file:line it follows (paths are relative to
the reference checkout).  Arithmetic is float64/complex128 exactly where the
reference's is (SURVEY.md quirks Q1-Q6 are reproduced, not "fixed").

Pinning: tests/test_oracle_golden.py checks every function here against the
fixtures in tests/golden/*.npz, which were produced by running the reference's
own modules (tests/golden/make_golden*.py), plus the reference's known-answer
material (hilbert.py:115-140 analytic identity, notch_filter.py:66-71, Heinzel
window constants echoed at windows.py:68-269).
Unpinned boundary: pybaseutils.utils.detrend_* is absent from the reference
checkout; mean / least-squares-line removal along axis 0 is assumed (the
matplotlib.mlab semantics) -- "parity unpinned" at that one call.
"""
import numpy as np

# --------------------------------------------------------------------------- #
# A1  windows.windows()                                    windows.py:57-297
# --------------------------------------------------------------------------- #
_COS_SUM = [  # (substring, coefficients, ROV)  -- order matters: windows.py:101-217 tests substrings in this order
    ("3f", (0.26526, -0.5, 0.23474), 0.667),
    ("4f", (0.21706, -0.42103, 0.28294, -0.07897), 0.75),
    ("5f", (0.1881, -0.36923, 0.28702, -0.13077, 0.02488), 0.785),
    ("3m", (0.28235, -0.52105, 0.19659), 0.655),
    ("4m", (0.241906, -0.460841, 0.255381, -0.041872), 0.721),
    ("5m", (0.209671, -0.407331, 0.281225, -0.092669, 0.0091036), 0.760),
    ("3a", (0.40897, -0.5, 0.09103), 0.612),
    ("3b", (0.4243801, -0.4973406, 0.0782793), 0.598),
    ("3", (0.375, -0.5, 0.125), 0.647),
    ("4a", (0.338946, -0.481973, 0.161054, -0.018027), 0.68),
    ("4b", (0.355768, -0.487396, 0.144232, -0.012604), 0.663),
    ("4c", (0.3635819, -0.4891775, 0.1365995, -0.0106411), 0.656),
    ("4", (0.3125, -0.46875, 0.1875, -0.03125), 0.705),
]


def _cos_sum(cc):
    def func(n):
        # windows.py:222-232 -- note the divisor is the *requested* length, so the
        # "periodic" variant (func(N+1)[:-1]) samples z = 2 pi k/(N+1)
        z = 2.0 * np.pi * np.arange(n) / n
        w = np.zeros(n, dtype=np.float64)
        for i, ci in enumerate(cc):
            w += ci if i == 0 else ci * np.cos(i * z)
        return w
    return func


def _window_func(name, beta=None):
    """(func(n) -> symmetric-or-whatever window of length n, ROV).  windows.py:63-272"""
    s = name.lower()
    if "hann" in s:
        return np.hanning, 0.50
    if "hamm" in s:
        return np.hamming, 0.50
    if "black" in s:
        return _cos_sum((0.35875, -0.48829, 0.14128, -0.01168)), 0.661     # windows.py:92-96
    if "nut" in s or "flat" in s or "sft" in s:
        for sub, cc, rov in _COS_SUM:
            if sub in s:
                return _cos_sum(cc), rov
        raise NameError("windows(): no coefficient set selected for %r (reference raises too)" % name)
    if "kaiser" in s:
        if beta is None:
            raise KeyError("beta")
        return (lambda n: np.kaiser(n, beta)), 2.0 / 3.0
    if "welch" in s:
        def func(n):                                                         # windows.py:249-252
            z = 2.0 * np.arange(n) / n
            return 1.0 - (z - 1.0) * (z - 1.0)
        return func, 0.293
    if "bart" in s:
        return np.bartlett, 0.50
    return (lambda n: np.ones(n, dtype=np.float64)), 0.0                     # windows.py:264-271


def windows(windowfunction, nwins=None, periodic=True, beta=None):
    """windows.py:57-297.  Without nwins: the recommended overlap (ROV)."""
    func, rov = _window_func(windowfunction, beta)
    if nwins is None:
        return rov
    if periodic:
        return func(nwins + 1)[:-1]          # windows.py:276-279
    return func(nwins)


# --------------------------------------------------------------------------- #
# A2  segment geometry and normalisation            fft_analysis.py:2412-2510
# --------------------------------------------------------------------------- #
def get_nwins(nsig, Navr, ov):               # :2412-2418
    nwins = int(np.floor(nsig * 1.0 / (Navr - Navr * ov + ov)))
    return nsig if nwins >= nsig else nwins


def get_noverlap(nwins, ov):                 # :2421-2422
    return int(np.ceil(ov * nwins))


def get_navr(nsig, nwins, noverlap):         # :2425-2429
    return 1 if nwins >= nsig else (nsig - noverlap) // (nwins - noverlap)


def get_nnyquist(nfft):                      # :2471-2484
    return (nfft + 1) // 2 if nfft % 2 else nfft // 2


def get_norms(win, Nnyquist, Fs):            # :2487-2510  (Q3: NENBW uses Nnyquist)
    S1 = np.sum(win)
    S2 = np.sum(win ** 2.0)
    return S1, S2, Nnyquist * 1.0 * S2 / (S1 ** 2), Fs * S2 / (S1 ** 2)


def get_fs(tvec):                            # :2376-2377
    return (len(tvec) - 1) / (tvec[-1] - tvec[0])


def get_ibounds(tvec, tbounds):              # :2380-2383
    Fs = get_fs(tvec)
    return [int(np.floor((tbounds[0] - tvec[0]) * Fs)), int(np.floor(1 + (tbounds[1] - tvec[0]) * Fs))]


# --------------------------------------------------------------------------- #
# A13  detrend                      pybaseutils.utils (absent) -- parity unpinned
# --------------------------------------------------------------------------- #
def detrend(x, style):
    """style >0 mean, 0 none, <0 linear; along axis 0  (fft_analysis.py:2539-2549)"""
    x = np.asarray(x)
    if style is None or style == 0:
        return x
    if style > 0:
        return x - x.mean(axis=0, keepdims=True)      # stays in the input dtype (Q6)
    n = x.shape[0]
    k = np.arange(n, dtype=np.float64)
    A = np.stack([k, np.ones(n)], axis=1)
    coef = np.linalg.lstsq(A, x.reshape(n, -1), rcond=None)[0]
    return x - (A @ coef).reshape(x.shape)


# --------------------------------------------------------------------------- #
# A3  fftanal.fft_win                                fft_analysis.py:2126-2203
# --------------------------------------------------------------------------- #
def fft_win(sig, tvec, win, nwins, noverlap, Navr, onesided, S1, S2, ENBW, detrend_style=1):
    """Returns (tt, freq, Xfft[Navr, nfft or Nnyquist] complex128, pseg[Navr])."""
    x_in = detrend(np.array(sig, copy=True), detrend_style)      # :2127, :2148 global detrend
    Fs = get_fs(tvec)
    nfft = nwins
    Nny = get_nnyquist(nfft)
    hop = nwins - noverlap
    Xfft = np.zeros((Navr, nfft), dtype=np.complex128)
    tt = np.zeros(Navr)
    pseg = np.zeros(Navr)
    for g in range(Navr):                                       # :2156-2176
        i0 = g * hop
        seg = win * x_in[i0:i0 + nwins]
        tt[g] = np.mean(tvec[i0:i0 + nwins])
        pseg[g] = np.trapezoid(seg * np.conj(seg), x=tvec[i0:i0 + nwins]).real
        Xfft[g] = np.fft.fft(seg, n=nfft)
    freq = np.fft.fftfreq(nfft, 1.0 / Fs)
    if onesided:                                                # :2179-2189  (Q1)
        freq = freq[:Nny]
        Xfft = Xfft[:, :Nny]
        Xfft[:, 1:-1] = np.sqrt(2) * Xfft[:, 1:-1]
        if nfft % 2:
            Xfft[:, -1] = np.sqrt(2) * Xfft[:, -1]
    else:                                                       # :2191-2192
        freq = np.fft.fftshift(freq)
        Xfft = np.fft.fftshift(Xfft, axes=-1)
    Xfft = Xfft / S1                                            # :2197
    pseg = pseg / S2                                            # :2198
    Xfft = Xfft / np.sqrt(ENBW)                                 # :2202
    return tt, freq, Xfft, pseg


def class_setup(tvec, sigx, Navr=None, windowfunction="Hanning", windowoverlap=None, tbounds=None,
                onesided=None, nwins=None, tper=None, minFreq=None, sigy=None, beta=None):
    """fftanal.init geometry  (fft_analysis.py:1713-1783).  `nwins=` overrides (Q5 escape hatch)."""
    if windowoverlap is None:
        windowoverlap = windows(windowfunction, beta=beta)
    if tbounds is None:
        tbounds = [tvec.min(), tvec.max()]
    if onesided is None:
        onesided = not (np.iscomplexobj(sigx) or (sigy is not None and np.iscomplexobj(sigy)))
    Fs = get_fs(tvec)
    ib = get_ibounds(tvec, tbounds)
    nsig = np.size(tvec[ib[0]:ib[1]])
    calc = False
    if Navr is None:
        calc = True
        Navr = 8
    if minFreq is not None:
        tper = 2.0 / minFreq
    if nwins is not None:
        calc = True
    elif tper is not None:
        nwins = int(Fs * tper)                                   # :1770 (Q5)
    else:
        calc = False
        nwins = get_nwins(nsig, Navr, windowoverlap)
    noverlap = get_noverlap(nwins, windowoverlap)
    if calc:
        Navr = get_navr(nsig, nwins, noverlap)
    win = windows(windowfunction, nwins=nwins, beta=beta)
    Nny = get_nnyquist(nwins)
    S1, S2, NENBW, ENBW = get_norms(win, Nny, Fs)
    return dict(Fs=Fs, ibounds=ib, nsig=nsig, nwins=nwins, noverlap=noverlap, Navr=Navr, win=win, Nnyquist=Nny,
                S1=S1, S2=S2, NENBW=NENBW, ENBW=ENBW, onesided=onesided, overlap=windowoverlap)


# --------------------------------------------------------------------------- #
# A4  Xstft / Pstft / averagewins                    fft_analysis.py:1924-2018
# --------------------------------------------------------------------------- #
def pwelch_class(tvec, sigx, sigy=None, detrend_style=1, **kw):
    """fftanal(...).pwelch() -> dict with the attributes the class sets."""
    g = class_setup(tvec, sigx, sigy=sigy, **kw)
    i0, i1 = g["ibounds"]
    t = tvec[i0:i1]
    out = dict(g)
    args = (g["win"], g["nwins"], g["noverlap"], g["Navr"], g["onesided"], g["S1"], g["S2"], g["ENBW"], detrend_style)
    out["tseg"], out["freq"], Xseg, out["Xpow"] = fft_win(sigx[i0:i1], t, *args)
    out["Xseg"] = Xseg
    out["Xfft"] = Xseg.mean(axis=0)                              # :1931
    out["Pxx_seg"] = Xseg * np.conj(Xseg)                        # :1946
    amp = np.sqrt(2) if g["onesided"] else 1.0
    out["Lxx_seg"] = amp * np.sqrt(np.abs(g["ENBW"] * out["Pxx_seg"]))
    out["Pxx"] = out["Pxx_seg"].mean(axis=0)                     # :1980
    out["varPxx"] = (out["Pxx"] / np.sqrt(g["Navr"])) ** 2.0     # :1988
    if sigy is not None and sigy is not sigx:
        _, _, Yseg, out["Ypow"] = fft_win(sigy[i0:i1], t, *args)
        out["Yseg"] = Yseg
        out["Yfft"] = Yseg.mean(axis=0)
        out["Pyy_seg"] = Yseg * np.conj(Yseg)                    # :1953
        out["Pxy_seg"] = Xseg * np.conj(Yseg)                    # :1960  (Q4: X.conj(Y) on the class path)
        out["phixy_seg"] = np.angle(out["Pxy_seg"])
        out["Pyy"] = out["Pyy_seg"].mean(axis=0)
        out["Pxy"] = out["Pxy_seg"].mean(axis=0)
        out["phi_xy"] = np.angle(out["Pxy"])
    return out


def crosscorr_class(r):
    """fftanal.crosscorr_stft (fft_analysis.py:1880-1920) and fftanal.crosscorr (:1840-1878) on a pwelch_class() result:
    correlations by inverse FFT of the per-segment / averaged spectra.  Everything the reference assigns before its
    corrcoef line (which needs the never-set self.nch for 1-D signals and raises) -- pinned by
    tests/golden/crosscorr_class.npz."""
    nfft, onesided = r["nwins"], r["onesided"]

    def back(P):
        tmp = np.array(P, dtype=np.complex128)
        if onesided:
            tmp[..., 1:-1] *= 0.5                                # :1893 / :1852
            if nfft % 2:
                tmp[..., -1] *= 0.5
            tmp = np.sqrt(nfft) * np.fft.irfft(tmp, n=nfft, axis=-1)
        else:
            tmp = np.sqrt(nfft) * np.fft.ifft(np.fft.ifftshift(tmp, axes=-1), n=nfft, axis=-1)
        return tmp
    out = {}
    for name in ("Pxx", "Pyy", "Pxy"):
        for suf in ("_seg", ""):
            if name + suf in r:
                tmp = back(r[name + suf])
                if name == "Pxx":
                    out["Ex" + suf] = tmp[..., 0].copy()
                if name == "Pyy":
                    out["Ey" + suf] = tmp[..., 0].copy()
                out["R" + name[1:] + suf] = np.fft.fftshift(tmp, axes=-1)
    out["lags"] = (np.arange(1, nfft + 1) - r["Nnyquist"]) / r["Fs"]
    return out


def welch_psd_stream(x, win, nfft, hop, nframes, Fs, detrend_style=1, chunk=4096):
    """Streaming two-sided (shifted) Welch PSD of the class path: identical arithmetic to
    fft_win -> Pstft -> averagewins (fft_analysis.py:2126-2203, :1946, :1980) without the [M,N]
    temporaries, so the 2^28-sample metric fits in RAM.  Used as bench.py's cpu_baseline ("port")."""
    x = np.asarray(x)
    if detrend_style and detrend_style > 0:
        x = x - x.mean()                                         # input dtype (Q6)
    S2 = np.sum(win ** 2.0)
    acc = np.zeros(nfft, dtype=np.float64)
    idx = np.arange(nfft)[None, :]
    for g0 in range(0, nframes, chunk):
        g1 = min(nframes, g0 + chunk)
        starts = (np.arange(g0, g1) * hop)[:, None]
        seg = win[None, :] * x[starts + idx]                     # float64 window => complex128 math
        X = np.fft.fft(seg, axis=-1)
        acc += (X.real ** 2 + X.imag ** 2).sum(axis=0)
    return np.fft.fftshift(acc) / (nframes * Fs * S2)            # /S1^2/ENBW == /(Fs*S2)


# --------------------------------------------------------------------------- #
# A5/A6  fft_pwelch                                    fft_analysis.py:36-648
# --------------------------------------------------------------------------- #
def cxy_cxy2(Pxx, Pyy, Pxy):                                     # :1662-1680
    Pxx = np.atleast_2d(Pxx.copy())
    if np.size(Pxx, axis=1) != np.size(Pyy, axis=1):
        Pxx = Pxx.T * np.ones((1, np.size(Pyy, axis=1)), dtype=Pxx.dtype)
    Cxy2 = Pxy * np.conj(Pxy) / (np.abs(Pxx) * np.abs(Pyy))
    Cxy = Pxy / np.sqrt(np.abs(Pxx) * np.abs(Pyy))
    return Cxy, Cxy2


def fft_pwelch(tvec, sigx, sigy, tbounds=None, Navr=None, windowoverlap=None, windowfunction=None,
               detrend_style=None, onesided=None, tper=None, minFreq=None, useMLAB=False):
    """fft_pwelch: the homebrew branch (useMLAB=False, :339-446) or the matplotlib.mlab.csd branch (useMLAB=True,
    :254-330).  Returns (freq, Pxy, Pxx, Pyy, Cxy, phi_xy, info-dict)."""
    calcNavr = Navr is None
    if windowfunction is None:
        windowfunction = "Hanning"
    if windowoverlap is None:
        windowoverlap = windows(windowfunction)
    if detrend_style is None:
        detrend_style = 1
    if tbounds is None:
        tbounds = [tvec[0], tvec[-1]]
    if onesided is None:
        onesided = not (np.iscomplexobj(sigx) or np.iscomplexobj(sigy))
    Fs = (len(tvec) - 1) / (tvec[-1] - tvec[0])                   # :136
    i0 = int(np.floor(Fs * (tbounds[0] - tvec[0])))               # :143
    i1 = int(np.floor(1 + Fs * (tbounds[1] - tvec[0])))           # :144
    nsig = np.size(tvec[i0:i1])
    sigy = np.atleast_2d(sigy)
    if sigy.shape[1] == len(tvec):
        sigy = sigy.T
    nch = sigy.shape[1]
    nTmodel = sigx.shape[0] != sigy.shape[0]                      # :169-176: sigx is one window of a model signal
    if nTmodel:
        if not calcNavr:
            raise UnboundLocalError("calcNavr")                   # the reference fails at :172 when Navr is given
        nwins = sigx.shape[0]
    elif minFreq is not None or tper is not None:
        if minFreq is not None:
            tper = 2.0 / minFreq                                  # :180-181
        nwins = int(Fs * tper)                                    # :183 (Q5)
    else:
        if Navr is None:
            Navr = 8
        calcNavr = False
        nwins = get_nwins(nsig, Navr, windowoverlap)              # :189
    noverlap = get_noverlap(nwins, windowoverlap)                 # :194
    reflecting = False
    if i0 == 0 and i1 == len(tvec):                               # :197-205 (Q2)
        reflecting = True
        sigx = np.concatenate((sigx[nwins - 1:0:-1, ...], sigx, sigx[-1:-nwins:-1, ...]), axis=0)
        sigy = np.concatenate((sigy[nwins - 1:0:-1, ...], sigy, sigy[-1:-nwins:-1, ...]), axis=0)
        nsig = sigx.shape[0]
    if calcNavr:
        Navr = get_navr(nsig, nwins, noverlap)                    # :209
    if nwins >= nsig:
        Navr = 1
        nwins = nsig
    nfft = nwins
    Nny = get_nnyquist(nfft)
    win = windows(windowfunction, nwins=nwins)
    S1, S2, NENBW, ENBW = get_norms(win, Nny, Fs)

    if useMLAB:
        # :254-330 -- mlab.csd(x, y_c, nfft, Fs, detrend=<per segment>, window=win, noverlap, sides, scale_by_freq=True):
        # conj(X) Y / Fs / sum(w^2), mlab's one-sided doubling (all but DC and the even-length Nyquist), mean over
        # (len - noverlap) // step segments; the reference then keeps the first Nnyquist bins (:317-326)
        x_in, y_in = np.asarray(sigx[i0:i1], dtype=np.float64), np.asarray(sigy[i0:i1, :], dtype=np.float64)
        step = nwins - noverlap
        nseg = (x_in.shape[0] - noverlap) // step
        idx = (np.arange(nseg) * step)[:, None] + np.arange(nfft)[None, :]
        kind = "none" if not detrend_style else ("mean" if detrend_style > 0 else "linear")
        X = np.fft.fft(win * _mlab_detrend(x_in[idx], kind), axis=-1)
        Y = np.stack([np.fft.fft(win * _mlab_detrend(y_in[:, c][idx], kind), axis=-1) for c in range(nch)])
        sc = 1.0 / (Fs * np.sum(win ** 2))
        Pxx = (np.conj(X) * X).mean(axis=0) * sc
        Pyy = (np.conj(Y) * Y).mean(axis=1) * sc
        Pxy = (np.conj(X)[None] * Y).mean(axis=1) * sc
        freq = np.fft.fftfreq(nfft, 1.0 / Fs)
        if onesided:
            nb = nfft // 2 + 1
            dbl = np.full(nb, 2.0)
            dbl[0] = 1.0
            if nfft % 2 == 0:
                dbl[-1] = 1.0
            Pxx, Pyy, Pxy = (Pxx[:nb] * dbl)[:Nny], (Pyy[:, :nb] * dbl)[:, :Nny], (Pxy[:, :nb] * dbl)[:, :Nny]
            freq = freq[:Nny]
        else:
            Pxx, Pyy, Pxy = np.fft.fftshift(Pxx), np.fft.fftshift(Pyy, axes=-1), np.fft.fftshift(Pxy, axes=-1)
            freq = np.fft.fftshift(freq)
        info = dict(S1=S1, S2=S2, NENBW=NENBW, ENBW=ENBW, nwins=nwins, noverlap=noverlap, Navr=Navr, Fs=Fs, nch=nch,
                    ibnds=[i0, i1], win=win, reflecting=reflecting, minFreq=2.0 * Fs / nwins)
        return pwelch_epilogue(freq, Pxx, Pyy.real.T, Pxy.T, info, onesided)

    x_in = detrend(sigx if nTmodel else sigx[i0:i1], detrend_style)   # :346-357 (nT-model: the whole one-window model)
    y_in = detrend(sigy[i0:i1, :], detrend_style)
    hop = nwins - noverlap
    Xfft = np.zeros((Navr, nfft), dtype=np.complex128)
    Yfft = np.zeros((nch, Navr, nfft), dtype=np.complex128)
    for g in range(Navr):                                         # :362-388
        a = g * hop
        Xfft[g] = np.fft.fft(win * (x_in if nTmodel else x_in[a:a + nwins]), n=nfft, axis=0)   # :366-369
        Yfft[:, g, :] = np.fft.fft(win[:, None] * y_in[a:a + nwins, :], n=nfft, axis=0).T
    Pxx_seg = Xfft * np.conj(Xfft)                                # :391-393
    Pyy_seg = Yfft * np.conj(Yfft)
    Pxy_seg = Yfft * np.conj(Xfft)[None, :, :]                    # Q4: Y.conj(X) on the function path
    freq = np.fft.fftfreq(nfft, 1.0 / Fs)
    if onesided:                                                  # :402-421
        freq = freq[:Nny]
        Pxx_seg = Pxx_seg[:, :Nny].copy()
        Pyy_seg = Pyy_seg[:, :, :Nny].copy()
        Pxy_seg = Pxy_seg[:, :, :Nny].copy()
        Pxx_seg[:, 1:-1] *= 2
        Pyy_seg[:, :, 1:-1] *= 2
        Pxy_seg[:, :, 1:-1] *= 2
        if nfft % 2:
            Pxx_seg[:, -1] *= 2
            Pyy_seg[:, :, -1] *= 2
            Pxy_seg[:, :, -1] *= 2
    else:                                                         # :423-427
        freq = np.fft.fftshift(freq)
        Pxx_seg = np.fft.fftshift(Pxx_seg, axes=-1)
        Pyy_seg = np.fft.fftshift(Pyy_seg, axes=-1)
        Pxy_seg = np.fft.fftshift(Pxy_seg, axes=-1)
    Pxx_seg = (1.0 / S1 ** 2) * Pxx_seg / ENBW                    # :432-440
    Pyy_seg = (1.0 / S1 ** 2) * Pyy_seg / ENBW
    Pxy_seg = (1.0 / S1 ** 2) * Pxy_seg / ENBW
    Pxx = Pxx_seg.mean(axis=0)                                    # :444-446
    Pyy = Pyy_seg.mean(axis=1).T
    Pxy = Pxy_seg.mean(axis=1).T
    info = dict(S1=S1, S2=S2, NENBW=NENBW, ENBW=ENBW, nwins=nwins, noverlap=noverlap, Navr=Navr, Fs=Fs, nch=nch,
                ibnds=[i0, i1], win=win, reflecting=reflecting, Xfft_seg=Xfft, Yfft_seg=Yfft, Pxy_seg=Pxy_seg,
                Pxx_seg=Pxx_seg, Pyy_seg=Pyy_seg, minFreq=2.0 * Fs / nwins)
    out = pwelch_epilogue(freq, Pxx, Pyy, Pxy, info, onesided)
    return out


def pwelch_epilogue(freq, Pxx, Pyy, Pxy, info, onesided):
    """fft_analysis.py:489-648 -- coherence, phase, amplitudes, correlations (host-side, length-N work)."""
    Navr, ENBW, nfft, Fs, nch = info["Navr"], info["ENBW"], info["nwins"], info["Fs"], info["nch"]
    Nny = get_nnyquist(nfft)
    Cxy, Cxy2 = cxy_cxy2(Pxx, Pyy, Pxy)                           # :489
    info["varCxy"] = ((1.0 - Cxy * np.conjugate(Cxy)) / np.sqrt(2 * Navr)) ** 2.0
    info["varCxy2"] = 4.0 * Cxy2 * info["varCxy"]
    info["varPxx"] = (Pxx / np.sqrt(Navr)) ** 2.0
    info["varPyy"] = (Pyy / np.sqrt(Navr)) ** 2.0
    info["varPxy"] = (Pxy / np.sqrt(Navr)) ** 2.0
    info["varPhxy"] = (np.sqrt(1.0 - np.abs(Cxy2))) / np.sqrt(2 * Navr * np.sqrt(np.abs(Cxy2))) ** 2.0
    phi_xy = np.arctan2(Pxy.imag, Pxy.real)                       # :520
    Lxx = np.sqrt(np.abs(ENBW * Pxx))
    Lyy = np.sqrt(np.abs(ENBW * Pyy))
    Lxy = np.sqrt(np.abs(ENBW * Pxy))
    if onesided:                                                  # :530-563
        Lxx[1:-1] *= np.sqrt(2)
        Lyy[1:-1, :] *= np.sqrt(2)
        Lxy[1:-1, :] *= np.sqrt(2)
        if nfft % 2:
            Lxx[-1] *= np.sqrt(2)
            Lyy[-1, :] *= np.sqrt(2)
            Lxy[-1, :] *= np.sqrt(2)

        def back(P):
            R = P.copy()
            R[1:-1, ...] *= 0.5
            if nfft % 2:
                R[-1, ...] *= 0.5
            return np.fft.irfft(R, n=nfft, axis=0)
        Rxx, Ryy, Rxy = back(Pxx), back(Pyy), back(Pxy)
        iCxy = np.fft.irfft(Cxy.copy(), n=nfft, axis=0)
    else:                                                         # :571-574
        Rxx = np.fft.ifft(np.fft.ifftshift(Pxx, axes=0), n=nfft, axis=0)
        Ryy = np.fft.ifft(np.fft.ifftshift(Pyy, axes=0), n=nfft, axis=0)
        Rxy = np.fft.ifft(np.fft.ifftshift(Pxy, axes=0), n=nfft, axis=0)
        iCxy = np.fft.ifft(np.fft.ifftshift(Cxy, axes=0), n=nfft, axis=0)
    s = np.sqrt(nfft)
    Rxx, Ryy, Rxy, iCxy = Rxx * s, Ryy * s, Rxy * s, iCxy * s     # :579-582
    Ex = Rxx[0, ...].copy()
    Ey = Ryy[0, ...].copy()
    corrcoef = Rxy / np.sqrt(np.ones((nfft, 1), dtype=Rxy.dtype) * (Ex * Ey))     # :590
    Rxx, Ryy, Rxy, iCxy, corrcoef = [np.fft.fftshift(a, axes=0) for a in (Rxx, Ryy, Rxy, iCxy, corrcoef)]
    lags = (np.asarray(range(1, nfft + 1), dtype=int) - Nny) / Fs  # :597
    info.update(Lxx=Lxx, Lyy=Lyy, Lxy=Lxy, Rxx=Rxx, Ryy=Ryy, Rxy=Rxy, iCxy=iCxy, Ex=Ex, Ey=Ey,
                corrcoef=corrcoef, lags=lags, Cxy2=Cxy2)
    info["varLxx"] = (Lxx ** 2) * (info["varPxx"] / np.abs(Pxx) ** 2)
    info["varLyy"] = (Lyy ** 2) * (info["varPyy"] / np.abs(Pyy) ** 2)
    info["varLxy"] = (Lxy ** 2) * (info["varPxy"] / np.abs(Pxy) ** 2)
    if nch == 1:                                                  # :605-631
        Pyy, Pxy, Cxy, phi_xy = Pyy.flatten(), Pxy.flatten(), Cxy.flatten(), phi_xy.flatten()
        for k in ("Cxy2", "lags", "Rxx", "Ryy", "Rxy", "corrcoef", "iCxy", "Lxx", "Lyy", "Lxy", "varLxx", "varLyy",
                  "varLxy", "varCxy", "varCxy2", "varPxx", "varPyy", "varPxy", "varPhxy"):
            info[k] = info[k].flatten()
    return freq, Pxy, Pxx, Pyy, Cxy, phi_xy, info


# --------------------------------------------------------------------------- #
# A8/A9  spectrogram.stft / specgram                     spectrogram.py:49-168
# --------------------------------------------------------------------------- #
def stft(tt, y, tper=None, returnclass=True, **kw):
    """spectrogram.py:140-168 -> fftanal.init(tper=...) + .stft() == pwelch() on the class path."""
    if tper is None:
        tper = (tt[-1] - tt[0]) / 20
    out = pwelch_class(tt, y, tper=tper, **kw)
    if returnclass:
        return out
    twin = np.linspace(tt[0], tt[-1], num=out["Navr"], endpoint=True)
    return twin, out["freq"], out["Xseg"]


def specgram(t, s, wl=512, hanning=True, overlap=True):
    """spectrogram.py:49-134 (windowAverage=None branch; the averaging branch is un-runnable on py3:
    float slice sizes at :116-118)."""
    s = s.flatten()
    n = len(s)
    dt = np.abs(t[1] - t[0])
    nWindows = (2 * (n - (n % wl)) // wl - 1) if overlap else ((n - (n % wl)) // wl - 1)
    out = np.zeros((wl, nWindows))
    w = np.hanning(wl)                                            # symmetric Hann (:109)
    for i in range(nWindows):
        a = i * wl // 2 if overlap else i * wl
        if hanning:
            out[:, i] = np.sqrt(8.0 / 3.0) * np.abs(np.fft.fft(w * s[a:a + wl])) ** 2 / wl
        else:
            out[:, i] = np.abs(np.fft.fft(s[a:a + wl])) ** 2 / wl
    fAxis = np.fft.fftfreq(wl, dt)
    if overlap:
        time = np.linspace(t[0] + wl * dt / 2, t[0] + wl * dt * ((nWindows / 2 - 1) + 1 / 2), num=nWindows)
    else:
        time = np.linspace(t[0] + wl * dt / 2, t[0] + wl * dt * ((nWindows - 1) + 1 / 2), num=nWindows)
    return time, fAxis, out


# --------------------------------------------------------------------------- #
# A10  hilbert                                                hilbert.py:22-112
# --------------------------------------------------------------------------- #
def hilbert(uin, nfft=None, axes=-1):
    if nfft is None:
        uin = np.atleast_1d(uin)
        nfft = np.shape(uin)[axes]
    nyq = (nfft + 1) // 2 if nfft % 2 else nfft // 2              # :47-50 (odd-N quirk, Q6)
    U = np.fft.fft(uin, n=nfft, axis=axes)                        # input-dtype precision under numpy>=2
    U[(slice(None),) * (axes % U.ndim) + (slice(nyq + 1, None),)] = 0.0
    U[(slice(None),) * (axes % U.ndim) + (slice(1, nyq),)] *= 2.0
    return np.fft.ifft(U, n=nfft, axis=axes).squeeze()


def hilbert_1d(uin, nfft=None):                                  # :70-112
    if nfft is None:
        uin = np.atleast_1d(uin)
        nfft = len(uin)
    nyq = (nfft + 1) // 2 if nfft % 2 else nfft // 2
    U = np.fft.fft(uin, n=nfft, axis=-1)
    h = np.zeros(nfft)
    h[0] = 1.0
    h[1:nyq] = 2.0
    h[nyq] = 1.0
    return np.fft.ifft(U * h, n=nfft, axis=-1)


# --------------------------------------------------------------------------- #
# A11  ccf                                                        ccf.py:66-77
# --------------------------------------------------------------------------- #
def ccf(x1, x2, fs):
    npts = len(x1)
    lags = np.arange(-npts + 1, npts)
    tau = -lags / float(fs)
    ccov = np.correlate(x1 - x1.mean(), x2 - x2.mean(), mode="full")
    return tau, ccov / (npts * x1.std() * x2.std())


def ccf_fft(x1, x2, fs):
    """Same result as ccf() through zero-padded FFTs (what the GPU kernel does); O(N log N) so the oracle
    also finishes at sizes where np.correlate cannot.  Equality with ccf() is asserted in the CPU tests."""
    n = len(x1)
    L = 1 << int(np.ceil(np.log2(2 * n)))
    a = np.fft.rfft(x1 - x1.mean(), L)
    b = np.fft.rfft(x2 - x2.mean(), L)
    r = np.fft.irfft(a * np.conj(b), L)
    ccov = np.concatenate([r[L - n + 1:], r[:n]])
    return -np.arange(-n + 1, n) / float(fs), ccov / (n * x1.std() * x2.std())


# --------------------------------------------------------------------------- #
# N4  derivative by FFT                              fft_analysis.py:1419-1587
# --------------------------------------------------------------------------- #
def fft_deriv(sig, xx=None, modified=True, detrend=None, window=None):
    """real(ifft(wavenumber * fft(win * scaled sig))) / win with one-sided differences at the two ends, on the scaled
    axes of rescale()/unscale() (fft_analysis.py:1419-1451, :1514-1565).  The lowpass/downsample branch (:1494-1509) is
    a no-op for the default arguments (Fs_new == Fs) and is not restated."""
    sig = np.asarray(sig, dtype=np.float64)
    xx = 1.0 * np.arange(len(sig)) if xx is None else np.asarray(xx, dtype=np.float64)
    slope = sig.max() - sig.min()
    offset = sig.min()
    slope = slope if slope != 0 else 1.0
    xslope = xx.max() - xx.min()
    xslope = xslope if xslope != 0 else 1.0
    xoffset = -1e-4
    y = (sig - offset) / slope
    x = (xx - xoffset) / xslope
    if detrend is not None:
        y = detrend(y)
    N = len(x)
    dx = x[1] - x[0]
    L = N * dx
    k = 2.0 * np.pi * np.fft.fftfreq(N, d=dx / L)
    wn = (1j * np.sin(k * dx) / dx if modified else 1j * k) / L
    win = np.ones(N) if window is None else window(N)
    y = win * y
    d0 = (y[1] - y[0]) / (x[1] - x[0])
    d1 = (y[-1] - y[-2]) / (x[-1] - x[-2])
    d = np.real(np.fft.ifft(wn * np.fft.fft(y))) / win
    d[0], d[-1] = d0, d1
    return d * slope / xslope, x * xslope + xoffset


# --------------------------------------------------------------------------- #
# N3  centre of gravity of a spectrum                         Doppler.py:43-81
# --------------------------------------------------------------------------- #
def cog(x, fs, fmin=None, fmax=None):
    """Power-weighted mean frequency of the two-sided spectrum (Doppler.py:43-58).  With fmin given, the band's frequencies
    are paired with the FIRST len(band) bins of the shifted spectrum: the reference's second mask is computed on the
    already selected frequency axis (Doppler.py:53-54), and this restatement keeps that."""
    if fmax is None:
        fmax = fs
    n = len(x)
    freq = np.fft.fftshift(np.fft.fftfreq(n, 1 / fs))
    spec = np.fft.fftshift(np.fft.fft(x)) / np.sqrt(n / 2)
    if fmin is not None:
        freq = freq[(np.abs(freq) >= fmin) & (np.abs(freq) <= fmax)]
        spec = spec[:len(freq)]
    if len(freq) > 0:
        p = np.abs(spec) ** 2
        return np.sum(p * freq) / np.sum(p)
    return 0.0


def cog_frames(t, x, fs, win=512, ov=0.5, fmin=None, fmax=None, window=None):
    """The window loop of cogspec (Doppler.py:61-81) over the complete windows at hop floor((1-ov) win): tcog = mean time,
    coge = cog of each window.  fmin/fmax here are a true band limit on |f| (not cog()'s pairing), window an optional
    taper: the build-defined extensions of pyfft_amd.doppler.cog_frames."""
    hop = int(np.floor((1.0 - ov) * win))
    nframes = (len(x) - win) // hop + 1
    freq = np.fft.fftfreq(win, 1 / fs)
    keep = np.ones(win, dtype=bool)
    if fmin is not None:
        keep &= np.abs(freq) >= fmin
    if fmax is not None:
        keep &= np.abs(freq) <= fmax
    w = np.ones(win) if window is None else np.asarray(window, dtype=np.float64)
    tcog = np.zeros(nframes)
    coge = np.zeros(nframes)
    for g in range(nframes):
        seg = np.asarray(x[g * hop:g * hop + win])
        p = np.abs(np.fft.fft(w * seg)) ** 2 * keep          # float64 (the reference transforms complex64 in single)
        den = p.sum()
        coge[g] = (p * freq).sum() / den if den > 0 else 0.0
        tcog[g] = np.mean(t[g * hop:g * hop + win])
    return tcog, coge


# --------------------------------------------------------------------------- #
# A12  notch / peak biquad design                      notch_filter.py:175-241
# --------------------------------------------------------------------------- #
def _design_notch_peak(w0, Q, ftype):
    w0 = float(w0)
    Q = float(Q)
    if w0 > 1.0 or w0 < 0.0:
        raise ValueError("w0 should be such that 0 < w0 < 1")
    bw = (w0 / Q) * np.pi
    w0 = w0 * np.pi
    gb = 1 / np.sqrt(2)
    if ftype == "notch":
        beta = (np.sqrt(1.0 - gb ** 2.0) / gb) * np.tan(bw / 2.0)
    else:
        beta = (gb / np.sqrt(1.0 - gb ** 2.0)) * np.tan(bw / 2.0)
    gain = 1.0 / (1.0 + beta)
    if ftype == "notch":
        b = gain * np.array([1.0, -2.0 * np.cos(w0), 1.0])
    else:
        b = (1.0 - gain) * np.array([1.0, 0.0, -1.0])
    a = np.array([1.0, -2.0 * gain * np.cos(w0), (2.0 * gain - 1.0)])
    return b, a


def iirnotch(w0, Q):
    return _design_notch_peak(w0, Q, "notch")


def iirpeak(w0, Q):
    return _design_notch_peak(w0, Q, "peak")


# --------------------------------------------------------------------------- #
# F1/F2  build-defined (absent from the reference): causal FIR by overlap-add, notch application
#        Nearest reference code: filters.py:282 (np.convolve FIR), ccf.py:283 (fftconvolve).
#        PARITY UNPINNED against the reference (nothing to pin to); pinned against scipy.signal.lfilter.
# --------------------------------------------------------------------------- #
def fftfilt(b, x):
    """y = lfilter(b, 1, x): causal FIR, output truncated to len(x)."""
    x = np.asarray(x)
    return np.convolve(x, np.asarray(b, dtype=np.float64))[:x.shape[0]]


def biquad_fir(b, a, ntaps):
    """First ntaps samples of the impulse response of b/a (second order), by the recursion itself."""
    h = np.zeros(ntaps)
    for n in range(ntaps):
        acc = b[n] if n < len(b) else 0.0
        for k in range(1, len(a)):
            if n - k >= 0:
                acc -= a[k] * h[n - k]
        h[n] = acc / a[0]
    return h


def notch_apply(x, w0, Q, ntaps=513, ftype="notch"):
    """Truncated-impulse-response FIR realisation of the designed biquad (SURVEY.md F2 option (i))."""
    b, a = _design_notch_peak(w0, Q, ftype)
    return fftfilt(biquad_fir(b, a, ntaps), x)


# --------------------------------------------------------------------------- #
# A7  plain transforms (convention: forward unnormalised e^{-j}, inverse 1/N)     dft.py:108-133,242-290
# --------------------------------------------------------------------------- #
def fft(x, n=None, axis=-1):
    return np.fft.fft(x, n=n, axis=axis)


def ifft(x, n=None, axis=-1):
    return np.fft.ifft(x, n=n, axis=axis)


# --------------------------------------------------------------------------- #
# multi-channel CSD (cfg5): build-defined generalisation of fft_pwelch's ref x channels loop
#   (fft_analysis.py:387-393, HeatPulse_Funcs.py:576-583).  G[k,i,j] = mean_g X_i[g,k] conj(X_j[g,k]) / (Fs*S2)
# --------------------------------------------------------------------------- #
def csd_matrix(x, win, nfft, hop, nframes, Fs, detrend_style=1):
    """x: [nch, nsig] real.  Returns one-sided-uncropped G[nfft//2+1, nch, nch] complex128 (no doubling)."""
    x = np.asarray(x, dtype=np.float64)
    if detrend_style and detrend_style > 0:
        x = x - x.mean(axis=1, keepdims=True)
    nch = x.shape[0]
    nb = nfft // 2 + 1
    S2 = np.sum(win ** 2.0)
    G = np.zeros((nb, nch, nch), dtype=np.complex128)
    for g in range(nframes):
        X = np.fft.rfft(win[None, :] * x[:, g * hop:g * hop + nfft], axis=-1)     # [nch, nb]
        G += X.T[:, :, None] * np.conj(X.T[:, None, :])
    return G / (nframes * Fs * S2)


# ---------------------------------------------------------------------------------------------
# matplotlib.mlab wrappers  psd / csd / coh / coh2  (fft_analysis.py:1060-1155).  The algorithm lives in matplotlib
# (mlab._spectral_helper, pinned here by version 3.10.8 of the build image): restated below from its published
# behaviour and pinned by tests/golden/mlab_wrappers.npz, which the reference itself produced (make_golden_mlab.py).
# Segments of NFFT samples, step NFFT - noverlap, count (len - noverlap) // step; detrend PER SEGMENT; symmetric Hann
# (mlab.window_hanning = np.hanning(NFFT) * x); conj(X) Y; / Fs / sum(w^2); one-sided doubling of every bin except DC
# (and Nyquist when NFFT is even); mean over segments; freqs = fftfreq with a positive Nyquist.
# ---------------------------------------------------------------------------------------------
def mlab_segments(n, nfft, noverlap):
    step = nfft - noverlap
    return (n - noverlap) // step, step


def _mlab_detrend(seg, kind):
    if kind in (None, "none"):
        return seg
    if kind == "mean":
        return seg - seg.mean(axis=-1, keepdims=True)
    if kind == "linear":
        k = np.arange(seg.shape[-1], dtype=np.float64)
        kc = k - k.mean()
        slope = (seg * kc).sum(axis=-1, keepdims=True) / (kc * kc).sum()
        return seg - seg.mean(axis=-1, keepdims=True) - slope * kc
    raise ValueError(kind)


def mlab_csd(x, y, nfft, fs, detrend="none", noverlap=0):
    """(Pxy, freqs) of matplotlib.mlab.csd(x, y, NFFT=nfft, Fs=fs, detrend=detrend, window=window_hanning,
    noverlap=noverlap) for real x, y (one-sided, scale_by_freq).  mlab.psd(x) = mlab_csd(x, x).real."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    nseg, step = mlab_segments(x.size, nfft, noverlap)
    idx = (np.arange(nseg) * step)[:, None] + np.arange(nfft)[None, :]
    w = np.hanning(nfft)
    nb = nfft // 2 + 1
    X = np.fft.fft(w * _mlab_detrend(x[idx], detrend), axis=-1)[:, :nb]
    Y = np.fft.fft(w * _mlab_detrend(y[idx], detrend), axis=-1)[:, :nb]
    res = np.conj(X) * Y / fs / np.sum(w ** 2)
    if nfft % 2 == 0:
        res[:, 1:-1] *= 2.0
    else:
        res[:, 1:] *= 2.0
    f = np.fft.fftfreq(nfft, 1.0 / fs)[:nb].copy()
    if nfft % 2 == 0:
        f[-1] *= -1.0
    return res.mean(axis=0), f


def _band(P, F, fmin, fmax, peak_threshold=None):
    keep = np.ones(P.shape, dtype=bool)
    if fmin is not None:
        keep &= F >= fmin
    if fmax is not None:
        keep &= F <= fmax
    if peak_threshold is not None:
        keep &= P > peak_threshold
    return P[keep], F[keep]


def mlab_psd_wrapper(x, fs, nfft=2048, fmin=None, fmax=None, detrend="none", peak_threshold=None, ov=0.67):
    """fft_analysis.psd (:1113-1131)"""
    P, F = mlab_csd(x, x, nfft, fs, detrend, int(np.floor(ov * nfft)))
    return _band(P.real, F, fmin, fmax, peak_threshold)


def mlab_csd_wrapper(x, y, fs, nfft=2048, fmin=0, fmax=500e3, detrend="none", peak_threshold=None, ov=0.67):
    """fft_analysis.csd (:1134-1155)"""
    P, F = mlab_csd(x, y, nfft, fs, detrend, int(np.floor(ov * nfft)))
    return _band(P, F, fmin, fmax, peak_threshold)


def mlab_coh_wrapper(x, y, fs, nfft=2048, fmin=0.0, fmax=500e3, detrend="mean", ov=0.67):
    """fft_analysis.coh (:1060-1088): sqrt of the magnitude-squared coherence"""
    nov = int(ov * nfft)
    Pxx, F = mlab_csd(x, x, nfft, fs, detrend, nov)
    Pyy, _ = mlab_csd(y, y, nfft, fs, detrend, nov)
    Pxy, _ = mlab_csd(x, y, nfft, fs, detrend, nov)
    c2 = np.abs(Pxy) ** 2 / (Pxx.real * Pyy.real)
    keep = (F <= fmax) & (F >= fmin)
    return np.sqrt(c2[keep]), F[keep]


def mlab_coh2_wrapper(x, y, fs, nfft=4096, fmin=0, fmax=500e3):
    """fft_analysis.coh2 (:1090-1110).  PARITY UNPINNED: the reference passes noverlap = nfft/2 as a float, which the
    matplotlib of this image rejects (TypeError), so the reference cannot produce a fixture; restated with nfft // 2."""
    fxx, f = mlab_csd(x, x, nfft, fs, "none", nfft // 2)
    fyy, _ = mlab_csd(y, y, nfft, fs, "none", nfft // 2)
    fxy, _ = mlab_csd(x, y, nfft, fs, "none", nfft // 2)
    keep = np.abs(f) <= fmax
    return {"coh": (np.abs(fxy * np.conj(fxy)) / (fxx * fyy)).real[keep], "f": f[keep], "PS": np.abs(fxx)[keep],
            "pha": np.arctan2(fxy.imag, fxy.real)[keep]}
