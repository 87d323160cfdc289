"""Drop-in for the reference's spectrogram.py hot functions: `specgram` (spectrogram.py:49-134) and `stft`
(spectrogram.py:140-168).  Frames are produced by the STFT kernel on the MI355X."""
import numpy as np

from . import engine as _E
from .fft_analysis import fftanal


def specgram(t, s, wl=512, hanning=True, overlap=True, windowAverage=None):
    """(time, fAxis, spectrogram[wl, nWindows]) -- sqrt(8/3)|FFT(hanning(wl) s_i)|^2 / wl per frame with hop
    wl/2 (or |FFT|^2/wl, hop wl, without window/overlap); symmetric np.hanning like the reference (:109).
    windowAverage=k averages k consecutive frames (the reference's branch needs py2 integer division, :116-122)."""
    if windowAverage is not None:
        overlap = False
    s = np.asarray(s).flatten()
    n = len(s)
    dt = np.abs(t[1] - t[0])
    if overlap:
        nWindows = 2 * (n - (n % wl)) // wl - 1
    else:
        nWindows = (n - (n % wl)) // wl - 1
    hop = wl // 2 if overlap else wl
    if hanning:
        win, amp = np.hanning(wl), np.sqrt(8.0 / 3.0) / wl
    else:
        win, amp = np.ones(wl), 1.0 / wl
    out, _ = _E.stft_frames(s, win, hop, nWindows, detrend=False, sided=_E.SIDED_RAW, amp_scale=amp, power=True,
                            bin_major=True)
    spectrogram = out.astype(np.float64)
    fAxis = np.fft.fftfreq(wl, dt)
    if windowAverage is not None:
        k = int(windowAverage)
        nA = nWindows // k
        spectrogram = spectrogram[:, :nA * k].reshape(wl, nA, k).mean(axis=2)
        time = np.linspace(t[0] + wl * dt / 2, t[0] + wl * dt * ((nWindows - 1) + 1 / 2), num=nA)
        return time, fAxis, spectrogram
    if overlap:
        time = np.linspace(t[0] + wl * dt / 2, t[0] + wl * dt * ((nWindows / 2 - 1) + 1 / 2), num=nWindows)
    else:
        time = np.linspace(t[0] + wl * dt / 2, t[0] + wl * dt * ((nWindows - 1) + 1 / 2), num=nWindows)
    return time, fAxis, spectrogram


def stft(tt, y_in, tper=None, returnclass=True, **kwargs):
    """fftanal().init(tt, y_in, tper=tper, **kwargs); .stft()  -> the object, or (twin, freq, Xseg)."""
    tt = np.asarray(tt)
    if tper is None:
        tper = (tt[-1] - tt[0]) / 20
        if tper < tt[2] - tt[1]:
            print("check your stft window size")
    kwargs.setdefault("verbose", False)
    Ystft = fftanal()
    Ystft.init(tt, y_in, tper=tper, **kwargs)
    Ystft.stft()
    if returnclass:
        return Ystft
    twin = np.linspace(tt[0], tt[-1], num=Ystft.Navr, endpoint=True)
    return twin, Ystft.freq, Ystft.Xseg
