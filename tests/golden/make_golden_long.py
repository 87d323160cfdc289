#!/usr/bin/env python3
"""Golden vectors for LONG segments -- the reference's default regime (Navr=8 -> nwins = floor(nsig/4.5),
fft_analysis.py:2412-2418) on its own test_fftanal input shape (fft_analysis.py:2950-2993: N = 2^19, df = 5 Hz, noisy
sines, Navr = 8, hamming, detrend_style = 1, onesided) -> nwins = 116 508, a transform 14x longer than one
workgroup handles, and not a power of two.

TEST INFRASTRUCTURE, build container only (see make_golden.py for the shims).  test_fftanal seeds its noise from the
clock (`np.random.seed()`); here the same signals are built from fixed seeds so that the test can regenerate the inputs
and only the outputs are stored -- sub-sampled (every bin of the first 1024, every 31st bin after that, the last 64;
same rule for the lag axis) to keep the fixture small.

Usage:  python tests/golden/make_golden_long.py        (writes tests/golden/pwelch_long_navr8.npz, stft_long.npz)
"""
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from make_golden import _install_shims, _load, save, c
from inputs_long import long_signals, stft_long_signal, subsample_index


def main():
    _install_shims()
    _load("windows")
    fa = _load("fft_analysis")
    sg = _load("spectrogram")

    tvec, sigx, sigy = long_signals()
    # ---- function path: exactly test_fftanal's call (fftanal(...).fftpwelch() -> fft_pwelch, :1798-1803)
    ft = fa.fftanal(tvec, sigx, sigy, tbounds=[tvec[0], tvec[-1]], Navr=8, windowfunction="hamming", useMLAB=False,
                    plotit=False, verbose=False, detrend_style=1, onesided=True)
    ft.fftpwelch()
    info = ft.fftinfo
    nb = ft.Pxx.shape[0]
    ib = subsample_index(nb)
    il = subsample_index(np.asarray(info.lags).shape[0])
    out = dict(N=np.int64(tvec.size), df=np.float64(5.0), seed=np.int64(20260401), ibin=ib, ilag=il,
               nwins=np.int64(info.nwins), noverlap=np.int64(info.noverlap), Navr=np.int64(info.Navr),
               Fs=np.float64(info.Fs), S1=np.float64(info.S1), S2=np.float64(info.S2), ENBW=np.float64(info.ENBW),
               NENBW=np.float64(info.NENBW), ibnds=np.asarray(info.ibnds), nbins=np.int64(nb),
               freq=c(np.asarray(ft.freq)[ib]))
    for name in ("Pxx", "Pyy", "Pxy", "Cxy", "phi_xy"):
        out[name] = c(np.asarray(getattr(ft, name)).reshape(nb, -1)[ib, 0])
    for name in ("Lxx", "Lyy", "Lxy", "Cxy2", "varPxx", "varPxy", "varCxy"):
        out["info_" + name] = c(np.asarray(getattr(info, name)).reshape(nb, -1)[ib, 0])
    for name in ("Rxx", "Ryy", "Rxy", "corrcoef", "lags"):
        a = np.asarray(getattr(info, name))
        out["info_" + name] = c(a.reshape(a.shape[0], -1)[il, 0])
    for name in ("Ex", "Ey"):
        out["info_" + name] = c(np.atleast_1d(np.asarray(getattr(info, name))).ravel())
    save("pwelch_long_navr8", **out)

    # ---- class path on the same record (init -> Xstft/Ystft/Pstft, :1924-1971; the three segment means of averagewins,
    # :1976-1988, taken directly because Cxy_Cxy2 fails on 1-D Pyy -- see make_golden.py), complex two-sided as well
    ft2 = fa.fftanal(tvec, sigx, sigy, tbounds=[tvec[0], tvec[-1]], Navr=8, windowoverlap=0.5, windowfunction="hamming",
                     useMLAB=False, plotit=False, verbose=False, detrend=1, onesided=True)
    ft2.Xstft()
    ft2.Ystft()
    ft2.Pstft()
    nb2 = ft2.Pxx_seg.shape[1]
    ib2 = subsample_index(nb2)
    save("welch_class_long_navr8", N=np.int64(tvec.size), df=np.float64(5.0), seed=np.int64(20260401), ibin=ib2,
         nwins=np.int64(ft2.nwins), noverlap=np.int64(ft2.noverlap), Navr=np.int64(ft2.Navr), Fs=np.float64(ft2.Fs),
         nbins=np.int64(nb2), freq=c(np.asarray(ft2.freq)[ib2]), tseg=c(ft2.tseg), Xpow=c(ft2.Xpow),
         Pxx=c(np.mean(ft2.Pxx_seg, axis=0)[ib2]), Pyy=c(np.mean(ft2.Pyy_seg, axis=0)[ib2]),
         Pxy=c(np.mean(ft2.Pxy_seg, axis=0)[ib2]), Xfft=c(np.asarray(ft2.Xfft)[ib2]),
         Xseg_first=c(ft2.Xseg[0][ib2]), Xseg_last=c(ft2.Xseg[-1][ib2]), Yseg_first=c(ft2.Yseg[0][ib2]))

    # ---- stft through spectrogram.stft on a shorter record with long, non-power-of-two windows (tper -> nwins = 10 000)
    k, xs = stft_long_signal()
    n = k.size
    st = sg.stft(k, xs, tper=10000.5, returnclass=True, windowfunction="Hanning", windowoverlap=0.5, verbose=False)
    nb3 = st.Xseg.shape[1]
    ib3 = subsample_index(nb3)
    save("stft_long_n10000", n=np.int64(n), seed=np.int64(77), ibin=ib3, nwins=np.int64(st.nwins),
         noverlap=np.int64(st.noverlap), Navr=np.int64(st.Navr), Fs=np.float64(st.Fs), nbins=np.int64(nb3),
         freq=c(np.asarray(st.freq)[ib3]), tseg=c(st.tseg), Xseg_sub=c(np.asarray(st.Xseg)[:, ib3]),
         Pxx=c(np.asarray(st.Pxx)[ib3]), Xpow=c(st.Xpow))


if __name__ == "__main__":
    main()
