#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's own numpy path.

TEST INFRASTRUCTURE.  Runs ONLY in the build container, where the reference
checkout lives at /root/reference.  It executes the reference modules
*unmodified* (loaded by file path, nothing copied, no bytecode written) on
seeded synthetic inputs and stores inputs + outputs as small .npz fixtures in
this directory.  The GPU box never sees /root/reference; tests there read
only the .npz files.

Shims -- needed because the reference targets an old
numpy and an un-vendored sibling package:
  * numpy.deprecate         removed in numpy 2 (used at windows.py:1052)
  * numpy.trapz             removed in numpy 2 (used at fft_analysis.py:2174)
  * pybaseutils             absent; Struct / detrend_* stand-ins with the
                            matplotlib.mlab semantics (mean / LS-line removal
                            along axis 0).  PARITY UNPINNED at that boundary:
                            no reference test pins detrend_* results.
  * package name `FFT`      the reference imports itself as `FFT.<module>`

Usage:  python tests/golden/make_golden.py        (rewrites tests/golden/*.npz)
"""
import os
import sys
import types
import importlib.util

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np
import scipy.signal

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# shims
# --------------------------------------------------------------------------- #
def _install_shims():
    if not hasattr(np, "deprecate"):
        np.deprecate = lambda *a, **k: (lambda f: f)
    if not hasattr(np, "trapz"):
        np.trapz = np.trapezoid

    class Struct(object):
        def __init__(self, d=None):
            if d is not None:
                for k, v in d.items():
                    setattr(self, k, v)

        def dict_from_class(self):
            return dict((k, v) for k, v in self.__dict__.items())

    def detrend_mean(x, axis=0):
        x = np.asarray(x)
        return x - x.mean(axis=axis, keepdims=True)

    def detrend_none(x, axis=0):
        return x

    def detrend_linear(x, axis=0):
        return scipy.signal.detrend(x, axis=axis, type="linear")

    pb = types.ModuleType("pybaseutils")
    pbs = types.ModuleType("pybaseutils.Struct")
    pbu = types.ModuleType("pybaseutils.utils")
    pbs.Struct = Struct
    pbu.detrend_mean = detrend_mean
    pbu.detrend_none = detrend_none
    pbu.detrend_linear = detrend_linear
    pb.Struct = pbs
    pb.utils = pbu
    sys.modules["pybaseutils"] = pb
    sys.modules["pybaseutils.Struct"] = pbs
    sys.modules["pybaseutils.utils"] = pbu

    pkg = types.ModuleType("FFT")
    pkg.__path__ = [REF]
    sys.modules["FFT"] = pkg

    # ccf.py imports FFT.dft (python-2 source, cannot be compiled); ccf.ccf
    # itself never calls it.
    dft = types.ModuleType("FFT.dft")
    dft.fft = np.fft.fft
    dft.ifft = np.fft.ifft
    sys.modules["FFT.dft"] = dft


def _load(name):
    spec = importlib.util.spec_from_file_location("FFT." + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["FFT." + name] = mod
    spec.loader.exec_module(mod)
    return mod


# --------------------------------------------------------------------------- #
# deterministic synthetic inputs (reduced sizes)
# --------------------------------------------------------------------------- #
def gauss(seed, n):
    return np.random.default_rng(seed).standard_normal(n)


def metric_signal(n, seed=0x5EED2024):
    """complex64 white noise + the two Heinzel section-13 tones scaled to fs=1"""
    g = gauss(seed, 2 * n)
    k = np.arange(n, dtype=np.float64)
    z = (g[0::2] + 1j * g[1::2]) / np.sqrt(2.0)
    z = z + 2.82842712 * np.exp(2j * np.pi * 0.1234 * k) + 1.0 * np.exp(2j * np.pi * 0.25002157 * k)
    return z.astype(np.complex64)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def c(x):
    return np.ascontiguousarray(x)


def main():
    _install_shims()
    win_mod = _load("windows")
    fa = _load("fft_analysis")
    sg = _load("spectrogram")
    hb = _load("hilbert")
    cc = _load("ccf")
    nf = _load("notch_filter")

    # ---------------- windows: every catalogue entry, periodic+symmetric, + ROV
    names = ["Hanning", "Hamming", "Blackman", "SFT3F", "SFT4F", "SFT5F", "SFT3M", "SFT4M", "SFT5M",
             "Nuttall3a", "Nuttall3b", "Nuttall3", "Nuttall4a", "Nuttall4b", "Nuttall4c", "Nuttall4",
             "Kaiser", "Welch", "Bartlett", "box"]
    d = {}
    for nm in names:
        kw = {"beta": 8.6} if nm == "Kaiser" else {}
        for N in (16, 255, 1024):
            d["%s_per_%d" % (nm, N)] = win_mod.windows(nm, nwins=N, verbose=False, **kw)
            d["%s_sym_%d" % (nm, N)] = win_mod.windows(nm, nwins=N, periodic=False, verbose=False, **kw)
        d["%s_rov" % nm] = np.float64(win_mod.windows(nm, verbose=False, **kw))
    save("windows", **d)

    # ---------------- geometry helpers / norms
    geo = []
    for nsig, Navr, ov in [(65536, 127, 0.5), (16384, 8, 0.5), (2 ** 20, 511, 0.5), (100000, 17, 0.661),
                           (5000, 3, 0.0), (4096, 1, 0.5), (1000, 2, 0.75)]:
        nw = fa.fftanal._getNwins(nsig, Navr, ov)
        no = fa.fftanal._getNoverlap(nw, ov)
        na = fa.fftanal._getNavr(nsig, nw, no)
        geo.append([nsig, Navr, ov, nw, no, na, fa.fftanal._getNnyquist(nw)])
    w = win_mod.windows("Hanning", nwins=4096, verbose=False)
    S1, S2, NENBW, ENBW = fa.fftanal._getNorms(w, 2048, 1.0)
    save("geometry", table=np.array(geo, dtype=np.float64), hann4096_norms=np.array([S1, S2, NENBW, ENBW]))

    # ---------------- class path: two-sided complex64 Welch (the metric shape, reduced)
    for tag, n, nfft in [("c64_2e16_n4096", 2 ** 16, 4096), ("c64_2e14_n1024", 2 ** 14, 1024)]:
        z = metric_signal(n)
        t = np.arange(n, dtype=np.float64)
        M = (n - nfft // 2) // (nfft // 2)
        ft = fa.fftanal(t, z, None, tbounds=[t[0], t[-1]], Navr=M, windowfunction="Hanning",
                        windowoverlap=0.5, verbose=False, plotit=False)
        # nwins derived from Navr may differ from nfft; force it like the metric does (tper path truncates, Q5)
        ft.nwins = nfft
        ft.noverlap = ft.getNoverlap()
        ft.Navr = ft.getNavr()
        ft.win, ft.winparams = ft.makewindowfn(ft.window, ft.nwins, False)
        ft.getNnyquist()
        ft.getNorms()
        ft.pwelch()
        save("welch_class_" + tag, x=z, Fs=np.float64(ft.Fs), nwins=np.int64(ft.nwins),
             noverlap=np.int64(ft.noverlap), Navr=np.int64(ft.Navr), Nnyquist=np.int64(ft.Nnyquist),
             S1=np.float64(ft.S1), S2=np.float64(ft.S2), ENBW=np.float64(ft.ENBW), NENBW=np.float64(ft.NENBW),
             freq=c(ft.freq), tseg=c(ft.tseg), Xpow=c(ft.Xpow), Xfft=c(ft.Xfft),
             Pxx=c(ft.Pxx), varPxx=c(ft.varPxx),
             Xseg_head=c(ft.Xseg[:4]), Xseg_tail=c(ft.Xseg[-2:]), Pxx_seg_head=c(ft.Pxx_seg[:2]))

    # ---------------- class path: one-sided real x and y (Q1 crop), Hamming + SFT3F, with cross terms
    n = 2 ** 15
    t = np.arange(n) / 1000.0
    x = np.sin(2 * np.pi * 50.0 * t) + 0.1 * gauss(11, n) + 3.0
    y = 0.5 * np.sin(2 * np.pi * 50.0 * t - np.pi / 4) + 0.1 * gauss(12, n) - 1.0
    for wname in ("Hamming", "SFT3F"):
        ft = fa.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=31, windowfunction=wname,
                        verbose=False, plotit=False)
        ft.Xstft()
        ft.Ystft()
        ft.Pstft()
        # averagewins() -> Cxy_Cxy2 needs 2-D Pyy for 1-D inputs (latent bug, SURVEY.md "latent bugs"): do
        # the three means it performs directly (fft_analysis.py:1976-1988)
        Pxx = np.mean(ft.Pxx_seg, axis=0)
        Pyy = np.mean(ft.Pyy_seg, axis=0)
        Pxy = np.mean(ft.Pxy_seg, axis=0)
        save("welch_class_real_" + wname, t=t, x=x, y=y, Fs=np.float64(ft.Fs), nwins=np.int64(ft.nwins),
             noverlap=np.int64(ft.noverlap), Navr=np.int64(ft.Navr), overlap=np.float64(ft.overlap),
             freq=c(ft.freq), tseg=c(ft.tseg), Pxx=c(Pxx), Pyy=c(Pyy), Pxy=c(Pxy),
             Xpow=c(ft.Xpow), Xfft=c(ft.Xfft), Xseg_head=c(ft.Xseg[:3]), Yseg_head=c(ft.Yseg[:3]),
             Lxx_seg_head=c(ft.Lxx_seg[:2]), phixy_seg_head=c(ft.phixy_seg[:2]))

    # ---------------- function path: fft_pwelch, BASELINE cfg1 (2^16 float64, 1024-pt Hann 50%)
    n = 2 ** 16 + 1     # one extra sample so tbounds=[t0, t[-2]] selects exactly 2^16 samples without reflection
    k = np.arange(n, dtype=np.float64)
    t = k / 1.0e4
    x = np.sin(2 * np.pi * 0.05 * k) + 0.1 * gauss(21, n)
    y = 0.5 * np.sin(2 * np.pi * 0.05 * k - np.pi / 4) + 0.1 * gauss(22, n)
    y2 = np.stack([y, 0.25 * np.sin(2 * np.pi * 0.11 * k + 0.3) + 0.2 * gauss(23, n) + 1.5], axis=1)

    def run_pwelch(tag, tt, xx, yy, **kw):
        freq, Pxy, Pxx, Pyy, Cxy, phi, info = fa.fft_pwelch(tt, xx, yy, plotit=False, verbose=False, **kw)
        out = dict(t=tt, x=xx, y=yy, freq=c(freq), Pxy=c(Pxy), Pxx=c(Pxx), Pyy=c(Pyy), Cxy=c(Cxy), phi_xy=c(phi))
        for a in ("S1", "S2", "ENBW", "NENBW", "nwins", "noverlap", "Navr", "Fs", "nch", "minFreq"):
            out["info_" + a] = np.asarray(getattr(info, a))
        for a in ("Lxx", "Lyy", "Lxy", "Rxx", "Ryy", "Rxy", "iCxy", "corrcoef", "lags", "Cxy2", "varCxy",
                  "varCxy2", "varPxx", "varPyy", "varPxy", "varPhxy", "varLxx", "varLyy", "varLxy", "Ex", "Ey"):
            out["info_" + a] = c(np.asarray(getattr(info, a)))
        out["info_ibnds"] = np.asarray(info.ibnds)
        if hasattr(info, "Xfft_seg"):
            out["Xfft_seg_head"] = c(info.Xfft_seg[:2])
            out["Pxy_seg_head"] = c(info.Pxy_seg[:, :2])
        save(tag, **out)

    # tbounds strictly inside -> no reflection (Q2); Navr=127 -> nwins=1024
    run_pwelch("pwelch_cfg1", t, x, y, tbounds=[t[0], t[-2]], Navr=127, windowoverlap=0.5, windowfunction="Hanning")
    # full-record tbounds -> reflection branch (Q2) when rounding gives i1==len
    run_pwelch("pwelch_reflect", t[:8192], x[:8192], y[:8192], tbounds=None, Navr=15, windowfunction="Hanning")
    # two channels, two-sided request on real data, mean detrend off
    run_pwelch("pwelch_2ch_twosided", t[:16384], x[:16384], y2[:16384], tbounds=[t[0], t[16382]], Navr=31,
               windowfunction="Hamming", onesided=False, detrend_style=0)
    # linear detrend, minFreq path (nwins=int(Fs*tper), Q5)
    run_pwelch("pwelch_minfreq_linear", t[:16384], x[:16384] + 0.3 * t[:16384], y[:16384],
               tbounds=[t[0], t[16382]], minFreq=2.0 * 1.0e4 / 1024.0 * 1.0000001, detrend_style=-1)

    # the reference's own deterministic self-test input (fft_analysis.py:2895-2944), homebrew branch
    df = 5.0
    N = 2 ** 14
    tvec = (1.0 / df) * np.arange(0.0, 1.0, 1.0 / N)
    sigx = 0.1 * scipy.signal.square(2.0 * np.pi * (df * 30.0) * tvec) + 7.0
    sigy = (0.007 * np.sin(2.0 * np.pi * (df * 30.0) * tvec - np.pi / 4.0) + 2.5)[:, None]
    run_pwelch("pwelch_selftest_navr8", tvec, sigx, sigy, tbounds=[tvec[0], tvec[-1]], Navr=8,
               windowfunction="hamming", detrend_style=1)
    run_pwelch("pwelch_selftest_minfreq", tvec, sigx, sigy, tbounds=[tvec[0], tvec[-1]], minFreq=15 * df,
               detrend_style=1)

    # ---------------- spectrogram.stft (cfg3 shape reduced: float32 chirp, 2048-pt Hann 75%)
    n = 2 ** 16
    k = np.arange(n, dtype=np.float64)
    f_inst = np.linspace(0.05, 0.2, n)
    chirp = (np.sin(2 * np.pi * np.cumsum(f_inst)) + 0.01 * gauss(31, n)).astype(np.float32)
    tt = k.copy()
    # tper chosen so int(Fs*tper)==2048 despite truncation (Q5)
    st = sg.stft(tt, chirp, tper=2048.5, returnclass=True, windowfunction="Hanning", windowoverlap=0.75,
                 verbose=False)
    save("stft_f32_n2048_ov75", t=tt, x=chirp, nwins=np.int64(st.nwins), noverlap=np.int64(st.noverlap),
         Navr=np.int64(st.Navr), Fs=np.float64(st.Fs), freq=c(st.freq), tseg=c(st.tseg),
         Xseg_head=c(st.Xseg[:3]), Xseg_mid=c(st.Xseg[st.Navr // 2: st.Navr // 2 + 2]), Xseg_tail=c(st.Xseg[-2:]),
         Pxx=c(st.Pxx), Xfft=c(st.Xfft), Xpow=c(st.Xpow))
    twin, freq, Xseg = sg.stft(tt[:8192], chirp[:8192].astype(np.float64), tper=256.5, returnclass=False,
                               windowfunction="Hamming", verbose=False)
    save("stft_tuple_n256", t=tt[:8192], x=chirp[:8192].astype(np.float64), twin=c(twin), freq=c(freq), Xseg=c(Xseg))

    # ---------------- spectrogram.specgram (symmetric Hann, hop wl/2; and boxcar/no-overlap)
    s = chirp[:20000].astype(np.float64)
    ts = np.arange(s.size) * 1e-3
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        time1, f1, sp1 = sg.specgram(ts, s, wl=512, hanning=True, overlap=True)
        time2, f2, sp2 = sg.specgram(ts, s, wl=500, hanning=False, overlap=False)
    save("specgram", t=ts, s=s, time1=c(time1), f1=c(f1), sp1=c(sp1), time2=c(time2), f2=c(f2), sp2=c(sp2))

    # ---------------- hilbert: KAT input of hilbert.py:115-140 + even/odd/2-D/float32
    Nk = 32
    yk = np.sin(2 * np.pi * np.arange(Nk) / Nk)
    u_even = gauss(41, 4096)
    u_odd = gauss(42, 1001)
    u_2d = gauss(43, 6 * 512).reshape(6, 512)
    u_f32 = gauss(44, 2048).astype(np.float32)
    save("hilbert", yk=yk, zk=c(hb.hilbert(yk)), zk1d=c(hb.hilbert_1d(yk)),
         u_even=u_even, z_even=c(hb.hilbert(u_even)), u_odd=u_odd, z_odd=c(hb.hilbert(u_odd)),
         z_odd_1d=c(hb.hilbert_1d(u_odd)),
         u_2d=u_2d, z_2d=c(hb.hilbert(u_2d)), z_2d_ax0=c(hb.hilbert(u_2d, axes=0)),
         u_f32=u_f32, z_f32=c(hb.hilbert(u_f32)), z_nfft=c(hb.hilbert(u_even[:1000], nfft=1024)))

    # ---------------- ccf (ccf.py:139-148 shape, seeded)
    fs = 1e5
    N = 2048
    tc = np.arange(N) / fs
    phi = 50 * np.pi / 180
    x1 = np.sin(2 * np.pi * 1e3 * tc) + gauss(51, N)
    x2 = np.sin(2 * np.pi * 1e3 * tc + phi) + gauss(52, N)
    tau, co = cc.ccf(x1, x2, fs)
    x3 = gauss(53, 777) + 2.0
    x4 = np.roll(x3, 5) + 0.5 * gauss(54, 777)
    tau2, co2 = cc.ccf(x3, x4, 1.0)
    save("ccf", x1=x1, x2=x2, fs=np.float64(fs), tau=c(tau), co=c(co), x3=x3, x4=x4, tau2=c(tau2), co2=c(co2))

    # ---------------- notch design (docstring example notch_filter.py:66-71 + a sweep)
    d = {}
    b, a = nf.iirnotch(60.0 / (200.0 / 2), 30.0)
    d["b_doc"], d["a_doc"] = b, a
    rows = []
    for w0 in (0.01, 0.12, 0.3, 0.5, 0.9):
        for Q in (0.7, 5.0, 30.0, 200.0):
            bn, an = nf.iirnotch(w0, Q)
            bp, ap = nf.iirpeak(w0, Q)
            rows.append(np.concatenate([[w0, Q], bn, an, bp, ap]))
    d["sweep"] = np.array(rows)
    save("notch", **d)


if __name__ == "__main__":
    main()
