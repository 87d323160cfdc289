"""GPU parity of the long-segment path (segments longer than one workgroup transform: k_long.hip + the multi-pass FFT) --
the reference's DEFAULT regime: Navr = 8 gives nwins = floor(nsig/4.5) (fft_analysis.py:2412-2418), 116 508 points for
its own test_fftanal input (fft_analysis.py:2950-2993).  Checked against the fixtures the reference produced
(tests/golden/make_golden_long.py; inputs rebuilt from seeds by tests/golden/inputs_long.py) and against the CPU oracle.
Tolerances: Welch bins rtol 2e-4 / atol 1e-6 max; spectra 1e-4 of the largest value."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden
from oracle import cpu_ref as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import inputs_long  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pyfft_amd
    from pyfft_amd import _ffi
    _ffi.init()
    return pyfft_amd


def close_rel(a, b, rel, what=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.max(np.abs(a - b))
    assert err <= rel * np.max(np.abs(b)), (what, err, np.max(np.abs(b)))


def test_fft_pwelch_long_navr8_golden(P):
    """the reference's own test_fftanal call: fftanal(...).fftpwelch() with N = 2^19, Navr = 8, hamming, one-sided"""
    g = load_golden("pwelch_long_navr8")
    tvec, sigx, sigy = inputs_long.long_signals(int(g["N"]), float(g["df"]), int(g["seed"]))
    ft = P.fftanal(tvec, sigx, sigy, tbounds=[tvec[0], tvec[-1]], Navr=8, windowfunction="hamming", useMLAB=False,
                   plotit=False, verbose=False, detrend_style=1, onesided=True)
    ft.fftpwelch()
    info = ft.fftinfo
    assert int(info.nwins) == int(g["nwins"]) == 116508 and int(info.Navr) == 8 and int(info.noverlap) == int(g["noverlap"])
    assert list(info.ibnds) == list(g["ibnds"])
    for k in ("S1", "S2", "ENBW", "NENBW", "Fs"):
        np.testing.assert_allclose(getattr(info, k), g[k], rtol=1e-12)
    ib, il = g["ibin"], g["ilag"]
    nb = int(g["nbins"])
    assert ft.Pxx.shape[0] == nb
    np.testing.assert_allclose(np.asarray(ft.freq)[ib], g["freq"], rtol=1e-12, atol=1e-12)
    for name in ("Pxx", "Pyy", "Pxy"):
        got = np.asarray(getattr(ft, name)).reshape(nb, -1)[ib, 0]
        np.testing.assert_allclose(got, g[name], rtol=2e-4, atol=1e-6 * np.max(np.abs(g[name])), err_msg=name)
    # epilogue: coherence, phase, amplitude spectra, correlations through the length-116508 inverse FFTs on the device
    close_rel(np.asarray(ft.Cxy).reshape(nb, -1)[ib, 0], g["Cxy"], 1e-3, "Cxy")
    m = np.abs(g["Cxy"]) > 0.5
    dphi = np.angle(np.exp(1j * (np.asarray(ft.phi_xy).reshape(nb, -1)[ib, 0][m] - g["phi_xy"][m])))
    assert np.max(np.abs(dphi)) < 2e-3
    for k in ("Lxx", "Lyy", "Lxy", "varPxx"):
        close_rel(np.asarray(getattr(info, k)).reshape(nb, -1)[ib, 0], g["info_" + k], 1e-3, k)
    for k in ("Rxx", "Ryy", "Rxy", "corrcoef", "lags"):
        a = np.asarray(getattr(info, k))
        close_rel(a.reshape(a.shape[0], -1)[il, 0], g["info_" + k], 1e-3, k)
    close_rel(np.atleast_1d(info.Ex).ravel(), g["info_Ex"], 1e-3, "Ex")
    close_rel(np.atleast_1d(info.Ey).ravel(), g["info_Ey"], 1e-3, "Ey")


def test_fft_pwelch_function_long_navr8(P):
    """the same through the module-level function, default arguments apart from the window (fft_analysis.py:36)"""
    g = load_golden("pwelch_long_navr8")
    tvec, sigx, sigy = inputs_long.long_signals(int(g["N"]), float(g["df"]), int(g["seed"]))
    freq, Pxy, Pxx, Pyy, Cxy, phi, info = P.fft_pwelch(tvec, sigx, sigy, [tvec[0], tvec[-1]], Navr=8,
                                                        windowfunction="hamming", detrend_style=1, onesided=True)
    ib = g["ibin"]
    for name, got in (("Pxx", Pxx), ("Pyy", Pyy), ("Pxy", Pxy)):
        np.testing.assert_allclose(np.asarray(got)[ib], g[name], rtol=2e-4, atol=1e-6 * np.max(np.abs(g[name])), err_msg=name)


def test_fftanal_class_long_navr8_golden(P):
    """class path fftanal.pwelch() (fft_analysis.py:1831, :1924-1990) on the same record: Xseg of 8 x 58 254 bins"""
    g = load_golden("welch_class_long_navr8")
    tvec, sigx, sigy = inputs_long.long_signals(int(g["N"]), float(g["df"]), int(g["seed"]))
    ft = P.fftanal(tvec, sigx, sigy, tbounds=[tvec[0], tvec[-1]], Navr=8, windowoverlap=0.5, windowfunction="hamming",
                   verbose=False, detrend=1, onesided=True)
    ft.pwelch()
    assert ft.onesided and ft.nwins == int(g["nwins"]) and ft.Navr == int(g["Navr"]) and ft.noverlap == int(g["noverlap"])
    ib = g["ibin"]
    np.testing.assert_allclose(np.asarray(ft.freq)[ib], g["freq"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ft.tseg, g["tseg"], rtol=1e-9)
    for k in ("Pxx", "Pyy", "Pxy"):
        np.testing.assert_allclose(np.asarray(getattr(ft, k))[ib], g[k], rtol=2e-4, atol=1e-6 * np.abs(g[k]).max(), err_msg=k)
    sc = np.abs(g["Xseg_first"]).max()
    assert np.max(np.abs(ft.Xseg[0][ib] - g["Xseg_first"])) <= 1e-4 * sc
    assert np.max(np.abs(ft.Xseg[-1][ib] - g["Xseg_last"])) <= 1e-4 * sc
    assert np.max(np.abs(ft.Yseg[0][ib] - g["Yseg_first"])) <= 1e-4 * np.abs(g["Yseg_first"]).max()
    np.testing.assert_allclose(ft.Xpow, g["Xpow"], rtol=2e-4)
    assert ft.Cxy.shape == ft.Pxy.shape


def test_stft_long_windows_golden(P):
    """spectrogram.stft with 10 000-point windows (not a power of two: chirp-z on a 32 768-point multi-pass transform,
    all 25 frames through each launch together)"""
    g = load_golden("stft_long_n10000")
    k, xs = inputs_long.stft_long_signal(int(g["n"]), int(g["seed"]))
    st = P.stft(k, xs, tper=10000.5, returnclass=True, windowfunction="Hanning", windowoverlap=0.5)
    assert st.nwins == 10000 and st.Navr == int(g["Navr"]) and st.noverlap == int(g["noverlap"])
    ib = g["ibin"]
    np.testing.assert_allclose(np.asarray(st.freq)[ib], g["freq"], rtol=1e-12)
    np.testing.assert_allclose(st.tseg, g["tseg"], rtol=1e-9)
    assert np.max(np.abs(np.asarray(st.Xseg)[:, ib] - g["Xseg_sub"])) <= 1e-4 * np.abs(g["Xseg_sub"]).max()
    np.testing.assert_allclose(np.asarray(st.Pxx)[ib], g["Pxx"], rtol=2e-4, atol=1e-6 * np.abs(g["Pxx"]).max())
    np.testing.assert_allclose(st.Xpow, g["Xpow"], rtol=2e-4)


# ---------------------------------------------------------------------------------------------- engine level vs oracle
@pytest.mark.parametrize("nfft,hop,nframes,cplx", [(16384, 8192, 5, True), (16384, 4096, 9, False), (10000, 5000, 7, True),
                                                    (5000, 2500, 40, False), (131072, 65536, 3, True),
                                                    (1 << 20, 1 << 19, 3, False), (300001, 150000, 2, True)])
def test_long_welch_psd_vs_oracle(P, nfft, hop, nframes, cplx):
    E = P.engine
    rng = np.random.default_rng(nfft % 9973)
    n = (nframes - 1) * hop + nfft + 17
    x = rng.standard_normal(n) + 0.7 * np.sin(2 * np.pi * 0.0123 * np.arange(n)) + 3.0
    if cplx:
        x = (x + 1j * (rng.standard_normal(n) - 1.5)).astype(np.complex64)
    else:
        x = x.astype(np.float32)
    win = O.windows("Hanning", nwins=nfft)
    S2 = float(np.sum(win ** 2))
    got = E.welch_psd(x, win, hop, nframes, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)
    ref = O.welch_psd_stream(x, win, nfft, hop, nframes, 1.0)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-6 * ref.max())
    # one-sided crop / doubling (Q1) and raw order agree with the two-sided result
    raw = E.welch_psd(x, win, hop, nframes, detrend=True, sided=E.SIDED_RAW, scale=1.0 / S2)
    np.testing.assert_allclose(np.fft.fftshift(raw), got, rtol=1e-12)
    one = E.welch_psd(x, win, hop, nframes, detrend=True, sided=E.SIDED_ONE, scale=1.0 / S2)
    nny = (nfft + 1) // 2 if nfft % 2 else nfft // 2
    exp = raw[:nny].copy()
    exp[1:-1] *= 2
    if nfft % 2:
        exp[-1] *= 2
    np.testing.assert_allclose(one, exp, rtol=1e-12)


def test_long_welch_detrend_modes(P):
    """none / linear / per-segment mean / per-segment line on 9 000-point segments against numpy float64"""
    E = P.engine
    rng = np.random.default_rng(5)
    nfft, hop, M = 9000, 3000, 6
    n = (M - 1) * hop + nfft
    k = np.arange(n)
    x = (rng.standard_normal(n) + 2.0 + 1e-4 * k).astype(np.float32)
    win = O.windows("Hamming", nwins=nfft)
    xd = x.astype(np.float64)

    def ref(prep, seg):
        acc = np.zeros(nfft)
        sig = prep(xd)
        for g in range(M):
            f = seg(sig[g * hop:g * hop + nfft])
            acc += np.abs(np.fft.fft(win * f)) ** 2
        return acc / M

    ident = lambda v: v                                                    # noqa: E731
    j = np.arange(nfft) - 0.5 * (nfft - 1)
    line = lambda v: v - v.mean() - j * (np.sum(j * v) / np.sum(j * j))    # noqa: E731
    kk = k - 0.5 * (n - 1)
    cases = {
        False: ref(ident, ident),
        "linear": ref(lambda v: v - v.mean() - kk * (np.sum(kk * v) / np.sum(kk * kk)), ident),
        "segmean": ref(ident, lambda v: v - v.mean()),
        "seglinear": ref(ident, line),
    }
    for mode, r in cases.items():
        got = E.welch_psd(x, win, hop, M, detrend=mode, sided=E.SIDED_RAW, scale=1.0)
        np.testing.assert_allclose(got, r, rtol=2e-4, atol=1e-6 * r.max(), err_msg=str(mode))


def test_long_welch_csd_multichannel(P):
    """reference signal against 3 channels, 12 000-point segments: Pxx, Pyy, Pxy = Y conj(X) (fft_analysis.py:393)"""
    E = P.engine
    rng = np.random.default_rng(11)
    nfft, hop, M, nch = 12000, 6000, 5, 3
    n = (M - 1) * hop + nfft + 5
    x = (rng.standard_normal(n) + 1.0).astype(np.float32)
    y = np.stack([0.5 * np.roll(x, 3 + c) + 0.3 * rng.standard_normal(n).astype(np.float32) - c for c in range(nch)]).astype(np.float32)
    win = O.windows("Hanning", nwins=nfft)
    pxx, pyy, pxy = E.welch_csd(x, y, win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0)
    xd = x.astype(np.float64) - x.astype(np.float64).mean()
    yd = y.astype(np.float64) - y.astype(np.float64).mean(axis=1, keepdims=True)
    nny = nfft // 2
    rxx = np.zeros(nfft)
    ryy = np.zeros((nch, nfft))
    rxy = np.zeros((nch, nfft), dtype=np.complex128)
    for g in range(M):
        X = np.fft.fft(win * xd[g * hop:g * hop + nfft])
        Y = np.fft.fft(win[None, :] * yd[:, g * hop:g * hop + nfft], axis=-1)
        rxx += np.abs(X) ** 2
        ryy += np.abs(Y) ** 2
        rxy += Y * np.conj(X)[None, :]

    def cut(P_):
        P_ = P_[..., :nny].copy() / M
        P_[..., 1:-1] *= 2
        return P_
    np.testing.assert_allclose(pxx, cut(rxx), rtol=2e-4, atol=1e-6 * cut(rxx).max())
    np.testing.assert_allclose(pyy, cut(ryy), rtol=2e-4, atol=1e-6 * cut(ryy).max())
    np.testing.assert_allclose(pxy, cut(rxy), rtol=2e-4, atol=2e-6 * np.abs(cut(rxy)).max())
    # complex two-sided input through the same path
    xc = (x[:n] + 1j * np.roll(x, 7)[:n]).astype(np.complex64)
    yc = (y[:1] * (1 + 0.5j)).astype(np.complex64)
    pxx2, pyy2, pxy2 = E.welch_csd(xc, yc, win, hop, M, detrend=False, sided=E.SIDED_TWO, scale=1.0)
    acc = np.zeros(nfft, dtype=np.complex128)
    for g in range(M):
        X = np.fft.fft(win * xc[g * hop:g * hop + nfft].astype(np.complex128))
        Y = np.fft.fft(win * yc[0, g * hop:g * hop + nfft].astype(np.complex128))
        acc += Y * np.conj(X)
    np.testing.assert_allclose(pxy2[0], np.fft.fftshift(acc) / M, rtol=2e-4, atol=2e-6 * np.abs(acc).max() / M)


def test_long_stft_layouts_and_pseg(P):
    """power output, bin-major layout and the per-frame time-domain power (fft_analysis.py:2174) on 8 500-point frames"""
    E = P.engine
    rng = np.random.default_rng(3)
    nfft, hop, M = 8500, 2125, 11
    n = (M - 1) * hop + nfft
    x = (rng.standard_normal(n) + 0.25).astype(np.float32)
    win = O.windows("Hanning", nwins=nfft)
    xd = x.astype(np.float64) - x.astype(np.float64).mean()
    frames = np.stack([win * xd[g * hop:g * hop + nfft] for g in range(M)])
    X = np.fft.fft(frames, axis=-1)
    out, pseg = E.stft_frames(x, win, hop, M, detrend=True, sided=E.SIDED_RAW, amp_scale=0.5, want_pseg=True)
    assert out.shape == (M, nfft)
    assert np.max(np.abs(out - 0.5 * X)) <= 1e-4 * np.abs(0.5 * X).max()
    np.testing.assert_allclose(pseg, np.trapezoid(frames ** 2, axis=-1), rtol=1e-4)
    pw, _ = E.stft_frames(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, amp_scale=2.0, power=True, bin_major=True)
    assert pw.shape == (nfft, M) and pw.dtype == np.float32
    ref = np.fft.fftshift(2.0 * np.abs(X) ** 2, axes=-1).T
    np.testing.assert_allclose(pw, ref, rtol=2e-4, atol=1e-6 * ref.max())
    one, _ = E.stft_frames(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, amp_scale=1.0)
    exp = X[:, :nfft // 2].copy()
    exp[:, 1:-1] *= np.sqrt(2.0)
    assert np.max(np.abs(one - exp)) <= 1e-4 * np.abs(exp).max()


def test_long_cog(P):
    """centre of gravity per frame (Doppler.py:43-58) with 20 000-point windows"""
    E = P.engine
    rng = np.random.default_rng(8)
    nfft, hop, M, fs = 20000, 10000, 4, 1.0e3
    n = (M - 1) * hop + nfft
    k = np.arange(n)
    x = (np.exp(2j * np.pi * (0.05 + 0.02 * k / n) * k) + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    win = np.ones(nfft)
    got = E.stft_cog(x, win, hop, M, fs)
    f = np.fft.fftfreq(nfft, 1.0 / fs)
    ref = []
    for g in range(M):
        p = np.abs(np.fft.fft(x[g * hop:g * hop + nfft].astype(np.complex128))) ** 2
        ref.append(np.sum(f * p) / np.sum(p))
    np.testing.assert_allclose(got, np.array(ref), rtol=2e-4, atol=1e-4)


def test_long_many_midsize_frames(P):
    """many frames of a mid-size long length go through each launch of the multi-pass transform as one batch, in slices"""
    E = P.engine
    rng = np.random.default_rng(21)
    nfft, hop, M = 16384, 2048, 700           # 700 x 16384 x 8 B = 92 MB of spectra: 2 chunks of the 192 MiB budget? no: 1; see below
    n = (M - 1) * hop + nfft
    x = rng.standard_normal(n).astype(np.float32)
    win = O.windows("Hanning", nwins=nfft)
    got = E.welch_psd(x, win, hop, M, detrend=False, sided=E.SIDED_RAW, scale=1.0)
    ref = np.zeros(nfft)
    xd = x.astype(np.float64)
    for g in range(M):
        ref += np.abs(np.fft.fft(win * xd[g * hop:g * hop + nfft])) ** 2
    ref /= M
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-6 * ref.max())
    # 4 000 frames of 5 000 points (chirp-z on 16 384): several chunks and several FFT slices
    nfft, hop, M = 5000, 1000, 4000
    n = (M - 1) * hop + nfft
    x = rng.standard_normal(n).astype(np.float32)
    win = O.windows("Hamming", nwins=nfft)
    got = E.welch_psd(x, win, hop, M, detrend=False, sided=E.SIDED_RAW, scale=1.0)
    xd = x.astype(np.float64)
    idx = np.arange(nfft)[None, :]
    ref = np.zeros(nfft)
    for g0 in range(0, M, 500):
        st = (np.arange(g0, min(M, g0 + 500)) * hop)[:, None]
        ref += (np.abs(np.fft.fft(win[None, :] * xd[st + idx], axis=-1)) ** 2).sum(axis=0)
    ref /= M
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-6 * ref.max())


def test_long_csd_matrix(P):
    """full CSD matrix with 9 000-point segments (not a power of two): the spectra come from the long path, the
    contraction is the same matrix-core kernel as at cfg5"""
    E = P.engine
    rng = np.random.default_rng(17)
    nch, nfft, hop, M = 5, 9000, 4500, 6
    n = (M - 1) * hop + nfft + 3
    common = rng.standard_normal(n)
    x = np.stack([(0.3 + 0.1 * c) * np.roll(common, 2 * c) + rng.standard_normal(n) + 0.2 * c for c in range(nch)]).astype(np.float32)
    win = O.windows("Hanning", nwins=nfft)
    got = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    ref = O.csd_matrix(x.astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    assert got.shape == ref.shape == (nfft // 2 + 1, nch, nch)
    assert np.max(np.abs(got - ref)) <= 2e-4 * np.abs(ref).max()
    # Hermitian, real diagonal = the long-path Welch PSD of each channel
    assert np.max(np.abs(got - np.conj(np.swapaxes(got, 1, 2)))) <= 1e-6 * np.abs(ref).max()
    p0 = E.welch_psd(x[0], win, hop, M, detrend=True, sided=E.SIDED_RAW, scale=1.0)[:nfft // 2 + 1]
    np.testing.assert_allclose(got[:, 0, 0].real, p0, rtol=2e-4, atol=1e-6 * p0.max())


def test_long_paths_on_device_tensors(P):
    """mem=1: torch CUDA tensors in, torch tensors out, for the long-segment Welch PSD / CSD / STFT"""
    import torch
    E = P.engine
    rng = np.random.default_rng(23)
    nfft, hop, M = 20000, 10000, 5
    n = (M - 1) * hop + nfft
    x = (rng.standard_normal(n) + 1.0).astype(np.float32)
    y = (0.5 * np.roll(x, 4) + 0.2 * rng.standard_normal(n)).astype(np.float32)
    win = O.windows("Hamming", nwins=nfft)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    p_t = E.welch_psd(xt, win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0)
    p_h = E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0)
    assert p_t.is_cuda and p_t.dtype == torch.float64
    np.testing.assert_allclose(p_t.cpu().numpy(), p_h, rtol=1e-12)
    a_t = E.welch_csd(xt, yt[None, :], win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0)
    a_h = E.welch_csd(x, y[None, :], win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0)
    for t, h in zip(a_t, a_h):
        assert t.is_cuda
        np.testing.assert_allclose(t.cpu().numpy(), h, rtol=1e-12, atol=1e-30)
    s_t, _ = E.stft_frames(xt, win, hop, M, detrend=True, sided=E.SIDED_ONE, amp_scale=1.0)
    s_h, _ = E.stft_frames(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, amp_scale=1.0)
    assert s_t.is_cuda and s_t.shape == (M, nfft // 2)
    np.testing.assert_allclose(s_t.cpu().numpy(), s_h, rtol=0, atol=0)


# ---- N2: the multi-channel driver (HeatPulse_Funcs._PWELCH_chloop) as ONE call -----------------------------------------
@pytest.mark.parametrize("Navr", [8, 64])
def test_pwelch_chloop_golden(P, Navr):
    """pyfft_amd.pwelch_chloop: reference signal x 16 channels in one fft_pwelch call on the GPU (reference transformed once),
    then integratespectra per harmonic band, all channels at once -- against the fixture that loops the reference's OWN
    fft_pwelch + integratespectra per channel (make_golden_chloop.py; HeatPulse_Funcs.py:576-583, :498-530).  Navr = 8 is the
    reference's default regime (7281-point segments: the long-segment path), Navr = 64 one fused kernel (1008 points,
    Bluestein).  trapz_var / reshapech: stand-ins, PARITY UNPINNED at that boundary (recorded in the generator)."""
    import inputs_chloop
    g = load_golden("chloop")
    tag = "n%d" % Navr
    tt, ref, sig = inputs_chloop.chloop_inputs(int(g["seed"]), int(g["nt"]), float(g["fs"]), int(g["nch"]), float(g["fmod"]))
    hp = P.pwelch_chloop(tt, ref, sig, float(g["fmod"]), harms=list(g["harms"]), fwid=float(g["fwid"]), Navr=Navr,
                         windowoverlap=0.5, windowfunction="hanning")
    assert hp.fftinfo.nwins == int(g["nwins_" + tag]) and hp.Navr == int(g["Navr_" + tag])
    np.testing.assert_allclose(hp.freq, g["freq_" + tag], rtol=1e-12, atol=1e-12)
    assert list(hp._ifk) == list(g["ifk_" + tag]) and hp._ifw == int(g["ifw_" + tag])
    sub = int(g["sub_" + tag])
    close_rel(hp.Pxx, g["Pxx_" + tag], 2e-4, "Pxx")
    close_rel(hp.Pyy[::sub], g["Pyy_" + tag], 2e-4, "Pyy")
    close_rel(hp.Pxy[::sub], g["Pxy_" + tag], 2e-4, "Pxy")
    # band integrals: float32 spectra integrated over a few bins around a strong line
    for k in ("Txy", "Amp", "Txx", "Tnn"):
        ref_k = g[k + "_" + tag]
        assert np.max(np.abs(getattr(hp, k) - ref_k) / np.abs(ref_k)) <= 5e-4, k
    for k in ("Vxy", "varA", "Vxx"):
        ref_k = g[k + "_" + tag]
        assert np.max(np.abs(getattr(hp, k) - ref_k)) <= 2e-3 * np.max(np.abs(ref_k)), k
    np.testing.assert_allclose(hp.Coh, g["Coh_" + tag], rtol=0, atol=2e-4)
    np.testing.assert_allclose(hp.varC, g["varC_" + tag], rtol=2e-2, atol=1e-7)
    np.testing.assert_allclose(hp.varP, g["varP_" + tag], rtol=2e-2, atol=1e-7)
    dphi = np.angle(np.exp(1j * (hp.Phase - g["Phase_" + tag])))
    assert np.max(np.abs(dphi)) <= 1e-3
