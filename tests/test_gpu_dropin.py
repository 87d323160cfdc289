"""GPU parity of the drop-in modules (pyfft_amd.fft_pwelch / fftanal / specgram / stft / hilbert / ccf / fftfilt)
against the golden fixtures captured from the reference (tests/golden/make_golden.py) -- the tests read like the
reference's own smoke functions, with assertions.  Tolerances: float32 device math vs float64 reference."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import cpu_ref as O

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


PW = {
    "pwelch_cfg1": dict(Navr=127, windowoverlap=0.5, windowfunction="Hanning", tb=-2),
    "pwelch_reflect": dict(Navr=15, windowfunction="Hanning", tb=None),
    "pwelch_2ch_twosided": dict(Navr=31, windowfunction="Hamming", onesided=False, detrend_style=0, tb=-2),
    "pwelch_minfreq_linear": dict(minFreq=2.0 * 1.0e4 / 1024.0 * 1.0000001, detrend_style=-1, tb=-2),
    "pwelch_selftest_navr8": dict(Navr=8, windowfunction="hamming", detrend_style=1, tb=-1),
    "pwelch_selftest_minfreq": dict(minFreq=75.0, detrend_style=1, tb=-1),
}


@pytest.mark.parametrize("tag", sorted(PW))
def test_fft_pwelch_golden(P, tag):
    g = load_golden(tag)
    kw = dict(PW[tag])
    tb = kw.pop("tb")
    t, x, y = g["t"], g["x"], g["y"]
    tbounds = None if tb is None else [t[0], t[tb]]
    freq, Pxy, Pxx, Pyy, Cxy, phi, info = P.fft_pwelch(t, x, y, tbounds, plotit=False, verbose=False, **kw)
    for k in ("nwins", "noverlap", "Navr", "nch"):
        assert int(getattr(info, k)) == int(g["info_" + k]), k
    assert list(info.ibnds) == list(g["info_ibnds"])
    for k in ("S1", "S2", "ENBW", "NENBW", "Fs", "minFreq"):
        np.testing.assert_allclose(getattr(info, k), g["info_" + k], rtol=1e-12)
    np.testing.assert_allclose(freq, g["freq"], rtol=1e-12, atol=1e-12)
    assert Pxx.dtype == np.complex128 and Pxy.dtype == np.complex128
    # Welch bins: rtol 2e-4 + atol 1e-6*max (the noise-free self-test signals have > 200 dB of dynamic range
    # in float64; float32 input quantisation sets the floor there -> atol relative to the peak)
    for name, got in (("Pxx", Pxx), ("Pyy", Pyy), ("Pxy", Pxy)):
        ref = g[name]
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-6 * np.max(np.abs(ref)), err_msg=name)
    # epilogue (fft_analysis.py:489-648), every tag.  Quantities that are ratios of spectra (coherence, phase, corrcoef)
    # are compared where the spectra stand above the float32 floor (1e-5 of their peak): the two self-test inputs are
    # noise-free, their off-harmonic bins are rounding noise over rounding noise in the float64 reference itself.
    nb = np.asarray(g["Pxx"]).shape[0]
    pxx_r = np.abs(np.asarray(g["Pxx"])).reshape(nb, -1)
    pyy_r = np.abs(np.asarray(g["Pyy"])).reshape(nb, -1)
    solid = (pxx_r > 1e-5 * pxx_r.max()) & (pyy_r > 1e-5 * pyy_r.max())
    cg, cr = np.asarray(Cxy).reshape(nb, -1), np.asarray(g["Cxy"]).reshape(nb, -1)
    assert np.max(np.abs(cg - cr)[solid]) <= 2e-3, "Cxy"
    m = solid & (np.abs(cr) > 0.5)
    pg, pr = np.asarray(phi).reshape(nb, -1), np.asarray(g["phi_xy"]).reshape(nb, -1)
    assert np.max(np.abs(np.angle(np.exp(1j * (pg[m] - pr[m]))))) < 2e-3, "phi_xy"
    for k in ("Lxx", "Lyy", "Lxy", "Rxx", "Ryy", "Rxy", "lags", "varPxx", "Ex", "Ey"):
        close_rel(np.atleast_1d(getattr(info, k)), g["info_" + k], 2e-3, k)
    cc_r = np.asarray(g["info_corrcoef"])
    close_rel(np.atleast_1d(info.corrcoef), cc_r, 2e-3, "corrcoef")


def test_fft_pwelch_segments(P):
    g = load_golden("pwelch_cfg1")
    t, x, y = g["t"], g["x"], g["y"]
    out = P.fft_pwelch(t, x, y, [t[0], t[-2]], Navr=127, windowoverlap=0.5, windowfunction="Hanning", plotit=False,
                       segments=True)
    info = out[-1]
    close_rel(info.Xfft_seg[:2], g["Xfft_seg_head"], 1e-4, "Xfft_seg")
    close_rel(info.Pxy_seg[:, :2], g["Pxy_seg_head"], 2e-4, "Pxy_seg")


@pytest.mark.parametrize("tag", ["c64_2e16_n4096", "c64_2e14_n1024"])
def test_fftanal_class_complex(P, tag):
    g = load_golden("welch_class_" + tag)
    x = g["x"]
    t = np.arange(x.size, dtype=np.float64)
    ft = P.fftanal(t, x, None, tbounds=[t[0], t[-1]], nwins=int(g["nwins"]), windowfunction="Hanning",
                   windowoverlap=0.5, verbose=False)
    ft.pwelch()
    assert ft.Navr == int(g["Navr"]) and ft.noverlap == int(g["noverlap"]) and not ft.onesided
    for k in ("S1", "S2", "ENBW", "NENBW", "Fs"):
        np.testing.assert_allclose(getattr(ft, k), g[k], rtol=1e-12)
    np.testing.assert_allclose(ft.freq, g["freq"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ft.tseg, g["tseg"], rtol=1e-9)
    np.testing.assert_allclose(ft.Pxx, g["Pxx"], rtol=2e-4, atol=1e-6 * np.abs(g["Pxx"]).max())
    np.testing.assert_allclose(ft.varPxx, g["varPxx"], rtol=5e-4, atol=1e-6 * np.abs(g["varPxx"]).max())
    close_rel(ft.Xseg[:4], g["Xseg_head"], 1e-4, "Xseg")
    close_rel(ft.Xseg[-2:], g["Xseg_tail"], 1e-4, "Xseg tail")
    close_rel(ft.Pxx_seg[:2], g["Pxx_seg_head"], 2e-4, "Pxx_seg")
    close_rel(ft.Xfft, g["Xfft"], 2e-4 * np.abs(g["Xseg_head"]).max() / np.abs(g["Xfft"]).max(), "Xfft")
    np.testing.assert_allclose(ft.Xpow, g["Xpow"], rtol=1e-4)
    # averaged spectra only (no [Navr, nfft] arrays): same Pxx
    ft2 = P.fftanal(t, x, None, tbounds=[t[0], t[-1]], nwins=int(g["nwins"]), windowfunction="Hanning",
                    windowoverlap=0.5, verbose=False, segments=False)
    ft2.pwelch()
    np.testing.assert_allclose(ft2.Pxx, ft.Pxx, rtol=1e-12)
    assert not hasattr(ft2, "Xseg")


@pytest.mark.parametrize("wname", ["Hamming", "SFT3F"])
def test_fftanal_class_real_xy(P, wname):
    g = load_golden("welch_class_real_" + wname)
    t, x, y = g["t"], g["x"], g["y"]
    ft = P.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=31, windowfunction=wname, verbose=False)
    ft.pwelch()
    assert ft.onesided and ft.nwins == int(g["nwins"]) and ft.Navr == int(g["Navr"])
    for k in ("Pxx", "Pyy", "Pxy"):
        np.testing.assert_allclose(getattr(ft, k), g[k], rtol=2e-4, atol=1e-6 * np.abs(g[k]).max(), err_msg=k)
    close_rel(ft.Xseg[:3], g["Xseg_head"], 1e-4, "Xseg")
    close_rel(ft.Yseg[:3], g["Yseg_head"], 1e-4, "Yseg")
    close_rel(ft.Lxx_seg[:2], g["Lxx_seg_head"], 1e-4, "Lxx_seg")
    np.testing.assert_allclose(ft.Xpow, g["Xpow"], rtol=2e-4)
    assert ft.Cxy.shape == ft.Pxy.shape


def test_stft_dropin(P):
    g = load_golden("stft_f32_n2048_ov75")
    st = P.stft(g["t"], g["x"], tper=2048.5, returnclass=True, windowfunction="Hanning", windowoverlap=0.75)
    assert st.nwins == 2048 and st.noverlap == 1536 and st.Navr == int(g["Navr"])
    np.testing.assert_allclose(st.freq, g["freq"], rtol=1e-12)
    np.testing.assert_allclose(st.tseg, g["tseg"], rtol=1e-9)
    sc = np.abs(g["Xseg_head"]).max()
    M = st.Navr
    assert np.max(np.abs(st.Xseg[:3] - g["Xseg_head"])) <= 1e-4 * sc
    assert np.max(np.abs(st.Xseg[M // 2:M // 2 + 2] - g["Xseg_mid"])) <= 1e-4 * sc
    assert np.max(np.abs(st.Xseg[-2:] - g["Xseg_tail"])) <= 1e-4 * sc
    np.testing.assert_allclose(st.Pxx, g["Pxx"], rtol=2e-4, atol=1e-6 * np.abs(g["Pxx"]).max())
    g2 = load_golden("stft_tuple_n256")
    twin, freq, Xseg = P.stft(g2["t"], g2["x"], tper=256.5, returnclass=False, windowfunction="Hamming")
    np.testing.assert_allclose(twin, g2["twin"], rtol=1e-12)
    np.testing.assert_allclose(freq, g2["freq"], rtol=1e-12)
    assert np.max(np.abs(Xseg - g2["Xseg"])) <= 1e-4 * np.abs(g2["Xseg"]).max()


def test_specgram_dropin(P):
    g = load_golden("specgram")
    time1, f1, sp1 = P.specgram(g["t"], g["s"], wl=512, hanning=True, overlap=True)
    np.testing.assert_allclose(time1, g["time1"], rtol=1e-12)
    np.testing.assert_allclose(f1, g["f1"], rtol=1e-12)
    np.testing.assert_allclose(sp1, g["sp1"], rtol=2e-4, atol=1e-6 * g["sp1"].max())
    time2, f2, sp2 = P.specgram(g["t"], g["s"], wl=500, hanning=False, overlap=False)     # wl not a power of two
    np.testing.assert_allclose(time2, g["time2"], rtol=1e-12)
    np.testing.assert_allclose(sp2, g["sp2"], rtol=2e-4, atol=1e-6 * g["sp2"].max())
    # windowAverage: block mean of consecutive frames
    ta, fa, spa = P.specgram(g["t"], g["s"], wl=500, hanning=False, windowAverage=3)
    nA = g["sp2"].shape[1] // 3
    np.testing.assert_allclose(spa, g["sp2"][:, :nA * 3].reshape(500, nA, 3).mean(axis=2), rtol=2e-4,
                               atol=1e-6 * g["sp2"].max())


def test_hilbert_dropin(P):
    g = load_golden("hilbert")
    for u, z in (("yk", "zk"), ("u_even", "z_even"), ("u_odd", "z_odd"), ("u_2d", "z_2d")):
        got = P.hilbert(g[u])
        assert got.shape == g[z].shape and got.dtype == g[z].dtype
        close_rel(got, g[z], 1e-4, z)
    close_rel(P.hilbert(g["u_2d"], axes=0), g["z_2d_ax0"], 1e-4, "axis 0")
    z32 = P.hilbert(g["u_f32"])
    assert z32.dtype == np.complex64
    close_rel(z32, g["z_f32"], 1e-4, "f32")
    close_rel(P.hilbert(g["u_even"][:1000], nfft=1024), g["z_nfft"], 1e-4, "nfft")
    close_rel(P.hilbert_1d(g["yk"]), g["zk1d"], 1e-5, "1d")
    close_rel(P.hilbert_1d(g["u_odd"]), g["z_odd_1d"], 1e-4, "1d odd")


def test_ccf_dropin(P):
    g = load_golden("ccf")
    tau, co = P.ccf(g["x1"], g["x2"], float(g["fs"]))
    np.testing.assert_allclose(tau, g["tau"], rtol=1e-12)
    close_rel(co, g["co"], 1e-4, "co")
    tau2, co2 = P.ccf(g["x3"], g["x4"], 1.0)
    close_rel(co2, g["co2"], 1e-4, "co2")
    # ccf.py:139-148: maximum near -phi/(2 pi f) = -138.9 us
    assert abs(tau[np.argmax(co)] - (-138.9e-6)) < 125e-6


def test_fftfilt_and_notch(P):
    import scipy.signal as ss
    rng = np.random.default_rng(4)
    n = 1 << 18
    k = np.arange(n)
    x = (rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 0.06 * k)).astype(np.float32)
    h = ss.firwin(513, 0.2)
    y = P.fftfilt(h, x)
    ref = ss.lfilter(h.astype(np.float32).astype(np.float64), 1.0, x.astype(np.float64))
    assert y.dtype == np.float32
    close_rel(y, ref, 1e-4, "fir")
    # notch at w0 = 0.12 (f = 0.06 cycles/sample): designed biquad == scipy's, applied as a 513-tap FIR
    b, a = P.iirnotch(0.12, 5.0)
    bs, as_ = ss.iirnotch(0.12, 5.0)
    np.testing.assert_allclose(b, bs, rtol=1e-14); np.testing.assert_allclose(a, as_, rtol=1e-14)
    yn = P.apply_notch(x, 0.12, 5.0, ntaps=513)
    exact = ss.lfilter(b, a, x.astype(np.float64))
    close_rel(yn, exact, 2e-4, "notch vs exact recursion")
    # default: the recurrence itself on the GPU (float64 state): float32 output rounding is all that differs
    close_rel(P.apply_notch(x, 0.12, 5.0), exact, 2e-7, "exact notch")
    close_rel(P.apply_notch(x, 0.12, 5.0, ntaps="auto"), exact, 1e-4, "auto-sized FIR notch")
    # the tone is gone: power at f = 0.06 drops by > 40 dB
    def tone_power(v):
        return np.abs(np.sum(v[4096:] * np.exp(-2j * np.pi * 0.06 * k[4096:]))) ** 2
    assert tone_power(yn) < 1e-4 * tone_power(x)
    # a NARROW notch (w0 = 0.01, Q = 30 -> pole radius 0.99948, |p|^513 = 0.77): the 513-tap FIR is refused, the
    # exact recurrence matches scipy.signal.lfilter on a float64 host recursion; ragged length (not a multiple of the tile)
    for (w0, Q, ftype, nn) in [(0.01, 30.0, "notch", n - 4321), (0.01, 200.0, "notch", n), (0.3, 50.0, "peak", 100003),
                               (0.5, 0.7, "notch", 777), (0.02, 30.0, "peak", 5)]:
        bq, aq = (P.iirnotch if ftype == "notch" else P.iirpeak)(w0, Q)
        xe = x[:nn]
        ex = ss.lfilter(bq, aq, xe.astype(np.float64))
        got = P.apply_notch(xe, w0, Q, ftype=ftype)
        assert got.dtype == np.float32 and got.shape == xe.shape
        close_rel(got, ex, 3e-7, "exact %s w0=%g Q=%g" % (ftype, w0, Q))
    with pytest.raises(ValueError, match="tail"):
        P.apply_notch(x, 0.01, 30.0, ntaps=513)
    with pytest.raises(ValueError, match="exact recurrence"):
        P.apply_notch(x, 0.01, 30.0, ntaps="auto")
    # general second-order sections through the engine entry (a[0] != 1, missing trailing coefficients)
    E = P.engine
    close_rel(E.biquad_filter([0.5, 0.25], [2.0, -1.0, 0.4], x), ss.lfilter([0.5, 0.25], [2.0, -1.0, 0.4], x.astype(np.float64)), 3e-7, "biquad")
    with pytest.raises(ValueError):
        E.biquad_filter([1.0], [0.0, 1.0], x)
    # smooth() of the reference == np.convolve(w/sum, reflect-padded, 'valid')
    from pyfft_amd.filters import smooth
    s = rng.standard_normal(3000)
    w = np.hanning(11)
    pad = np.r_[s[10:0:-1], s, s[-2:-12:-1]]
    close_rel(smooth(s, 11, "hanning"), np.convolve(w / w.sum(), pad, mode="valid"), 1e-5, "smooth")


def test_mlab_wrappers_psd_csd_coh():
    """psd / csd / coh / coh2 (the reference's matplotlib.mlab wrappers) against the fixture the reference produced;
    per-segment mean detrend, symmetric Hann, arbitrary hop, even and non-power-of-two nfft"""
    import pyfft_amd as P
    g = load_golden("mlab_wrappers")
    x, y, fs = g["x"], g["y"], float(g["fs"])
    p, f = P.psd(x, fs)
    np.testing.assert_allclose(f, g["psd_f"], rtol=1e-12)
    np.testing.assert_allclose(p, g["psd_p"], rtol=2e-4, atol=1e-6 * g["psd_p"].max())
    p, f = P.psd(x, fs, nfft=500, fmin=20.0, fmax=300.0, detrend="mean", ov=0.5)
    np.testing.assert_allclose(f, g["psd2_f"], rtol=1e-12)
    np.testing.assert_allclose(p, g["psd2_p"], rtol=2e-4, atol=1e-6 * g["psd2_p"].max())
    p, f = P.csd(x, y, fs)
    np.testing.assert_allclose(f, g["csd_f"], rtol=1e-12)
    assert np.max(np.abs(p - g["csd_p"])) <= 3e-4 * np.abs(g["csd_p"]).max()
    p, f = P.csd(x, y, fs, nfft=1024, fmin=None, fmax=None, detrend="mean", ov=0.75)
    assert np.max(np.abs(p - g["csd2_p"])) <= 3e-4 * np.abs(g["csd2_p"]).max()
    p, f = P.psd(x, fs, nfft=1024, detrend="linear", ov=0.5)
    np.testing.assert_allclose(p, g["psd3_p"], rtol=2e-4, atol=1e-6 * g["psd3_p"].max())
    p, f = P.csd(x, y, fs, nfft=600, fmin=None, fmax=None, detrend="linear", ov=0.25)
    assert np.max(np.abs(p - g["csd3_p"])) <= 3e-4 * np.abs(g["csd3_p"]).max()
    c, f = P.coh(x, y, fs)
    np.testing.assert_allclose(f, g["coh_f"], rtol=1e-12)
    np.testing.assert_allclose(c, g["coh_c"], rtol=2e-3, atol=2e-4)
    c, f = P.coh(x, y, fs, nfft=512, fmin=10.0, fmax=400.0, detrend="none", ov=0.5)
    np.testing.assert_allclose(c, g["coh2_c"], rtol=2e-3, atol=2e-4)
    r = P.coh2(x, y, fs)
    ref = O.mlab_coh2_wrapper(x, y, fs)                      # parity unpinned: the reference's coh2 raises (float noverlap)
    np.testing.assert_allclose(r["f"], ref["f"], rtol=1e-12)
    np.testing.assert_allclose(r["coh"], ref["coh"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(r["PS"], ref["PS"], rtol=2e-4, atol=1e-6 * ref["PS"].max())


@pytest.mark.parametrize("onesided", [True, False])
def test_fftanal_crosscorr_against_reference_fixture(P, onesided):
    """fftanal.crosscorr_stft / crosscorr (fft_analysis.py:1840-1920) against the fixture the reference produced
    (tests/golden/make_golden_xcorr.py); inverse FFTs of length 1333 (not a power of two) on the device"""
    g = load_golden("crosscorr_class")
    rng = np.random.default_rng(int(g["seed"]))
    n, fs = int(g["n"]), float(g["fs"])
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 50 * t) + 0.3 * rng.standard_normal(n)
    y = np.sin(2 * np.pi * 50 * t + 0.7) + 0.3 * rng.standard_normal(n)
    tag = "one" if onesided else "two"
    ft = P.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=8, windowoverlap=0.5, windowfunction="hanning",
                   onesided=onesided, plotit=False, verbose=False, segments=True)
    ft.pwelch()
    assert ft.nwins == int(g["nwins_" + tag]) and ft.Nnyquist == int(g["Nnyquist_" + tag])
    ft.crosscorr_stft()
    for k in ("Rxx_seg", "Ryy_seg", "Rxy_seg", "Ex_seg", "Ey_seg"):
        close_rel(np.asarray(getattr(ft, k)), g[k + "_" + tag], 2e-4, k)
    ft.crosscorr()
    for k in ("Rxx", "Ryy", "Rxy"):
        close_rel(np.asarray(getattr(ft, k)).ravel(), g[k + "_" + tag].ravel(), 2e-4, k)
    close_rel(np.atleast_1d(ft.Ex).ravel(), g["Ex_" + tag].ravel(), 2e-4, "Ex")
    # corrcoef[_seg]: the reference's own line raises (self.nch unset, fixture err_* = AttributeError); the drop-in
    # evaluates the formula it states, Rxy / sqrt(Ex Ey)
    assert str(g["err_stft_" + tag]) == "AttributeError"
    ref_cc = g["Rxy_seg_" + tag] / np.sqrt(g["Ex_seg_" + tag] * g["Ey_seg_" + tag])[:, None]
    close_rel(ft.corrcoef_seg, ref_cc, 5e-4, "corrcoef_seg")


def test_fftanal_crosscorr_stft_and_getters():
    """crosscorr_stft (fft_analysis.py:1880-1920) restated with numpy's irfft/ifft, and the reference-named getters"""
    import pyfft_amd as P
    rng = np.random.default_rng(4)
    n, fs = 6000, 1.0e3
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 50 * t) + 0.3 * rng.standard_normal(n)
    y = np.sin(2 * np.pi * 50 * t + 0.7) + 0.3 * rng.standard_normal(n)
    for onesided in (True, False):
        ft = P.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=8, windowoverlap=0.5, windowfunction="hanning",
                       onesided=onesided, plotit=False, verbose=False, segments=True)
        ft.pwelch()
        ft.crosscorr_stft()
        nfft = ft.nwins
        for name in ("Pxx_seg", "Pxy_seg"):
            tmp = np.array(getattr(ft, name), dtype=np.complex128)
            if onesided:
                tmp[..., 1:-1] *= 0.5
                if nfft % 2:
                    tmp[..., -1] *= 0.5
                ref = np.sqrt(nfft) * np.fft.irfft(tmp, n=nfft, axis=-1)
            else:
                ref = np.sqrt(nfft) * np.fft.ifft(np.fft.ifftshift(tmp, axes=-1), n=nfft, axis=-1)
            ref = np.fft.fftshift(ref, axes=-1)
            got = getattr(ft, "R" + name[1:])
            assert got.shape == ref.shape
            assert np.max(np.abs(got - ref)) <= 5e-6 * np.abs(ref).max()
        assert ft.corrcoef_seg.shape == ft.Rxy_seg.shape and ft.lags.shape == (nfft,)
        assert ft.getNnyquist() == P.fftanal._getNnyquist(nfft)
        assert ft.getNoverlap() == P.fftanal._getNoverlap(nfft, 0.5) and ft.getNavr() == P.fftanal._getNavr(ft.nsig, nfft, ft.noverlap)


def test_fftanal_static_fft_win_matches_instance():
    """fftanal._fft_win (the static multi-channel twin, fft_analysis.py:2554-2640) == fft_win per channel"""
    import pyfft_amd as P
    rng = np.random.default_rng(5)
    n, fs = 5000, 2.0e3
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 70 * t) + 0.2 * rng.standard_normal(n) + 0.5
    y = np.cos(2 * np.pi * 70 * t) + 0.2 * rng.standard_normal(n) - 0.1
    ft = P.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=6, windowoverlap=0.5, windowfunction="hanning", onesided=True,
                   plotit=False, verbose=False)
    i0, i1 = ft.ibounds
    tt, freq, X1, p1 = ft.fft_win(x[i0:i1], t[i0:i1])
    _, _, Y1, q1 = ft.fft_win(y[i0:i1], t[i0:i1])
    kw = dict(tvec=t[i0:i1], onesided=True, win=ft.win, nwins=ft.nwins, Navr=ft.Navr, noverlap=ft.noverlap,
              Nnyquist=ft.Nnyquist, detrend_style=ft.detrendstyle, S1=ft.S1, S2=ft.S2, ENBW=ft.ENBW, Fs=ft.Fs)
    tt2, freq2, X2, p2 = P.fftanal._fft_win(np.stack([x[i0:i1], y[i0:i1]], axis=1), **kw)
    assert X2.shape == (2,) + X1.shape and p2.shape == (2,) + p1.shape
    np.testing.assert_allclose(tt2, tt)
    np.testing.assert_allclose(freq2, freq)
    np.testing.assert_allclose(X2[0], X1, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(X2[1], Y1, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(p2[1], q1, rtol=1e-6)


def test_fft_win_detrendwin_linear():
    """fft_win(detrendwin=True) with the linear style (detrend_style < 0): every window's own least-squares line removed"""
    import pyfft_amd as P
    rng = np.random.default_rng(8)
    n, fs = 6000, 1.0e3
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 33 * t) + 0.2 * rng.standard_normal(n) + 1.0 + 5.0 * t ** 2
    ft = P.fftanal(t, x, tbounds=[t[0], t[-1]], Navr=7, windowoverlap=0.5, windowfunction="hanning", onesided=False,
                   detrend=-1, plotit=False, verbose=False)
    i0, i1 = ft.ibounds
    tt, freq, X, pseg = ft.fft_win(x[i0:i1], t[i0:i1], detrendwin=True)
    xs = x[i0:i1]
    hop = ft.nwins - ft.noverlap
    k = np.arange(ft.nwins)
    ref = []
    for g in range(ft.Navr):
        seg = xs[g * hop: g * hop + ft.nwins]
        seg = seg - np.polyval(np.polyfit(k, seg, 1), k)
        ref.append(np.fft.fftshift(np.fft.fft(ft.win * seg)) / ft.S1 / np.sqrt(ft.ENBW))
    ref = np.array(ref)
    assert np.max(np.abs(X - ref)) <= 5e-5 * np.abs(ref).max()


def test_fft_win_detrendwin_mean():
    """fft_win(detrendwin=True) (fft_analysis.py:2171): every window's own mean removed instead of the global mean"""
    import pyfft_amd as P
    rng = np.random.default_rng(6)
    n, fs = 7000, 1.0e3
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 40 * t) + 0.2 * rng.standard_normal(n) + 2.0 + 3.0 * t      # strong drift
    for onesided in (True, False):
        ft = P.fftanal(t, x, tbounds=[t[0], t[-1]], Navr=9, windowoverlap=0.5, windowfunction="hanning", onesided=onesided,
                       plotit=False, verbose=False)
        i0, i1 = ft.ibounds
        tt, freq, X, pseg = ft.fft_win(x[i0:i1], t[i0:i1], detrendwin=True)
        xs = x[i0:i1]
        hop = ft.nwins - ft.noverlap
        ref = []
        for g in range(ft.Navr):
            seg = xs[g * hop: g * hop + ft.nwins]
            F = np.fft.fft(ft.win * (seg - seg.mean()))
            if onesided:
                F = F[:ft.Nnyquist].copy()
                F[1:-1] *= np.sqrt(2)
                if ft.nwins % 2:
                    F[-1] *= np.sqrt(2)
            else:
                F = np.fft.fftshift(F)
            ref.append(F / ft.S1 / np.sqrt(ft.ENBW))
        ref = np.array(ref)
        assert X.shape == ref.shape
        assert np.max(np.abs(X - ref)) <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("tag,kw", [("pwelch_usemlab_onesided", dict(Navr=15, windowoverlap=0.5, windowfunction="Hamming")),
                                    ("pwelch_usemlab_twosided_linear", dict(Navr=9, windowoverlap=0.5, windowfunction="Hanning",
                                                                            onesided=False, detrend_style=-1))])
def test_fft_pwelch_usemlab_on_device(tag, kw):
    """fft_pwelch(useMLAB=True): the reference's matplotlib.mlab.csd branch, per-segment detrend, on the device"""
    import pyfft_amd as P
    g = load_golden(tag)
    t, x, y = g["t"], g["x"], g["y"]
    freq, Pxy, Pxx, Pyy, Cxy, phi, info = P.fft_pwelch(t, x, y, [t[0], t[-2]], useMLAB=True, plotit=False, verbose=False, **kw)
    np.testing.assert_allclose(freq, g["freq"], rtol=1e-12, atol=1e-9)
    assert Pxx.shape == g["Pxx"].shape and Pyy.shape == g["Pyy"].shape and Pxy.shape == g["Pxy"].shape
    assert np.max(np.abs(Pxx - g["Pxx"])) <= 3e-4 * np.abs(g["Pxx"]).max()
    assert np.max(np.abs(Pyy - g["Pyy"])) <= 3e-4 * np.abs(g["Pyy"]).max()
    assert np.max(np.abs(Pxy - g["Pxy"])) <= 3e-4 * np.abs(g["Pxy"]).max()
    np.testing.assert_allclose(Cxy, g["Cxy"], rtol=5e-3, atol=5e-4)
    assert np.max(np.abs(info.Rxy - g["info_Rxy"])) <= 5e-4 * np.abs(g["info_Rxy"]).max()


def test_fftanal_fftpwelch_usemlab():
    """the class path hands useMLAB through to fft_pwelch (fft_analysis.py:1796-1803)"""
    import pyfft_amd as P
    g = load_golden("pwelch_usemlab_onesided")
    t, x, y = g["t"], g["x"], g["y"]
    ft = P.fftanal(t, x, y, tbounds=[t[0], t[-2]], Navr=15, windowoverlap=0.5, windowfunction="Hamming", useMLAB=True,
                   plotit=False, verbose=False)
    ft.fftpwelch()
    assert np.max(np.abs(ft.Pxx - g["Pxx"])) <= 3e-4 * np.abs(g["Pxx"]).max()
    assert np.max(np.abs(ft.Pxy - g["Pxy"])) <= 3e-4 * np.abs(g["Pxy"]).max()


def test_hilbert_complex_input():
    """complex input (the reference's code path has no real-only restriction): linearity of FFT -> mask -> IFFT"""
    import pyfft_amd as P
    rng = np.random.default_rng(9)
    u = rng.standard_normal((3, 700)) + 1j * rng.standard_normal((3, 700))
    z = P.hilbert(u)
    ref = O.hilbert(u)
    assert z.shape == ref.shape and np.max(np.abs(z - ref)) <= 2e-5 * np.abs(ref).max()
    z1 = P.hilbert_1d(u[0])
    assert np.max(np.abs(z1 - O.hilbert_1d(u[0]))) <= 2e-5 * np.abs(ref).max()


# ---- N3: Doppler centre of gravity (Doppler.py:43-81) -----------------------------------------------------------------
def test_doppler_cog_whole_vector(P):
    """pyfft_amd.cog against values the reference's Doppler.cog produced (tests/golden/make_golden_doppler.py).  Tolerance:
    float32 device transform; the real-input centre is a cancellation of +f against -f, so it is absolute in fs."""
    g = load_golden("doppler_cog")
    fs, z, r = float(g["fs"]), g["z"], g["r"]
    for tag in ("4096", "1000", "40000", "16384"):          # one-workgroup (pow2, Bluestein) and long transforms
        m = int(tag)
        assert abs(P.cog(z[:m], fs) - float(g["cog_z_" + tag])) <= 2e-6 * fs, tag
        assert abs(P.cog(r[:m], fs) - float(g["cog_r_" + tag])) <= 2e-6 * fs, tag
    # band form: the reference's pairing of band frequencies with the leading bins
    assert abs(P.cog(z[:4096], fs, fmin=50e3, fmax=200e3) - float(g["cog_z_band"])) <= 1e-5 * abs(float(g["cog_z_band"]))
    assert abs(P.cog(z[:1000], fs, fmin=100e3) - float(g["cog_z_band_nofmax"])) <= 1e-5 * abs(float(g["cog_z_band_nofmax"]))
    assert P.cog(z[:1000], fs, fmin=2e6, fmax=3e6) == 0.0


@pytest.mark.parametrize("tag,win,ov", [("512", 512, 0.5), ("200", 200, 0.75)])
def test_doppler_cog_frames(P, tag, win, ov):
    """the window loop of cogspec: every complete window's cog from the fused kernel vs the reference's cog per window"""
    g = load_golden("doppler_cog")
    fs, z, r = float(g["fs"]), g["z"], g["r"]
    t = np.arange(len(z)) / fs
    tc, cg = P.cog_frames(t, z, fs, win=win, ov=ov)
    assert cg.dtype == np.float64 and cg.shape == g["frames_z_" + tag].shape
    np.testing.assert_allclose(tc, O.cog_frames(t, z, fs, win=win, ov=ov)[0], rtol=1e-12)
    np.testing.assert_allclose(cg, g["frames_z_" + tag], rtol=0, atol=2e-6 * fs)
    tc, cg = P.cog_frames(t, r, fs, win=win, ov=ov)
    np.testing.assert_allclose(cg, g["frames_r_" + tag], rtol=0, atol=2e-6 * fs)


def test_doppler_cog_frames_band_window_and_device_tensor(P):
    """build-defined extensions: true band limit, taper, device-resident input; checked against the oracle.  Plus the
    size-independent property at a large size: a pure tone's centre of gravity is the tone."""
    import torch
    from pyfft_amd.windows import windows
    g = load_golden("doppler_cog")
    fs, z = float(g["fs"]), g["z"]
    t = np.arange(len(z)) / fs
    w = windows("hann", nwins=256, verbose=False)
    for kw in (dict(fmin=60e3, fmax=250e3), dict(fmin=None, fmax=120e3), dict(window=w), dict(window=w, fmin=70e3, fmax=300e3)):
        _, ref = O.cog_frames(t, z, fs, win=256, ov=0.5, **kw)
        _, got = P.cog_frames(t, z, fs, win=256, ov=0.5, **kw)
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6 * fs, err_msg=str(kw.keys()))
    zt = torch.from_numpy(z).cuda()
    _, got_t = P.cog_frames(t, zt, fs, win=256, ov=0.5)
    _, got_h = P.cog_frames(t, z, fs, win=256, ov=0.5)
    assert got_t.is_cuda and np.array_equal(got_t.cpu().numpy(), got_h)
    # empty band -> 0, like cog()
    _, e = P.cog_frames(t, z, fs, win=256, ov=0.5, fmin=0.6 * fs, fmax=0.7 * fs)
    assert np.all(e == 0.0)
    # 2^24 samples, 4096-point windows: tone at bin 333.0 of 4096 -> cog = 333/4096 fs in every window
    n = 1 << 24
    k = torch.arange(n, device="cuda", dtype=torch.float64)
    tone = torch.exp(2j * np.pi * (333.0 / 4096.0) * k).to(torch.complex64)
    _, cg = P.cog_frames(np.arange(n) / fs, tone, fs, win=4096, ov=0.5)
    assert cg.shape == (n // 2048 - 1,)
    np.testing.assert_allclose(cg.cpu().numpy(), 333.0 / 4096.0 * fs, rtol=0, atol=1e-5 * fs)
    with pytest.raises(ValueError):
        P.cog_frames(t, z, fs, win=1 << 20)


# ---- N4: derivative by FFT (fft_analysis.py:1453-1587) ------------------------------------------------------------------
def test_fft_deriv_golden(P):
    """pyfft_amd.fft_deriv against the reference's outputs on the inputs of its own test_fft_deriv.  Tolerance: the
    wavenumber multiplies float32 rounding noise by up to 1/dx = N on the scaled axis, so atol = 2e-7 * N * max|d| floor
    1e-4 * max|d| (the end points are exact one-sided differences)."""
    from golden.make_golden_deriv import cases
    g = load_golden("fft_deriv")
    for name, (yy, xx, kw) in cases().items():
        d, xo = P.fft_deriv(yy, xx, **kw)
        ref = g[name + "_d"]
        assert d.shape == ref.shape and d.dtype == np.float64, name
        scale = np.max(np.abs(ref))
        tol = max(1e-4, 2e-7 * len(yy)) * scale
        assert np.max(np.abs(d - ref)) <= tol, (name, np.max(np.abs(d - ref)), tol)
        assert d[0] == pytest.approx(ref[0], rel=1e-9, abs=1e-12 * scale) and d[-1] == pytest.approx(ref[-1], rel=1e-9, abs=1e-12 * scale)
        if name + "_x" in g.files:
            np.testing.assert_allclose(xo, g[name + "_x"], rtol=1e-12, atol=1e-12)
    yy, xx, _ = cases()["sine_aperiodic"]
    d, _ = P.fft_deriv(yy, xx, detrend=P.detrend_mean)
    ref = g["sine_aperiodic_detrend_d"]
    assert np.max(np.abs(d - ref)) <= 4e-4 * np.max(np.abs(ref))
    with pytest.raises(NotImplementedError):
        P.fft_deriv(yy, xx, lowpass=0.01)                    # would need the absent downsampling pre-filter


def test_spectral_filter_rows_vs_numpy(P):
    """sp_spectral_filter = IFFT(H FFT(x)): batched rows, zero-padding, Bluestein and long rows, device tensors"""
    import torch
    rng = np.random.default_rng(5)
    for n, batch in ((256, 7), (1000, 3), (4096, 64), (8192, 2), (30000, 1), (1 << 17, 2)):
        x = rng.standard_normal((batch, n)).astype(np.float32)
        H = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.ifft(H[None, :] * np.fft.fft(x.astype(np.float64), axis=-1), axis=-1)
        got = P.engine.spectral_filter_rows(x, H)
        assert got.shape == ref.shape and got.dtype == np.complex64
        err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
        assert err <= 5e-6, (n, err)
        got_t = P.engine.spectral_filter_rows(torch.from_numpy(x).cuda(), H)
        assert np.array_equal(got_t.cpu().numpy(), got), n
    # H = analytic-signal mask reproduces hilbert()
    x = rng.standard_normal((4, 2048)).astype(np.float32)
    h = np.zeros(2048)
    h[0] = h[1024] = 1.0
    h[1:1024] = 2.0
    np.testing.assert_allclose(P.engine.spectral_filter_rows(x, h), P.engine.hilbert_rows(x, 2048), rtol=0, atol=1e-6)


# ---- A5: the nT-model branch of fft_pwelch (fft_analysis.py:169-176, :346-393) ---------------------------------------
NT_CASES = {"one_mean": dict(detrend_style=1), "one_linear_hamming": dict(detrend_style=-1, windowfunction="hamming"),
            "two_none": dict(detrend_style=0, onesided=False)}


@pytest.mark.parametrize("tag", sorted(NT_CASES))
def test_fft_pwelch_ntmodel_golden(P, tag):
    """a one-window model signal against every window of sigy: Pxy = mean_g(Y_g) conj(X) through the time-domain frame sum
    (sp_frame_sum), vs what the reference returned for the same inputs (tests/golden/make_golden_ntmodel.py)"""
    from golden.make_golden_ntmodel import inputs
    g = load_golden("pwelch_ntmodel")
    fs, t, xm, y = inputs()
    tb = list(g["tb"])
    for ych, ytag in ((y[:, 0], "1ch"), (y, "2ch")):
        r = P.fft_pwelch(t, xm, ych, tb, **NT_CASES[tag])
        p = "%s_%s_" % (tag, ytag)
        np.testing.assert_allclose(r[0], g[p + "freq"], rtol=1e-12, atol=1e-9)
        for nm, v in zip(("Pxy", "Pxx", "Pyy"), r[1:4]):
            close_rel(np.asarray(v).reshape(g[p + nm].shape), g[p + nm], 2e-5, p + nm)
        info = r[6]
        assert (info.Navr, info.nwins, info.noverlap) == (int(g[p + "Navr"]), int(g[p + "nwins"]), int(g[p + "noverlap"]))
        for k in ("Lxy", "Rxy", "corrcoef"):          # same bound as test_fft_pwelch_golden (sqrt / normalisation of small bins)
            close_rel(np.asarray(getattr(info, k)).reshape(g[p + k].shape), g[p + k], 2e-3, p + k)
        # coherence where the model has power (elsewhere it is a ratio of rounding noise)
        strong = np.abs(g[p + "Pxx"]) > 1e-6 * np.abs(g[p + "Pxx"]).max()
        c_ref = g[p + "Cxy"]
        c_got = np.asarray(r[4]).reshape(c_ref.shape)
        assert np.max(np.abs(c_got[strong] - c_ref[strong])) <= 1e-3
    with pytest.raises(UnboundLocalError):
        P.fft_pwelch(t, xm, y[:, 0], tb, Navr=37)                # the reference's behaviour, recorded in the fixture
    with pytest.raises(ValueError):
        P.fft_pwelch(t, xm, y[:, 0], None)
    # per-segment arrays: the model spectrum repeated
    r = P.fft_pwelch(t, xm, y, tb, segments=True, **NT_CASES[tag])
    assert r[6].Xfft_seg.shape == (int(g["%s_2ch_Navr" % tag]), 1024) and np.array_equal(r[6].Xfft_seg[0], r[6].Xfft_seg[-1])


def test_frame_sum_vs_numpy(P):
    """sp_frame_sum: time-domain sum of all frames per channel, with none / mean / linear detrend, real and complex,
    ragged hop, device tensors; and the linearity it exists for: FFT(win * c) = sum_g FFT(win * frame_g)."""
    import torch
    import scipy.signal
    rng = np.random.default_rng(9)
    for nch, nsig, nfft, hop, cplx in ((1, 5000, 256, 128, False), (3, 20011, 1000, 333, False), (2, 9000, 512, 512, True),
                                        (5, 1 << 18, 4096, 1024, False)):
        y = rng.standard_normal((nch, nsig)) + 0.3 + 1e-4 * np.arange(nsig)
        if cplx:
            y = y + 1j * (rng.standard_normal((nch, nsig)) - 0.2)
        y = y.astype(np.complex64 if cplx else np.float32)
        M = (nsig - nfft) // hop + 1
        idx = (np.arange(M) * hop)[:, None] + np.arange(nfft)[None, :]
        for det in (False, True, "linear"):
            y64 = y.astype(np.complex128 if cplx else np.float64)
            if det is True:
                y64 = y64 - y64.mean(axis=1, keepdims=True)
            elif det == "linear":
                y64 = scipy.signal.detrend(y64.real, axis=1) + (1j * scipy.signal.detrend(y64.imag, axis=1) if cplx else 0)
            ref = np.stack([y64[c][idx].sum(axis=0) for c in range(nch)])
            got = P.engine.frame_sum(y, nfft, hop, M, detrend=det)
            assert got.shape == ref.shape
            scale = np.max(np.abs(y64)) * np.sqrt(M)
            assert np.max(np.abs(got - ref)) <= 2e-5 * scale * np.sqrt(M), (nch, nsig, det)
        got_t = P.engine.frame_sum(torch.from_numpy(y).cuda(), nfft, hop, M, detrend=True)
        np.testing.assert_allclose(got_t.cpu().numpy(), P.engine.frame_sum(y, nfft, hop, M, detrend=True), rtol=1e-12, atol=1e-9)


def test_doppler_cog_streaming_and_generic_paths_agree(P, monkeypatch):
    """sp_stft_cog has two forms: the register-carried streaming kernel (power-of-two window, hop = win/4, /2 or win) and the
    generic frame kernel (everything else; SP_COG_GENERIC=1 forces it).  Same moments, float32 rounding apart."""
    g = load_golden("doppler_cog")
    fs, z, r = float(g["fs"]), g["z"], g["r"]
    t = np.arange(len(z)) / fs
    for x in (z, r):
        for win, ov in ((256, 0.5), (512, 0.75), (1024, 0.0), (4096, 0.5)):
            for kw in (dict(), dict(fmin=40e3, fmax=260e3), dict(detrend=True)):
                monkeypatch.delenv("SP_COG_GENERIC", raising=False)
                _, a = P.cog_frames(t, x, fs, win=win, ov=ov, **kw)
                monkeypatch.setenv("SP_COG_GENERIC", "1")
                _, b = P.cog_frames(t, x, fs, win=win, ov=ov, **kw)
                monkeypatch.delenv("SP_COG_GENERIC", raising=False)
                np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * fs, err_msg="%d %.2f %s" % (win, ov, sorted(kw)))
                _, ref = O.cog_frames(t, x - (x.mean() if kw.get("detrend") else 0), fs, win=win, ov=ov,
                                      **{k: v for k, v in kw.items() if k != "detrend"})
                np.testing.assert_allclose(a, ref, rtol=0, atol=3e-6 * fs)


@pytest.mark.parametrize("tag,kw", [("one_mean", dict(onesided=True, detrend=1)), ("two_none", dict(onesided=False, detrend=0)),
                                    ("one_linear", dict(onesided=True, detrend=-1))])
def test_fftanal_stft_usemlab_scipy_branch(P, tag, kw):
    """fftanal.stft() with useMLAB=True = the scipy.signal.stft branch (fft_analysis.py:1805-1824): zero-extended
    boundaries, padded frames, 1/sum(window) scaling, detrend across segments; fixture from the reference
    (make_golden_stftmlab.py), including the IndexError its averagewins step ends in"""
    g = load_golden("stft_usemlab")
    rng = np.random.default_rng(int(g["seed"]))
    n = int(g["n"])
    t = np.arange(n) / 2.0e3
    x = np.sin(2 * np.pi * 120.0 * t) + 0.2 * rng.standard_normal(n) + 0.7
    y = np.cos(2 * np.pi * 120.0 * t + 0.4) + 0.2 * rng.standard_normal(n) - 0.3 + 0.1 * t
    ft = P.fftanal(t, x, y, tbounds=[t[0], t[-1]], Navr=12, windowoverlap=0.5, windowfunction="hamming", useMLAB=True,
                   plotit=False, verbose=False, **kw)
    assert ft.nwins == int(g["nwins_" + tag]) and ft.noverlap == int(g["noverlap_" + tag])
    with pytest.raises(IndexError) as ei:
        ft.stft()
    assert str(g["err_" + tag]) == "IndexError"
    # numpy's own exception from the same np.size(Pyy, axis=1) call the reference makes on the 1-D means (fft_analysis.py:1669;
    # np.size indexes shape[axis]: a plain IndexError "tuple index out of range" under numpy 2.2 too -- ADVICE r2 expected an
    # AxisError; whatever numpy raises there is what propagates now, the hand-written raise behind it is not reached)
    assert "tuple index out of range" in str(ei.value) and "reference behaviour" not in str(ei.value)
    # the working half of the branch is public
    f2, t2, Z2 = ft.scipy_stft()
    np.testing.assert_allclose(f2, g["freq_" + tag], rtol=1e-12, atol=1e-12)
    assert Z2.shape == g["Xseg_" + tag].shape
    np.testing.assert_allclose(ft.freq, g["freq_" + tag], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ft.tseg, g["tseg_" + tag], rtol=1e-12, atol=1e-12)
    for k in ("Xseg", "Yseg"):
        ref = g[k + "_" + tag]
        assert getattr(ft, k).shape == ref.shape
        assert np.max(np.abs(getattr(ft, k) - ref)) <= 1e-4 * np.abs(ref).max(), k
    for k in ("Pxx", "Pxy", "varPxx"):
        ref = g[k + "_" + tag]
        np.testing.assert_allclose(getattr(ft, k), ref, rtol=3e-4, atol=1e-6 * np.abs(ref).max(), err_msg=k)
