"""Pin the CPU oracle (oracle/cpu_ref.py) to the reference: every fixture in tests/golden/ was produced by
running the reference's own modules (tests/golden/make_golden.py); plus the reference's known-answer material.
CPU-only.  float64 everywhere -> rtol 1e-12 unless stated."""
import numpy as np
import pytest

from oracle import cpu_ref as O
from conftest import load_golden

RT = 1e-12


def close(a, b, rtol=RT, atol=0.0):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = np.max(np.abs(b)) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol + 1e-13 * scale)


# ------------------------------------------------------------------ windows (A1)
WNAMES = ["Hanning", "Hamming", "Blackman", "SFT3F", "SFT4F", "SFT5F", "SFT3M", "SFT4M", "SFT5M", "Nuttall3a",
          "Nuttall3b", "Nuttall3", "Nuttall4a", "Nuttall4b", "Nuttall4c", "Nuttall4", "Kaiser", "Welch", "Bartlett",
          "box"]


@pytest.mark.parametrize("name", WNAMES)
def test_windows_catalogue(name):
    g = load_golden("windows")
    kw = {"beta": 8.6} if name == "Kaiser" else {}
    for N in (16, 255, 1024):
        close(O.windows(name, nwins=N, **kw), g["%s_per_%d" % (name, N)])
        close(O.windows(name, nwins=N, periodic=False, **kw), g["%s_sym_%d" % (name, N)])
    assert O.windows(name, **kw) == float(g["%s_rov" % name])


def test_window_constants_heinzel():
    # windows.py:68-70 / Heinzel: Hann ROV 50 %, NENBW 1.5 bins (true N), S1 = N/2, S2 = 3N/8 for the periodic form
    w = O.windows("Hanning", nwins=4096)
    assert O.windows("Hanning") == 0.5
    assert abs(w.sum() - 2048.0) < 1e-9 and abs((w ** 2).sum() - 1536.0) < 1e-9
    assert abs(4096 * (w ** 2).sum() / w.sum() ** 2 - 1.5) < 1e-12


# ------------------------------------------------------------------ geometry (A2)
def test_geometry_table():
    g = load_golden("geometry")
    for nsig, Navr, ov, nw, no, na, nyq in g["table"]:
        nwins = O.get_nwins(int(nsig), int(Navr), ov)
        assert nwins == int(nw)
        assert O.get_noverlap(nwins, ov) == int(no)
        assert O.get_navr(int(nsig), nwins, int(no)) == int(na)
        assert O.get_nnyquist(nwins) == int(nyq)
    close(np.array(O.get_norms(O.windows("Hanning", nwins=4096), 2048, 1.0)), g["hann4096_norms"])


# ------------------------------------------------------------------ class path (A3, A4)
@pytest.mark.parametrize("tag", ["c64_2e16_n4096", "c64_2e14_n1024"])
def test_class_welch_complex(tag):
    g = load_golden("welch_class_" + tag)
    x = g["x"]
    t = np.arange(x.size, dtype=np.float64)
    r = O.pwelch_class(t, x, nwins=int(g["nwins"]), windowfunction="Hanning", windowoverlap=0.5,
                       tbounds=[t[0], t[-1]])
    assert r["Navr"] == int(g["Navr"]) and r["noverlap"] == int(g["noverlap"]) and not r["onesided"]
    for k in ("S1", "S2", "ENBW", "NENBW", "Fs"):
        close(r[k], g[k])
    close(r["freq"], g["freq"])
    close(r["tseg"], g["tseg"])
    close(r["Pxx"], g["Pxx"], rtol=1e-11)
    close(r["varPxx"], g["varPxx"], rtol=1e-11)
    close(r["Xfft"], g["Xfft"], rtol=1e-9, atol=1e-12)
    close(r["Xpow"], g["Xpow"], rtol=1e-11)
    close(r["Xseg"][:4], g["Xseg_head"], rtol=1e-10, atol=1e-12)
    close(r["Xseg"][-2:], g["Xseg_tail"], rtol=1e-10, atol=1e-12)
    close(r["Pxx_seg"][:2], g["Pxx_seg_head"], rtol=1e-10)
    # the streaming restatement used for the CPU baseline is the same arithmetic
    ps = O.welch_psd_stream(x, r["win"], r["nwins"], r["nwins"] - r["noverlap"], r["Navr"], r["Fs"], chunk=7)
    close(ps, g["Pxx"].real, rtol=1e-11)


@pytest.mark.parametrize("wname", ["Hamming", "SFT3F"])
def test_class_welch_real_onesided(wname):
    g = load_golden("welch_class_real_" + wname)
    t, x, y = g["t"], g["x"], g["y"]
    r = O.pwelch_class(t, x, y, Navr=31, windowfunction=wname, tbounds=[t[0], t[-1]])
    assert r["onesided"] and r["nwins"] == int(g["nwins"]) and r["noverlap"] == int(g["noverlap"])
    assert r["Navr"] == int(g["Navr"])
    close(r["freq"], g["freq"])
    for k in ("Pxx", "Pyy", "Pxy"):
        close(r[k], g[k], rtol=1e-10, atol=1e-18)
    close(r["Xseg"][:3], g["Xseg_head"], rtol=1e-10, atol=1e-12)
    close(r["Yseg"][:3], g["Yseg_head"], rtol=1e-10, atol=1e-12)
    close(r["Lxx_seg"][:2], g["Lxx_seg_head"], rtol=1e-10, atol=1e-12)
    close(r["Xpow"], g["Xpow"], rtol=1e-11)


# ------------------------------------------------------------------ function path (A5, A6)
PW = {
    "pwelch_cfg1": dict(Navr=127, windowoverlap=0.5, windowfunction="Hanning", tb=-2),
    "pwelch_reflect": dict(Navr=15, windowfunction="Hanning", tb=None),
    "pwelch_2ch_twosided": dict(Navr=31, windowfunction="Hamming", onesided=False, detrend_style=0, tb=-2),
    "pwelch_minfreq_linear": dict(minFreq=2.0 * 1.0e4 / 1024.0 * 1.0000001, detrend_style=-1, tb=-2),
    "pwelch_selftest_navr8": dict(Navr=8, windowfunction="hamming", detrend_style=1, tb=-1),
    "pwelch_selftest_minfreq": dict(minFreq=75.0, detrend_style=1, tb=-1),
}


@pytest.mark.parametrize("tag", sorted(PW))
def test_fft_pwelch(tag):
    g = load_golden(tag)
    kw = dict(PW[tag])
    tb = kw.pop("tb")
    t, x, y = g["t"], g["x"], g["y"]
    tbounds = None if tb is None else [t[0], t[tb]]
    freq, Pxy, Pxx, Pyy, Cxy, phi, info = O.fft_pwelch(t, x, y, tbounds=tbounds, **kw)
    for k in ("nwins", "noverlap", "Navr", "nch"):
        assert int(info[k]) == int(g["info_" + k]), k
    assert list(info["ibnds"]) == list(g["info_ibnds"])
    for k in ("S1", "S2", "ENBW", "NENBW", "Fs", "minFreq"):
        close(info[k], g["info_" + k])
    close(freq, g["freq"])
    # linear detrend goes through a different LS solver than scipy's -> slightly looser
    rt = 1e-7 if kw.get("detrend_style", 1) < 0 else 1e-9
    at = 1e-9 * float(np.max(np.abs(g["Pxx"])))
    close(Pxx, g["Pxx"], rtol=rt, atol=at)
    close(Pyy, g["Pyy"], rtol=rt, atol=at)
    close(Pxy, g["Pxy"], rtol=rt, atol=at)
    close(info["Xfft_seg"][:2], g["Xfft_seg_head"], rtol=rt, atol=1e-9 * np.abs(g["Xfft_seg_head"]).max())
    if tag in ("pwelch_cfg1", "pwelch_2ch_twosided", "pwelch_reflect"):
        # epilogue quantities: only where no bin is ~0/0 (the self-test inputs are noise-free -> coherence of
        # rounding noise is ill-conditioned; those are compared through Pxx/Pyy/Pxy above)
        close(Cxy, g["Cxy"], rtol=1e-7, atol=1e-9)
        close(phi, g["phi_xy"], rtol=1e-6, atol=1e-7)
        for k in ("Lxx", "Lyy", "Lxy", "Rxx", "Ryy", "Rxy", "corrcoef", "lags", "Ex", "Ey", "varPxx", "varPyy"):
            gk = g["info_" + k]
            # (np.ascontiguousarray in the generator promoted 0-d values to 1-d)
            close(np.atleast_1d(info[k]), gk, rtol=1e-7, atol=1e-9 * float(np.max(np.abs(gk))))


# ------------------------------------------------------------------ stft / specgram (A8, A9)
def test_stft_class_f32():
    g = load_golden("stft_f32_n2048_ov75")
    r = O.stft(g["t"], g["x"], tper=2048.5, windowfunction="Hanning", windowoverlap=0.75)
    assert r["nwins"] == 2048 and r["noverlap"] == 1536 and r["Navr"] == int(g["Navr"]) and r["onesided"]
    close(r["freq"], g["freq"])
    close(r["tseg"], g["tseg"])
    M = r["Navr"]
    at = 1e-9 * float(np.abs(g["Xseg_head"]).max())
    close(r["Xseg"][:3], g["Xseg_head"], rtol=1e-9, atol=at)
    close(r["Xseg"][M // 2:M // 2 + 2], g["Xseg_mid"], rtol=1e-9, atol=at)
    close(r["Xseg"][-2:], g["Xseg_tail"], rtol=1e-9, atol=at)
    close(r["Pxx"], g["Pxx"], rtol=1e-9, atol=1e-9 * float(np.abs(g["Pxx"]).max()))
    close(r["Xpow"], g["Xpow"], rtol=1e-9)


def test_stft_tuple():
    g = load_golden("stft_tuple_n256")
    twin, freq, Xseg = O.stft(g["t"], g["x"], tper=256.5, returnclass=False, windowfunction="Hamming")
    close(twin, g["twin"])
    close(freq, g["freq"])
    close(Xseg, g["Xseg"], rtol=1e-10, atol=1e-12)


def test_specgram():
    g = load_golden("specgram")
    time1, f1, sp1 = O.specgram(g["t"], g["s"], wl=512, hanning=True, overlap=True)
    close(time1, g["time1"]); close(f1, g["f1"]); close(sp1, g["sp1"], rtol=1e-11, atol=1e-13)
    time2, f2, sp2 = O.specgram(g["t"], g["s"], wl=500, hanning=False, overlap=False)
    close(time2, g["time2"]); close(f2, g["f2"]); close(sp2, g["sp2"], rtol=1e-11, atol=1e-13)


# ------------------------------------------------------------------ hilbert (A10)
def test_hilbert_golden():
    g = load_golden("hilbert")
    close(O.hilbert(g["yk"]), g["zk"], atol=1e-15)
    close(O.hilbert_1d(g["yk"]), g["zk1d"], atol=1e-15)
    close(O.hilbert(g["u_even"]), g["z_even"], atol=1e-14)
    close(O.hilbert(g["u_odd"]), g["z_odd"], atol=1e-14)
    close(O.hilbert_1d(g["u_odd"]), g["z_odd_1d"], atol=1e-14)
    close(O.hilbert(g["u_2d"]), g["z_2d"], atol=1e-14)
    close(O.hilbert(g["u_2d"], axes=0), g["z_2d_ax0"], atol=1e-14)
    z32 = O.hilbert(g["u_f32"])
    assert z32.dtype == g["z_f32"].dtype
    close(z32, g["z_f32"], rtol=1e-6, atol=1e-6)
    close(O.hilbert(g["u_even"][:1000], nfft=1024), g["z_nfft"], atol=1e-14)


def test_hilbert_known_answer():
    # hilbert.py:115-140: one-cycle sine, analytic signal = y - j cos
    N = 32
    ph = 2 * np.pi * np.arange(N) / N
    z = O.hilbert(np.sin(ph))
    assert np.max(np.abs(z - (np.sin(ph) - 1j * np.cos(ph)))) < 1e-14


# ------------------------------------------------------------------ ccf (A11)
def test_ccf_golden_and_fft_form():
    g = load_golden("ccf")
    tau, co = O.ccf(g["x1"], g["x2"], float(g["fs"]))
    close(tau, g["tau"]); close(co, g["co"], rtol=1e-11, atol=1e-14)
    tau2, co2 = O.ccf(g["x3"], g["x4"], 1.0)
    close(tau2, g["tau2"]); close(co2, g["co2"], rtol=1e-11, atol=1e-14)
    # the FFT formulation the GPU kernel uses is the same function
    tf, cf = O.ccf_fft(g["x1"], g["x2"], float(g["fs"]))
    close(tf, g["tau"]); close(cf, g["co"], rtol=1e-9, atol=1e-13)
    # ccf.py:139-148: expected lag of the maximum = -phi/(2 pi f) = -138.9 us (noisy input: within one period/8)
    assert abs(tau[np.argmax(co)] - (-138.9e-6)) < 125e-6


# ------------------------------------------------------------------ notch design (A12)
def test_notch_design():
    g = load_golden("notch")
    b, a = O.iirnotch(60.0 / 100.0, 30.0)
    close(b, g["b_doc"]); close(a, g["a_doc"])
    for row in g["sweep"]:
        w0, Q = row[0], row[1]
        bn, an = O.iirnotch(w0, Q)
        bp, ap = O.iirpeak(w0, Q)
        close(np.concatenate([bn, an, bp, ap]), row[2:])
    import scipy.signal as ss
    bs, as_ = ss.iirnotch(0.6, 30.0)
    close(b, bs); close(a, as_)
    with pytest.raises(ValueError):
        O.iirnotch(1.5, 3.0)


# ------------------------------------------------------------------ build-defined FIR (F1/F2) vs scipy
def test_fftfilt_and_notch_apply_vs_scipy():
    import scipy.signal as ss
    rng = np.random.default_rng(5)
    x = rng.standard_normal(5000)
    h = ss.firwin(513, 0.2)
    close(O.fftfilt(h, x), ss.lfilter(h, 1.0, x), rtol=1e-10, atol=1e-12)
    b, a = O.iirnotch(0.12, 5.0)
    close(O.biquad_fir(b, a, 64), ss.lfilter(b, a, np.r_[1.0, np.zeros(63)]), rtol=1e-11, atol=1e-14)
    # truncation error bound of the 513-tap realisation against the exact recursion
    err = np.max(np.abs(O.notch_apply(x, 0.12, 5.0) - ss.lfilter(b, a, x)))
    assert err < 1e-6 * np.max(np.abs(x))


def test_mlab_wrappers_against_reference_fixture():
    """psd / csd / coh as the reference (through matplotlib.mlab) computed them"""
    g = load_golden("mlab_wrappers")
    x, y, fs = g["x"], g["y"], float(g["fs"])
    p, f = O.mlab_psd_wrapper(x, fs)
    np.testing.assert_allclose(f, g["psd_f"], rtol=1e-13, atol=0)
    np.testing.assert_allclose(p, g["psd_p"], rtol=1e-10)
    p, f = O.mlab_psd_wrapper(x, fs, nfft=500, fmin=20.0, fmax=300.0, detrend="mean", ov=0.5)
    np.testing.assert_allclose(f, g["psd2_f"], rtol=1e-13)
    np.testing.assert_allclose(p, g["psd2_p"], rtol=1e-10)
    p, f = O.mlab_csd_wrapper(x, y, fs)
    np.testing.assert_allclose(f, g["csd_f"], rtol=1e-13)
    np.testing.assert_allclose(p, g["csd_p"], rtol=1e-9, atol=1e-12 * np.abs(g["csd_p"]).max())
    p, f = O.mlab_csd_wrapper(x, y, fs, nfft=1024, fmin=None, fmax=None, detrend="mean", ov=0.75)
    np.testing.assert_allclose(p, g["csd2_p"], rtol=1e-9, atol=1e-12 * np.abs(g["csd2_p"]).max())
    p, f = O.mlab_psd_wrapper(x, fs, nfft=1024, detrend="linear", ov=0.5)
    np.testing.assert_allclose(p, g["psd3_p"], rtol=1e-9)
    p, f = O.mlab_csd_wrapper(x, y, fs, nfft=600, fmin=None, fmax=None, detrend="linear", ov=0.25)
    np.testing.assert_allclose(p, g["csd3_p"], rtol=1e-8, atol=1e-12 * np.abs(g["csd3_p"]).max())
    c, f = O.mlab_coh_wrapper(x, y, fs)
    np.testing.assert_allclose(f, g["coh_f"], rtol=1e-13)
    np.testing.assert_allclose(c, g["coh_c"], rtol=1e-9)
    c, f = O.mlab_coh_wrapper(x, y, fs, nfft=512, fmin=10.0, fmax=400.0, detrend="none", ov=0.5)
    np.testing.assert_allclose(c, g["coh2_c"], rtol=1e-9)
    assert int(g["cohb_ok"]) == 0          # the reference's coh2 raises under this matplotlib: no fixture


@pytest.mark.parametrize("tag,kw", [("pwelch_usemlab_onesided", dict(Navr=15, windowoverlap=0.5, windowfunction="Hamming")),
                                    ("pwelch_usemlab_twosided_linear", dict(Navr=9, windowoverlap=0.5, windowfunction="Hanning",
                                                                            onesided=False, detrend_style=-1))])
def test_fft_pwelch_usemlab_branch(tag, kw):
    """fft_pwelch(useMLAB=True) (fft_analysis.py:254-330) as the reference computed it through matplotlib.mlab.csd"""
    g = load_golden(tag)
    t, x, y = g["t"], g["x"], g["y"]
    freq, Pxy, Pxx, Pyy, Cxy, phi, info = O.fft_pwelch(t, x, y, tbounds=[t[0], t[-2]], useMLAB=True, **kw)
    np.testing.assert_allclose(freq, g["freq"], rtol=1e-13, atol=1e-9)
    for a, name in ((Pxx, "Pxx"), (Pyy, "Pyy"), (Pxy, "Pxy"), (Cxy, "Cxy")):
        assert np.max(np.abs(np.asarray(a) - g[name])) <= 1e-11 * np.abs(g[name]).max(), name
    np.testing.assert_allclose(info["Rxy"], g["info_Rxy"], rtol=1e-8, atol=1e-11 * np.abs(g["info_Rxy"]).max())


def test_doppler_cog_against_reference_fixture():
    """Doppler.cog (Doppler.py:43-58) whole-vector, banded (reference pairing) and per-window values"""
    g = load_golden("doppler_cog")
    fs, z, r = float(g["fs"]), g["z"], g["r"]
    t = np.arange(len(z)) / fs
    for tag in ("4096", "1000", "40000", "16384"):
        m = int(tag)
        np.testing.assert_allclose(O.cog(z[:m], fs), g["cog_z_" + tag], rtol=1e-10, atol=1e-7)
        np.testing.assert_allclose(O.cog(r[:m], fs), g["cog_r_" + tag], rtol=1e-8, atol=1e-6)
    np.testing.assert_allclose(O.cog(z[:4096], fs, fmin=50e3, fmax=200e3), g["cog_z_band"], rtol=1e-10)
    np.testing.assert_allclose(O.cog(z[:1000], fs, fmin=100e3), g["cog_z_band_nofmax"], rtol=1e-10)
    assert O.cog(z[:1000], fs, fmin=2e6, fmax=3e6) == float(g["cog_z_band_empty"]) == 0.0
    for tag, win, ov in (("512", 512, 0.5), ("200", 200, 0.75)):
        tc, cg = O.cog_frames(t, z, fs, win=win, ov=ov)
        # numpy transforms the complex64 / float32 frames in single precision in the reference; cog_frames works in double
        np.testing.assert_allclose(cg, g["frames_z_" + tag], rtol=1e-7, atol=1e-6)
        np.testing.assert_allclose(tc, g["frames_t_" + tag], rtol=1e-12)
        tc, cg = O.cog_frames(t, r, fs, win=win, ov=ov)
        np.testing.assert_allclose(cg, g["frames_r_" + tag], rtol=1e-6, atol=1e-7 * fs)


def test_fft_deriv_against_reference_fixture():
    """fft_deriv (fft_analysis.py:1453-1587) on the inputs of the reference's own test_fft_deriv"""
    from golden.make_golden_deriv import cases
    g = load_golden("fft_deriv")
    for name, (yy, xx, kw) in cases().items():
        d, xo = O.fft_deriv(yy, xx, **kw)
        scale = np.max(np.abs(g[name + "_d"]))
        np.testing.assert_allclose(d, g[name + "_d"], rtol=0, atol=1e-9 * scale, err_msg=name)
        if name + "_x" in g.files:
            np.testing.assert_allclose(xo, g[name + "_x"], rtol=1e-12, atol=1e-12, err_msg=name)
    yy, xx, _ = cases()["sine_aperiodic"]
    d, _ = O.fft_deriv(yy, xx, detrend=lambda v: v - v.mean())
    np.testing.assert_allclose(d, g["sine_aperiodic_detrend_d"], rtol=0, atol=1e-9 * np.max(np.abs(d)))


NT_CASES = {"one_mean": dict(detrend_style=1), "one_linear_hamming": dict(detrend_style=-1, windowfunction="hamming"),
            "two_none": dict(detrend_style=0, onesided=False)}


@pytest.mark.parametrize("tag", sorted(NT_CASES))
def test_fft_pwelch_ntmodel_branch(tag):
    """nT-model branch (fft_analysis.py:169-176, :346-393): a one-window model signal against every window of sigy"""
    from golden.make_golden_ntmodel import inputs
    g = load_golden("pwelch_ntmodel")
    fs, t, xm, y = inputs()
    tb = list(g["tb"])
    for ych, ytag in ((y[:, 0], "1ch"), (y, "2ch")):
        r = O.fft_pwelch(t, xm, ych, tb, **NT_CASES[tag])
        p = "%s_%s_" % (tag, ytag)
        for nm, v in zip(("freq", "Pxy", "Pxx", "Pyy", "Cxy", "phi_xy"), r[:6]):
            ref = g[p + nm]
            np.testing.assert_allclose(np.asarray(v).reshape(ref.shape), ref, rtol=1e-9, atol=1e-12 * np.max(np.abs(ref)), err_msg=p + nm)
        for k in ("S1", "S2", "ENBW", "NENBW", "Navr", "nwins", "noverlap", "Lxy", "Rxy", "corrcoef", "lags"):
            ref = g[p + k]
            np.testing.assert_allclose(np.asarray(r[6][k]).reshape(ref.shape), ref, rtol=1e-8, atol=1e-11 * max(1.0, np.max(np.abs(ref))), err_msg=p + k)
    assert list(g["errors"]) == ["UnboundLocalError", "ValueError"]
    with pytest.raises(UnboundLocalError):
        O.fft_pwelch(t, xm, y[:, 0], tb, Navr=37)
    with pytest.raises(ValueError):
        O.fft_pwelch(t, xm, y[:, 0], None)


# ------------------------------------------------------------------ long segments (the reference's default Navr=8 regime)
def _long_inputs():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs_long
    return inputs_long


def test_fft_pwelch_long_navr8():
    """test_fftanal's own call shape (fft_analysis.py:2950-2993): N = 2^19, Navr = 8 -> nwins = 116 508"""
    g = load_golden("pwelch_long_navr8")
    tvec, sigx, sigy = _long_inputs().long_signals(int(g["N"]), float(g["df"]), int(g["seed"]))
    freq, Pxy, Pxx, Pyy, Cxy, phi, info = O.fft_pwelch(tvec, sigx, sigy, tbounds=[tvec[0], tvec[-1]], Navr=8,
                                                        windowfunction="hamming", detrend_style=1, onesided=True)
    assert int(info["nwins"]) == int(g["nwins"]) == 116508 and int(info["Navr"]) == 8
    assert int(info["noverlap"]) == int(g["noverlap"]) and list(info["ibnds"]) == list(g["ibnds"])
    for k in ("S1", "S2", "ENBW", "NENBW", "Fs"):
        close(info[k], g[k])
    ib, il = g["ibin"], g["ilag"]
    assert Pxx.shape[0] == int(g["nbins"])
    close(freq[ib], g["freq"])
    at = 1e-9 * float(np.max(np.abs(g["Pxx"])))
    for name, got in (("Pxx", Pxx), ("Pyy", Pyy), ("Pxy", Pxy)):
        close(np.asarray(got).reshape(len(freq), -1)[ib, 0], g[name], rtol=1e-8, atol=at)
    close(np.asarray(Cxy).reshape(len(freq), -1)[ib, 0], g["Cxy"], rtol=1e-7, atol=1e-9)
    close(np.asarray(phi).reshape(len(freq), -1)[ib, 0], g["phi_xy"], rtol=1e-6, atol=1e-7)
    for k in ("Lxx", "Lyy", "Lxy", "varPxx"):
        gk = g["info_" + k]
        close(np.asarray(info[k]).reshape(len(freq), -1)[ib, 0], gk, rtol=1e-7, atol=1e-9 * float(np.max(np.abs(gk))))
    for k in ("Rxx", "Ryy", "Rxy", "corrcoef", "lags"):
        gk = g["info_" + k]
        a = np.asarray(info[k])
        close(a.reshape(a.shape[0], -1)[il, 0], gk, rtol=1e-7, atol=1e-9 * float(np.max(np.abs(gk))))


def test_class_welch_long_navr8():
    g = load_golden("welch_class_long_navr8")
    tvec, sigx, sigy = _long_inputs().long_signals(int(g["N"]), float(g["df"]), int(g["seed"]))
    r = O.pwelch_class(tvec, sigx, sigy, Navr=8, windowfunction="hamming", windowoverlap=0.5, tbounds=[tvec[0], tvec[-1]])
    assert r["onesided"] and r["nwins"] == int(g["nwins"]) and r["Navr"] == int(g["Navr"])
    ib = g["ibin"]
    close(r["freq"][ib], g["freq"])
    close(r["tseg"], g["tseg"])
    for k in ("Pxx", "Pyy", "Pxy"):
        close(np.asarray(r[k])[ib], g[k], rtol=1e-8, atol=1e-9 * float(np.max(np.abs(g[k]))))
    sc = float(np.abs(g["Xseg_first"]).max())
    close(r["Xseg"][0][ib], g["Xseg_first"], rtol=1e-8, atol=1e-9 * sc)
    close(r["Xseg"][-1][ib], g["Xseg_last"], rtol=1e-8, atol=1e-9 * sc)
    close(r["Yseg"][0][ib], g["Yseg_first"], rtol=1e-8, atol=1e-9 * sc)
    close(r["Xpow"], g["Xpow"], rtol=1e-9)


def test_stft_long_windows():
    g = load_golden("stft_long_n10000")
    k, xs = _long_inputs().stft_long_signal(int(g["n"]), int(g["seed"]))
    r = O.stft(k, xs, tper=10000.5, windowfunction="Hanning", windowoverlap=0.5)
    assert r["nwins"] == int(g["nwins"]) == 10000 and r["Navr"] == int(g["Navr"])
    ib = g["ibin"]
    close(r["freq"][ib], g["freq"])
    close(r["tseg"], g["tseg"])
    close(np.asarray(r["Xseg"])[:, ib], g["Xseg_sub"], rtol=1e-8, atol=1e-9 * float(np.abs(g["Xseg_sub"]).max()))
    close(np.asarray(r["Pxx"])[ib], g["Pxx"], rtol=1e-8, atol=1e-9 * float(np.abs(g["Pxx"]).max()))


def _xcorr_inputs(g):
    rng = np.random.default_rng(int(g["seed"]))
    n, fs = int(g["n"]), float(g["fs"])
    t = np.arange(n) / fs
    x = np.sin(2 * np.pi * 50 * t) + 0.3 * rng.standard_normal(n)
    y = np.sin(2 * np.pi * 50 * t + 0.7) + 0.3 * rng.standard_normal(n)
    return t, x, y


@pytest.mark.parametrize("onesided", [True, False])
def test_class_crosscorr_against_reference_fixture(onesided):
    """fftanal.crosscorr_stft / crosscorr (fft_analysis.py:1840-1920): reference-generated fixture (make_golden_xcorr.py)"""
    g = load_golden("crosscorr_class")
    t, x, y = _xcorr_inputs(g)
    tag = "one" if onesided else "two"
    r = O.pwelch_class(t, x, y, Navr=8, windowfunction="hanning", windowoverlap=0.5, tbounds=[t[0], t[-1]], onesided=onesided)
    assert r["nwins"] == int(g["nwins_" + tag]) and r["Nnyquist"] == int(g["Nnyquist_" + tag])
    c = O.crosscorr_class(r)
    for k in ("Rxx_seg", "Ryy_seg", "Rxy_seg", "Ex_seg", "Ey_seg", "Rxx", "Ryy", "Rxy"):
        ref = g[k + "_" + tag]
        close(np.asarray(c[k]).reshape(ref.shape), ref, rtol=1e-9, atol=1e-12 * float(np.abs(ref).max()))
    # the reference's own last line fails for 1-D signals (self.nch is never set): recorded, not reproduced
    assert str(g["err_stft_" + tag]) == "AttributeError" and str(g["err_avg_" + tag]) == "AttributeError"
