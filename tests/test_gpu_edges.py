"""Edge cases through the C ABI and the drop-in layer: degenerate sizes, a single segment (nwins >= nsig), ragged
tails, maximum workgroup sizes, argument errors.  GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O


@pytest.fixture(scope="module")
def P():
    import pyfft_amd
    from pyfft_amd import _ffi
    _ffi.init()
    return pyfft_amd


def test_tiny_and_degenerate_transforms(P):
    E = P.engine
    assert np.allclose(E.fft(np.array([3.0 + 1j], dtype=np.complex64)), [3.0 + 1j])            # n = 1
    x = np.array([[1, 2], [3, 4]], dtype=np.complex64)
    np.testing.assert_allclose(E.fft(x), np.fft.fft(x), atol=1e-6)                              # n = 2
    assert E.fft(np.zeros((0, 8), dtype=np.complex64)).shape == (0, 8)                          # empty batch
    z = np.zeros(64, dtype=np.complex64)
    assert np.all(E.fft(z) == 0)
    big = np.full(256, 1e30, dtype=np.float32)                                                  # no overflow to inf/nan
    assert np.all(np.isfinite(E.fft(big * 1e-3)))


def test_single_segment_welch_and_nwins_ge_nsig(P):
    """fft_pwelch with Navr=1: nwins collapses to nsig (fft_analysis.py:214-216), any length up to the workgroup limit"""
    rng = np.random.default_rng(2)
    for n in (100, 1000, 4096, 3001):
        t = np.arange(n + 1) / 100.0
        x = rng.standard_normal(n + 1)
        y = rng.standard_normal(n + 1)
        freq, Pxy, Pxx, Pyy, Cxy, phi, info = P.fft_pwelch(t, x, y, [t[0], t[-2]], Navr=1, windowoverlap=0.0,
                                                          windowfunction="box", plotit=False)
        r = O.fft_pwelch(t, x, y, tbounds=[t[0], t[-2]], Navr=1, windowoverlap=0.0, windowfunction="box")
        assert info.Navr == 1 == r[6]["Navr"] and info.nwins == r[6]["nwins"] == n
        np.testing.assert_allclose(Pxx, r[2], rtol=2e-4, atol=1e-6 * np.abs(r[2]).max())
        np.testing.assert_allclose(Pxy, r[1], rtol=2e-4, atol=1e-6 * np.abs(r[1]).max())
        # one segment: coherence is exactly 1 in magnitude
        assert np.allclose(np.abs(Cxy[1:-1]), 1.0, atol=1e-3)


def test_limits_raise_cleanly(P):
    E = P.engine
    Err = P._ffi.SpectralError
    x = np.zeros(40000, dtype=np.float32)
    # segments longer than one workgroup transform go through the multi-kernel path (tests/test_gpu_long.py) ...
    assert np.all(E.welch_psd(x + 1, np.ones(16384), 8192, 2, detrend=False, sided=E.SIDED_RAW)[1:] < 1e-3)
    assert E.welch_psd(x, np.ones(5000), 2500, 3).shape == (5000,)
    # ... up to a 2^26-point transform (chirp-z needs next_pow2(2n-1) points)
    with pytest.raises(Err, match="limit is 2\\^26"):
        E.fft(np.zeros((1 << 25) + 1, dtype=np.complex64))
    with pytest.raises(Err):
        E.welch_psd(x, np.ones(256), 128, 100000)                # frames run past the end
    with pytest.raises(Err):
        E.welch_psd(x, np.ones(256), 0, 4)                       # hop 0
    with pytest.raises(Err):
        E.fir_filter(np.ones(5000), x, nfft=4096)                # block shorter than 2*(taps-1)
    with pytest.raises(Err):
        E.hilbert_rows(np.zeros((2, 8), dtype=np.float32), 1)
    with pytest.raises(ValueError):
        E.xcorr_normalised(np.zeros(8), np.zeros(9))
    # the library is still usable after errors
    assert np.allclose(E.fft(np.ones(8, dtype=np.complex64))[0], 8.0)


def test_maximum_workgroup_sizes(P):
    E = P.engine
    rng = np.random.default_rng(3)
    x = rng.standard_normal(8192 * 5 + 17).astype(np.float32)
    win = O.windows("Hanning", nwins=8192)
    M = (x.size - 8192) // 4096 + 1
    ref = O.welch_psd_stream(x, win, 8192, 4096, M, 1.0) * np.sum(win ** 2)
    np.testing.assert_allclose(E.welch_psd(x, win, 4096, M, detrend=True, sided=E.SIDED_TWO, scale=1.0), ref, rtol=2e-4,
                               atol=1e-6 * ref.max())
    w2 = O.windows("Hamming", nwins=4095)                         # largest fused chirp-z: L = 8192
    M2 = (x.size - 4095) // 1000 + 1
    ref2 = O.welch_psd_stream(x, w2, 4095, 1000, M2, 1.0) * np.sum(w2 ** 2)
    np.testing.assert_allclose(E.welch_psd(x, w2, 1000, M2, detrend=True, sided=E.SIDED_TWO, scale=1.0), ref2, rtol=2e-4,
                               atol=2e-6 * ref2.max())
    h = rng.standard_normal(4097)
    y = E.fir_filter(h, x, nfft=8192)                             # most taps one block can take
    assert np.max(np.abs(y - O.fftfilt(h.astype(np.float32).astype(np.float64), x.astype(np.float64)))) < 2e-4 * np.abs(y).max()


def test_class_methods_crosscorr_and_amplitudes(P):
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "pwelch_cfg1.npz"))
    t, x, y = g["t"], g["x"], g["y"]
    ft = P.fftanal(t, x, y, tbounds=[t[0], t[-2]], Navr=127, windowoverlap=0.5, windowfunction="Hanning", verbose=False)
    ft.pwelch()
    ft.crosscorr()
    ft.convert2amplitudes()
    # class path conjugation X conj(Y) (fft_analysis.py:1960) = conj of the function path's Y conj(X) (:393)
    np.testing.assert_allclose(ft.Pxy, np.conj(g["Pxy"]), rtol=2e-4, atol=1e-6 * np.abs(g["Pxy"]).max())
    np.testing.assert_allclose(ft.Lxx, g["info_Lxx"], rtol=2e-4, atol=1e-6 * g["info_Lxx"].max())
    np.testing.assert_allclose(ft.Rxx, g["info_Rxx"], rtol=0, atol=2e-4 * np.abs(g["info_Rxx"]).max())
    np.testing.assert_allclose(ft.lags, g["info_lags"], rtol=1e-12)
    # fftpwelch() == fft_pwelch
    ft2 = P.fftanal(t, x, y, tbounds=[t[0], t[-2]], Navr=127, windowoverlap=0.5, windowfunction="Hanning", verbose=False)
    ft2.fftpwelch()
    np.testing.assert_allclose(ft2.Pxx, g["Pxx"], rtol=2e-4, atol=1e-6 * np.abs(g["Pxx"]).max())
    assert ft2.fftinfo.Navr == 127 and hasattr(ft2, "Cxy2")
    # transforms on the object
    np.testing.assert_allclose(ft.ifft(ft.fft(x[:1024])), x[:1024], atol=1e-5 * np.abs(x).max())


def test_hilbert_edge_shapes(P):
    assert P.hilbert(np.array([1.0, 2.0, 3.0, 4.0])).shape == (4,)
    z = P.hilbert(np.ones((1, 16)))                               # squeeze like the reference
    assert z.shape == (16,)
    np.testing.assert_allclose(P.hilbert(np.array([1.0, -1.0])), O.hilbert(np.array([1.0, -1.0])), atol=1e-6)
    np.testing.assert_allclose(P.hilbert(np.arange(5.0)), O.hilbert(np.arange(5.0)), atol=2e-6)     # odd quirk
    zc = np.arange(8.0) + 1j * np.cos(np.arange(8.0))        # complex input: by linearity, like the reference's fft/ifft
    np.testing.assert_allclose(P.hilbert(zc), O.hilbert(zc), atol=5e-6)


def test_long_fft_two_pass_option_matches(monkeypatch):
    """SP_BIGFFT_2PASS=1 (strided two-pass long transform, an option that measured slower) == the default five-pass form"""
    from pyfft_amd import engine as E
    rng = np.random.default_rng(21)
    for n in (1 << 14, 1 << 17):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        a = E.fft(x)
        ai = E.ifft(a)
        monkeypatch.setenv("SP_BIGFFT_2PASS", "1")
        b = E.fft(x)
        bi = E.ifft(b)
        monkeypatch.delenv("SP_BIGFFT_2PASS")
        assert np.max(np.abs(a - b)) <= 2e-6 * np.abs(a).max()
        assert np.max(np.abs(bi - x)) <= 3e-6 * np.abs(x).max() and np.max(np.abs(ai - x)) <= 3e-6 * np.abs(x).max()


@pytest.mark.parametrize("lg", [20, 21, 22])
def test_long_fft_three_pass_matches_numpy(lg, monkeypatch):
    """the three-pass long transform (N = A B C, default from 2^20 points) against numpy, forward and inverse, and
    against the five-pass form"""
    from pyfft_amd import engine as E
    rng = np.random.default_rng(lg)
    n = 1 << lg
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    a = E.fft(x)
    assert np.max(np.abs(a - ref)) <= 4e-6 * np.abs(ref).max()
    assert np.max(np.abs(E.ifft(a) - x)) <= 4e-6 * np.abs(x).max()
    monkeypatch.setenv("SP_BIGFFT_5PASS", "1")
    b = E.fft(x)
    assert np.max(np.abs(a - b)) <= 3e-6 * np.abs(ref).max()


def test_long_hilbert_three_pass_fused_mask(monkeypatch):
    """2^20- and 2^21-point analytic signals: three-pass transforms with the mask fused into the inverse's first pass,
    against the oracle and against the five-pass form with the separate mask kernel"""
    from pyfft_amd import engine as E
    rng = np.random.default_rng(77)
    for n in (1 << 20, 1 << 21):
        u = rng.standard_normal((1, n))
        z = E.hilbert_rows(u, n)
        ref = O.hilbert(u)
        assert np.max(np.abs(z - ref)) <= 1e-4 * np.abs(ref).max()
        assert np.max(np.abs(z.real - u)) <= 1e-4 * np.abs(u).max()
        monkeypatch.setenv("SP_BIGFFT_5PASS", "1")
        z5 = E.hilbert_rows(u, n)
        monkeypatch.delenv("SP_BIGFFT_5PASS")
        assert np.max(np.abs(z - z5)) <= 2e-5 * np.abs(ref).max()


def test_edge_shapes_of_cog_frame_sum_spectral_filter_deriv(P):
    """tests/edge_probe.py: tiny / odd / Bluestein / multi-wave / long shapes of cog_frames, frame_sum,
    spectral_filter_rows, fft_deriv and cog against the oracle or numpy (41 cases)"""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("edge_probe", os.path.join(ROOT, "tests", "edge_probe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run() == []


def test_table_cache_eviction_keeps_live_tables(P):
    """with the device-table cache full, a miss inside a call must not free a table the same call already
    holds (window cached as a plain table, FFT(window) missing), nor the tables of a pending sp_welch_accum."""
    E = P.engine
    rng = np.random.default_rng(99)
    n, nfft, hop = 1 << 16, 4096, 2048
    M = (n - nfft) // hop + 1
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n) + (0.5 - 0.25j)).astype(np.complex64)
    win0 = O.windows("Hanning", nwins=nfft)
    ref = O.welch_psd_stream(x, win0, nfft, hop, M, 1.0)
    S2 = float(np.sum(win0 ** 2))
    xs = x[:8 * nfft]

    def churn(count, seed):
        # `count` distinct windows, each cached as one plain table
        for i in range(count):
            w = np.hanning(nfft) * (1.0 + 1e-3 * (seed + i))
            E.stft_frames(xs, w, nfft, 2, detrend=False, sided=E.SIDED_RAW)

    churn(63, 0)
    E.stft_frames(xs, win0, nfft, 2, detrend=False, sided=E.SIDED_RAW)      # win0 cached as a plain table: cache full (64)
    got = E.welch_psd(x, win0, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)   # hit on win0, miss on FFT(win0)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-6 * ref.max())
    # pending accumulate across 70 evicting calls
    s = E.welch_accum(x, win0, hop, M)
    churn(70, 1000)
    mean = s / n
    got2 = E.welch_finish(nfft, mean, M, sided=E.SIDED_TWO, scale=1.0 / S2)
    np.testing.assert_allclose(got2, ref, rtol=2e-4, atol=1e-6 * ref.max())


def test_device_tensor_argument_validation(P):
    """the device-tensor (mem=1) paths check dtype / shape / device like the numpy paths do"""
    import torch
    E = P.engine
    x = torch.zeros(4096, device="cuda")
    yc = torch.zeros(4096, dtype=torch.complex64, device="cuda")
    w = np.ones(256)
    with pytest.raises(TypeError):
        E.welch_csd(x, yc, w, 128, 4)                          # float32 x with complex64 y
    with pytest.raises(TypeError):
        E.welch_csd(x, np.zeros(4096, dtype=np.float32), w, 128, 4)
    with pytest.raises(ValueError):
        E.welch_csd(x, x[:100], w, 128, 4)                     # channel shorter than x
    with pytest.raises(ValueError):
        E.xcorr_normalised(x, x[:100])                         # would read past the end of x2
    with pytest.raises(TypeError):
        E.xcorr_normalised(yc, yc)
    with pytest.raises(TypeError):
        P.fft_analysis.psd(np.zeros(4096, dtype=np.complex128), 1.0)     # was: imaginary part silently dropped
    pxx, pyy, pxy = E.welch_csd(x + 1, (x + 2)[None, :], w, 128, 4, detrend=False)
    assert pxx.is_cuda and pyy.shape == (1, 128)


def test_unstable_biquad_is_refused():
    """the blocked scan raises the state map to powers up to n -- an unstable section (|p| > 1) would overflow them
    and NaN every output, where scipy.signal.lfilter returns valid early samples; both layers refuse with a clear error"""
    from pyfft_amd import engine as E
    from pyfft_amd import _ffi
    x = np.ones(4096, dtype=np.float32)
    with pytest.raises(ValueError, match="unstable"):
        E.biquad_filter([1.0, 0.0, 0.0], [1.0, -2.1, 1.1025], x)            # double pole at 1.05
    b = np.array([1.0, 0.0, 0.0]); a = np.array([1.0, -2.1, 1.1025]); y = np.empty_like(x)
    rc = _ffi.lib().sp_biquad(_ffi.ptr(b), _ffi.ptr(a), _ffi.ptr(x), x.size, _ffi.ptr(y), 0)
    assert rc != 0 and b"unstable" in _ffi.lib().sp_last_error()
    # marginally stable (|p| = 1: an integrator) and a stable resonator still run
    yi = E.biquad_filter([1.0, 0.0, 0.0], [1.0, -1.0, 0.0], x)
    np.testing.assert_allclose(yi, np.arange(1, 4097, dtype=np.float64), rtol=1e-6)
    ys = E.biquad_filter([1.0, 0.0, 0.0], [1.0, -1.8, 0.9], x)
    assert np.all(np.isfinite(ys))


def test_sharded_psd_refuses_long_segments_cleanly():
    """the sharded / split Welch entry points have no long-segment path (nfft > 8192: few, large frames --
    nothing to shard); they must say so instead of failing somewhere inside (from dist.welch_psd_sharded too)"""
    from pyfft_amd import engine as E
    from pyfft_amd._ffi import SpectralError
    from pyfft_amd.dist import shard_plan, welch_psd_sharded
    nfft, hop = 16384, 8192
    x = (np.random.default_rng(0).standard_normal(nfft * 5)).astype(np.float32)
    win = np.hanning(nfft)
    plan = shard_plan(x.size, nfft, hop, 1, 0)
    for fn in (lambda: welch_psd_sharded(x, win, plan, 1.0),
               lambda: E.welch_export(x, win, hop, plan.frames),
               lambda: E.welch_accum(x, win, hop, plan.frames),
               lambda: E.welch_apply(np.zeros(5 * nfft + 8), win, plan.frames)):
        with pytest.raises(SpectralError, match="not sharded"):
            fn()
    # the one-GPU call takes the long-segment path
    p = E.welch_psd(x, win, hop, plan.frames, detrend=True, sided=E.SIDED_ONE, scale=1.0)
    assert p.shape == (nfft // 2,) and np.all(np.isfinite(p))


def test_csd_epilogue_is_scale_invariant():
    """the epilogue's inverse transforms run in float32 -- rows are normalised first, so spectra of a 1e-10-volt
    signal (P ~ 1e-26: float32 denormals when cast as they stand) or of a 1e+12 one give the O(1) case's correlations times
    the scale, and the same (scale-free) corrcoef"""
    from pyfft_amd import engine as E
    rng = np.random.default_rng(2)
    nfft, nb, nch = 512, 256, 3
    pxx = rng.random(nb) + 0.1
    pyy = rng.random((nch, nb)) + 0.1
    pxy = (rng.standard_normal((nch, nb)) + 1j * rng.standard_normal((nch, nb))) * 0.2
    base = E.csd_epilogue(pxx, pyy, pxy, nfft, True, 1.5)
    for sc in (1e-26, 1e+24):
        r = E.csd_epilogue(pxx * sc, pyy * sc, pxy * sc, nfft, True, 1.5)
        for k in ("Rxx", "Ryy", "Rxy"):
            assert np.all(np.isfinite(r[k]))
            np.testing.assert_allclose(np.asarray(r[k]) / sc, base[k], rtol=1e-5, atol=1e-6 * np.abs(base[k]).max(), err_msg=k)
        np.testing.assert_allclose(r["corrcoef"], base["corrcoef"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r["Cxy2"], base["Cxy2"], rtol=1e-9)
