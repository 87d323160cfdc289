"""GPU parity of the HIP kernels (through the C ABI, via pyfft_amd.engine) against the CPU oracle on the same
seeded inputs and against the committed golden fixtures.  float32 device math vs float64 oracle:
tolerances are the ones BASELINE.md / SURVEY.md section 8d state, written next to each check."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O


@pytest.fixture(scope="module")
def E():
    from pyfft_amd import engine
    from pyfft_amd import _ffi
    _ffi.init()
    return engine


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b)))


# ---------------------------------------------------------------- A7 batched FFT
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_fft_forward_inverse(E, n):
    rng = np.random.default_rng(n)
    batch = 37 if n <= 1024 else 5
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    X = E.fft(x)
    ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    # forward vs float64: <= 2e-6*sqrt(log2 N) relative to ||X||_inf
    assert relerr(X, ref) <= 2e-6 * max(1.0, np.sqrt(np.log2(n)))
    xr = E.ifft(X)
    # round trip: max|ifft(fft(x)) - x| / max|x| <= 5e-6
    assert relerr(xr, x) <= 5e-6
    # inverse alone vs float64
    assert relerr(E.ifft(x), np.fft.ifft(x.astype(np.complex128), axis=-1)) <= 2e-6 * max(1.0, np.sqrt(np.log2(n)))


@pytest.mark.parametrize("n", [3, 5, 6, 7, 12, 100, 255, 777, 1000, 1023, 2184, 3640, 4095])
def test_fft_arbitrary_length(E, n):
    """Bluestein path (lengths that are not powers of two, e.g. the reference's nwins 1023 / 2184 / 3640)."""
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((4, n)) + 1j * rng.standard_normal((4, n))).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    assert relerr(E.fft(x), ref) <= 4e-6 * max(1.0, np.sqrt(np.log2(n)))
    assert relerr(E.ifft(E.fft(x)), x) <= 1e-5
    assert relerr(E.ifft(x), np.fft.ifft(x.astype(np.complex128), axis=-1)) <= 4e-6 * max(1.0, np.sqrt(np.log2(n)))


@pytest.mark.parametrize("n", [1 << 14, 1 << 15, 1 << 17, 1 << 20, 10000, 50001, 100003, 131071])
def test_fft_long(E, n):
    """longer than one workgroup: multi-pass four-step FFT (powers of two) and chirp-z on top of it (other lengths)"""
    rng = np.random.default_rng(n % 9973)
    x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    X = E.fft(x)
    assert relerr(X, ref) <= 4e-6 * np.sqrt(np.log2(n))
    assert relerr(E.ifft(X), x) <= 2e-5
    assert relerr(E.ifft(x), np.fft.ifft(x.astype(np.complex128), axis=-1)) <= 4e-6 * np.sqrt(np.log2(n))
    # a pure tone lands in one bin (phase accuracy of the long twiddles)
    k0 = n // 3
    tone = np.exp(2j * np.pi * k0 * np.arange(n) / n).astype(np.complex64)[None, :]
    T = E.fft(tone)[0]
    assert abs(T[k0] - n) < 2e-4 * n and np.max(np.abs(np.delete(T, k0))) < 2e-4 * n


def test_fft_linearity_and_impulse(E):
    n = 4096
    x = np.zeros((3, n), dtype=np.complex64)
    x[0, 0] = 1.0          # impulse -> all ones
    x[1, 1] = 1.0          # shifted impulse -> exp(-2 pi i k/N)
    x[2, :] = 1.0          # constant -> N delta
    X = E.fft(x)
    k = np.arange(n)
    assert np.max(np.abs(X[0] - 1.0)) < 1e-6
    assert np.max(np.abs(X[1] - np.exp(-2j * np.pi * k / n))) < 2e-6
    assert abs(X[2, 0] - n) < 1e-2 and np.max(np.abs(X[2, 1:])) < 1e-2


def test_fft_axis_and_pad(E):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((64, 6)).astype(np.float32)
    X = E.fft(x, n=128, axis=0)
    assert relerr(X, np.fft.fft(x.astype(np.float64), n=128, axis=0)) < 5e-6


# ---------------------------------------------------------------- A3+A4 Welch PSD (class path)
@pytest.mark.parametrize("tag", ["c64_2e16_n4096", "c64_2e14_n1024"])
def test_welch_psd_golden_complex(E, tag):
    g = load_golden("welch_class_" + tag)
    x = g["x"]
    nfft, nov, M = int(g["nwins"]), int(g["noverlap"]), int(g["Navr"])
    win = O.windows("Hanning", nwins=nfft)
    scale = 1.0 / (float(g["Fs"]) * float(g["S2"]))
    P = E.welch_psd(x, win, nfft - nov, M, detrend=True, sided=E.SIDED_TWO, scale=scale)
    ref = g["Pxx"].real
    # Welch PSD bins: rtol 2e-4, atol 1e-6*max(Pxx)
    np.testing.assert_allclose(P, ref, rtol=2e-4, atol=1e-6 * ref.max())
    # natural-order variant is the un-shifted spectrum
    Praw = E.welch_psd(x, win, nfft - nov, M, detrend=True, sided=E.SIDED_RAW, scale=scale)
    np.testing.assert_allclose(np.fft.fftshift(Praw), P, rtol=1e-12)
    # detrend off + explicit mean
    P0 = E.welch_psd(x, win, nfft - nov, M, detrend=False, sided=E.SIDED_TWO, scale=scale)
    ref0 = O.welch_psd_stream(x, win, nfft, nfft - nov, M, float(g["Fs"]), detrend_style=0)
    np.testing.assert_allclose(P0, ref0, rtol=2e-4, atol=1e-6 * ref0.max())
    Pm = E.welch_psd(x, win, nfft - nov, M, detrend=True, sided=E.SIDED_TWO, scale=scale,
                     mean_value=complex(x.astype(np.complex128).mean()))
    np.testing.assert_allclose(Pm, ref, rtol=2e-4, atol=1e-6 * ref.max())


@pytest.mark.parametrize("wname", ["Hamming", "SFT3F"])
def test_welch_psd_real_onesided(E, wname):
    g = load_golden("welch_class_real_" + wname)
    x = g["x"]
    nfft, nov, M = int(g["nwins"]), int(g["noverlap"]), int(g["Navr"])
    win = O.windows(wname, nwins=nfft)
    S2 = np.sum(win ** 2)
    P = E.welch_psd(x, win, nfft - nov, M, detrend=True, sided=E.SIDED_ONE, scale=1.0 / (float(g["Fs"]) * S2))
    ref = g["Pxx"].real
    np.testing.assert_allclose(P, ref, rtol=2e-4, atol=1e-6 * ref.max())


@pytest.mark.parametrize("nfft,hop", [(64, 16), (256, 256), (512, 100), (2048, 512), (8192, 4096), (1024, 1)])
def test_welch_psd_shapes(E, nfft, hop):
    """ragged run partition, hop that is not a divisor, single frame, real and complex input."""
    rng = np.random.default_rng(nfft + hop)
    nsig = nfft + hop * 37 + 5
    for cplx in (False, True):
        x = rng.standard_normal(nsig) + (1j * rng.standard_normal(nsig) if cplx else 0) + 0.5
        x = x.astype(np.complex64 if cplx else np.float32)
        M = (nsig - nfft) // hop + 1
        win = O.windows("Nuttall4c", nwins=nfft)
        for frames in (M, 1):
            P = E.welch_psd(x, win, hop, frames, detrend=True, sided=E.SIDED_TWO, scale=1.0)
            ref = O.welch_psd_stream(x, win, nfft, hop, frames, 1.0) * np.sum(win ** 2)
            np.testing.assert_allclose(P, ref, rtol=2e-4, atol=2e-6 * ref.max())


@pytest.mark.parametrize("nfft,hop", [(1023, 511), (1001, 250), (3640, 1820), (30, 7), (4095, 4095)])
def test_welch_psd_arbitrary_length(E, nfft, hop):
    """non power-of-two and odd segment lengths, one- and two-sided (odd n doubles the last kept bin too)."""
    rng = np.random.default_rng(nfft)
    nsig = nfft + hop * 23 + 3
    x = (rng.standard_normal(nsig) + 0.7).astype(np.float32)
    M = (nsig - nfft) // hop + 1
    win = O.windows("Hamming", nwins=nfft)
    xd = x.astype(np.float64) - x.astype(np.float64).mean()
    idx = (np.arange(M) * hop)[:, None] + np.arange(nfft)[None, :]
    P2 = (np.abs(np.fft.fft(win * xd[idx], axis=-1)) ** 2).mean(axis=0)
    got2 = E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    np.testing.assert_allclose(got2, np.fft.fftshift(P2), rtol=2e-4, atol=1e-6 * P2.max())
    nny = (nfft + 1) // 2 if nfft % 2 else nfft // 2
    P1 = P2[:nny].copy()
    P1[1:-1] *= 2
    if nfft % 2:
        P1[-1] *= 2
    got1 = E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, scale=1.0)
    assert got1.shape == (nny,)
    np.testing.assert_allclose(got1, P1, rtol=2e-4, atol=1e-6 * P1.max())


def test_welch_linear_detrend(E):
    rng = np.random.default_rng(5)
    n, nfft, hop = 50000, 1024, 512
    k = np.arange(n)
    x = (np.sin(0.2 * k) + 0.3 * rng.standard_normal(n) + 2.0 + 1e-4 * k).astype(np.float32)
    M = (n - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    xd = O.detrend(x.astype(np.float64), -1)
    ref = O.welch_psd_stream(xd, win, nfft, hop, M, 1.0, detrend_style=0) * np.sum(win ** 2)
    got = E.welch_psd(x, win, hop, M, detrend="linear", sided=E.SIDED_TWO, scale=1.0)
    np.testing.assert_allclose(got, ref, rtol=5e-4, atol=3e-6 * ref.max())
    z = (x + 1j * (0.5 * x[::-1] - 3e-5 * k)).astype(np.complex64)
    zd = O.detrend(z.astype(np.complex128).real, -1) + 1j * O.detrend(z.astype(np.complex128).imag, -1)
    refz = O.welch_psd_stream(zd, win, nfft, hop, M, 1.0, detrend_style=0) * np.sum(win ** 2)
    gotz = E.welch_psd(z, win, hop, M, detrend="linear", sided=E.SIDED_TWO, scale=1.0)
    np.testing.assert_allclose(gotz, refz, rtol=5e-4, atol=3e-6 * refz.max())


def test_welch_carry_vs_generic_kernel(E):
    """the register-carried metric kernel and the generic kernel are the same function"""
    import os
    rng = np.random.default_rng(12)
    for nfft, hop in ((4096, 2048), (2048, 512), (1024, 1024), (256, 64)):
        nsig = nfft + hop * 301
        x = (rng.standard_normal(nsig) + 1j * rng.standard_normal(nsig) + (0.2 + 0.1j)).astype(np.complex64)
        M = (nsig - nfft) // hop + 1
        win = O.windows("Hanning", nwins=nfft)
        a = E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
        assert E.profile_last_kernel().startswith("k_welch_carry")
        os.environ["SP_WELCH_GENERIC"] = "1"
        try:
            b = E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
            assert E.profile_last_kernel() == "k_welch"
        finally:
            del os.environ["SP_WELCH_GENERIC"]
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-7 * b.max())
        ref = O.welch_psd_stream(x, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
        np.testing.assert_allclose(a, ref, rtol=2e-4, atol=1e-6 * ref.max())


@pytest.mark.parametrize("nfft,hop,cplx", [(4096, 2048, True), (4096, 1024, True), (2048, 2048, False),
                                           (1024, 512, False), (256, 64, True), (8192, 4096, True)])
def test_welch_onepass_detrend(E, nfft, hop, cplx):
    """global-mean detrend done in one pass (mean estimate + exact epilogue correction) == two-pass == oracle,
    including a mean far larger than the signal, a drifting mean, and very few frames."""
    import os
    rng = np.random.default_rng(nfft + hop)
    for nframes_extra, offset, drift in ((301, 0.3, 0.0), (57, 50.0, 0.0), (130, 2.0, 3.0), (0, 1.0, 0.0), (1, -4.0, 0.0)):
        nsig = nfft + hop * nframes_extra + 11
        k = np.arange(nsig) / nsig
        x = rng.standard_normal(nsig) + offset + drift * k
        if cplx:
            x = x + 1j * (rng.standard_normal(nsig) - 0.5 * offset + 2 * drift * k ** 2)
        x = x.astype(np.complex64 if cplx else np.float32)
        M = (nsig - nfft) // hop + 1
        win = O.windows("Hanning", nwins=nfft)
        one = E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
        # complex input: the one-pass carry kernel; real input: the two-frames-per-transform kernel (mean pre-pass)
        if cplx:
            assert "onepass" in E.profile_last_kernel()
        elif M >= 2:
            assert E.profile_last_kernel() == "k_welch_rp"
        os.environ["SP_WELCH_TWOPASS"] = "1"
        os.environ["SP_NO_REALPAIR"] = "1"
        try:
            two = E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
            assert "onepass" not in E.profile_last_kernel() and E.profile_last_kernel() != "k_welch_rp"
        finally:
            del os.environ["SP_WELCH_TWOPASS"]
            del os.environ["SP_NO_REALPAIR"]
        ref = O.welch_psd_stream(x, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
        np.testing.assert_allclose(two, ref, rtol=2e-4, atol=1e-6 * ref.max())
        np.testing.assert_allclose(one, ref, rtol=2e-4, atol=1e-6 * ref.max())


def test_welch_accum_finish_sharded(E):
    """two shards of one stream, each accumulated separately and finished with the GLOBAL mean, add up to the
    single-pass PSD of the whole stream (what pyfft_amd.dist does across GPUs)."""
    rng = np.random.default_rng(99)
    nfft, hop = 4096, 2048
    nsig = nfft + hop * 200
    x = (rng.standard_normal(nsig) + 1j * rng.standard_normal(nsig) + (1.5 - 0.7j)).astype(np.complex64)
    x[: nsig // 2] += 2.0                     # the two halves have different means
    M = (nsig - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    ref = O.welch_psd_stream(x, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    Ma = M // 2
    xa = x[: (Ma - 1) * hop + nfft]           # shard A: frames [0, Ma) (+ halo)
    xb = x[Ma * hop:]                         # shard B: frames [Ma, M)
    own_a, own_b = Ma * hop, nsig - Ma * hop
    sa = E.welch_accum(xa, win, hop, Ma, nmean=own_a)
    np.testing.assert_allclose(sa[0] + 1j * sa[1], x[:own_a].astype(np.complex128).sum(), rtol=1e-6)
    mean = None
    # finish A needs the global mean: second shard's sum first (only one accumulation may be pending)
    sb_direct = x[Ma * hop:].astype(np.complex128).sum()
    gm = (sa[0] + 1j * sa[1] + sb_direct) / nsig
    pa = E.welch_finish(nfft, np.array([gm.real, gm.imag]), M, sided=E.SIDED_TWO, scale=1.0)
    sb = E.welch_accum(xb, win, hop, M - Ma, nmean=own_b)
    np.testing.assert_allclose(sb[0] + 1j * sb[1], sb_direct, rtol=1e-6)
    pb = E.welch_finish(nfft, np.array([gm.real, gm.imag]), M, sided=E.SIDED_TWO, scale=1.0)
    np.testing.assert_allclose(pa + pb, ref, rtol=2e-4, atol=1e-6 * ref.max())


@pytest.mark.parametrize("sided", ["two", "one", "raw"])
def test_welch_export_apply_sharded(E, sided):
    """one-collective form: the states of three unequal shards (different means, a mean 30x the signal) add up, and
    sp_welch_apply turns the sum into the single-pass PSD of the whole stream; numpy and device inputs"""
    import torch
    sd = {"two": E.SIDED_TWO, "one": E.SIDED_ONE, "raw": E.SIDED_RAW}[sided]
    rng = np.random.default_rng(123)
    nfft, hop = 2048, 512
    nsig = nfft + hop * 301
    x = (rng.standard_normal(nsig) + 1j * rng.standard_normal(nsig) + (30.0 - 11.0j)).astype(np.complex64)
    x[: nsig // 3] += 2.0
    x[-nsig // 5:] -= 1.0j
    M = (nsig - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    ref = E.welch_psd(x, win, hop, M, detrend=True, sided=sd, scale=3.0)
    cuts = [0, M // 4, M // 4 + M // 3, M]
    states = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        xs = x[a * hop: (b - 1) * hop + nfft] if b < M else x[a * hop:]
        own = (b - a) * hop if b < M else nsig - a * hop
        if a == cuts[1]:
            st = E.welch_export(torch.from_numpy(xs).cuda(), win, hop, b - a, nmean=own).cpu().numpy()
        else:
            st = E.welch_export(xs, win, hop, b - a, nmean=own)
        assert st.shape == (5 * nfft + 8,)
        states.append(st)
    tot = np.sum(states, axis=0)
    assert tot[5 * nfft + 5] == M and tot[5 * nfft + 6] == nsig
    p = E.welch_apply(tot, win, M, sided=sd, scale=3.0)
    np.testing.assert_allclose(p, ref, rtol=2e-4, atol=2e-6 * ref.max())
    # against the oracle too (two-sided)
    if sided == "two":
        oref = O.welch_psd_stream(x, win, nfft, hop, M, 1.0) * np.sum(win ** 2) * 3.0
        np.testing.assert_allclose(p, oref, rtol=2e-4, atol=2e-6 * oref.max())


def test_dist_world1_on_device(E):
    """pyfft_amd.dist with device tensors and no process group (world = 1) == welch_psd"""
    import torch
    from pyfft_amd.dist import shard_plan, welch_psd_sharded
    rng = np.random.default_rng(7)
    nfft, hop = 4096, 2048
    nsig = nfft + hop * 150 + 9
    x = (rng.standard_normal(nsig) + 1j * rng.standard_normal(nsig) + (0.3 + 0.2j)).astype(np.complex64)
    win = O.windows("Hanning", nwins=nfft)
    plan = shard_plan(nsig, nfft, hop, 1, 0)
    assert plan.nsamples == nsig and plan.own_samples == nsig
    xt = torch.from_numpy(x).cuda()
    p = welch_psd_sharded(xt, win, plan, scale=1.0)
    assert p.is_cuda and p.dtype == torch.float64
    ref = O.welch_psd_stream(x, win, nfft, hop, plan.frames_total, 1.0) * np.sum(win ** 2)
    np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=2e-4, atol=1e-6 * ref.max())
    # device-tensor entry points of the other kernels
    X = E.fft(xt[: 8 * nfft].reshape(8, nfft))
    assert relerr(X.cpu().numpy(), np.fft.fft(x[: 8 * nfft].reshape(8, nfft).astype(np.complex128), axis=-1)) < 6e-6
    h = E.hilbert_rows(xt.real[: 4 * 1024].reshape(4, 1024).contiguous(), 1024)
    assert relerr(h.cpu().numpy(), O.hilbert(x.real[: 4 * 1024].reshape(4, 1024).astype(np.float64))) < 1e-4


def test_real_pair_packing_equals_plain_path(E):
    """real input: the two-frames-per-transform kernels (Welch power, STFT) against the one-frame kernels and the
    oracle; odd and even frame counts, one- and two-sided, non power-of-two Welch length"""
    import os
    rng = np.random.default_rng(21)
    for nfft, hop, extra in ((1024, 256, 41), (2048, 512, 40), (256, 256, 7), (1000, 500, 13)):
        nsig = nfft + hop * extra + 3
        x = (rng.standard_normal(nsig) + 0.4).astype(np.float32)
        M = (nsig - nfft) // hop + 1
        win = O.windows("Hamming", nwins=nfft)
        ref = O.welch_psd_stream(x, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
        for sided in (E.SIDED_TWO, E.SIDED_ONE):
            a = E.welch_psd(x, win, hop, M, detrend=True, sided=sided, scale=1.0)
            assert E.profile_last_kernel() == "k_welch_rp"
            os.environ["SP_NO_REALPAIR"] = "1"
            try:
                b = E.welch_psd(x, win, hop, M, detrend=True, sided=sided, scale=1.0)
                assert E.profile_last_kernel() != "k_welch_rp"
            finally:
                del os.environ["SP_NO_REALPAIR"]
            np.testing.assert_allclose(a, b, rtol=3e-5, atol=1e-7 * b.max())
        np.testing.assert_allclose(E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0), ref, rtol=2e-4,
                                   atol=1e-6 * ref.max())
        if nfft & (nfft - 1):
            continue
        for sided in (E.SIDED_ONE, E.SIDED_RAW):
            Xa, pa = E.stft_frames(x, win, hop, M, detrend=True, sided=sided, amp_scale=0.5, want_pseg=True)
            os.environ["SP_NO_REALPAIR"] = "1"
            try:
                Xb, pb = E.stft_frames(x, win, hop, M, detrend=True, sided=sided, amp_scale=0.5, want_pseg=True)
            finally:
                del os.environ["SP_NO_REALPAIR"]
            assert np.max(np.abs(Xa - Xb)) <= 2e-6 * np.abs(Xb).max()
            np.testing.assert_allclose(pa, pb, rtol=1e-5)
        Pa, _ = E.stft_frames(x, win, hop, M, detrend=False, sided=E.SIDED_RAW, amp_scale=1.0, power=True, bin_major=True)
        xd = x.astype(np.float64)
        idx = (np.arange(M) * hop)[:, None] + np.arange(nfft)[None, :]
        refp = (np.abs(np.fft.fft(win * xd[idx], axis=-1)) ** 2).T
        np.testing.assert_allclose(Pa, refp, rtol=2e-4, atol=1e-6 * refp.max())


def test_welch_errors(E):
    from pyfft_amd._ffi import SpectralError
    x = np.zeros(1000, dtype=np.float32)
    with pytest.raises(SpectralError):
        E.welch_psd(x, np.ones(512), 256, 10)            # frames run past the signal


# ---------------------------------------------------------------- A5 fft_pwelch core
def test_welch_csd_cfg1(E):
    g = load_golden("pwelch_cfg1")
    t, x, y = g["t"], g["x"], g["y"]
    i0, i1 = [int(v) for v in g["info_ibnds"]]
    nfft, nov, M = int(g["info_nwins"]), int(g["info_noverlap"]), int(g["info_Navr"])
    win = O.windows("Hanning", nwins=nfft)
    scale = 1.0 / (float(g["info_Fs"]) * float(g["info_S2"]))
    pxx, pyy, pxy = E.welch_csd(x[i0:i1], y[i0:i1][None, :], win, nfft - nov, M, detrend=True, sided=E.SIDED_ONE,
                                scale=scale)
    np.testing.assert_allclose(pxx, g["Pxx"].real, rtol=2e-4, atol=1e-6 * g["Pxx"].real.max())
    np.testing.assert_allclose(pyy[0], g["Pyy"].real, rtol=2e-4, atol=1e-6 * g["Pyy"].real.max())
    np.testing.assert_allclose(pxy[0], g["Pxy"], rtol=2e-4, atol=1e-6 * np.abs(g["Pxy"]).max())


def test_welch_csd_multichannel_twosided(E):
    rng = np.random.default_rng(77)
    n, nfft, hop = 20000, 512, 256
    k = np.arange(n)
    x = (np.sin(0.3 * k) + 0.2 * rng.standard_normal(n) + 1.0).astype(np.float32)
    y = np.stack([0.5 * np.sin(0.3 * k - 0.7) + 0.1 * rng.standard_normal(n),
                  rng.standard_normal(n) - 2.0,
                  0.25 * x + 0.1 * rng.standard_normal(n)]).astype(np.float32)
    M = (n - nfft) // hop + 1
    win = O.windows("Hamming", nwins=nfft)
    pxx, pyy, pxy = E.welch_csd(x, y, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    xd = x.astype(np.float64) - x.astype(np.float64).mean()
    yd = y.astype(np.float64) - y.astype(np.float64).mean(axis=1, keepdims=True)
    idx = (np.arange(M) * hop)[:, None] + np.arange(nfft)[None, :]
    X = np.fft.fftshift(np.fft.fft(win * xd[idx], axis=-1), axes=-1)
    np.testing.assert_allclose(pxx, (np.abs(X) ** 2).mean(axis=0), rtol=2e-4, atol=1e-6 * (np.abs(X) ** 2).mean(axis=0).max())
    for c in range(3):
        Y = np.fft.fftshift(np.fft.fft(win * yd[c][idx], axis=-1), axes=-1)
        ryy = (np.abs(Y) ** 2).mean(axis=0)
        rxy = (Y * np.conj(X)).mean(axis=0)
        np.testing.assert_allclose(pyy[c], ryy, rtol=2e-4, atol=1e-6 * ryy.max())
        np.testing.assert_allclose(pxy[c], rxy, rtol=2e-4, atol=2e-6 * np.abs(rxy).max())


def _csd_per_bin_excess(G, ref, rtol=2e-4, atol_rel=1e-6):
    """worst |G_ij[k] - ref_ij[k]| in units of its allowance rtol * sqrt(ref_ii[k] ref_jj[k]) + atol_rel * max_k sqrt(ref_ii[k]
    ref_jj[k]): the float32 tolerance of SURVEY section 8(d) (rtol 2e-4, atol 1e-6 max) applied per bin and per pair -- the
    geometric mean of the two auto-spectra is the natural scale of a cross-spectrum (|G_ij| <= it), so weakly coherent pairs
    and quiet bins are held to their own level, not to the global peak.  <= 1 passes."""
    d = np.sqrt(np.abs(np.einsum("kii->ki", ref).real))                      # [nb, nch]
    gm = d[:, :, None] * d[:, None, :]                                       # sqrt(ref_ii ref_jj)  [nb, nch, nch]
    allow = rtol * gm + atol_rel * gm.max(axis=0, keepdims=True)
    return float(np.max(np.abs(G - ref) / allow))


def _coloured_record(nch, nsig, seed, line_db=None, weak_pair=(3, 7), weak_gamma2=0.01):
    """real multi-channel record for the cfg5 parity tests: every channel an independent AR(1) ("red", ~45 dB between the
    spectrum's ends) floor; one pair shares a weak common component (mean-squared coherence weak_gamma2); optionally a line
    `line_db` dB above the floor in every channel (a different amplitude and phase per channel), a fraction of a bin off
    centre so that it leaks"""
    from scipy.signal import lfilter
    rng = np.random.default_rng(seed)
    x = np.empty((nch, nsig))
    for c in range(nch):
        x[c] = lfilter([1.0], [1.0, -0.97], rng.standard_normal(nsig))
    a2 = np.sqrt(weak_gamma2) / (1.0 - np.sqrt(weak_gamma2))                 # gamma^2 = (a2 / (1 + a2))^2
    common = lfilter([1.0], [1.0, -0.97], rng.standard_normal(nsig)) * np.sqrt(a2)
    for c in weak_pair:
        x[c] += common
    if line_db is not None:
        k = np.arange(nsig)
        # floor PSD of the AR(1) process at f = 0.237: 1 / |1 - 0.97 e^{-i w}|^2; a sinusoid of amplitude A in a Hann-windowed
        # periodogram stands A^2 S1^2 / (4 S2) over a floor of that height
        w = 2 * np.pi * 0.237
        floor = 1.0 / abs(1.0 - 0.97 * np.exp(-1j * w)) ** 2
        for c in range(nch):
            amp = np.sqrt(floor * 10.0 ** (line_db / 10.0) * 4.0 * 1.5 / 256.0) * (1.0 + 0.1 * c)   # (nfft 256: S1^2/S2 = N/1.5)
            x[c] += amp * np.cos(w * k + 0.4 * c)
        x += 0.3
    return x.astype(np.float32)


@pytest.mark.parametrize("line_db", [None, 80.0])
@pytest.mark.parametrize("nch", [16, 64])
def test_csd_matrix_per_bin_parity_dynamic_range(E, nch, line_db):
    """cfg5 parity per bin and pair: coloured floor, a weakly coherent pair (gamma^2 ~ 0.01), with and
    without a line 80 dB above the floor, 2200 frames = 1100 frame pairs so that the TWO-piece bf16 contraction
    (k_csdm_bf16<4>, operands rounded to 16 significant bits) is the one that runs; against oracle.csd_matrix (float64,
    generalises fft_analysis.py:387-393).  Also forced three-piece and float32-MFMA forms."""
    import os
    nfft, hop, M = 256, 128, 2200
    nsig = (M - 1) * hop + nfft
    x = _coloured_record(nch, nsig, seed=17 + nch, line_db=line_db)
    win = O.windows("Hanning", nwins=nfft)
    ref = O.csd_matrix(x.astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    d = np.abs(np.einsum("kii->ki", ref).real)
    if line_db is not None:
        assert d[:, 0].max() / np.median(d[:, 0]) > 1e7                      # the record really spans > 70 dB
    g2 = np.abs(ref[:, 3, 7]) ** 2 / (d[:, 3] * d[:, 7])
    if line_db is None:
        assert 0.003 < np.median(g2) < 0.03                                  # the weak pair is weak (and not lost)
    worst = {}
    for tag, env in (("two_piece", {}), ("three_piece", {"SP_CSDM_SPLIT3": "1"}), ("fp32_mfma", {"SP_CSDM_FP32": "1"})):
        os.environ.update(env)
        try:
            G = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
        finally:
            for k in env:
                del os.environ[k]
        worst[tag] = _csd_per_bin_excess(G, ref)
    assert max(worst.values()) <= 1.0, worst


@pytest.mark.parametrize("nch,nfft,hop,nsig", [(5, 256, 128, 6000), (64, 1024, 512, 20000), (70, 512, 256, 9000),
                                                (3, 1000, 300, 7000), (3, 32, 16, 819), (64, 64, 32, 10688),
                                                (2, 8192, 4096, 40960)])
def test_csd_matrix(E, nch, nfft, hop, nsig):
    """cfg5 shape (reduced): common component with per-channel gain/delay + independent noise; full nch x nch matrix,
    including more than one 64-channel block, a ragged last frame chunk and a non power-of-two segment"""
    rng = np.random.default_rng(nch * nfft)
    k = np.arange(nsig)
    common = np.sin(0.13 * k) + 0.5 * rng.standard_normal(nsig)
    x = np.stack([(0.3 + 0.05 * c) * np.roll(common, c % 7) + rng.standard_normal(nsig) + 0.1 * c
                  for c in range(nch)]).astype(np.float32)
    M = (nsig - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    G = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    ref = O.csd_matrix(x.astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    assert G.shape == ref.shape == (nfft // 2 + 1, nch, nch)
    assert _csd_per_bin_excess(G, ref) <= 1.0            # per bin and pair, not relative to the global peak
    # Hermitian in (i, j), real non-negative diagonal
    assert np.max(np.abs(G - np.conj(np.swapaxes(G, 1, 2)))) <= 1e-6 * np.abs(G).max()
    # diagonal == Welch PSD of each channel (rfft layout)
    p0 = E.welch_psd(x[0], win, hop, M, detrend=True, sided=E.SIDED_RAW, scale=1.0)[: nfft // 2 + 1]
    np.testing.assert_allclose(G[:, 0, 0].real, p0, rtol=1e-4, atol=1e-6 * p0.max())


@pytest.mark.parametrize("nch,nfft,hop,nsig", [(64, 4096, 2048, 4096 + 2048 * 300), (37, 512, 256, 512 + 256 * 1001),
                                                (64, 1024, 256, 1024 + 256 * 77)])
def test_csd_matrix_bf16_split_is_float32_accurate(E, nch, nfft, hop, nsig):
    """the contraction on the bf16 matrix cores (k_csdm_bf16: x = h + m + l, six piece products) against the float32-MFMA
    form (SP_CSDM_FP32=1) and the float64 oracle: the split must not cost accuracy.  Odd frame counts (a half-filled last
    pair), fewer than 64 channels (masked channel slots) and few bin groups (frame-pair slices added atomically)"""
    import os
    rng = np.random.default_rng(nch + nfft)
    k = np.arange(nsig)
    common = np.sin(0.11 * k) + rng.standard_normal(nsig)
    x = np.stack([(0.5 + 0.03 * c) * np.roll(common, c % 5) + 0.3 * rng.standard_normal(nsig) + 0.2 * c
                  for c in range(nch)]).astype(np.float32)
    M = (nsig - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    G = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    os.environ["SP_CSDM_FP32"] = "1"
    try:
        G32 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    finally:
        del os.environ["SP_CSDM_FP32"]
    ref = O.csd_matrix(x.astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    peak = np.abs(ref).max()
    e16, e32 = np.max(np.abs(G - ref)) / peak, np.max(np.abs(G32 - ref)) / peak
    assert e16 <= 2e-5 and e16 <= 3.0 * e32 + 1e-6, (e16, e32)
    assert np.max(np.abs(G - np.conj(np.swapaxes(G, 1, 2)))) <= 1e-6 * peak


def test_csd_matrix_packed_spectra_path(E):
    """nfft 4096 at 50 % overlap: the spectra stage runs on the pipeline of specialised waves and writes the PACKED pair
    spectra; the contraction runs on them and the mirror combination G[k] = (H[k] + conj H[N-k]) / 2 is taken once on the
    sums.  Against the per-frame spectra path (SP_CSDM_NOPIPESPEC=1) and the oracle; odd frame count (a lone last frame), odd
    pair count (a lone last pair), fewer than 64 channels"""
    import os
    # (tail: samples after the last frame, which count for the mean; dc: per-channel offsets, so that the one-pass mean
    # detrend's correction terms are far above the comparison threshold)
    for nch, M, seed, tail, dc in ((64, 301, 77, 0, 0.1), (64, 303, 78, 777, 3.0), (20, 830, 79, 5000, -2.0)):
        rng = np.random.default_rng(seed)
        nfft, hop = 4096, 2048
        nsig = (M - 1) * hop + nfft + tail
        x = (rng.standard_normal((nch, nsig)) + 0.7 * rng.standard_normal(nsig)[None, :]
             + dc * (1 + np.arange(nch))[:, None] / nch).astype(np.float32)
        win = O.windows("Hanning", nwins=nfft)
        G1 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
        os.environ["SP_CSDM_NOPIPESPEC"] = "1"
        try:
            G0 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
        finally:
            del os.environ["SP_CSDM_NOPIPESPEC"]
        os.environ["SP_CSDM_TWOPASS"] = "1"           # packed spectra, means by the separate pass
        try:
            G2 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
        finally:
            del os.environ["SP_CSDM_TWOPASS"]
        assert np.max(np.abs(G1 - G0)) <= 3e-6 * np.abs(G0).max(), (nch, M)
        assert np.max(np.abs(G2 - G0)) <= 3e-6 * np.abs(G0).max(), (nch, M)
        assert np.max(np.abs(G1 - np.conj(np.swapaxes(G1, 1, 2)))) <= 1e-6 * np.abs(G1).max()
        # no detrend and caller-supplied means take the plain packed path
        Gn = E.csd_matrix(x, win, hop, M, detrend=False, scale=1.0)
        os.environ["SP_CSDM_NOPIPESPEC"] = "1"
        try:
            Gn0 = E.csd_matrix(x, win, hop, M, detrend=False, scale=1.0)
        finally:
            del os.environ["SP_CSDM_NOPIPESPEC"]
        assert np.max(np.abs(Gn - Gn0)) <= 3e-6 * np.abs(Gn0).max(), (nch, M)
    ref = O.csd_matrix(x[:3].astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    assert np.max(np.abs(G1[:, :3, :3] - ref)) <= 2e-5 * np.abs(ref).max()


def test_csd_matrix_two_piece_contraction(E):
    """From 1024 frame pairs on the contraction splits its operands into two bf16 pieces (16 bits) instead of three: against
    the three-piece form (SP_CSDM_SPLIT3=1) and the float64 oracle, on noise and on a line whose spectrum repeats from frame
    to frame (the case where the operand rounding does not average out)"""
    import os
    nfft, hop, nch, M = 4096, 2048, 16, 2051
    nsig = (M - 1) * hop + nfft
    win = O.windows("Hanning", nwins=nfft)
    rng = np.random.default_rng(31)
    t = np.arange(nsig)
    for kind in (0, 1):
        if kind == 0:
            x = (rng.standard_normal((nch, nsig)) + 0.7 * rng.standard_normal(nsig)[None, :] + 0.3).astype(np.float32)
        else:
            x = np.stack([np.sin(2 * np.pi * 200 * t / nfft + 0.1 * c) * (1 + 0.01 * c) for c in range(nch)])
            x = (x + 1e-3 * rng.standard_normal((nch, nsig))).astype(np.float32)
        G2 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
        os.environ["SP_CSDM_SPLIT3"] = "1"
        try:
            G3 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
        finally:
            del os.environ["SP_CSDM_SPLIT3"]
        assert np.max(np.abs(G2 - G3)) > 0                                  # (two different kernels ran)
        assert np.max(np.abs(G2 - G3)) <= 5e-6 * np.abs(G3).max(), kind
        ref = O.csd_matrix(x[:2].astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
        assert np.max(np.abs(G2[:, :2, :2] - ref)) <= 5e-6 * np.abs(ref).max(), kind
        assert np.max(np.abs(G3[:, :2, :2] - ref)) <= 5e-6 * np.abs(ref).max(), kind


def test_welch_csd_real_pair_equals_plain(E):
    import os
    rng = np.random.default_rng(8)
    n, nfft, hop = 30000, 1024, 256
    k = np.arange(n)
    x = (np.sin(0.21 * k) + 0.3 * rng.standard_normal(n) - 0.7).astype(np.float32)
    y = np.stack([0.4 * np.sin(0.21 * k + 1.0) + 0.2 * rng.standard_normal(n) + c for c in range(3)]).astype(np.float32)
    M = (n - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    for sided in (E.SIDED_ONE, E.SIDED_TWO, E.SIDED_RAW):
        a = E.welch_csd(x, y, win, hop, M, detrend=True, sided=sided, scale=2.0)
        os.environ["SP_NO_REALPAIR"] = "1"
        try:
            b = E.welch_csd(x, y, win, hop, M, detrend=True, sided=sided, scale=2.0)
        finally:
            del os.environ["SP_NO_REALPAIR"]
        for u, v in zip(a, b):
            assert np.max(np.abs(u - v)) <= 3e-5 * np.abs(v).max()


# ---------------------------------------------------------------- A8/A9 STFT / specgram
def test_stft_golden_f32(E):
    g = load_golden("stft_f32_n2048_ov75")
    x = g["x"]
    nfft, nov, M, Fs = int(g["nwins"]), int(g["noverlap"]), int(g["Navr"]), float(g["Fs"])
    win = O.windows("Hanning", nwins=nfft)
    S1, S2 = win.sum(), (win ** 2).sum()
    amp = 1.0 / (S1 * np.sqrt(Fs * S2 / S1 ** 2))
    Xseg, pseg = E.stft_frames(x, win, nfft - nov, M, detrend=True, sided=E.SIDED_ONE, amp_scale=amp, want_pseg=True)
    assert Xseg.shape == (M, nfft // 2)
    scale = np.abs(g["Xseg_head"]).max()
    # STFT: rtol 1e-4 relative to ||.||_inf
    assert np.max(np.abs(Xseg[:3] - g["Xseg_head"])) <= 1e-4 * scale
    assert np.max(np.abs(Xseg[M // 2:M // 2 + 2] - g["Xseg_mid"])) <= 1e-4 * scale
    assert np.max(np.abs(Xseg[-2:] - g["Xseg_tail"])) <= 1e-4 * scale
    # Xpow = trapz(|w x|^2, t)/S2 with dt = 1/Fs
    np.testing.assert_allclose(pseg / Fs / S2, g["Xpow"], rtol=1e-4)
    # mean |Xseg|^2 == Pxx
    P = (np.abs(Xseg.astype(np.complex128)) ** 2).mean(axis=0)
    np.testing.assert_allclose(P, g["Pxx"].real, rtol=2e-4, atol=1e-6 * g["Pxx"].real.max())


@pytest.mark.parametrize("nfft,hop,M", [(2048, 512, 1001), (2048, 1024, 300), (1024, 256, 77), (4096, 1024, 64), (8192, 2048, 9)])
def test_stft_one_sided_compile_time_form(E, nfft, hop, M):
    """spectrogram.stft's shape (one-sided complex rows, no per-frame power) takes k_stft_rp<.., FAST>: the one-sided crop, the
    doubling of [1:-1] and the store pattern as compile-time facts.  Against the generic form (SP_STFT_NOFAST=1) to rounding and
    against the float64 frames (fft_analysis.py:2156-2203 restated); odd and even frame counts (a lone last frame), both the
    register-carried (hop = nfft/4) and the plain fetch."""
    import os
    rng = np.random.default_rng(nfft + hop + M)
    n = (M - 1) * hop + nfft + 3
    x = (rng.standard_normal(n) + 0.3).astype(np.float32)
    win = O.windows("Hanning", nwins=nfft)
    X, _ = E.stft_frames(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, amp_scale=0.5)
    os.environ["SP_STFT_NOFAST"] = "1"
    try:
        X0, _ = E.stft_frames(x, win, hop, M, detrend=True, sided=E.SIDED_ONE, amp_scale=0.5)
    finally:
        del os.environ["SP_STFT_NOFAST"]
    assert X.shape == X0.shape == (M, nfft // 2)
    assert np.max(np.abs(X - X0)) <= 1e-6 * np.abs(X0).max()          # (same transform; the scale is applied in another order)
    xd = x.astype(np.float64) - x.astype(np.float64).mean()
    fr = np.stack([xd[g * hop: g * hop + nfft] * win for g in range(M)])
    ref = np.fft.fft(fr, axis=1)[:, : nfft // 2] * 0.5
    ref[:, 1:-1] *= np.sqrt(2.0)
    assert np.max(np.abs(X - ref)) <= 1e-4 * np.abs(ref).max()


def test_stft_twosided_complex_and_power_binmajor(E):
    g = load_golden("welch_class_c64_2e16_n4096")
    x = g["x"]
    nfft, nov, M, Fs = int(g["nwins"]), int(g["noverlap"]), int(g["Navr"]), float(g["Fs"])
    win = O.windows("Hanning", nwins=nfft)
    amp = 1.0 / np.sqrt(Fs * float(g["S2"]))
    Xseg, _ = E.stft_frames(x, win, nfft - nov, M, detrend=True, sided=E.SIDED_TWO, amp_scale=amp)
    scale = np.abs(g["Xseg_head"]).max()
    assert np.max(np.abs(Xseg[:4] - g["Xseg_head"])) <= 1e-4 * scale
    assert np.max(np.abs(Xseg[-2:] - g["Xseg_tail"])) <= 1e-4 * scale
    # specgram-style output: power, natural order, [bins, frames]
    s = load_golden("specgram")
    sig, wl = s["s"], 512
    nW = s["sp1"].shape[1]
    out, _ = E.stft_frames(sig, np.hanning(wl), wl // 2, nW, detrend=False, sided=E.SIDED_RAW,
                           amp_scale=np.sqrt(8.0 / 3.0) / wl, power=True, bin_major=True)
    assert out.shape == s["sp1"].shape
    np.testing.assert_allclose(out, s["sp1"], rtol=2e-4, atol=1e-6 * s["sp1"].max())


# ---------------------------------------------------------------- A10 hilbert
def test_hilbert_rows(E):
    g = load_golden("hilbert")
    z = E.hilbert_rows(g["u_even"][None, :], 4096)[0]
    assert np.max(np.abs(z - g["z_even"])) <= 1e-4 * np.abs(g["z_even"]).max()
    z2 = E.hilbert_rows(g["u_2d"], 512)
    assert np.max(np.abs(z2 - g["z_2d"])) <= 1e-4 * np.abs(g["z_2d"]).max()
    zk = E.hilbert_rows(g["yk"][None, :], 32)[0]
    ph = 2 * np.pi * np.arange(32) / 32
    assert np.max(np.abs(zk - (np.sin(ph) - 1j * np.cos(ph)))) < 2e-6      # hilbert.py:115-140 KAT
    zp = E.hilbert_rows(g["u_even"][None, :1000], 1024)[0]                    # zero-padded transform length
    assert np.max(np.abs(zp - g["z_nfft"])) <= 1e-4 * np.abs(g["z_nfft"]).max()
    # real part of the analytic signal is the input
    assert np.max(np.abs(z.real - g["u_even"])) < 1e-5 * np.abs(g["u_even"]).max()
    # odd length: the reference leaves bin (N+1)/2 un-zeroed (differs from scipy) -- reproduced
    zo = E.hilbert_rows(g["u_odd"][None, :], g["u_odd"].size)[0]
    assert np.max(np.abs(zo - g["z_odd"])) <= 1e-4 * np.abs(g["z_odd"]).max()
    z6 = E.hilbert_rows(np.ascontiguousarray(g["u_2d"].T), 6)           # tiny non power-of-two rows
    assert np.max(np.abs(z6.T - g["z_2d_ax0"])) <= 1e-4 * np.abs(g["z_2d_ax0"]).max()


def test_hilbert_long_rows(E):
    """whole-signal analytic signal longer than one workgroup (transform length = the axis length)"""
    rng = np.random.default_rng(31)
    for n in (1 << 16, 30001, 100000):
        u = rng.standard_normal((2, n))
        z = E.hilbert_rows(u, n)
        ref = O.hilbert(u)
        assert np.max(np.abs(z - ref)) <= 1e-4 * np.abs(ref).max()
    u = rng.standard_normal((1, 40000))
    z = E.hilbert_rows(u, 1 << 16)                     # zero-padded transform length
    assert np.max(np.abs(z[0] - O.hilbert(u[0], nfft=1 << 16))) <= 1e-4 * np.abs(u).max()


def test_hilbert_long_half_length(E):
    """power-of-two lengths from 2^21: the real input goes through ONE half-length transform each way (pair load fused
    into the first pass, k_hilbert_mid, analytic signal written by the last pass); full, zero-padded (odd sample count)
    and truncated rows, two rows with an odd row pitch"""
    rng = np.random.default_rng(32)
    n = 1 << 21
    u = rng.standard_normal((2, n + 5))[:, : n + 5]
    for nuse in (n, 1500001, n + 5):
        z = E.hilbert_rows(np.ascontiguousarray(u[:, :nuse]), n)
        ref = O.hilbert(u[:, :nuse], nfft=n)
        assert z.shape == (2, n)
        assert np.max(np.abs(z - ref)) <= 2e-5 * np.abs(ref).max(), nuse
    u = rng.standard_normal(1 << 22) + 3.0                      # a mean: DC stays in the real part only
    z = E.hilbert_rows(u[None, :], 1 << 22)[0]
    ref = O.hilbert(u)
    assert np.max(np.abs(z - ref)) <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("log2n", [21, 22, 23, 24, 25])
def test_hilbert_long_middle_step_in_the_row_pass(E, log2n):
    """round 3: forward row FFT -> the (k, M - k) middle step through one LDS exchange -> inverse row FFT in ONE kernel that
    owns mirror row pairs, the inverse as the adjoint column passes in reversed order (k_hilbert_rowsmid, k_fft_cols_inv).
    Every three-pass split (A, B, C) from 2^20 to 2^24 complex points; against the separate-middle-step form
    (SP_HILBERT_NOFUSEMID=1) everywhere and the float64 oracle (hilbert.py:54-67) where that is quick."""
    import os
    rng = np.random.default_rng(log2n)
    n = 1 << log2n
    u = (rng.standard_normal(n) + 0.7).astype(np.float32)
    z = E.hilbert_rows(u[None, :], n)[0]
    os.environ["SP_HILBERT_NOFUSEMID"] = "1"
    try:
        z0 = E.hilbert_rows(u[None, :], n)[0]
    finally:
        del os.environ["SP_HILBERT_NOFUSEMID"]
    assert np.max(np.abs(z - z0)) <= 5e-6 * np.abs(z0).max()
    assert np.array_equal(z.real, u)                                  # the real part is the input itself
    if log2n <= 23:
        ref = O.hilbert(u.astype(np.float64))
        assert np.max(np.abs(z - ref)) <= 2e-5 * np.abs(ref).max()
    if log2n in (21, 24):
        # the full, aligned row takes the predicate-free end passes (plain first pass on the samples seen as complex pairs,
        # k_fft_cols_inv<., 3>); against the predicated forms
        os.environ["SP_COLS_NOHALF"] = os.environ["SP_HILBERT_PAIRLOAD"] = "1"
        try:
            z1 = E.hilbert_rows(u[None, :], n)[0]
        finally:
            del os.environ["SP_COLS_NOHALF"], os.environ["SP_HILBERT_PAIRLOAD"]
        assert np.max(np.abs(z - z1)) <= 1e-6 * np.abs(z1).max()
    nuse = n - 12345                                                  # truncated / zero-padded row
    z = E.hilbert_rows(u[None, :nuse], n)[0]
    os.environ["SP_HILBERT_NOFUSEMID"] = "1"
    try:
        z0 = E.hilbert_rows(u[None, :nuse], n)[0]
    finally:
        del os.environ["SP_HILBERT_NOFUSEMID"]
    assert np.max(np.abs(z - z0)) <= 5e-6 * np.abs(z0).max()


# ---------------------------------------------------------------- A11 ccf
def test_xcorr_golden(E):
    g = load_golden("ccf")
    co = E.xcorr_normalised(g["x1"], g["x2"])
    assert co.shape == g["co"].shape
    assert np.max(np.abs(co - g["co"])) <= 1e-4 * np.abs(g["co"]).max()
    co2 = E.xcorr_normalised(g["x3"], g["x4"])         # odd length 777, non-zero mean
    assert np.max(np.abs(co2 - g["co2"])) <= 1e-4 * np.abs(g["co2"]).max()
    # autocorrelation: symmetric, 1 at zero lag
    a = E.xcorr_normalised(g["x1"], g["x1"])
    n = g["x1"].size
    assert abs(a[n - 1] - 1.0) < 1e-5 and np.max(np.abs(a - a[::-1])) < 1e-5


@pytest.mark.parametrize("n", [5000, 70001, 1 << 18, 300001, 600001, 1100003])
def test_xcorr_long(E, n):
    """more lags than one workgroup holds: the reference's O(N^2) np.correlate is infeasible here; the oracle's
    FFT formulation (asserted equal to the direct form on the CPU) is the checker.  300001: three-pass transforms of 2^20
    points; 600001 and 1100003: the inverse at half length (2^20 and 2^21 points for 2^21 and 2^22 lags)"""
    rng = np.random.default_rng(n % 1000)
    k = np.arange(n)
    x1 = np.sin(0.01 * k) + rng.standard_normal(n) + 1.5
    x2 = np.roll(x1, 37) + 0.5 * rng.standard_normal(n) - 0.5
    co = E.xcorr_normalised(x1, x2)
    tau, ref = O.ccf_fft(x1.astype(np.float32).astype(np.float64), x2.astype(np.float32).astype(np.float64), 1.0)
    assert co.shape == ref.shape
    assert np.max(np.abs(co - ref)) <= 1e-4 * np.abs(ref).max()
    assert np.argmax(co) == np.argmax(ref)


@pytest.mark.parametrize("n", [600001, 1100003, 1 << 21, (1 << 22) + 7, 1 << 23, 1 << 24])
def test_xcorr_long_middle_step_in_the_row_pass(E, n):
    """round 3: the forward transform's row pass, k_xc_mid_half's arithmetic and the first pass of the half-length transform in ONE
    kernel that owns mirror row pairs (k_xc_rowsmid), the lags written by the last column pass (k_fft_cols_lag).  Every
    three-pass split from 2^21 to 2^25 points; against the separate-middle-step form (SP_XC_NOFUSEMID=1) everywhere and the
    oracle's FFT formulation (ccf.py:66-77 restated) where that is quick."""
    import os
    rng = np.random.default_rng(n % 997)
    k = np.arange(n)
    x1 = (np.sin(0.01 * k) + rng.standard_normal(n) + 1.5).astype(np.float32)
    x2 = (np.roll(x1, 41) + 0.5 * rng.standard_normal(n) - 0.5).astype(np.float32)
    co = E.xcorr_normalised(x1, x2)
    os.environ["SP_XC_NOFUSEMID"] = "1"
    try:
        co0 = E.xcorr_normalised(x1, x2)
    finally:
        del os.environ["SP_XC_NOFUSEMID"]
    assert co.shape == co0.shape == (2 * n - 1,)
    assert np.max(np.abs(co - co0)) <= 2e-5 * np.abs(co0).max()
    if n & (n - 1) == 0:
        # power-of-two sample counts end exactly at the middle row of the first pass: the predicate-free pack (k_fft_cols<., 4>)
        # against the predicated one
        os.environ["SP_COLS_NOHALF"] = "1"
        try:
            co1 = E.xcorr_normalised(x1, x2)
        finally:
            del os.environ["SP_COLS_NOHALF"]
        assert np.max(np.abs(co - co1)) <= 1e-6 * np.abs(co1).max()
    assert np.argmax(co) == np.argmax(co0) == (n - 1) - 41
    if n <= (1 << 22) + 7:
        tau, ref = O.ccf_fft(x1.astype(np.float64), x2.astype(np.float64), 1.0)
        assert np.max(np.abs(co - ref)) <= 1e-4 * np.abs(ref).max()


# ---------------------------------------------------------------- F1 FIR
@pytest.mark.parametrize("ntaps,n,nfft", [(513, 40000, 4096), (513, 40000, 0), (31, 5000, 1024), (2, 100, 64),
                                           (1, 1000, 256), (4097, 20000, 8192)])
def test_fir_filter(E, ntaps, n, nfft):
    rng = np.random.default_rng(ntaps + n)
    h = rng.standard_normal(ntaps) / np.sqrt(ntaps)
    x = rng.standard_normal(n)
    y = E.fir_filter(h, x, nfft=nfft)
    ref = O.fftfilt(h.astype(np.float32).astype(np.float64), x.astype(np.float32).astype(np.float64))
    assert y.shape == ref.shape
    assert np.max(np.abs(y - ref)) <= 1e-4 * np.abs(ref).max()


def test_mean(E):
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(100001) + 3.0).astype(np.float32)
    assert abs(E.mean(x) - x.astype(np.float64).mean()) < 1e-9
    z = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000) + (1 - 2j)).astype(np.complex64)
    assert abs(E.mean(z) - z.astype(np.complex128).mean()) < 1e-9


@pytest.mark.parametrize("nch,nfft,hop,nsig", [(64, 512, 256, 40000), (37, 256, 64, 9000), (130, 256, 128, 5000)])
def test_csd_matrix_paths_agree(E, nch, nfft, hop, nsig, monkeypatch):
    """The three contraction paths -- fused MFMA (default for nch <= 64), MFMA on the transposed copy (default above
    64 channels), VALU (A/B only) -- give the same matrix"""
    rng = np.random.default_rng(nch + nfft)
    x = (rng.standard_normal((nch, nsig)) + 0.3 * rng.standard_normal(nsig)[None, :] + 0.2).astype(np.float32)
    M = (nsig - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    g0 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    monkeypatch.setenv("SP_CSDM_TRANSPOSED", "1")
    g1 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    monkeypatch.delenv("SP_CSDM_TRANSPOSED")
    monkeypatch.setenv("SP_CSDM_VALU", "1")
    g2 = E.csd_matrix(x, win, hop, M, detrend=True, scale=1.0)
    scale = np.abs(g2).max()
    assert np.max(np.abs(g0 - g2)) <= 2e-6 * scale
    assert np.max(np.abs(g1 - g2)) <= 2e-6 * scale


@pytest.mark.parametrize("nfft,hop,n,nch,detrend", [(1024, 512, 1024 + 512 * 40, 5, True), (256, 64, 256 + 64 * 37, 2, "linear"),
                                                    (4096, 2048, 4096 + 2048 * 9, 3, True)])
def test_welch_csd_reference_once_path(E, nfft, hop, n, nch, detrend, monkeypatch):
    """reference-once pair form (default for >= 2 real channels) == x + i y_c form == plain complex form; odd and even
    frame counts, mean and linear detrend"""
    rng = np.random.default_rng(nfft + nch)
    k = np.arange(n)
    x = (np.sin(0.11 * k) + 0.3 * rng.standard_normal(n) - 0.4 + 1e-4 * k).astype(np.float32)
    y = np.stack([0.5 * np.sin(0.11 * k + 0.3 * c) + 0.2 * rng.standard_normal(n) + 0.1 * c for c in range(nch)]).astype(np.float32)
    M = (n - nfft) // hop + 1
    win = O.windows("Hanning", nwins=nfft)
    a = E.welch_csd(x, y, win, hop, M, detrend=detrend, sided=E.SIDED_TWO, scale=1.0)
    monkeypatch.setenv("SP_CSD_XIY", "1")
    b = E.welch_csd(x, y, win, hop, M, detrend=detrend, sided=E.SIDED_TWO, scale=1.0)
    monkeypatch.setenv("SP_NO_REALPAIR", "1")
    c = E.welch_csd(x, y, win, hop, M, detrend=detrend, sided=E.SIDED_TWO, scale=1.0)
    for u, v, w in zip(a, b, c):
        assert np.max(np.abs(u - w)) <= 3e-5 * np.abs(w).max()
        assert np.max(np.abs(v - w)) <= 3e-5 * np.abs(w).max()


@pytest.mark.parametrize("M,tail,nch", [(301, 0, 5), (300, 777, 3), (2051, 5000, 2), (301, 100, 9)])
def test_welch_csd_reference_once_one_pass_means(E, M, tail, nch, monkeypatch):
    """reference against >= 2 real channels at nfft 4096 / 50 % overlap with mean detrend: the channels' means are taken in the
    pass that forms the spectra (estimate + block sums + exact correction in the finish kernel).  Against the separate mean
    pass (SP_CSD_TWOPASS=1) and the float64 oracle; channel offsets far above the noise, samples after the last frame (they
    count for the mean), odd and even frame counts"""
    nfft, hop = 4096, 2048
    n = (M - 1) * hop + nfft + tail
    rng = np.random.default_rng(M + nch)
    k = np.arange(n)
    x = (np.sin(0.11 * k) + 0.3 * rng.standard_normal(n) - 0.4).astype(np.float32)
    y = np.stack([0.5 * np.sin(0.11 * k + 0.3 * c) + 0.2 * rng.standard_normal(n) + 1.5 * (c + 1) for c in range(nch)]).astype(np.float32)
    win = O.windows("Hanning", nwins=nfft)
    if nch < 8:
        monkeypatch.setenv("SP_CSD_ONEPASS", "1")          # (the default takes the form from 8 channels on)
    for sided in (E.SIDED_TWO, E.SIDED_ONE, E.SIDED_RAW):
        a = E.welch_csd(x, y, win, hop, M, detrend=True, sided=sided, scale=1.0)
        monkeypatch.setenv("SP_CSD_TWOPASS", "1")
        b = E.welch_csd(x, y, win, hop, M, detrend=True, sided=sided, scale=1.0)
        monkeypatch.delenv("SP_CSD_TWOPASS")
        for u, v in zip(a, b):
            assert np.max(np.abs(u - v)) <= 3e-6 * np.abs(v).max()
        assert any(np.max(np.abs(u - v)) > 0 for u, v in zip(a, b))           # (two different paths ran)
    pxx, pyy, pxy = E.welch_csd(x, y, win, hop, M, detrend=True, sided=E.SIDED_RAW, scale=1.0)       # (unshifted bins)
    x64 = x.astype(np.float64) - x.astype(np.float64).mean()
    y64 = y.astype(np.float64) - y.astype(np.float64).mean(axis=1, keepdims=True)
    fx = np.stack([np.fft.fft(win * x64[g * hop:g * hop + nfft]) for g in range(M)])
    for c in range(nch):
        fy = np.stack([np.fft.fft(win * y64[c, g * hop:g * hop + nfft]) for g in range(M)])
        ryy = np.mean(np.abs(fy) ** 2, axis=0)
        rxy = np.mean(fy * np.conj(fx), axis=0)
        assert np.max(np.abs(pyy[c] - ryy)) <= 2e-5 * ryy.max()
        assert np.max(np.abs(pxy[c] - rxy)) <= 2e-5 * np.abs(rxy).max()


@pytest.mark.parametrize("hop,cplx,wname", [(2048, True, "Hanning"), (1024, True, "Hamming"), (2048, False, "Hanning")])
def test_cog_one_pass_mean_detrend(E, hop, cplx, wname, monkeypatch):
    """Centre of gravity per frame with mean detrend, nfft 4096: for a cosine-sum window the mean is not taken by a pass of its
    own -- the kernel detrends by an estimate, keeps the 7 lowest bins of every frame and k_cog_finish_op corrects the moments
    with the exact mean.  Against the separate mean pass (SP_COG_TWOPASS=1) and numpy float64 on sampled frames; a DC offset
    far above the signal (it would drag every centre of gravity to zero if mishandled)"""
    nfft, M, fs = 4096, 9000, 1000.0
    n = (M - 1) * hop + nfft + 333
    rng = np.random.default_rng(hop + int(cplx))
    t = np.arange(n)
    ph = 2 * np.pi * (0.11 * t + 0.04 * n / (2 * np.pi * 3) * np.sin(2 * np.pi * 3 * t / n))
    if cplx:
        x = (np.exp(1j * ph) + 0.3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) + (2.0 - 1.5j)).astype(np.complex64)
    else:
        x = (np.cos(ph) + 0.3 * rng.standard_normal(n) + 2.5).astype(np.float32)
    win = O.windows(wname, nwins=nfft)
    a = E.stft_cog(x, win, hop, M, fs, detrend=True)
    monkeypatch.setenv("SP_COG_TWOPASS", "1")
    b = E.stft_cog(x, win, hop, M, fs, detrend=True)
    monkeypatch.delenv("SP_COG_TWOPASS")
    assert np.max(np.abs(a - b)) > 0                      # (two different paths ran)
    assert np.max(np.abs(a - b)) <= 2e-5 * fs
    x64 = x.astype(np.complex128 if cplx else np.float64)
    x64 = x64 - x64.mean()
    ks = np.fft.fftfreq(nfft, 1.0 / nfft)
    for g in range(0, M, 997):
        P = np.abs(np.fft.fft(win * x64[g * hop:g * hop + nfft])) ** 2
        ref = fs / nfft * np.sum(ks * P) / np.sum(P)
        assert abs(a[g] - ref) <= 2e-5 * fs, (g, a[g], ref)
    # a window whose spectrum is not confined to a few bins keeps the separate pass
    wk = np.kaiser(nfft, 8.0)
    c = E.stft_cog(x, wk, hop, M, fs, detrend=True)
    monkeypatch.setenv("SP_COG_TWOPASS", "1")
    d = E.stft_cog(x, wk, hop, M, fs, detrend=True)
    assert np.array_equal(c, d)


# ---------------------------------------------------------------- A6 / N1 epilogue on device-resident spectra
@pytest.mark.parametrize("nfft,onesided", [(1024, True), (1333, True), (512, False), (777, False)])
def test_csd_epilogue_on_device(E, nfft, onesided):
    """coherence / phase / amplitude spectra / correlations from averaged spectra that never leave the GPU
    (fft_analysis.py:489-648), against the same algebra in numpy float64"""
    import torch
    rng = np.random.default_rng(nfft)
    nch = 3
    nb = ((nfft + 1) // 2 if nfft % 2 else nfft // 2) if onesided else nfft
    pxx = 1.0 + rng.random(nb)
    pyy = 0.5 + rng.random((nch, nb))
    pxy = (rng.standard_normal((nch, nb)) + 1j * rng.standard_normal((nch, nb))) * 0.4
    enbw = 3.7
    r = E.csd_epilogue(torch.from_numpy(pxx).cuda(), torch.from_numpy(pyy).cuda(), torch.from_numpy(pxy).cuda(), nfft, onesided, enbw)
    assert all(v.is_cuda for v in r.values())
    g = {k: v.cpu().numpy() for k, v in r.items()}
    h = E.csd_epilogue(pxx, pyy, pxy, nfft, onesided, enbw)              # host arrays through the same kernels
    for k in g:
        np.testing.assert_allclose(g[k], h[k], rtol=0, atol=0)
    den = np.abs(pxx)[None, :] * np.abs(pyy)
    np.testing.assert_allclose(g["Cxy"], pxy / np.sqrt(den), rtol=1e-12)
    np.testing.assert_allclose(g["Cxy2"], np.abs(pxy) ** 2 / den, rtol=1e-12)
    np.testing.assert_allclose(g["phi"], np.angle(pxy), rtol=1e-12)
    amp = np.ones(nb)
    if onesided:
        amp[1:-1] = np.sqrt(2)
        if nfft % 2:
            amp[-1] = np.sqrt(2)
    np.testing.assert_allclose(g["Lxx"], amp * np.sqrt(enbw * pxx), rtol=1e-12)
    np.testing.assert_allclose(g["Lxy"], amp * np.sqrt(enbw * np.abs(pxy)), rtol=1e-12)

    def back(P, halve=True):
        P = np.array(P, dtype=np.complex128)
        if onesided:
            if halve:
                P[..., 1:-1] *= 0.5
                if nfft % 2:
                    P[..., -1] *= 0.5
            return np.sqrt(nfft) * np.fft.irfft(P, n=nfft, axis=-1)
        return np.sqrt(nfft) * np.fft.ifft(np.fft.ifftshift(P, axes=-1), n=nfft, axis=-1)
    Rxx, Ryy, Rxy = back(pxx), back(pyy), back(pxy)
    sc = np.abs(Rxx).max()
    for name, ref in (("Rxx", Rxx), ("Ryy", Ryy), ("Rxy", Rxy), ("iCxy", back(pxy / np.sqrt(den), halve=False))):
        assert np.max(np.abs(g[name] - np.fft.fftshift(ref, axes=-1))) <= 2e-6 * max(sc, np.abs(ref).max()), name
    Ex, Ey = Rxx[0], Ryy[:, 0]
    np.testing.assert_allclose(g["Ex"], Ex, rtol=1e-5)
    np.testing.assert_allclose(g["Ey"], Ey, rtol=1e-5)
    cc = np.fft.fftshift(Rxy, axes=-1) / np.sqrt(Ex * Ey)[:, None]
    assert np.max(np.abs(g["corrcoef"] - cc)) <= 1e-5 * np.abs(cc).max()
