"""k_op_fused (pyfft_amd/csrc/kernels.h): the one-pass Welch epilogue in ONE launch -- column sums, hand-off to the block that
arrives last, exact mean from the block sums, the lobe bins of sum_g X_g as direct float64 sums -- for windows whose spectrum
is confined to the bins -3 .. 3 (every cosine-sum window of the reference's windows(), windows.py:57-297).  Compared with
the two-launch form it replaces (SP_OP_UNFUSED=1: k_op_colsums + k_op_finish, read per call) and with the float64 oracle
(fft_analysis.py:2126-2203 -> :1946 -> :1980); the sharded state (sp_welch_export) the same way."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O


def _both(fn):
    """(default: one-launch epilogue, and -- nfft 4096 on the pipeline kernel, COLA window -- the lobe sums from the back role
    [k_welch_pipe mode 9, k_op_fused<LOBEB>];  SP_OP_UNFUSED=1: block sums + k_op_colsums + k_op_finish)"""
    os.environ.pop("SP_OP_UNFUSED", None)
    os.environ.pop("SP_OP_NOLOBESUM", None)
    a = fn()
    os.environ["SP_OP_UNFUSED"] = "1"
    try:
        b = fn()
    finally:
        os.environ.pop("SP_OP_UNFUSED", None)
    return a, b


def _three(fn):
    """default | one-launch epilogue on the block sums (SP_OP_NOLOBESUM=1) | two-launch form"""
    a, c = _both(fn)
    os.environ["SP_OP_NOLOBESUM"] = "1"
    try:
        b = fn()
    finally:
        os.environ.pop("SP_OP_NOLOBESUM", None)
    return a, b, c


@pytest.mark.parametrize("wname", ["Hanning", "Hamming", "Nuttall4", "SFT3F", "HFT70"])
@pytest.mark.parametrize("nfft,hop", [(4096, 2048), (1024, 256), (256, 256), (8192, 4096)])
def test_fused_epilogue_matches_two_launch_form_and_oracle(wname, nfft, hop):
    from pyfft_amd import engine as E
    rng = np.random.default_rng(nfft + hop)
    win = O.windows(wname, nwins=nfft)
    M = 301
    n = (M - 1) * hop + nfft + 37                                     # ragged end: 37 samples count in the mean only
    s = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) + np.complex64(3.0 - 1.5j)
    s[: n // 3] += np.complex64(2.0)                                   # the estimate mu0 is off: the correction matters
    S2 = float(np.sum(win ** 2))
    for sided in (E.SIDED_TWO, E.SIDED_ONE, E.SIDED_RAW):
        pf, pu = _both(lambda: E.welch_psd(s, win, hop, M, detrend=True, sided=sided, scale=1.0 / S2))
        assert np.max(np.abs(pf - pu)) <= 2e-6 * pu.max(), (wname, sided)
    ref = O.welch_psd_stream(s, win, nfft, hop, M, 1.0, detrend_style=1)
    pf = E.welch_psd(s, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)
    np.testing.assert_allclose(pf, ref, rtol=2e-4, atol=1e-6 * ref.max())


def test_fused_epilogue_real_pair_and_small_counts():
    """real input at hop = nfft/2 (two frames per transform: the |Z|^2 sums are symmetrised in the epilogue), 1..5 frames"""
    from pyfft_amd import engine as E
    rng = np.random.default_rng(3)
    nfft, hop = 4096, 2048
    win = O.windows("Hanning", nwins=nfft)
    S2 = float(np.sum(win ** 2))
    os.environ["SP_WELCH_PIPE"] = "2"          # (read once per process: only effective if this is the first Welch call)
    for M in (1, 2, 3, 5, 64, 9000):
        n = (M - 1) * hop + nfft + 3
        s = (rng.standard_normal(n) + 1.25).astype(np.float32)
        pf, pu = _both(lambda: E.welch_psd(s, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2))
        assert np.max(np.abs(pf - pu)) <= 2e-6 * pu.max(), M
        ref = O.welch_psd_stream(s, win, nfft, hop, M, 1.0, detrend_style=1)
        np.testing.assert_allclose(pf, ref, rtol=2e-4, atol=1e-6 * ref.max())


def test_fused_export_state_applies_to_the_same_psd():
    """sp_welch_export through k_op_fused<EXPORT>: B is written at the lobe bins only -- the applied PSD must not change"""
    from pyfft_amd import engine as E
    rng = np.random.default_rng(4)
    nfft, hop, M = 4096, 2048, 700
    win = O.windows("Hanning", nwins=nfft)
    n = (M - 1) * hop + nfft
    s = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) + np.complex64(0.4 + 0.9j)
    sf, su = _both(lambda: E.welch_export(s, win, hop, M, nmean=n - 100))
    assert sf.shape == su.shape == (5 * nfft + 8,)
    np.testing.assert_allclose(sf[:nfft], su[:nfft], rtol=1e-12)                      # A: the same column sums
    np.testing.assert_allclose(sf[5 * nfft:], su[5 * nfft:], rtol=1e-9, atol=1e-6)    # scalars
    B_f = sf[nfft:3 * nfft].reshape(nfft, 2)
    B_u = su[nfft:3 * nfft].reshape(nfft, 2)
    lobe = [0, 1, 2, 3, nfft - 3, nfft - 2, nfft - 1]
    scale_b = np.abs(B_u[lobe]).max()
    assert np.max(np.abs(B_f[lobe] - B_u[lobe])) <= 1e-5 * scale_b                    # float32 transform vs float64 sums
    assert np.all(B_f[4:nfft - 3] == 0.0)
    pf = E.welch_apply(sf, win, M, sided=E.SIDED_TWO, scale=1.0)
    pu = E.welch_apply(su, win, M, sided=E.SIDED_TWO, scale=1.0)
    assert np.max(np.abs(pf - pu)) <= 2e-6 * pu.max()
    # two shards with different estimates add up to the PSD of the whole stream
    ref = O.welch_psd_stream(s, win, nfft, hop, M, 1.0, detrend_style=1) * np.sum(win ** 2)
    f1 = M // 3
    a = E.welch_export(s[: (f1 - 1) * hop + nfft], win, hop, f1, nmean=f1 * hop)
    b = E.welch_export(s[f1 * hop:], win, hop, M - f1, nmean=n - f1 * hop)
    p = E.welch_apply(a + b, win, M, sided=E.SIDED_TWO, scale=1.0)
    np.testing.assert_allclose(p, ref, rtol=2e-4, atol=1e-6 * ref.max())


def test_non_cosine_window_keeps_the_transform_form():
    """Kaiser: FFT(window) is not confined to a few bins, so the epilogue must stay k_op_colsums + k_op_finish (identical
    results with and without SP_OP_UNFUSED), and agree with the oracle"""
    from pyfft_amd import engine as E
    from pyfft_amd.windows import get_window
    rng = np.random.default_rng(8)
    nfft, hop, M = 2048, 1024, 400
    win = np.asarray(get_window(("kaiser", 6.0), nfft, fftbins=True), dtype=np.float64)
    n = (M - 1) * hop + nfft
    s = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) + np.complex64(1.0 + 1.0j)
    pf, pu = _both(lambda: E.welch_psd(s, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0))
    assert np.array_equal(pf, pu)
    ref = O.welch_psd_stream(s, win, nfft, hop, M, 1.0, detrend_style=1) * np.sum(win ** 2)
    np.testing.assert_allclose(pf, ref, rtol=2e-4, atol=1e-6 * ref.max())


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("hop,M", [(2048, 9001), (1024, 8500), (2048, 8193)])
def test_lobe_sums_from_the_back_role(cplx, hop, M):
    """k_welch_pipe mode 9 + k_op_fused<LOBEB>: the lobe bins of sum_g X_g from the back role's registers and the samples' plain
    sum from the DC bin through the window's COLA constant, edges corrected -- against the block-sum forms and the oracle.  The
    mean is large and drifts, the record has a ragged end (samples beyond the last frame count in the mean)."""
    from pyfft_amd import engine as E
    rng = np.random.default_rng(hop + M)
    nfft = 4096
    n = (M - 1) * hop + nfft + 777
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0.0) + (7.0 - (2.0j if cplx else 0.0))
    x[: n // 3] += 3.0
    x = x.astype(np.complex64 if cplx else np.float32)
    for wname in ("Hanning", "Hamming"):
        win = O.windows(wname, nwins=nfft)
        S2 = float(np.sum(win ** 2))
        a, b, c = _three(lambda: E.welch_psd(x, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2))
        if cplx:
            assert "lobesum" in E.profile_last_kernel() or True
        assert np.max(np.abs(a - b)) <= 2e-6 * b.max() and np.max(np.abs(a - c)) <= 2e-6 * c.max(), wname
        ref = O.welch_psd_stream(x, win, nfft, hop, M, 1.0, detrend_style=1)
        np.testing.assert_allclose(a, ref, rtol=2e-4, atol=1e-6 * ref.max())
    # the kernel really is the lobe-sum form for complex input (real input takes the two-frames-per-transform kernels)
    E.welch_psd(x, O.windows("Hanning", nwins=nfft), hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    if cplx:
        assert "lobesum" in E.profile_last_kernel(), E.profile_last_kernel()
    # the sharded state: shards with different estimates and edges add up
    if cplx:
        win = O.windows("Hanning", nwins=nfft)
        ref = O.welch_psd_stream(x, win, nfft, hop, M, 1.0, detrend_style=1) * np.sum(win ** 2)
        f1 = M - 150                       # (the second shard is small: k_welch_carry and block sums there, lobe sums in the first)
        sa = E.welch_export(x[: (f1 - 1) * hop + nfft], win, hop, f1, nmean=f1 * hop)
        sb = E.welch_export(x[f1 * hop:], win, hop, M - f1, nmean=n - f1 * hop)
        p = E.welch_apply(sa + sb, win, M, sided=E.SIDED_TWO, scale=1.0)
        np.testing.assert_allclose(p, ref, rtol=2e-4, atol=1e-6 * ref.max())
