"""k_welch_pipe (pyfft_amd/csrc/k_welch_pipe.hip): the wave-specialised pipeline that runs the nfft-4096 one-pass Welch PSD
(fft_analysis.py:2126-2203 fft_win -> :1946 Pstft -> :1980 averagewins).  By default it takes over from 32 frames per CU
on; SP_WELCH_PIPE=2 forces it for any frame count, which is how its fill/drain and tail handling (1..5 frames, frame
counts that do not divide by the grid, groups without work) are reached at sizes the oracle finishes in seconds.  Child
process: the switch is read once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from pyfft_amd import engine as E
from oracle import cpu_ref as O
rng = np.random.default_rng(11)
nfft = 4096
win = O.windows("Hanning", nwins=nfft)
S2 = float(np.sum(win ** 2))
worst = 0.0
for hop in (2048, 1024, 4096):
    for M in (1, 2, 3, 5, 255, 257, 1000, 2049):
        n = (M - 1) * hop + nfft + 17                       # 17 samples past the last frame: they count in the mean
        s = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) + np.complex64(0.7 - 0.2j)
        p = E.welch_psd(s, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)
        assert E.profile_last_kernel().startswith("k_welch_pipe"), E.profile_last_kernel()
        ref = O.welch_psd_stream(s, win, nfft, hop, M, 1.0, detrend_style=1)
        err = float(np.max(np.abs(p - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))
        worst = max(worst, err)
        assert err <= 1.0, (hop, M, err)
# plain accumulation (no detrend / a caller-supplied constant): the same pipeline without the one-pass epilogue
for hop, M in ((2048, 333), (1024, 64), (4096, 7)):
    n = (M - 1) * hop + nfft
    s = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) + np.complex64(0.3 + 0.1j)
    p = E.welch_psd(s, win, hop, M, detrend=False, sided=E.SIDED_TWO, scale=1.0 / S2)
    assert E.profile_last_kernel() == "k_welch_pipe", E.profile_last_kernel()
    ref = O.welch_psd_stream(s, win, nfft, hop, M, 1.0, detrend_style=0)
    err = float(np.max(np.abs(p - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))
    assert err <= 1.0, ("plain", hop, M, err)
    mv = complex(0.25, 0.125)                                # exactly representable: the oracle subtracts it in complex64
    p = E.welch_psd(s, win, hop, M, detrend=True, mean_value=mv, sided=E.SIDED_TWO, scale=1.0 / S2)
    assert E.profile_last_kernel() == "k_welch_pipe", E.profile_last_kernel()
    ref = O.welch_psd_stream((s - np.complex64(mv)).astype(np.complex64), win, nfft, hop, M, 1.0, detrend_style=0)
    err = float(np.max(np.abs(p - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))
    assert err <= 1.0, ("const", hop, M, err)
# real input at hop = nfft/2: two frames per transform (modes 3 / 4), even and odd frame counts, all detrend forms
for M in (2, 3, 8, 9, 500, 1001):
    n = (M - 1) * 2048 + nfft + 5
    s = (rng.standard_normal(n) + 1.7).astype(np.float32)
    for det, mv in ((True, None), (False, None), (True, 1.5)):
        p = E.welch_psd(s, win, 2048, M, detrend=det, mean_value=mv, sided=E.SIDED_TWO, scale=1.0 / S2)
        assert "realpair" in E.profile_last_kernel(), (E.profile_last_kernel(), M, det, mv)
        sr = s if mv is None else (s - np.float32(mv))
        ref = O.welch_psd_stream(sr, win, nfft, 2048, M, 1.0, detrend_style=1 if (det and mv is None) else 0)
        err = float(np.max(np.abs(p - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))
        assert err <= 1.0, ("realpair", M, det, mv, err)
    p1 = E.welch_psd(s, win, 2048, M, detrend=True, sided=E.SIDED_ONE, scale=1.0 / S2)
    assert p1.shape[0] in (nfft // 2, nfft // 2 + 1) and np.all(np.isfinite(p1))
# centre of gravity per frame (Doppler.cog / cogspec): the back role reduces each frame's moments instead of summing |X|^2
fs = 1.0e3
for hop, M in ((2048, 200), (1024, 131), (4096, 65)):
    n = (M - 1) * hop + nfft
    tt = np.arange(n) / fs
    f_inst = 120.0 * np.sin(2 * np.pi * tt / (n / fs))
    s = (np.exp(2j * np.pi * np.cumsum(f_inst) / fs) + 0.1 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    c = E.stft_cog(s, np.ones(nfft), hop, M, fs)
    _, ref = O.cog_frames(tt, s, fs, win=nfft, ov=1.0 - hop / nfft)
    assert c.shape == ref.shape
    assert np.max(np.abs(c - ref)) <= 2e-4 * fs, ("cog", hop, M, float(np.max(np.abs(c - ref))))
# the split ABI (shards: accumulate against the local estimate, finish with a mean handed in), real and complex input
for cplx in (True, False):
    hop, M = 2048, 700
    n = (M - 1) * hop + nfft
    s = rng.standard_normal(n).astype(np.float32) + np.float32(2.0)
    if cplx:
        s = (s + 1j * rng.standard_normal(n)).astype(np.complex64)
    tot = E.welch_accum(s, win, hop, M)
    assert E.profile_last_kernel().startswith("k_welch_pipe"), E.profile_last_kernel()
    ssum = s.astype(np.complex128).sum()
    assert abs(complex(tot[0], tot[1]) - ssum) <= 1e-6 * abs(ssum) + 1e-3
    p = E.welch_finish(nfft, np.array([ssum.real / n, ssum.imag / n]), M, sided=E.SIDED_TWO, scale=1.0 / S2)
    ref = O.welch_psd_stream(s.astype(np.complex64), win, nfft, hop, M, 1.0, detrend_style=1)
    err = float(np.max(np.abs(p - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))
    assert err <= 1.0, (cplx, err)
print("pipe ok worst/tolerance %%.4f" %% worst)
""" % ROOT


def test_pipeline_kernel_parity_forced():
    env = dict(os.environ, SP_WELCH_PIPE="2")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pipe ok" in r.stdout


def test_pipeline_kernel_is_the_default_at_the_metric_shape():
    """2^25 complex64 samples (16383 frames of 4096, 50 %% overlap) take the pipeline by default; identical input through
    SP_WELCH_PIPE=0 (k_welch_carry) agrees to float32 accumulation noise"""
    child = r"""
import numpy as np, sys, torch
sys.path.insert(0, %r)
from pyfft_amd import engine as E
from oracle import cpu_ref as O
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = torch.view_as_complex(torch.randn((1 << 25, 2), generator=g, device="cuda", dtype=torch.float32)) + (0.25 - 0.5j)
win = O.windows("Hanning", nwins=4096)
M = ((1 << 25) - 4096) // 2048 + 1
p = E.welch_psd(x, win, 2048, M, detrend=True, sided=E.SIDED_TWO, scale=1.0).cpu().numpy()
print("KERNEL", E.profile_last_kernel())
np.save(sys.argv[1], p)
# no overlap: the one-pass mean detrend stays on the symmetric kernel (the pipeline's front role spills there and ran at half
# its rate), the plain accumulation takes the pipeline
M0 = (1 << 25) // 4096
q = E.welch_psd(x, win, 4096, M0, detrend=True, sided=E.SIDED_TWO, scale=1.0)
print("KERNEL0 onepass", E.profile_last_kernel())
q = E.welch_psd(x, win, 4096, M0, detrend=False, sided=E.SIDED_TWO, scale=1.0)
print("KERNEL0 plain", E.profile_last_kernel())
""" % ROOT
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        outs = {}
        for mode in ("1", "0"):
            f = os.path.join(d, "p%s.npy" % mode)
            r = subprocess.run([sys.executable, "-c", child, f], env=dict(os.environ, SP_WELCH_PIPE=mode),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout + r.stderr
            assert ("KERNEL k_welch_pipe" in r.stdout) == (mode == "1"), r.stdout
            assert "KERNEL0 onepass k_welch_pipe" not in r.stdout, r.stdout
            assert ("KERNEL0 plain k_welch_pipe" in r.stdout) == (mode == "1"), r.stdout
            outs[mode] = np.load(f)
    assert np.max(np.abs(outs["1"] - outs["0"])) <= 2e-5 * outs["0"].max()
