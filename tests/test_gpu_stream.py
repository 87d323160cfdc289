"""The streaming engine (sp_welch_dist_submit / _flush without a communicator; pyfft_amd.dist.NativeWelchPipeline): the epilogue of
step k runs on the library's own stream beside the main kernel of step k + 1, results arrive one submit late.  Shapes beyond
the metric's: every supported hop, real and complex input, transform lengths 256 .. 8192 (k_welch_carry as the main kernel),
changing shapes from step to step, few frames, interleaving with ordinary calls on the launch stream, error paths.  Reference:
the same path as sp_welch_psd (fft_analysis.py:2126-2203 -> :1946 -> :1980)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O


def _plan(total, nfft, hop):
    from pyfft_amd.dist import shard_plan
    return shard_plan(total, nfft, hop, 1, 0)


def _excess(got, ref):
    return float(np.max(np.abs(got - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))


@pytest.mark.parametrize("nfft,hop,M,cplx", [(4096, 2048, 9000, True), (4096, 1024, 8300, True), (4096, 2048, 17000, False),
                                             (1024, 512, 300, True), (256, 64, 1000, False), (8192, 8192, 40, True),
                                             (2048, 1024, 3, True), (4096, 2048, 1, True)])
def test_streamed_steps_match_oracle(nfft, hop, M, cplx):
    import torch
    from pyfft_amd import engine as E
    from pyfft_amd.dist import NativeWelchPipeline
    rng = np.random.default_rng(nfft + hop + M)
    total = (M - 1) * hop + nfft + 11
    win = O.windows("Hanning", nwins=nfft)
    plan = _plan(total, nfft, hop)
    assert plan.frames == M
    pipe = NativeWelchPipeline(win, plan, scale=1.0, sided=E.SIDED_TWO)
    xs, refs = [], []
    for k in range(4):
        x = rng.standard_normal(total) + (1j * rng.standard_normal(total) if cplx else 0.0) + (0.5 + 0.25 * k)
        x = x.astype(np.complex64 if cplx else np.float32)
        xs.append(torch.from_numpy(x).cuda())
        refs.append(O.welch_psd_stream(x, win, nfft, hop, M, 1.0, detrend_style=1) * np.sum(win ** 2))
    got = []
    for x in xs:
        r = pipe.submit(x)
        if r is not None:
            got.append(r)
    got += pipe.flush_all()
    assert len(got) == 4 and pipe.flush() is None
    for g, ref in zip(got, refs):
        assert _excess(g.cpu().numpy(), ref) <= 1.0


def test_streamed_steps_change_shape_and_interleave_with_plain_calls():
    """consecutive steps with different transform lengths, windows and sidedness; ordinary library calls (which use the shared
    scratch and the launch stream) between the submits; the outputs must not be disturbed"""
    import torch
    from pyfft_amd import engine as E
    from pyfft_amd.dist import NativeWelchPipeline
    rng = np.random.default_rng(5)
    cases = [(4096, 2048, "Hanning", E.SIDED_TWO), (1024, 256, "Hamming", E.SIDED_ONE), (4096, 2048, "Hanning", E.SIDED_RAW),
             (512, 512, "Hanning", E.SIDED_TWO)]
    outs, refs, keep = [], [], []
    for nfft, hop, wname, sided in cases:
        M = 8200 if nfft == 4096 else 257
        total = (M - 1) * hop + nfft
        x = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (1.0 - 0.5j)).astype(np.complex64)
        win = O.windows(wname, nwins=nfft)
        pipe = NativeWelchPipeline(win, _plan(total, nfft, hop), scale=1.0, sided=sided)
        xd = torch.from_numpy(x).cuda()
        assert pipe.submit(xd) is None           # (each pipeline object shares the ONE engine of the library)
        keep.append((pipe, xd))
        # an ordinary call in between: another Welch PSD and an FFT on the launch stream
        y = E.welch_psd(xd[: 64 * nfft], win, hop, (64 * nfft - nfft) // hop + 1, detrend=True, sided=E.SIDED_TWO, scale=1.0)
        z = E.fft(xd[:nfft].reshape(1, nfft))
        assert torch.isfinite(y).all() and torch.isfinite(z.real).all()
        full = O.welch_psd_stream(x, win, nfft, hop, M, 1.0, detrend_style=1) * np.sum(win ** 2)      # two-sided, shifted
        if sided == E.SIDED_ONE:
            ref = np.fft.ifftshift(full)[: nfft // 2].copy()
            ref[1:-1] *= 2.0
        elif sided == E.SIDED_RAW:
            ref = np.fft.ifftshift(full)
        else:
            ref = full
        refs.append(ref)
    # one engine, several pipeline objects: every step is reported to the object that submitted it
    for (p, _), ref in zip(keep, refs):
        res = outs if False else p.flush_all()
        assert len(res) == 1, len(res)
        assert _excess(res[0].cpu().numpy(), ref) <= 1.0


def test_stream_error_paths_and_empty_flush():
    from pyfft_amd import engine as E
    from pyfft_amd import _ffi
    from pyfft_amd._ffi import SpectralError
    import torch
    assert E.welch_dist_flush() == 0                                   # nothing in flight
    x = torch.zeros(16384 * 5, dtype=torch.complex64, device="cuda")
    with pytest.raises(SpectralError, match="not sharded"):
        E.welch_dist_submit(x, np.hanning(16384), 8192, 9, x.numel(), 9)          # long segments
    with pytest.raises(SpectralError, match="hop"):
        E.welch_dist_submit(x, np.hanning(1024), 100, 10, x.numel(), 10)          # unsupported hop
    with pytest.raises(SpectralError, match="frames_total"):
        E.welch_dist_submit(x, np.hanning(1024), 512, 10, x.numel(), 5)           # frames_total < nframes
    assert E.welch_dist_flush() == 0
    assert E.comm_info() == (0, -1)


@pytest.mark.parametrize("two_lanes", [False, True])
def test_main_kernel_lanes_follow_the_launch_stream(two_lanes, monkeypatch):
    """SP_DIST_TWO_LANES=1: the main kernels run on two lanes of the engine's own (even / odd steps) instead of the launch stream: every
    step's input is PRODUCED on the launch stream right before its submit (a large torch kernel, no synchronisation), so a lane
    that did not wait for the launch stream would read stale samples; twelve steps, each with its own mean and amplitude"""
    import torch
    from pyfft_amd import engine as E
    from pyfft_amd.dist import NativeWelchPipeline
    if two_lanes:
        monkeypatch.setenv("SP_DIST_TWO_LANES", "1")
    nfft, hop, M = 4096, 2048, 6000
    total = (M - 1) * hop + nfft
    rng = np.random.default_rng(77)
    base = (rng.standard_normal(total) + 1j * rng.standard_normal(total)).astype(np.complex64)
    win = O.windows("Hanning", nwins=nfft)
    ref0 = O.welch_psd_stream(base, win, nfft, hop, M, 1.0, detrend_style=1) * np.sum(win ** 2)     # detrended: the mean drops out
    bd = torch.from_numpy(base).cuda()
    pipe = NativeWelchPipeline(win, _plan(total, nfft, hop), scale=1.0, sided=E.SIDED_TWO)
    bufs = [torch.zeros_like(bd) for _ in range(3)]
    got, amps = [], []
    for k in range(12):
        a = 1.0 + 0.5 * k
        amps.append(a)
        buf = bufs[k % 3]                       # (a buffer is rewritten three submits later: its step was reported by then)
        torch.mul(bd, a, out=buf)
        buf += complex(0.3 * k, -0.1 * k)
        r = pipe.submit(buf)
        if r is not None:
            got.append(r)
    got += pipe.flush_all()
    assert len(got) == 12
    for g, a in zip(got, amps):
        assert _excess(g.cpu().numpy(), a * a * ref0) <= 1.0, a
