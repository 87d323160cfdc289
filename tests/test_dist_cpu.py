"""N > 1 path on CPU: two gloo ranks run pyfft_amd.dist.welch_psd_sharded with the CPU oracle standing in for the
device kernels (the HIP library cannot run here); checks the shard plan (halo, ownership), the single collective, and that the sharded result equals the single-process PSD of the whole stream."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

from pyfft_amd.dist import shard_plan, welch_psd_sharded
from oracle import cpu_ref as O


def test_shard_plan_partitions_frames_and_samples():
    for total, nfft, hop, world in [(1 << 16, 4096, 2048, 2), (100000, 1024, 256, 3), (5000, 1000, 1000, 4),
                                    (1 << 20, 4096, 2048, 8), (12345, 256, 64, 5)]:
        plans = [shard_plan(total, nfft, hop, world, r) for r in range(world)]
        M = (total - nfft) // hop + 1
        assert sum(p.frames for p in plans) == M and plans[0].first_frame == 0
        assert sum(p.own_samples for p in plans) == total
        for a, b in zip(plans, plans[1:]):
            assert a.first_frame + a.frames == b.first_frame
            # halo: a reads into b's territory by nfft - hop samples
            assert a.first_sample + a.nsamples == b.first_sample + (nfft - hop)
        assert plans[-1].first_sample + plans[-1].nsamples == total
        for p in plans:
            assert (p.frames - 1) * hop + nfft <= p.nsamples
    with pytest.raises(ValueError):
        shard_plan(1000, 4096, 2048, 2, 0)
    with pytest.raises(ValueError):
        shard_plan(8192, 4096, 2048, 8, 0)


def _oracle_backend(win, nfft):
    """(export, apply) with the semantics of sp_welch_export / sp_welch_apply, computed in float64 by the oracle's
    building blocks: spectra against a deliberately rough local mean estimate mu0 (the mean of the shard's first 100
    samples), the additive state, and the formula that applies the global mean to the summed state."""
    W = np.fft.fft(np.asarray(win, dtype=np.float64))

    def export(x, w, hop, frames, nmean):
        x = np.asarray(x).astype(np.complex128)
        mu0 = x[:100].mean()
        idx = (np.arange(frames) * hop)[:, None] + np.arange(nfft)[None, :]
        X = np.fft.fft(np.asarray(w, dtype=np.float64) * (x[idx] - mu0), axis=-1)
        A, B = (np.abs(X) ** 2).sum(axis=0), X.sum(axis=0)
        C = np.conj(mu0) * B
        s = x[:nmean].sum()
        sc = [frames * mu0.real, frames * mu0.imag, frames * abs(mu0) ** 2, s.real, s.imag, float(frames), float(nmean), 0.0]
        return np.concatenate([A, np.stack([B.real, B.imag], axis=1).ravel(), np.stack([C.real, C.imag], axis=1).ravel(), sc])

    def apply(st, w, frames_total, sided, scale):
        n = nfft
        A, B, C, sc = st[:n], st[n:3 * n].reshape(n, 2), st[3 * n:5 * n].reshape(n, 2), st[5 * n:]
        B, C = B[:, 0] + 1j * B[:, 1], C[:, 0] + 1j * C[:, 1]
        mu = (sc[3] + 1j * sc[4]) / sc[6]
        S1 = sc[0] + 1j * sc[1]
        P = A - 2 * np.real(np.conj(W) * (np.conj(mu) * B - C)) + np.abs(W) ** 2 * (abs(mu) ** 2 * sc[5]
                                                                                   - 2 * np.real(np.conj(mu) * S1) + sc[2])
        assert sided == 2
        return np.fft.fftshift(P) * scale / frames_total
    return export, apply


def _worker(rank, world, port, total, nfft, hop, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)
    stream = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (0.4 - 1.1j)).astype(np.complex64)
    stream[: total // 3] += 1.0                       # shard means differ
    win = O.windows("Hanning", nwins=nfft)
    plan = shard_plan(total, nfft, hop, world, rank)
    x_local = stream[plan.first_sample: plan.first_sample + plan.nsamples]
    p = welch_psd_sharded(x_local, win, plan, scale=1.0, backend=_oracle_backend(win, nfft))
    np.save(os.path.join(out_dir, "p%d.npy" % rank), p)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_welch_matches_single_process(tmp_path, world):
    total, nfft, hop = 40000, 1024, 512
    port = 29000 + os.getpid() % 2000 + world
    mp.spawn(_worker, args=(world, port, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(1234)
    stream = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (0.4 - 1.1j)).astype(np.complex64)
    stream[: total // 3] += 1.0
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1
    ref = O.welch_psd_stream(stream, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    for r in range(world):
        p = np.load(os.path.join(str(tmp_path), "p%d.npy" % r))
        # (the reference subtracts the mean in the input dtype, complex64 here -- Q6 -- hence 1e-6, not 1e-12)
        np.testing.assert_allclose(p, ref, rtol=1e-6, atol=1e-9 * ref.max())


# ---- cfg5: frame-sharded CSD matrix, and the channel-sharded reference-vs-channels CSD -----------------------------
def _csd_record(nch, total):
    rng = np.random.default_rng(99)
    common = rng.standard_normal(total)
    x = np.stack([(0.2 + 0.1 * c) * np.roll(common, c) + rng.standard_normal(total) + 0.3 * c for c in range(nch)])
    x[:, : total // 4] += 0.7                                     # shard means differ from the record means
    return x.astype(np.float32)


def _oracle_csd_backend(nfft):
    def means(x, n):
        return np.asarray(x, dtype=np.float64)[:, :n].mean(axis=1)

    def matrix(x, win, hop, frames, gmean, scale):
        xd = np.asarray(x, dtype=np.float64) - np.asarray(gmean)[:, None]
        return O.csd_matrix(xd, win, nfft, hop, frames, 1.0, detrend_style=0) * np.sum(win ** 2) * scale
    return means, matrix


def _csd_worker(rank, world, port, nch, total, nfft, hop, out_dir):
    from pyfft_amd.dist import csd_matrix_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = _csd_record(nch, total)
    win = O.windows("Hanning", nwins=nfft)
    plan = shard_plan(total, nfft, hop, world, rank)
    x_local = x[:, plan.first_sample: plan.first_sample + plan.nsamples]
    g = csd_matrix_sharded(x_local, win, plan, scale=1.0, backend=_oracle_csd_backend(nfft), compact=False)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), g)
    # compact: only the Hermitian upper triangle travels, in complex64 (34 MB instead of 134 MB at cfg5)
    gc = csd_matrix_sharded(x_local, win, plan, scale=1.0, backend=_oracle_csd_backend(nfft), compact=True)
    np.save(os.path.join(out_dir, "gc%d.npy" % rank), gc)
    # default with a caller-supplied (float64) backend: the full-precision exchange (no silent float32)
    gd = csd_matrix_sharded(x_local, win, plan, scale=1.0, backend=_oracle_csd_backend(nfft))
    np.save(os.path.join(out_dir, "gd%d.npy" % rank), gd)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_csd_matrix_matches_single_process(tmp_path, world):
    nch, total, nfft, hop = 5, 9000, 256, 128
    port = 31000 + os.getpid() % 2000 + world
    mp.spawn(_csd_worker, args=(world, port, nch, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    x = _csd_record(nch, total)
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1
    ref = O.csd_matrix(x.astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "g%d.npy" % r))
        assert np.max(np.abs(g - ref)) <= 1e-10 * np.abs(ref).max()
        gc = np.load(os.path.join(str(tmp_path), "gc%d.npy" % r))
        assert gc.dtype == np.complex128 and gc.shape == ref.shape
        assert np.max(np.abs(gc - ref)) <= 5e-7 * np.abs(ref).max()            # float32 exchange
        assert np.max(np.abs(gc - np.conj(np.swapaxes(gc, 1, 2)))) <= 1e-7 * np.abs(ref).max()   # Hermitian by construction
        gd = np.load(os.path.join(str(tmp_path), "gd%d.npy" % r))
        assert np.max(np.abs(gd - ref)) <= 1e-10 * np.abs(ref).max()


def _chan_worker(rank, world, port, nch, total, nfft, hop, out_dir):
    from pyfft_amd.dist import welch_csd_channel_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec = _csd_record(nch + 1, total).astype(np.float64)
    x, y = rec[0], rec[1:]
    per = nch // world
    y_local = y[rank * per:(rank + 1) * per]
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1

    def backend(a, b):          # oracle: one-row CSD from the full-matrix restatement
        g = O.csd_matrix(np.vstack([a[None, :], b]), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
        return g[:, 0, 0].real, np.stack([g[:, c, c].real for c in range(1, b.shape[0] + 1)]), \
            np.stack([g[:, c, 0] for c in range(1, b.shape[0] + 1)])
    pxx, pyy, pxy = welch_csd_channel_sharded(x, y_local, win, hop, M, backend=backend)
    np.savez(os.path.join(out_dir, "c%d.npz" % rank), pxx=pxx, pyy=pyy, pxy=pxy)
    dist.destroy_process_group()


def test_channel_sharded_csd_gathers_all_channels(tmp_path):
    world, nch, total, nfft, hop = 2, 4, 6000, 256, 128
    port = 33000 + os.getpid() % 2000
    mp.spawn(_chan_worker, args=(world, port, nch, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    rec = _csd_record(nch + 1, total).astype(np.float64)
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1
    g = O.csd_matrix(rec, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "c%d.npz" % r))
        assert d["pyy"].shape == (nch, nfft // 2 + 1) and d["pxy"].shape == (nch, nfft // 2 + 1)
        for c in range(nch):
            np.testing.assert_allclose(d["pyy"][c], g[:, c + 1, c + 1].real, rtol=1e-12)
            np.testing.assert_allclose(d["pxy"][c], g[:, c + 1, 0], rtol=1e-10, atol=1e-12 * np.abs(g).max())


def _cog_stream(total):
    rng = np.random.default_rng(77)
    k = np.arange(total)
    f = 0.05 + 0.2 * k / total
    return np.exp(2j * np.pi * np.cumsum(f)) + 0.2 * (rng.standard_normal(total) + 1j * rng.standard_normal(total))


def _cog_backend(x, w, hop, frames, fs, lo, hi, mv):      # oracle in place of the HIP kernels
    xx = np.asarray(x) - (mv if mv is not None else 0.0)
    t = np.arange(len(xx)) / fs
    nfft = len(w)
    assert (len(xx) - nfft) // hop + 1 >= frames
    ov = 1.0 - hop / nfft
    return O.cog_frames(t, xx, fs, win=nfft, ov=ov, fmin=lo if lo else None, fmax=hi, window=w)[1][:frames]


def _cog_worker(rank, world, port, total, nfft, hop, out_dir):
    import torch.distributed as dist
    from pyfft_amd.dist import shard_plan, cog_frames_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = _cog_stream(total)
    plan = shard_plan(total, nfft, hop, world, rank)
    x_local = z[plan.first_sample:plan.first_sample + plan.nsamples]
    w = O.windows("Hanning", nwins=nfft)
    full = cog_frames_sharded(x_local, w, plan, 1.0e3, mean_value=z.mean(), backend=_cog_backend)
    own = cog_frames_sharded(x_local, w, plan, 1.0e3, mean_value=z.mean(), gather=False, backend=_cog_backend)
    np.savez(os.path.join(out_dir, "g%d.npz" % rank), full=full, own=own, f0=plan.first_frame)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_cog_frames_needs_no_exchange_and_gathers(tmp_path, world):
    """frames are independent: every rank's own results equal its slice of the single-process vector; gather=True returns
    the whole vector on every rank (uneven frame counts: 3 ranks over 77 frames)"""
    total, nfft, hop = 5120, 256, 64
    port = 35000 + os.getpid() % 2000 + world
    mp.spawn(_cog_worker, args=(world, port, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    z = _cog_stream(total)
    M = (total - nfft) // hop + 1
    ref = _cog_backend(z, O.windows("Hanning", nwins=nfft), hop, M, 1.0e3, 0.0, None, z.mean())
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "g%d.npz" % r))
        assert d["full"].shape == (M,)
        np.testing.assert_allclose(d["full"], ref, rtol=1e-9, atol=1e-9)
        f0 = int(d["f0"])
        np.testing.assert_allclose(d["own"], ref[f0:f0 + len(d["own"])], rtol=1e-9, atol=1e-9)


# ---- bench.py launches its own ranks when asked for N > 1 GPUs without a launcher ----------------------------------
def test_bench_self_launch_dryrun():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself.  SP_BENCH_DRYRUN=1
    replaces the GPU work with one gloo all-reduce so that the launch plumbing is exercised on a machine without GPUs."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SP_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dryrun"] and d["n_gpus"] == 2 and d["rank_sum"] == 3.0 and d["steps"] == 3
    # the strong-scaling split of the 2^28-sample stream deals out every frame exactly once
    assert d["strong_frames_total"] == (2 ** 28 - 4096) // 2048 + 1 == d["strong_frames_sum"]
    # a failing child makes the parent exit non-zero
    env["SP_BENCH_DRYRUN"] = "0"
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                         "--settle-steps", "0", "--cpu-log2n", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0


# ---- the pipelined form: the collective of step i is consumed in step i+1 --------------------------------------------
def _pipe_worker(rank, world, port, total, nfft, hop, out_dir):
    from pyfft_amd.dist import WelchPipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    win = O.windows("Hanning", nwins=nfft)
    plan = shard_plan(total, nfft, hop, world, rank)
    pipe = WelchPipeline(win, plan, scale=1.0, backend=_oracle_backend(win, nfft))
    outs = []
    for step in range(3):                                   # three different streams through the pipeline
        rng = np.random.default_rng(500 + step)
        stream = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (0.1 * step - 0.3j)).astype(np.complex64)
        r = pipe.submit(stream[plan.first_sample: plan.first_sample + plan.nsamples])
        assert (r is None) == (step == 0)
        if r is not None:
            outs.append(r)
    outs.append(pipe.flush())
    assert pipe.flush() is None
    np.save(os.path.join(out_dir, "pp%d.npy" % rank), np.stack(outs))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_welch_pipeline_returns_each_step_one_submit_later(tmp_path, world):
    total, nfft, hop = 20000, 512, 256
    port = 36000 + os.getpid() % 2000 + world
    mp.spawn(_pipe_worker, args=(world, port, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "pp%d.npy" % r))
        assert got.shape[0] == 3
        for step in range(3):
            rng = np.random.default_rng(500 + step)
            stream = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (0.1 * step - 0.3j)).astype(np.complex64)
            ref = O.welch_psd_stream(stream, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
            np.testing.assert_allclose(got[step], ref, rtol=1e-6, atol=1e-9 * ref.max())


# ---- paths without a reduction: overlap-save FIR and STFT of one long stream dealt out to the ranks --------------------
def _nored_worker(rank, world, port, total, ntaps, nfft, hop, out_dir):
    from pyfft_amd.dist import sample_shard_plan, fftfilt_sharded, stft_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(4242)
    x = rng.standard_normal(total)
    h = rng.standard_normal(ntaps) / ntaps
    sp = sample_shard_plan(total, ntaps, world, rank)
    xl = x[sp.read_first: sp.read_first + sp.nread]
    own = fftfilt_sharded(h, xl, sp, backend=O.fftfilt)
    full = fftfilt_sharded(h, xl, sp, gather=True, backend=O.fftfilt)
    fp = shard_plan(total, nfft, hop, world, rank)
    win = O.windows("Hamming", nwins=nfft)

    def stft_backend(xx, w, hp, frames, d, mv, sided, amp, power):       # oracle in place of engine.stft_frames
        xx = np.asarray(xx, dtype=np.float64) - (mv if mv is not None else 0.0)
        idx = (np.arange(frames) * hp)[:, None] + np.arange(len(w))[None, :]
        return amp * np.fft.fft(w[None, :] * xx[idx], axis=-1)[:, :len(w) // 2]
    S = stft_sharded(x[fp.first_sample: fp.first_sample + fp.nsamples], win, fp, mean_value=x.mean(), backend=stft_backend)
    np.savez(os.path.join(out_dir, "n%d.npz" % rank), own=own, full=full, first=sp.first, last=sp.last, S=S, f0=fp.first_frame)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fir_and_stft_need_no_exchange(tmp_path, world):
    total, ntaps, nfft, hop = 10007, 65, 256, 64
    port = 38000 + os.getpid() % 2000 + world
    mp.spawn(_nored_worker, args=(world, port, total, ntaps, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(4242)
    x = rng.standard_normal(total)
    h = rng.standard_normal(ntaps) / ntaps
    y = O.fftfilt(h, x)
    win = O.windows("Hamming", nwins=nfft)
    M = (total - nfft) // hop + 1
    idx = (np.arange(M) * hop)[:, None] + np.arange(nfft)[None, :]
    Sref = np.fft.fft(win[None, :] * (x - x.mean())[idx], axis=-1)[:, :nfft // 2]
    covered = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "n%d.npz" % r))
        a, b = int(d["first"]), int(d["last"])
        np.testing.assert_allclose(d["own"], y[a:b], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(d["full"], y, rtol=1e-10, atol=1e-12)
        covered += b - a
        f0 = int(d["f0"])
        np.testing.assert_allclose(d["S"], Sref[f0:f0 + d["S"].shape[0]], rtol=1e-10, atol=1e-10)
    assert covered == total
