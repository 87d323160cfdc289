"""Two processes sharing the one GPU of the test box (gloo; RCCL refuses two ranks per device) run the sharded Welch
PSD through the real HIP kernels on device tensors; every rank must hold the single-process PSD of the whole stream."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O


def _worker(rank, world, port, total, nfft, hop, out_dir):
    import torch
    import torch.distributed as dist
    from pyfft_amd.dist import shard_plan, welch_psd_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rng = np.random.default_rng(77)
    stream = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (0.8 - 0.3j)).astype(np.complex64)
    stream[: total // 2] += 1.5
    plan = shard_plan(total, nfft, hop, world, rank)
    x_local = torch.from_numpy(stream[plan.first_sample: plan.first_sample + plan.nsamples]).cuda()
    win = O.windows("Hanning", nwins=nfft)
    p = welch_psd_sharded(x_local, win, plan, scale=1.0)
    assert p.is_cuda
    np.save(os.path.join(out_dir, "p%d.npy" % rank), p.cpu().numpy())
    dist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_welch(tmp_path):
    import torch.multiprocessing as mp
    world, total, nfft, hop = 2, 4096 + 2048 * 600, 4096, 2048
    port = 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(world, port, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(77)
    stream = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (0.8 - 0.3j)).astype(np.complex64)
    stream[: total // 2] += 1.5
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1
    ref = O.welch_psd_stream(stream, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    for r in range(world):
        p = np.load(os.path.join(str(tmp_path), "p%d.npy" % r))
        np.testing.assert_allclose(p, ref, rtol=2e-4, atol=1e-6 * ref.max())


def _csd_worker(rank, world, port, nch, total, nfft, hop, out_dir):
    import torch
    import torch.distributed as dist
    from pyfft_amd.dist import shard_plan, csd_matrix_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x = _csd_record(nch, total)
    plan = shard_plan(total, nfft, hop, world, rank)
    x_local = torch.from_numpy(np.ascontiguousarray(x[:, plan.first_sample: plan.first_sample + plan.nsamples])).cuda()
    win = O.windows("Hanning", nwins=nfft)
    g = csd_matrix_sharded(x_local, win, plan, scale=1.0)
    assert g.is_cuda
    np.save(os.path.join(out_dir, "g%d.npy" % rank), g.cpu().numpy())
    dist.destroy_process_group()


def _csd_record(nch, total):
    rng = np.random.default_rng(5)
    common = rng.standard_normal(total)
    x = np.stack([(0.2 + 0.02 * c) * np.roll(common, c) + rng.standard_normal(total) + 0.05 * c for c in range(nch)])
    x[:, : total // 3] += 0.9                                     # shard means differ from the record means
    return x.astype(np.float32)


def test_two_ranks_one_gpu_sharded_csd_matrix(tmp_path):
    """cfg5 shape (reduced): two frame shards of a 64-channel record, HIP kernels on device tensors, the matrix
    all-reduced; every rank must hold the single-process matrix of the whole record (global-mean detrend)"""
    import torch.multiprocessing as mp
    world, nch, total, nfft, hop = 2, 64, 256 + 128 * 300, 256, 128
    port = 29900 + os.getpid() % 300
    mp.spawn(_csd_worker, args=(world, port, nch, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    x = _csd_record(nch, total)
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1
    ref = O.csd_matrix(x.astype(np.float64), win, nfft, hop, M, 1.0) * np.sum(win ** 2)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "g%d.npy" % r))
        assert np.max(np.abs(g - ref)) <= 2e-4 * np.abs(ref).max()


def test_channel_means_and_given_means(tmp_path):
    """sp_channel_means / sp_csd_matrix_means: a matrix computed with the caller's constants equals the self-detrended
    one when the constants are the channel means"""
    from pyfft_amd import engine as E
    x = _csd_record(7, 5000)
    m = E.channel_means(x)
    np.testing.assert_allclose(m, x.astype(np.float64).mean(axis=1), rtol=1e-6, atol=1e-7)
    m2 = E.channel_means(x, 1234)
    np.testing.assert_allclose(m2, x[:, :1234].astype(np.float64).mean(axis=1), rtol=1e-6, atol=1e-7)
    win = O.windows("Hanning", nwins=256)
    M = (5000 - 256) // 128 + 1
    g0 = E.csd_matrix(x, win, 128, M, detrend=True)
    g1 = E.csd_matrix(x, win, 128, M, means=m)
    assert np.max(np.abs(g0 - g1)) <= 1e-6 * np.abs(g0).max()


def _cog_stream(total):
    rng = np.random.default_rng(78)
    k = np.arange(total)
    f = 0.05 + 0.2 * k / total
    z = np.exp(2j * np.pi * np.cumsum(f)) + 0.2 * (rng.standard_normal(total) + 1j * rng.standard_normal(total))
    return z.astype(np.complex64)


def _cog_worker(rank, world, port, total, nfft, hop, out_dir):
    import torch
    import torch.distributed as dist
    from pyfft_amd.dist import shard_plan, cog_frames_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    z = _cog_stream(total)
    plan = shard_plan(total, nfft, hop, world, rank)
    x_local = torch.from_numpy(z[plan.first_sample: plan.first_sample + plan.nsamples]).cuda()
    c = cog_frames_sharded(x_local, np.ones(nfft), plan, 1.0e3)
    assert c.is_cuda and c.shape == (plan.frames_total,)
    np.save(os.path.join(out_dir, "c%d.npy" % rank), c.cpu().numpy())
    dist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_cog_frames(tmp_path):
    """frames dealt out to two ranks, the streaming moments kernel on each shard (+ halo), results all-gathered: every rank
    holds the single-process vector of the whole stream"""
    import torch.multiprocessing as mp
    world, total, nfft, hop = 2, 1024 + 512 * 801, 1024, 512
    port = 30300 + os.getpid() % 300
    mp.spawn(_cog_worker, args=(world, port, total, nfft, hop, str(tmp_path)), nprocs=world, join=True)
    z = _cog_stream(total)
    _, ref = O.cog_frames(np.arange(total) / 1.0e3, z, 1.0e3, win=nfft, ov=0.5)
    for r in range(world):
        c = np.load(os.path.join(str(tmp_path), "c%d.npy" % r))
        np.testing.assert_allclose(c, ref, rtol=0, atol=3e-6 * 1.0e3)


def _pipe_fir_worker(rank, world, port, total, nfft, hop, ntaps, out_dir):
    import torch
    import torch.distributed as dist
    from pyfft_amd.dist import shard_plan, WelchPipeline, sample_shard_plan, fftfilt_sharded, stft_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    win = O.windows("Hanning", nwins=nfft)
    plan = shard_plan(total, nfft, hop, world, rank)
    pipe = WelchPipeline(win, plan, scale=1.0)
    outs = []
    for step in range(3):
        rng = np.random.default_rng(900 + step)
        stream = (rng.standard_normal(total) + 1j * rng.standard_normal(total) + (0.2 * step - 0.4j)).astype(np.complex64)
        r = pipe.submit(torch.from_numpy(stream[plan.first_sample: plan.first_sample + plan.nsamples]).cuda())
        if r is not None:
            outs.append(r.cpu().numpy())
    outs.append(pipe.flush().cpu().numpy())
    # overlap-save FIR and STFT of one real stream dealt out to the ranks: no exchange on the data path
    rng = np.random.default_rng(31)
    xr = rng.standard_normal(total).astype(np.float32)
    h = (rng.standard_normal(ntaps) / ntaps).astype(np.float32)
    sp = sample_shard_plan(total, ntaps, world, rank)
    y = fftfilt_sharded(h, torch.from_numpy(xr[sp.read_first: sp.read_first + sp.nread]).cuda(), sp)
    S = stft_sharded(torch.from_numpy(xr[plan.first_sample: plan.first_sample + plan.nsamples]).cuda(), win, plan,
                     mean_value=float(xr.astype(np.float64).mean()))
    assert y.is_cuda and S.is_cuda
    np.savez(os.path.join(out_dir, "q%d.npz" % rank), psd=np.stack(outs), y=y.cpu().numpy(), first=sp.first, last=sp.last,
             S=S.cpu().numpy(), f0=plan.first_frame)
    dist.destroy_process_group()


def test_two_ranks_one_gpu_pipeline_fir_stft(tmp_path):
    """WelchPipeline (async all-reduce consumed one submit later), fftfilt_sharded (halo = ntaps - 1) and stft_sharded with
    the HIP kernels on device tensors; results against the single-process oracle"""
    import torch.multiprocessing as mp
    world, total, nfft, hop, ntaps = 2, 4096 + 2048 * 150, 4096, 2048, 129
    port = 30700 + os.getpid() % 300
    mp.spawn(_pipe_fir_worker, args=(world, port, total, nfft, hop, ntaps, str(tmp_path)), nprocs=world, join=True)
    win = O.windows("Hanning", nwins=nfft)
    M = (total - nfft) // hop + 1
    rng = np.random.default_rng(31)
    xr = rng.standard_normal(total).astype(np.float32)
    h = (rng.standard_normal(ntaps) / ntaps).astype(np.float32)
    yref = O.fftfilt(h.astype(np.float64), xr.astype(np.float64))
    xd = xr.astype(np.float64) - xr.astype(np.float64).mean()
    covered = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "q%d.npz" % r))
        for step in range(3):
            g = np.random.default_rng(900 + step)
            stream = (g.standard_normal(total) + 1j * g.standard_normal(total) + (0.2 * step - 0.4j)).astype(np.complex64)
            ref = O.welch_psd_stream(stream, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
            np.testing.assert_allclose(d["psd"][step], ref, rtol=2e-4, atol=1e-6 * ref.max())
        a, b = int(d["first"]), int(d["last"])
        assert np.max(np.abs(d["y"] - yref[a:b])) <= 1e-4 * np.abs(yref).max()
        covered += b - a
        f0 = int(d["f0"])
        nf = d["S"].shape[0]
        idx = (np.arange(f0, f0 + nf) * hop)[:, None] + np.arange(nfft)[None, :]
        Sref = np.fft.fft(win[None, :] * xd[idx], axis=-1)[:, :nfft // 2]
        Sref[:, 1:-1] *= np.sqrt(2.0)
        assert np.max(np.abs(d["S"] - Sref)) <= 1e-4 * np.abs(Sref).max()
    assert covered == total
