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
