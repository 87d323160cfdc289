"""N > 1 path on CPU: two gloo ranks run pyfft_amd.dist.welch_psd_sharded with the CPU oracle standing in for the
device kernels (the HIP library cannot run here); checks the shard plan (halo, ownership), the order/size of the two
collectives, and that the sharded result equals the single-process PSD of the whole stream."""
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
    """(accum, finish) with the semantics of sp_welch_accum / sp_welch_finish, computed by the oracle."""
    state = {}

    def accum(x, w, hop, frames, nmean):
        state.update(x=np.asarray(x), hop=hop, frames=frames)
        s = np.asarray(x[:nmean]).astype(np.complex128).sum()
        return np.array([s.real, s.imag])

    def finish(n, mean, frames_total, sided, scale, like):
        x = state["x"].astype(np.complex128) - (mean[0] + 1j * mean[1])
        p = O.welch_psd_stream(x, win, n, state["hop"], state["frames"], 1.0, detrend_style=0) * np.sum(win ** 2)
        return p * state["frames"] / frames_total * scale
    return accum, finish


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
