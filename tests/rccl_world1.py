"""Child process of tests/test_gpu_rccl.py: an RCCL ("nccl") process group of ONE rank on cuda:0, created before any other
GPU call of the process, drives the sharded code paths -- WelchPipeline.submit/flush (async all-reduce consumed one step
later), welch_psd_sharded, csd_matrix_sharded(compact) -- with the collectives really issued (force_collective), and
compares with the single-process results / the CPU oracle.  Prints one JSON line."""
import json
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # first GPU call of the process
    torch.cuda.set_device(0)
    from pyfft_amd import engine as E
    from pyfft_amd.dist import shard_plan, WelchPipeline, welch_psd_sharded, csd_matrix_sharded
    from oracle import cpu_ref as O
    import synth

    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    nfft, hop = 4096, 2048
    win = O.windows("Hanning", nwins=nfft)
    S2 = float(np.sum(win ** 2))

    # (1) pipelined PSD, 5 steps over 5 different shards of the counter-based stream: result k must equal the one-call PSD
    total = nfft + hop * 9000                     # > 32 frames per CU: the pipeline kernel (k_welch_pipe) runs
    plan = shard_plan(total, nfft, hop, 1, 0)
    pipe = WelchPipeline(win, plan, scale=1.0 / S2, sided=E.SIDED_TWO, force_collective=True)
    xs = [synth.stream_torch(k * 1000003, total, dev) for k in range(5)]
    got = []
    for x in xs:
        r = pipe.submit(x)
        if r is not None:
            got.append(r)
    got.append(pipe.flush())
    worst = 0.0
    for x, g in zip(xs, got):
        one = E.welch_psd(x, win, hop, plan.frames, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)
        worst = max(worst, float(torch.max(torch.abs(g - one) / (2e-4 * torch.abs(one) + 1e-6 * one.max())).item()))
    out["pipeline_steps"] = len(got)
    out["pipeline_vs_single_call"] = worst

    # (2) welch_psd_sharded with the collective issued, against the CPU oracle on host-generated samples
    ng = 1 << 20
    pg = shard_plan(ng, nfft, hop, 1, 0)
    p = welch_psd_sharded(synth.stream_torch(0, ng, dev), win, pg, 1.0 / S2, sided=E.SIDED_TWO, force_collective=True).cpu().numpy()
    ref = O.welch_psd_stream(synth.stream_numpy(0, ng), win, nfft, hop, pg.frames_total, 1.0)
    out["sharded_vs_oracle"] = float(np.max(np.abs(p - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))

    # (3) cfg5 shape (reduced): the compact Hermitian all-reduce in complex64
    rng = np.random.default_rng(5)
    nch, n5, nf5, h5 = 16, 256 + 128 * 400, 256, 128
    common = rng.standard_normal(n5)
    rec = np.stack([(0.2 + 0.02 * c) * np.roll(common, c) + rng.standard_normal(n5) + 0.05 * c for c in range(nch)]).astype(np.float32)
    w5 = O.windows("Hanning", nwins=nf5)
    p5 = shard_plan(n5, nf5, h5, 1, 0)
    xd = torch.from_numpy(rec).to(dev)
    gc = csd_matrix_sharded(xd, w5, p5, scale=1.0, compact=True, force_collective=True)
    gf = csd_matrix_sharded(xd, w5, p5, scale=1.0, compact=False, force_collective=True)
    g0 = E.csd_matrix(xd, w5, h5, p5.frames, scale=1.0)
    refm = O.csd_matrix(rec.astype(np.float64), w5, nf5, h5, p5.frames, 1.0) * np.sum(w5 ** 2)
    pk = float(np.abs(refm).max())
    out["csd_compact_vs_local"] = float(torch.max(torch.abs(gc - g0)).item()) / pk
    out["csd_full_vs_local"] = float(torch.max(torch.abs(gf - g0)).item()) / pk
    out["csd_compact_vs_oracle"] = float(np.max(np.abs(gc.cpu().numpy() - refm))) / pk
    # (3b) the streaming engine WITHOUT a communicator (one GPU: the epilogue of step k beside the main kernel of step k + 1)
    from pyfft_amd.dist import NativeWelchPipeline as _NP
    spipe = _NP(win, plan, scale=1.0 / S2, sided=E.SIDED_TWO)
    got = []
    for x in xs + xs:                              # ten steps: both scratch sets are reused several times
        r = spipe.submit(x)
        if r is not None:
            got.append(r)
    got.extend(spipe.flush_all())
    worst = 0.0
    for x, g in zip(xs + xs, got):
        one = E.welch_psd(x, win, hop, plan.frames, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)
        worst = max(worst, float(torch.max(torch.abs(g - one) / (2e-4 * torch.abs(one) + 1e-6 * one.max())).item()))
    out["stream_steps"] = len(got)
    out["stream_vs_single_call"] = worst
    # real input (two frames per transform) and a non-cosine window (two-launch epilogue on the engine's stream)
    xr = [torch.randn(total, device=dev) + 0.5 * k for k in range(3)]
    from pyfft_amd.windows import get_window as _gw
    wk2 = np.asarray(_gw(("kaiser", 5.0), nfft, fftbins=True), dtype=np.float64)
    worst = 0.0
    for w_ in (win, wk2):
        rp = _NP(w_, plan, scale=1.0, sided=E.SIDED_TWO)
        res = [r for r in (rp.submit(x) for x in xr) if r is not None] + rp.flush_all()
        assert len(res) == 3
        for x, g in zip(xr, res):
            one = E.welch_psd(x, w_, hop, plan.frames, detrend=True, sided=E.SIDED_TWO, scale=1.0)
            worst = max(worst, float(torch.max(torch.abs(g - one) / (2e-4 * torch.abs(one) + 1e-6 * one.max())).item()))
    out["stream_real_and_kaiser"] = worst
    # (4) the same pipeline with the collective issued by libspectral itself (sp_comm_init + sp_welch_dist_submit/flush)
    from pyfft_amd.dist import native_comm_init, NativeWelchPipeline
    out["native_comm"] = list(native_comm_init(device=0))
    out["native_comm_info"] = list(E.comm_info())
    npipe = NativeWelchPipeline(win, plan, scale=1.0 / S2, sided=E.SIDED_TWO)
    got = []
    for x in xs:
        r = npipe.submit(x)
        if r is not None:
            got.append(r)
    got.extend(npipe.flush_all())                # (with a communicator results arrive two submits late: two are left)
    assert npipe.flush() is None
    worst = 0.0
    for x, g in zip(xs, got):
        one = E.welch_psd(x, win, hop, plan.frames, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)
        worst = max(worst, float(torch.max(torch.abs(g - one) / (2e-4 * torch.abs(one) + 1e-6 * one.max())).item()))
    out["native_pipeline_steps"] = len(got)
    out["native_pipeline_vs_single_call"] = worst
    # the partition used with more than one rank: the main kernel over ncu - 4 CUs, the rest left to RCCL's kernel (which cannot
    # run beside k_welch_pipe); forced here for the group of one rank
    os.environ["SP_DIST_RESERVE_CUS"] = "4"
    try:
        rpipe = NativeWelchPipeline(win, plan, scale=1.0 / S2, sided=E.SIDED_TWO)
        got = [r for r in (rpipe.submit(x) for x in xs) if r is not None] + rpipe.flush_all()
    finally:
        del os.environ["SP_DIST_RESERVE_CUS"]
    worst = 0.0
    for x, g in zip(xs, got):
        one = E.welch_psd(x, win, hop, plan.frames, detrend=True, sided=E.SIDED_TWO, scale=1.0 / S2)
        worst = max(worst, float(torch.max(torch.abs(g - one) / (2e-4 * torch.abs(one) + 1e-6 * one.max())).item()))
    out["native_reserved_steps"] = len(got)
    out["native_reserved_cus_vs_single_call"] = worst
    # a non-cosine-sum window (Kaiser): the epilogue takes the two-launch form (k_op_colsums + k_op_finish<EXPORT>)
    from pyfft_amd.windows import get_window
    wk = np.asarray(get_window(("kaiser", 8.0), nfft, fftbins=True), dtype=np.float64)
    pk = NativeWelchPipeline(wk, plan, scale=1.0, sided=E.SIDED_TWO)
    assert pk.submit(xs[0]) is None
    gk = pk.flush()
    onek = E.welch_psd(xs[0], wk, hop, plan.frames, detrend=True, sided=E.SIDED_TWO, scale=1.0)
    out["native_kaiser_vs_single_call"] = float(torch.max(torch.abs(gk - onek) / (2e-4 * torch.abs(onek) + 1e-6 * onek.max())).item())
    torch.cuda.synchronize()
    E.comm_destroy()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
