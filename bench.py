#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of Welch PSD, 4096-pt periodic Hann, 50 % overlap, complex64 stream
(BASELINE.json metric; reference path fftanal.fft_win -> Pstft -> averagewins, fft_analysis.py:2126-2203,
:1944-1990), on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: launches its own N ranks through
                                                          torch.distributed.run before anything touches a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input: every rank runs the fused Welch PSD
(global-mean detrend + window + overlapped FFT + |X|^2 + segment average, one pass over the samples) over its own
2^28-sample segment of the stream, device-resident; at N>1 the shards' additive states (|X|^2 accumulator + what is
needed to apply the mean of the whole stream: 5*4096+8 doubles) are summed with ONE RCCL all-reduce (the only
exchange the path has), issued asynchronously so that it overlaps the kernels of the next step (pyfft_amd.dist.
WelchPipeline; the timed region holds K submits and the final flush = exactly K steps of work).  Weak scaling: per-GPU
work is fixed, value = all samples / max-over-ranks time.

Prints ONE JSON line on rank 0.  Extra objects: roofline (dominant kernel k_welch, HIP events on the launch
stream; traffic from the PMC record of THIS build under profiles/, else null), cpu_baseline (the CPU oracle
`welch_psd_stream` on the host cores, bounded sample: all usable cores and one core; N=1 only).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pyfft_amd import engine as E          # noqa: E402
from pyfft_amd.windows import windows      # noqa: E402
from pyfft_amd.dist import shard_plan, WelchPipeline   # noqa: E402

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
PMC_RECORD = os.path.join(ROOT, "profiles", "pmc_metric_kernel_current.json")


def kernel_source_digest():
    """sha256 over the kernel sources: ties a PMC record to the build it was taken from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pyfft_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def pmc_traffic(log2n, nfft):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC record of the CURRENT kernel sources
    (tools/pmc_record.py writes it: FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections as calibrated in
    profiles/); None when there is no record for this build and workload -- never a remembered number."""
    try:
        with open(PMC_RECORD) as f:
            rec = json.load(f)
        if rec.get("source_sha256") == kernel_source_digest() and rec.get("log2n") == log2n and rec.get("nfft") == nfft:
            return float(rec["traffic_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(ngpus):
    """--gpus N > 1 without a launcher: start N ranks (one per GPU) through torch.distributed.run as a child process.
    Nothing in this process has touched a GPU yet (imports only).  Rank 0 of the children prints the JSON line; the
    exit code is the children's."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def synth_stream(n0, n, device, seed):
    """complex64 samples [n0, n0+n) of the synthetic stream: unit-variance complex white
    noise + the two Heinzel section-13 tones at fs = 1.  Noise comes from torch's device generator seeded per
    segment; phases are evaluated in float64."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = torch.empty(n, dtype=torch.complex64, device=device)
    chunk = 1 << 24
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        k = torch.arange(n0 + a, n0 + b, dtype=torch.float64, device=device)
        noise = torch.randn((b - a, 2), generator=gen, dtype=torch.float32, device=device) * (0.5 ** 0.5)
        z = torch.view_as_complex(noise).to(torch.complex128)
        z = z + 2.82842712 * torch.exp(2j * np.pi * torch.remainder(0.1234 * k, 1.0))
        z = z + 1.0 * torch.exp(2j * np.pi * torch.remainder(0.25002157 * k, 1.0))
        z = z + (0.05 - 0.02j)                       # a small offset so the detrend has something to remove
        out[a:b] = z.to(torch.complex64)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-steps", type=int, default=60,
                    help="untimed steps before the warm-up steps so that the GPU clocks have ramped: the first ~40 "
                         "steps after idle run 10-25 %% slower (tools/steptrace.py).  A fixed count, not a time, so "
                         "that every rank issues the same collectives.  0 = off")
    ap.add_argument("--log2n", type=int, default=28, help="samples per GPU = 2^log2n")
    ap.add_argument("--nfft", type=int, default=4096)
    ap.add_argument("--cpu-log2n", type=int, default=28, help="CPU-baseline sample = first 2^k samples (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the all-cores CPU baseline "
                                                             "(0 = the cores this process may run on, at most 32)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("SP_BENCH_DRYRUN", "") not in ("", "0"):
        # launch-plumbing rehearsal for machines without a GPU (tests/test_dist_cpu.py): rendezvous over gloo, one
        # all-reduce, one JSON line from rank 0 -- no kernels, not a measurement
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "rank_sum": float(t.item()), "steps": args.steps}))
        dist.destroy_process_group()
        return
    dist = None
    # rehearsal on a one-GPU box: SP_BENCH_ONE_GPU=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the driver's real runs use one GPU per rank over RCCL
    one_gpu = os.environ.get("SP_BENCH_ONE_GPU", "") not in ("", "0")
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nfft = args.nfft
    hop = nfft // 2
    S = 1 << args.log2n                                  # samples per GPU (weak scaling)
    total = S * world                                    # whole stream
    plan = shard_plan(total, nfft, hop, world, rank)     # contiguous frame ranges + (nfft-hop)-sample halo
    M_total, M_local, n_local = plan.frames_total, plan.frames, plan.nsamples
    x = synth_stream(plan.first_sample, n_local, dev, seed=0x5EED2024 + rank)

    win = windows("Hanning", nwins=nfft, verbose=False)
    S2 = float(np.sum(win ** 2))
    Fs = 1.0
    scale = 1.0 / (Fs * S2)

    pipe = WelchPipeline(win, plan, scale=scale, sided=E.SIDED_TWO) if world > 1 else None

    def step():
        # the whole hot path: global-mean detrend (fft_analysis.py:2148) + window + overlapped FFT + |X|^2 + segment
        # average, in ONE pass over the samples; at N>1: one all_reduce of the shard states, started asynchronously and
        # consumed one step later (pyfft_amd/dist.py: WelchPipeline) -- submit() returns the PREVIOUS step's PSD
        if world == 1:
            return E.welch_psd(x, win, hop, M_local, detrend=True, sided=E.SIDED_TWO, scale=scale)
        return pipe.submit(x)

    def drain():
        return pipe.flush() if pipe is not None else None

    E.profile_enable(False)          # the HIP-event hook is only switched on for the roofline measurement below
    for _ in range(max(0, args.settle_steps)):
        step()
    drain()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        pxx = step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pxx = step()
    last = drain()                   # the K-th step's all-reduce and finish kernel are inside the timed region
    if last is not None:
        pxx = last
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant-kernel duration, HIP events on the launch stream, measured in separate (untimed-region) steps so
    # the event waits do not perturb the timed loop
    kd = []
    E.profile_enable(True)
    for _ in range(max(5, min(args.steps, 20))):
        step()
        kd.append(E.profile_last_ms())
    drain()
    torch.cuda.synchronize()
    k_ms = float(np.mean(kd))

    ms_per_step = 1e3 * elapsed / args.steps
    value = (total / 1e6) / (elapsed / args.steps)       # Msamples/s, whole job
    alg_bytes = 8.0 * n_local                            # 8 B per complex64 input sample, read once
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9           # GB/s

    result = {
        "metric": "Msamples/sec Welch-PSD 4096-pt Hann 50% overlap, complex64",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Welch PSD, 2^%d complex64 samples per GPU, nfft=%d periodic Hann, hop=%d, "
                               "global mean detrend, two-sided" % (args.log2n, nfft, hop),
                   "samples_per_gpu": S, "frames_per_gpu": M_local, "settle_steps": args.settle_steps, "parallelism": "segment-sharded x%d, "
                   "one RCCL all-reduce of the shard state (%d doubles)" % (world, 5 * nfft + 8)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(args.log2n, nfft) if world == 1 else None,
                     "kernel": "%s<%d,complex64>" % (E.profile_last_kernel(), nfft),
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "VALU + LDS co-bound (ablation table, DESIGN.md); traffic = rocprofv3 PMC bytes of this build "
                             "(profiles/pmc_metric_kernel_current.json) or null"},
    }

    if rank == 0 and world == 1 and args.cpu_log2n > 0:
        # CPU baseline: the oracle's streaming restatement of the same path ("port": numpy float64 pocketfft, identical
        # arithmetic to the reference's fft_win -> Pstft -> averagewins) on the first 2^k samples of the same stream,
        # (i) on ONE core -- the reference's own CPU path is single-threaded -- and (ii) on every core this process may
        # use (frame ranges dealt to worker processes; oracle/cpu_parallel.py); the one-core run doubles as the in-run
        # parity gate.
        from oracle import cpu_ref as O
        from oracle import cpu_parallel
        nc = min(S, 1 << args.cpu_log2n)
        Mc = (nc - nfft) // hop + 1
        xc = x[:nc].cpu().numpy()
        t1 = time.perf_counter()
        ref = O.welch_psd_stream(xc, win, nfft, hop, Mc, Fs)
        tc = time.perf_counter() - t1
        got = E.welch_psd(x[:nc], win, hop, Mc, detrend=True, sided=E.SIDED_TWO, scale=scale).cpu().numpy()
        err = float(np.max(np.abs(got - ref) / (2e-4 * np.abs(ref) + 1e-6 * ref.max())))
        one = {"value": (nc / 1e6) / tc, "unit": "Msamples/s", "cores": 1, "seconds": tc}
        cores = cpu_parallel.usable_cores(args.cpu_cores)
        result["cpu_baseline"] = dict(one, kind="port",
                                      sample="first 2^%d samples of the same stream (%d frames), numpy float64 pocketfft, "
                                             "%.1f s" % (args.cpu_log2n, Mc, tc),
                                      host_cores_available=os.cpu_count())
        if cores > 1:
            try:
                par, tp = cpu_parallel.welch_psd_parallel(xc, win, nfft, hop, Mc, Fs, cores)
                perr = float(np.max(np.abs(par - ref)) / ref.max())
                result["cpu_baseline"] = {"value": (nc / 1e6) / tp, "unit": "Msamples/s", "cores": cores, "kind": "port",
                                          "sample": "first 2^%d samples of the same stream (%d frames), numpy float64 "
                                                    "pocketfft, frame ranges over %d worker processes, %.1f s (agrees "
                                                    "with the one-core result to %.1e of the peak)"
                                                    % (args.cpu_log2n, Mc, cores, tp, perr),
                                          "one_core": one, "host_cores_available": os.cpu_count()}
            except Exception as exc:                       # a box that refuses worker processes keeps the one-core line
                result["cpu_baseline"]["all_cores_error"] = repr(exc)
        result["parity"] = {"vs": "oracle.welch_psd_stream on the CPU sample", "rtol": 2e-4, "atol_rel_max": 1e-6,
                            "worst_over_tolerance": err, "ok": bool(err <= 1.0)}
        if err > 1.0:
            result["value"] = 0.0
            result["error"] = "parity gate failed"

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
