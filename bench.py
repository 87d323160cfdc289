#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of Welch PSD, 4096-pt periodic Hann, 50 % overlap, complex64 stream
(BASELINE.json metric; reference path fftanal.fft_win -> Pstft -> averagewins, fft_analysis.py:2126-2203,
:1944-1990), on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: launches its own N ranks through
                                                          torch.distributed.run before anything touches a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input: the fused Welch PSD (global-mean detrend + window
+ overlapped FFT + |X|^2 + segment average, one pass over the samples) of ONE 2^28-sample complex64 stream, device-resident.
At N > 1 the stream is split N ways by frames (pyfft_amd.dist.shard_plan: contiguous frame ranges + a 2048-sample halo) --
STRONG scaling, the configuration BASELINE.json's metric names ("a 2^28-sample complex64 stream at 1, 2, 4 and 8 GPUs") and
the headline `value`; the shards' additive states (|X|^2 accumulator + what is needed to apply the mean of the WHOLE stream:
5*4096+8 doubles) are summed with ONE RCCL all-reduce per step (the only exchange the path has), issued asynchronously so
that it overlaps the kernels of the next step (pyfft_amd.dist.WelchPipeline; the timed region holds K submits and the final
flush = exactly K steps of work).  The weak-scaling figure (2^28 samples per GPU) is measured in the same run and reported
under "weak_scaling".  The stream is counter-based (synth.py: splitmix64(seed ^ k) -> Box-Muller + two tones), so every
rank generates exactly its own samples of the same stream and results are comparable across N: at N > 1 rank 0 gates the
timed result against one GPU's PSD of the whole stream and a 2^22-sample prefix through the same sharded path against the
CPU oracle.

Prints ONE JSON line on rank 0.  Extra objects: roofline (dominant kernel k_welch_pipe, HIP events on the launch
stream; traffic from the PMC record of THIS build under profiles/, else null), cpu_baseline (the CPU oracle
`welch_psd_stream` on the host cores, bounded sample: all usable cores and one core; N=1 only).
SP_BENCH_FORCE_DIST=1 (with --gpus 1): the sharded code path with an RCCL process group of one rank.
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
from pyfft_amd.dist import shard_plan, WelchPipeline, NativeWelchPipeline, native_comm_init   # noqa: E402
import synth                              # noqa: E402

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


def measure(E, x, plan, win, scale, world, dist, dev, args, collective, native):
    """settle + warm-up + EXACTLY args.steps timed steps of the hot path over this rank's shard `x` (barrier + synchronize on
    both sides, max over ranks), then the dominant kernel's duration by HIP events in further untimed steps."""
    if not collective and not native:
        pipe = None        # plain calls: E.welch_psd per step (SP_BENCH_STREAM=0)
    elif native:           # the streaming engine of libspectral (sp_welch_dist_submit): one call per step; the epilogue -- and at
        #                    N > 1 the ncclAllReduce of the shard state -- run on the library's stream beside the next main kernel
        pipe = NativeWelchPipeline(win, plan, scale=scale, sided=E.SIDED_TWO)
    else:                  # torch.distributed carries the state (gloo rehearsals; SP_BENCH_NATIVE_COMM=0)
        pipe = WelchPipeline(win, plan, scale=scale, sided=E.SIDED_TWO, force_collective=collective)

    def step():
        # the whole hot path: global-mean detrend (fft_analysis.py:2148) + window + overlapped FFT + |X|^2 + segment
        # average, in ONE pass over the samples; sharded: one all_reduce of the shard states, started asynchronously and
        # consumed one step later (pyfft_amd/dist.py: WelchPipeline) -- submit() returns the PREVIOUS step's PSD
        if pipe is None:
            return E.welch_psd(x, win, plan.hop, plan.frames, detrend=True, sided=E.SIDED_TWO, scale=scale)
        return pipe.submit(x)

    def drain():
        return pipe.flush() if pipe is not None else None

    def barrier():
        if world > 1:
            dist.barrier()

    E.profile_enable(False)          # the HIP-event hook is only switched on for the roofline measurement below
    # (the clocks need ~30 ms of work: small shards get proportionally more settle steps -- a count every rank agrees on)
    for _ in range(max(0, args.settle_steps) * max(1, (1 << 28) // max(1, plan.total_samples // plan.world))):
        step()
    drain()
    torch.cuda.synchronize()
    pxx = None
    for _ in range(args.warmup):
        pxx = step()
    drain()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pxx = step()
    last = drain()                   # the K-th step's all-reduce and finish kernel are inside the timed region
    t_enq = time.perf_counter() - t0
    if last is not None:
        pxx = last
    torch.cuda.synchronize()
    barrier()
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
    E.profile_enable(False)
    return {"elapsed": elapsed, "enqueue": t_enq, "kernel_ms": float(np.mean(kd)), "pxx": pxx,
            "kernel": E.profile_last_kernel()}


def tol_ratio(got, ref, rtol=2e-4, atol_rel=1e-6):
    """worst |got - ref| in units of the allowance rtol |ref| + atol_rel max(ref) (<= 1 passes)"""
    return float(np.max(np.abs(got - ref) / (rtol * np.abs(ref) + atol_rel * ref.max())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-steps", type=int, default=60,
                    help="untimed steps before the warm-up steps so that the GPU clocks have ramped: the first ~40 "
                         "steps after idle run 10-25 %% slower (tools/steptrace.py).  A fixed count, not a time, so "
                         "that every rank issues the same collectives.  0 = off")
    ap.add_argument("--log2n", type=int, default=28,
                    help="the stream has 2^log2n samples: split over the GPUs in the strong-scaling (headline) mode, per GPU "
                         "in the weak-scaling mode")
    ap.add_argument("--mode", choices=("auto", "strong", "weak", "both"), default="auto",
                    help="N > 1: 'strong' = ONE 2^log2n-sample stream split N ways (BASELINE.json's metric; the headline "
                         "value), 'weak' = 2^log2n samples per GPU, 'both' (= auto) = strong as the headline with the weak "
                         "figure as an extra key.  N = 1: the two coincide")
    ap.add_argument("--nfft", type=int, default=4096)
    ap.add_argument("--cpu-log2n", type=int, default=28, help="CPU-baseline sample = first 2^k samples (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the all-cores CPU baseline "
                                                             "(0 = the cores this process may run on, at most 32)")
    ap.add_argument("--gate-log2n", type=int, default=22,
                    help="sharded runs: the first 2^k samples of the stream go through the same sharded path and are compared "
                         "with the CPU oracle on host-generated samples (0 = skip)")
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
        # all-reduce, the shard plans of both modes, one JSON line from rank 0 -- no kernels, not a measurement
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        ps = shard_plan(1 << args.log2n, args.nfft, args.nfft // 2, world, rank)
        fr = torch.tensor([float(ps.frames)], dtype=torch.float64)
        dist.all_reduce(fr)
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "rank_sum": float(t.item()), "steps": args.steps,
                              "strong_frames_total": ps.frames_total, "strong_frames_sum": float(fr.item())}))
        dist.destroy_process_group()
        return
    dist = None
    # rehearsal on a one-GPU box: SP_BENCH_ONE_GPU=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the driver's real runs use one GPU per rank over RCCL.  SP_BENCH_FORCE_DIST=1 runs the
    # sharded code path (WelchPipeline: export, all-reduce over an RCCL group, apply) with a world of ONE rank
    one_gpu = os.environ.get("SP_BENCH_ONE_GPU", "") not in ("", "0")
    force_dist = os.environ.get("SP_BENCH_FORCE_DIST", "") not in ("", "0")
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    collective = world > 1 or force_dist
    # the library's own RCCL communicator (one ctypes call per step, no host work between the kernels and the collective)
    native = collective and not one_gpu and os.environ.get("SP_BENCH_NATIVE_COMM", "1") not in ("", "0")
    native_note = None
    if native:
        try:
            native_comm_init(device=local)
        except Exception as exc:          # (collective: every rank sees the same failure) -> torch.distributed carries the state
            native, native_note = False, "native RCCL communicator unavailable (%r): torch.distributed path" % (exc,)
    has_comm = native
    # one GPU, no collective: the same streaming engine without a communicator (SP_BENCH_STREAM=0: plain sp_welch_psd calls)
    if not collective and os.environ.get("SP_BENCH_STREAM", "1") not in ("", "0"):
        native = True

    nfft = args.nfft
    hop = nfft // 2
    S = 1 << args.log2n
    mode = args.mode
    if mode == "auto":
        mode = "both" if world > 1 else "strong"
    win = windows("Hanning", nwins=nfft, verbose=False)
    S2 = float(np.sum(win ** 2))
    Fs = 1.0
    scale = 1.0 / (Fs * S2)

    def run(total):
        plan = shard_plan(total, nfft, hop, world, rank)     # contiguous frame ranges + (nfft-hop)-sample halo
        x = synth.stream_torch(plan.first_sample, plan.nsamples, dev)
        m = measure(E, x, plan, win, scale, world, dist, dev, args, collective, native)
        m["plan"] = plan
        m["x"] = x
        return m

    runs = {}
    if mode in ("strong", "both"):
        runs["strong"] = run(S)
    if mode in ("weak", "both") and (world > 1 or mode == "weak"):
        if "strong" in runs:
            runs["strong"].pop("x")                          # free the strong shard before the weak one is made
        runs["weak"] = run(S * world)
    head = "strong" if "strong" in runs else "weak"
    m = runs[head]
    plan = m["plan"]
    total = plan.total_samples
    elapsed, k_ms = m["elapsed"], m["kernel_ms"]

    ms_per_step = 1e3 * elapsed / args.steps
    value = (total / 1e6) / (elapsed / args.steps)       # Msamples/s, whole job
    alg_bytes = 8.0 * plan.nsamples                      # 8 B per complex64 input sample, read once
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9           # GB/s

    if world == 1:
        workload = "Welch PSD, 2^%d complex64 samples, nfft=%d periodic Hann, hop=%d, global mean detrend, two-sided" \
                   % (args.log2n, nfft, hop)
    elif head == "strong":
        workload = ("Welch PSD of ONE 2^%d-sample complex64 stream split over %d GPUs (2^%d/%d = %d samples + a %d-sample halo "
                    "per GPU), nfft=%d periodic Hann, hop=%d, global mean detrend, two-sided"
                    % (args.log2n, world, args.log2n, world, S // world, nfft - hop, nfft, hop))
    else:
        workload = "Welch PSD, 2^%d complex64 samples per GPU (one stream of %d samples), nfft=%d periodic Hann, hop=%d, " \
                   "global mean detrend, two-sided" % (args.log2n, total, nfft, hop)
    result = {
        "metric": "Msamples/sec Welch-PSD 4096-pt Hann 50% overlap, complex64",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": head if world > 1 else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "total_samples": total, "samples_per_gpu": plan.nsamples,
                   "frames_per_gpu": plan.frames, "frames_total": plan.frames_total, "settle_steps": args.settle_steps,
                   "stream": "counter-based: splitmix64(seed ^ k) -> Box-Muller + two tones + offset (synth.py), identical "
                             "samples on every rank",
                   "parallelism": "segment-sharded x%d, one RCCL all-reduce of the shard state (%d doubles) per step, "
                                  "overlapped with the next step's kernels; %s" % (world, 5 * nfft + 8,
                                  "issued by libspectral on its own stream (sp_welch_dist_submit; communicator limited to "
                                  "%s workgroups, main kernels over all but %s CUs so that RCCL's kernel runs beside them)"
                                  % (os.environ.get("SP_DIST_RCCL_CTAS", "4"),
                                     os.environ.get("SP_DIST_RESERVE_CUS", "8" if world > 1 else "0")) if native else
                                  "issued through torch.distributed (WelchPipeline)")
                                  if collective else ("single GPU, no collective; steps streamed through sp_welch_dist_submit (the "
                                  "epilogue of step k runs beside the main kernel of step k+1; K submits + the flush are inside "
                                  "the timed region)" if native else "single GPU, no collective; one sp_welch_psd call per step")},
        "host_enqueue_ms_per_step": 1e3 * m["enqueue"] / args.steps,
        **({"note": native_note} if native_note else {}),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(args.log2n, nfft) if (world == 1 and not force_dist) else None,
                     "kernel": "%s<%d,complex64>" % (m["kernel"], nfft),
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "rank 0's launch over its own shard; VALU + LDS co-bound (ablation table, DESIGN.md); traffic = "
                             "rocprofv3 PMC bytes of this build (profiles/pmc_metric_kernel_current.json) or null"},
    }
    if "weak" in runs and head == "strong":
        w = runs["weak"]
        result["weak_scaling"] = {"value": (w["plan"].total_samples / 1e6) / (w["elapsed"] / args.steps),
                                  "unit": "Msamples/s", "ms_per_step": 1e3 * w["elapsed"] / args.steps,
                                  "samples_per_gpu": w["plan"].nsamples, "kernel_ms": w["kernel_ms"],
                                  "note": "2^%d samples per GPU, same K steps" % args.log2n}

    # ---- parity gates of the sharded path (outside the timed region) ---------------------------------------------------
    if collective:
        from pyfft_amd.dist import welch_psd_sharded
        gates = {}
        if args.gate_log2n > 0:
            # (a) the first 2^k samples of the stream through the SAME sharded path (shard plan over `world` ranks, state
            # all-reduce, apply) against the CPU oracle on host-generated samples of the same formula
            ng = 1 << args.gate_log2n
            pg = shard_plan(ng, nfft, hop, world, rank)
            xg = synth.stream_torch(pg.first_sample, pg.nsamples, dev)
            got = welch_psd_sharded(xg, win, pg, scale, sided=E.SIDED_TWO, force_collective=force_dist).cpu().numpy()
            if rank == 0:
                from oracle import cpu_ref as O
                ref = O.welch_psd_stream(synth.stream_numpy(0, ng), win, nfft, hop, pg.frames_total, Fs)
                r = tol_ratio(got, ref)
                gates["prefix_vs_oracle"] = {"samples": ng, "vs": "oracle.welch_psd_stream on host-generated samples",
                                             "rtol": 2e-4, "atol_rel_max": 1e-6, "worst_over_tolerance": r, "ok": bool(r <= 1.0)}
        # (b) the timed result itself: the PSD of the whole stream from N shards against ONE GPU's PSD of the same stream
        # (rank 0 generates all of it; the single-GPU path is oracle-gated by the N = 1 run and by tests/)
        if rank == 0:
            xa = synth.stream_torch(0, total, dev)
            one = E.welch_psd(xa, win, hop, plan.frames_total, detrend=True, sided=E.SIDED_TWO, scale=scale).cpu().numpy()
            del xa
            r = tol_ratio(m["pxx"].cpu().numpy(), one)
            gates["sharded_vs_single_gpu"] = {"samples": total, "rtol": 2e-4, "atol_rel_max": 1e-6,
                                              "worst_over_tolerance": r, "ok": bool(r <= 1.0)}
            result["parity"] = gates
            if not all(g["ok"] for g in gates.values()):
                result["value"] = 0.0
                result["error"] = "parity gate failed"

    if rank == 0 and world == 1 and not force_dist and args.cpu_log2n > 0:
        # CPU baseline: the oracle's streaming restatement of the same path ("port": numpy float64 pocketfft, identical
        # arithmetic to the reference's fft_win -> Pstft -> averagewins) on the first 2^k samples of the same stream,
        # (i) on ONE core -- the reference's own CPU path is single-threaded -- and (ii) on every core this process may
        # use (frame ranges dealt to worker processes; oracle/cpu_parallel.py); the one-core run doubles as the in-run
        # parity gate.
        from oracle import cpu_ref as O
        from oracle import cpu_parallel
        x = m["x"]
        nc = min(S, 1 << args.cpu_log2n)
        Mc = (nc - nfft) // hop + 1
        xc = x[:nc].cpu().numpy()
        gen = float(np.max(np.abs(xc[:1 << 16] - synth.stream_numpy(0, 1 << 16))))     # device and host generator agree
        t1 = time.perf_counter()
        ref = O.welch_psd_stream(xc, win, nfft, hop, Mc, Fs)
        tc = time.perf_counter() - t1
        got = E.welch_psd(x[:nc], win, hop, Mc, detrend=True, sided=E.SIDED_TWO, scale=scale).cpu().numpy()
        err = tol_ratio(got, ref)
        if nc == S:           # the timed loop's own last result (streamed steps) against the same oracle
            err = max(err, tol_ratio(m["pxx"].cpu().numpy(), ref))
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
                            "worst_over_tolerance": err, "ok": bool(err <= 1.0),
                            "device_vs_host_generator_max_abs": gen}
        if err > 1.0 or gen > 1e-5:
            result["value"] = 0.0
            result["error"] = "parity gate failed"

    if rank == 0:
        print(json.dumps(result))
    if has_comm:
        torch.cuda.synchronize()
        E.comm_destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
