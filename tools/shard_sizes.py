#!/usr/bin/env python3
"""Step time of the headline path as a function of the shard size, on ONE GPU: bench.py at --log2n 25..28,
(a) the single-GPU path (E.welch_psd: k_welch_pipe -> finish) and (b) the sharded path with an RCCL group of one rank
(SP_BENCH_FORCE_DIST=1: export kernels -> all-reduce -> apply, pipelined), which is what a rank of an N-GPU run executes on a
2^28/N-sample shard.  Fits t_step(n) = t_fixed + n / rate and prints the predicted strong-scaling curve
t_step(2^28 / N), N = 1, 2, 4, 8 (the all-reduce of 160 KiB is inside (b)'s step at world 1 already; its xGMI latency at
N > 1 is overlapped with the next step by WelchPipeline and is NOT in this prediction).
    python tools/shard_sizes.py [--steps 200] > profiles/r03_shard_sizes.txt"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(log2n, steps, force_dist, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    if force_dist:
        env["SP_BENCH_FORCE_DIST"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "20",
           "--log2n", str(log2n), "--cpu-log2n", "0", "--gate-log2n", "0" if not force_dist else "20"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit("bench failed: " + r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--sizes", type=int, nargs="+", default=[28, 27, 26, 25, 24])
    ap.add_argument("--env", nargs="*", default=[], help="NAME=VALUE pairs passed to bench.py (A/B of a build knob)")
    args = ap.parse_args()
    extra0 = dict(kv.split("=", 1) for kv in args.env)
    for label, fd, env2 in (("single GPU, one sp_welch_psd call per step (SP_BENCH_STREAM=0)", False, {"SP_BENCH_STREAM": "0"}),
                            ("single GPU, steps streamed through sp_welch_dist_submit (epilogue beside the next main kernel)", False, {}),
                            ("sharded path, RCCL group of one rank (export -> all-reduce -> apply, streamed)", True, {})):
        extra = dict(extra0, **env2)
        print("== %s%s" % (label, ("  [" + " ".join(args.env) + "]") if args.env else ""))
        print("%6s %12s %12s %12s %12s %10s" % ("log2n", "ms/step", "kernel ms", "host ms", "Msamples/s", "roof frac"))
        ns, ts, ks = [], [], []
        for k in args.sizes:
            d = bench(k, args.steps, fd, extra)
            ns.append(2.0 ** k)
            ts.append(d["ms_per_step"])
            ks.append(d["roofline"]["kernel_ms"])
            print("%6d %12.4f %12.4f %12.4f %12.0f %10.3f" % (k, d["ms_per_step"], d["roofline"]["kernel_ms"],
                                                            d["host_enqueue_ms_per_step"], d["value"], d["roofline"]["frac"]))
            sys.stdout.flush()
        A = np.vstack([np.ones(len(ns)), np.array(ns)]).T
        (t0, sl), *_ = np.linalg.lstsq(A, np.array(ts), rcond=None)
        (k0, ksl), *_ = np.linalg.lstsq(A, np.array(ks), rcond=None)
        print("fit: t_step = %.4f ms + n / (%.0f Msamples/s);  kernel = %.4f ms + n / (%.0f Msamples/s)"
              % (t0, 1e-3 / sl, k0, 1e-3 / ksl))
        t1 = t0 + sl * 2.0 ** 28
        print("predicted strong scaling of the 2^28-sample stream (step = fit at 2^28 / N):")
        for n in (1, 2, 4, 8):
            tn = t0 + sl * 2.0 ** 28 / n
            print("   N = %d: %.4f ms/step  %9.0f Msamples/s  efficiency %.2f" % (n, tn, 2.0 ** 28 / 1e3 / tn, t1 / (n * tn)))
        print()


if __name__ == "__main__":
    main()
