"""TEST / MEASUREMENT INFRASTRUCTURE (like the rest of oracle/): the CPU baseline of bench.py on all host cores.

The reference's Welch path (fft_analysis.py:2126-2203 fft_win -> :1946 Pstft -> :1980 averagewins) is single-threaded
numpy; SURVEY 8(d)(ii) asks for the same arithmetic spread over the host's cores as a second baseline: contiguous frame
ranges are dealt to worker processes, each runs the streaming restatement of `cpu_ref.welch_psd_stream` on its range
against the GLOBAL mean, and the partial |X|^2 sums are added.  Workers are plain child processes of this file
(`python cpu_parallel.py --worker ...`: numpy only -- never a fork of the parent, which has initialised HIP, and no
re-import of the parent's main module); samples, window and partial sums travel through files in /dev/shm."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np


def usable_cores(requested=0):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    if requested and requested > 0:
        return max(1, min(requested, n))
    return max(1, min(n, 32))


def _partial(x, win, nfft, hop, g0, g1, mean):
    acc = np.zeros(nfft, dtype=np.float64)
    idx = np.arange(nfft)[None, :]
    for a in range(g0, g1, 2048):
        b = min(g1, a + 2048)
        st = (np.arange(a, b) * hop)[:, None]
        seg = win[None, :] * (x[st + idx] - mean)            # input-dtype subtraction (Q6), float64 window
        X = np.fft.fft(seg, axis=-1)
        acc += (X.real ** 2 + X.imag ** 2).sum(axis=0)
    return acc


def _worker_main(argv):
    job = json.loads(argv[0])
    x = np.load(job["x"], mmap_mode="r")
    win = np.load(job["win"])
    mean = np.asarray(complex(job["mean_re"], job["mean_im"]) if job["cplx"] else job["mean_re"]).astype(x.dtype)
    acc = _partial(x, win, job["nfft"], job["hop"], job["g0"], job["g1"], mean)
    np.save(job["out"], acc)


def welch_psd_parallel(x, win, nfft, hop, nframes, Fs, cores):
    """two-sided shifted PSD like cpu_ref.welch_psd_stream (global-mean detrend); returns (psd, seconds).  The clock
    covers the workers' whole life (interpreter start-up included) but not the copy of the samples into shared memory,
    which the one-core baseline does not pay either."""
    x = np.asarray(x)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="spwelch_", dir=shm)
    try:
        xp, wp = os.path.join(d, "x.npy"), os.path.join(d, "win.npy")
        np.save(xp, x)
        np.save(wp, np.asarray(win, dtype=np.float64))
        mean = x.mean().astype(x.dtype)                      # the reference subtracts the mean in the input dtype (Q6)
        bounds = np.linspace(0, nframes, cores + 1).astype(np.int64)
        jobs = []
        for i in range(cores):
            if bounds[i + 1] > bounds[i]:
                jobs.append({"x": xp, "win": wp, "nfft": int(nfft), "hop": int(hop), "g0": int(bounds[i]),
                             "g1": int(bounds[i + 1]), "cplx": bool(np.iscomplexobj(x)), "mean_re": float(np.real(mean)),
                             "mean_im": float(np.imag(mean)), "out": os.path.join(d, "p%d.npy" % i)})
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", json.dumps(j)], env=env) for j in jobs]
        rcs = [p.wait() for p in procs]
        dt = time.perf_counter() - t0
        if any(rcs):
            raise RuntimeError("cpu_parallel worker failed: %r" % (rcs,))
        acc = np.sum([np.load(j["out"]) for j in jobs], axis=0)
    finally:
        for name in os.listdir(d):
            try:
                os.unlink(os.path.join(d, name))
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass
    S2 = np.sum(np.asarray(win, dtype=np.float64) ** 2.0)
    return np.fft.fftshift(acc) / (nframes * Fs * S2), dt


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--worker":
        _worker_main(sys.argv[2:])
