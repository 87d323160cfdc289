#!/usr/bin/env python3
"""Where the host's time per streamed step goes (sp_welch_dist_submit through pyfft_amd.engine): Python wrapper pieces and the C call,
perf_counter around each, GPU idle-free (a 2^25-sample shard, 2000 submits)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import engine as E, _ffi
from pyfft_amd.dist import shard_plan, NativeWelchPipeline
from pyfft_amd.windows import windows

n, nfft, hop = 1 << 25, 4096, 2048
x = torch.view_as_complex(torch.randn((n, 2), device="cuda", dtype=torch.float32))
win = windows("Hanning", nwins=nfft, verbose=False)
plan = shard_plan(n, nfft, hop, 1, 0)
pipe = NativeWelchPipeline(win, plan, scale=1.0, sided=E.SIDED_TWO)
for _ in range(50):
    pipe.submit(x)
pipe.flush(); torch.cuda.synchronize()
K = 2000
t0 = time.perf_counter()
for _ in range(K):
    pipe.submit(x)
t1 = time.perf_counter()
pipe.flush(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("submit loop: %.1f us per step on the host; with the drain %.1f us per step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
# pieces of the wrapper
w = E._win32(win)
def piece(name, fn, k=K):
    t = time.perf_counter()
    for _ in range(k):
        fn()
    print("  %-40s %6.2f us" % (name, (time.perf_counter() - t) / k * 1e6))
piece("_win32(win)", lambda: E._win32(win))
piece("_bind_stream(x)", lambda: E._bind_stream(x))
piece("_torch_samples(x)", lambda: E._torch_samples(x))
piece("torch.empty(out)", lambda: torch.empty(4096, dtype=torch.float64, device=x.device))
xs = E._torch_samples(x)
out = torch.empty(4096, dtype=torch.float64, device=x.device)
nd = _ffi.C.c_int(0)
L = _ffi.lib()
def ccall():
    L.sp_welch_dist_submit(_ffi.ptr(xs.data_ptr()), E._tcode(xs), xs.numel(), _ffi.ptr(w), w.size, hop, plan.frames, plan.own_samples, plan.frames_total, E.SIDED_TWO, 1.0, _ffi.ptr(out.data_ptr()), _ffi.C.byref(nd))
piece("the C call alone (ctypes)", ccall)
E.welch_dist_flush(); torch.cuda.synchronize()
