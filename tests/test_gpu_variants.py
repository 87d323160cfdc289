"""The packed-fp32 build of the FFT core (SP_PACKED=1, pyfft_amd/lib/libspectral_packed.so) gives the same results as
the default library: run in a child process because a process binds one library."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from pyfft_amd import engine as E
from oracle import cpu_ref as O
rng = np.random.default_rng(3)
for n in (256, 1024, 4096, 8192):
    x = (rng.standard_normal((5, n)) + 1j * rng.standard_normal((5, n))).astype(np.complex64)
    X = E.fft(x)
    ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    assert np.max(np.abs(X - ref)) <= 3e-6 * np.abs(ref).max(), n
    assert np.max(np.abs(E.ifft(X) - x)) <= 2e-6 * np.abs(x).max(), n
nfft, hop = 4096, 2048
s = (rng.standard_normal(nfft + hop * 50) + 1j * rng.standard_normal(nfft + hop * 50) + 0.3).astype(np.complex64)
win = O.windows("Hanning", nwins=nfft)
M = 51
p = E.welch_psd(s, win, hop, M, detrend=True, sided=E.SIDED_TWO, scale=1.0)
ref = O.welch_psd_stream(s, win, nfft, hop, M, 1.0) * np.sum(win ** 2)
assert np.max(np.abs(p - ref)) <= 2e-4 * ref.max()
print("variant ok", E.lib_path() if hasattr(E, "lib_path") else "")
""" % ROOT


def test_packed_core_library_matches():
    lib = os.path.join(ROOT, "pyfft_amd", "lib", "libspectral_packed.so")
    if not os.path.exists(lib):
        pytest.fail("libspectral_packed.so is missing: run `make` (or __graft_entry__.build())")
    env = dict(os.environ, SP_LIB_PATH=lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "variant ok" in r.stdout
