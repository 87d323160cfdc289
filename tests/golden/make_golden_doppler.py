#!/usr/bin/env python3
"""Golden vectors for the reference's centre-of-gravity function Doppler.cog (Doppler.py:43-58).

TEST INFRASTRUCTURE, build container only (needs /root/reference and matplotlib).  Doppler.py imports `FFT.stft`, a
module the reference does not contain (its stft lives in spectrogram.py); the import is satisfied with an alias module
holding the reference's own spectrogram.stft.  cog itself is called unmodified.  The per-window vector reproduces the
loop body of cogspec (Doppler.py:73-81) by calling the reference's cog on each complete window.
Usage:  python tests/golden/make_golden_doppler.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
from make_golden import _install_shims, _load, gauss, save


def main():
    _install_shims()
    _load("windows")
    _load("fft_analysis")
    spg = _load("spectrogram")
    _load("filters")
    alias = types.ModuleType("FFT.stft")
    alias.stft = spg.stft
    sys.modules["FFT.stft"] = alias
    D = _load("Doppler")

    fs = 1.0e6
    n = 40000
    t = np.arange(n) / fs
    # IQ signal whose Doppler line sweeps 80 -> 180 kHz, plus a weak image line and noise
    f_inst = 80e3 + 100e3 * t / t[-1]
    ph = 2 * np.pi * np.cumsum(f_inst) / fs
    z = np.exp(1j * ph) + 0.2 * np.exp(-2j * np.pi * 50e3 * t) + 0.15 * (gauss(21, n) + 1j * gauss(22, n))
    z = z.astype(np.complex64)
    r = (np.cos(ph) + 0.3 * gauss(23, n)).astype(np.float32)
    d = {"fs": np.array(fs), "z": z, "r": r}          # t = arange(n)/fs is rebuilt by the tests
    # whole-vector cog: lengths that fit one workgroup transform (pow2 / not) and ones that do not
    for tag, m in (("4096", 4096), ("1000", 1000), ("40000", 40000), ("16384", 16384)):
        d["cog_z_" + tag] = np.array(D.cog(z[:m], fs))
        d["cog_r_" + tag] = np.array(D.cog(r[:m], fs))
    # the band-limited form (reference pairing, see oracle/cpu_ref.py cog)
    d["cog_z_band"] = np.array(D.cog(z[:4096], fs, fmin=50e3, fmax=200e3))
    d["cog_z_band_nofmax"] = np.array(D.cog(z[:1000], fs, fmin=100e3))
    d["cog_z_band_empty"] = np.array(D.cog(z[:1000], fs, fmin=2e6, fmax=3e6))
    # per-window: win 512 / ov 0.5 (cogspec defaults) and win 200 / ov 0.75
    for tag, win, ov in (("512", 512, 0.5), ("200", 200, 0.75)):
        hop = int(np.floor((1.0 - ov) * win))
        nfr = (n - win) // hop + 1
        d["frames_z_" + tag] = np.array([D.cog(z[g * hop:g * hop + win], fs) for g in range(nfr)])
        d["frames_r_" + tag] = np.array([D.cog(r[g * hop:g * hop + win], fs) for g in range(nfr)])
        d["frames_t_" + tag] = np.array([np.mean(t[g * hop:g * hop + win]) for g in range(nfr)])
    save("doppler_cog", **d)


if __name__ == "__main__":
    main()
