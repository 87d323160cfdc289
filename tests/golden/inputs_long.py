"""Seeded inputs of the long-segment fixtures (make_golden_long.py stores outputs only; the tests rebuild the inputs
with these functions).  Pure numpy, no reference code."""
import numpy as np


def long_signals(N=2 ** 19, df=5.0, seed=20260401):
    """the signals of the reference's test_fftanal (fft_analysis.py:2955-2975), with seeded instead of clock-seeded noise"""
    tvec = (1.0 / df) * np.arange(0.0, 1.0, 1.0 / N)
    rng = np.random.default_rng(seed)
    sigx = 0.005 * np.sin(2.0 * np.pi * (df * 30.0) * tvec) + 7.0 + 0.02 * rng.standard_normal(tvec.shape[0])
    sigy = 0.005 * np.sin(2.0 * np.pi * (df * 30.0) * tvec - np.pi / 4.0) + 0.02 * rng.standard_normal(tvec.shape[0]) + 2.5
    return tvec, sigx, sigy


def stft_long_signal(n=2 ** 17, seed=77):
    """chirp 0.01 -> 0.05 cycles/sample + noise, unit sample spacing"""
    rng = np.random.default_rng(seed)
    k = np.arange(n, dtype=np.float64)
    xs = np.sin(2 * np.pi * np.cumsum(np.linspace(0.01, 0.05, n))) + 0.05 * rng.standard_normal(n)
    return k, xs


def subsample_index(n):
    """bins kept in the fixtures: the first 1024, every 31st after that, the last 64"""
    idx = np.concatenate([np.arange(min(n, 1024)), np.arange(1024, n, 31), np.arange(max(n - 64, 0), n)])
    return np.unique(idx)
