"""Counter-based synthetic stream of the headline benchmark: measurement / test infrastructure, not
part of the product.

    z[n] = (g(2n) + j g(2n+1)) / sqrt(2) + 2.82842712 e^{j 2 pi 0.1234 n} + 1.0 e^{j 2 pi 0.25002157 n} + (0.05 - 0.02j)
    g(k) = Box-Muller on the two 32-bit halves of splitmix64(seed XOR k),  seed = 0x5EED2024,  cast to complex64

Every sample is a pure function of its index n, so any rank can produce any range [n0, n0 + n) of ONE stream: the halos of
neighbouring shards are the same samples, and a result of N ranks can be compared with a single-rank result or with the CPU
oracle on host-generated samples.  (The constant offset gives the global-mean detrend something to remove; the two tones are
Heinzel section 13's at fs = 1.)  Two implementations of the same formula: numpy on the host (`stream_numpy`) and torch on
a device (`stream_torch`, int64 arithmetic with wrap-around = uint64 arithmetic, logical shifts emulated by masks); both
evaluate Box-Muller and the tone phases in float64, so they agree to the last float32 bit or two of the cast.
"""
import numpy as np

SEED = 0x5EED2024
TONES = ((2.82842712, 0.1234), (1.0, 0.25002157))
OFFSET = 0.05 - 0.02j

_M64 = (1 << 64) - 1
_C0, _C1, _C2 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _splitmix64_numpy(x):
    """x: uint64 array -> uint64 array (one splitmix64 output per counter)"""
    with np.errstate(over="ignore"):
        z = x + np.uint64(_C0)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        return z ^ (z >> np.uint64(31))


def gauss_numpy(k, seed=SEED):
    """g(k) for an integer array k (float64)"""
    z = _splitmix64_numpy(np.uint64(seed) ^ np.asarray(k).astype(np.uint64))
    hi = (z >> np.uint64(32)).astype(np.float64)
    lo = (z & np.uint64(0xFFFFFFFF)).astype(np.float64)
    u1 = (hi + 1.0) * (1.0 / 4294967296.0)              # (0, 1]
    u2 = lo * (1.0 / 4294967296.0)                      # [0, 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def stream_numpy(n0, n, seed=SEED):
    """complex64 samples [n0, n0 + n) of the stream, on the host"""
    out = np.empty(n, dtype=np.complex64)
    chunk = 1 << 20
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        idx = np.arange(n0 + a, n0 + b, dtype=np.int64)
        z = (gauss_numpy(2 * idx, seed) + 1j * gauss_numpy(2 * idx + 1, seed)) * np.sqrt(0.5)
        kf = idx.astype(np.float64)
        for amp, f in TONES:
            z = z + amp * np.exp(2j * np.pi * np.remainder(f * kf, 1.0))
        out[a:b] = (z + OFFSET).astype(np.complex64)
    return out


def _s64(v):
    """a 64-bit constant as the signed python int torch's int64 arithmetic wants"""
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(z, s):
    """logical shift right of an int64 tensor"""
    return (z >> s) & ((1 << (64 - s)) - 1)


def gauss_torch(k, seed=SEED):
    """g(k) for an int64 tensor k (float64 tensor on the same device)"""
    import torch
    z = (k ^ _s64(seed)) + _s64(_C0)
    z = (z ^ _lsr(z, 30)) * _s64(_C1)
    z = (z ^ _lsr(z, 27)) * _s64(_C2)
    z = z ^ _lsr(z, 31)
    hi = _lsr(z, 32).to(torch.float64)
    lo = (z & 0xFFFFFFFF).to(torch.float64)
    u1 = (hi + 1.0) * (1.0 / 4294967296.0)
    u2 = lo * (1.0 / 4294967296.0)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)


def stream_torch(n0, n, device, seed=SEED):
    """complex64 samples [n0, n0 + n) of the stream as a tensor on `device`"""
    import torch
    out = torch.empty(n, dtype=torch.complex64, device=device)
    chunk = 1 << 23
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        idx = torch.arange(n0 + a, n0 + b, dtype=torch.int64, device=device)
        re = gauss_torch(2 * idx, seed)
        im = gauss_torch(2 * idx + 1, seed)
        z = torch.complex(re, im) * (0.5 ** 0.5)
        del re, im
        kf = idx.to(torch.float64)
        for amp, f in TONES:
            z = z + amp * torch.exp(2j * np.pi * torch.remainder(f * kf, 1.0))
        out[a:b] = (z + OFFSET).to(torch.complex64)
        del z, kf, idx
    return out
