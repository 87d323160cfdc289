"""Input recipe of tests/golden/chloop.npz, shared by the generator (make_golden_chloop.py, build container) and the tests
(GPU box): the arrays are rebuilt from the seed, only the reference's OUTPUTS are stored."""
import numpy as np

SEED, NT, FS, NCH, FMOD = 41, 1 << 15, 1.0e4, 16, 37.0


def chloop_inputs(seed=SEED, nt=NT, fs=FS, nch=NCH, fmod=FMOD):
    """tt, ref (a modulated heating power: thresholded sine + noise), sig[nt, nch] (delayed, attenuated responses at the
    first three harmonics + noise + offsets)"""
    rng = np.random.default_rng(seed)
    tt = np.arange(nt) / fs
    ref = (np.sin(2 * np.pi * fmod * tt) > 0.2).astype(np.float64) + 0.02 * rng.standard_normal(nt)
    sig = np.empty((nt, nch))
    for ch in range(nch):
        delay = 0.4e-3 * (ch + 1)
        resp = np.zeros(nt)
        for h in (1, 2, 3):
            resp += (0.8 ** ch) / h ** 1.5 * np.sin(2 * np.pi * h * fmod * (tt - delay * np.sqrt(h)) - 0.3 * h)
        sig[:, ch] = resp + (0.3 + 0.05 * ch) * rng.standard_normal(nt) + 1.0 + 0.1 * ch
    return tt, ref, sig
