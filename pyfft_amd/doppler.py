"""Centre-of-gravity (power-weighted mean frequency) of Doppler spectra: drop-in for `Doppler.cog` (Doppler.py:43-58) and
the per-window loop of `Doppler.cogspec` (Doppler.py:61-81), on the device.

`cog_frames` is where the work is: one fused kernel transforms every frame and reduces sum f|X|^2 and sum |X|^2 in
registers, so no spectrogram is written to HBM (sp_stft_cog, include/spectral.h).
"""
import numpy as np

from . import engine as _E


def _device_frame_ok(n):
    """Does a length-n transform fit one workgroup (powers of two up to max_wg_fft, other lengths up to half of it)?"""
    mx = _E.max_wg_fft()
    return n >= 2 and (n <= mx if (n & (n - 1)) == 0 else n <= mx // 2)


def cog(x, fs, fmin=None, fmax=None):
    """Centre of gravity of the two-sided power spectrum of x: sum(|X|^2 f) / sum(|X|^2), f = fftfreq(n, 1/fs).

    fmin=None (the reference's default and the only form its callers use): every bin counts, fmax is ignored.
    fmin given: the reference selects the band fmin <= |f| <= fmax on the frequency axis and then indexes the spectrum
    with a mask computed from the ALREADY selected frequencies (Doppler.py:53-54), i.e. it pairs the band's frequencies
    with the first len(band) bins of the shifted spectrum.  That behaviour is kept bit for bit in meaning (same bins,
    same pairing) so results match the reference; use `cog_frames(..., fmin=, fmax=)` for a true band limit."""
    x = np.asarray(x)
    n = len(x)
    if fmax is None:
        fmax = fs
    if fmin is None and _device_frame_ok(n):
        return float(_E.stft_cog(x, np.ones(n), n, 1, fs)[0])
    spec = np.fft.fftshift(_E.fft(x)).astype(np.complex128) / np.sqrt(n / 2)
    freq = np.fft.fftshift(np.fft.fftfreq(n, 1 / fs))
    if fmin is not None:
        freq = freq[(np.abs(freq) >= fmin) & (np.abs(freq) <= fmax)]
        spec = spec[:len(freq)]
    if len(freq) > 0:
        p = np.abs(np.square(spec))
        return float(np.sum(p * freq) / np.sum(p))
    return 0.0


def cog_frames(t, x, fs, win=512, ov=0.5, fmin=None, fmax=None, window=None, detrend=False):
    """(tcog, coge): centre of gravity of every length-`win` window of x, hop = floor((1-ov) win) — the loop of
    `cogspec` (Doppler.py:61-81: `coge[ii] = cog(x[i0:i1], fs)`, `tcog[ii] = mean(t[i0:i1])`) over the complete windows.
    The reference takes its window start/stop indices from pybaseutils.utils.sliding_window_1d, which is absent; complete
    windows at that hop are this build's stated choice.

    window: optional taper table of length `win` (default boxcar = the reference); fmin/fmax: a true band limit on |f|."""
    x = np.asarray(x) if not _E._is_torch(x) else x
    nsig = int(x.shape[0])
    win = int(win)
    hop = int(np.floor((1.0 - ov) * win))
    if hop < 1 or win < 2 or nsig < win:
        raise ValueError("cog_frames: need 2 <= win <= len(x) and an overlap below 1")
    if not _device_frame_ok(win):
        raise ValueError("cog_frames: window length %d exceeds the one-workgroup transform" % win)
    nframes = (nsig - win) // hop + 1
    w = np.ones(win) if window is None else np.asarray(window, dtype=np.float64)
    if w.shape != (win,):
        raise ValueError("cog_frames: window table must have length win")
    coge = _E.stft_cog(x, w, hop, nframes, fs, fmin=0.0 if fmin is None else fmin, fmax=fmax, detrend=detrend)
    t = np.asarray(t, dtype=np.float64)
    cs = np.concatenate(([0.0], np.cumsum(t)))
    i0 = np.arange(nframes) * hop
    tcog = (cs[i0 + win] - cs[i0]) / win
    return tcog, coge
