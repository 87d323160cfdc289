"""FFT filtering.  The reference's filters.py has no FFT/overlap-add filter (only scipy IIR wrappers and a
np.convolve smoother, filters.py:226-358); `fftfilt` is the build-defined hot function for that slot:
causal FIR y = lfilter(b, 1, x) by overlap-save on the MI355X."""
import numpy as np

from . import engine as _E
from .notch_filter import apply_notch  # noqa: F401  (notch application lives with the design)


def fftfilt(b, x, nfft=None):
    """y[n] = sum_k b[k] x[n-k], same length as x (float32 math on the GPU; float64 returned for float64 x)."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("fftfilt: 1-D signal expected")
    y = _E.fir_filter(np.asarray(b, dtype=np.float64), x, nfft=0 if nfft is None else int(nfft))
    return y.astype(np.float64) if x.dtype != np.float32 else y


def smooth(x, window_len=11, window="hanning"):
    """Window-FIR smoother of the reference (filters.py:226-283: reflect-pad, convolve with w/sum(w), 'valid')
    with the convolution done by the GPU FIR kernel."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 1:
        raise ValueError("smooth only accepts 1 dimension arrays.")
    if x.size < window_len:
        raise ValueError("Input vector needs to be bigger than window size.")
    if window_len < 3:
        return x
    if window not in ("flat", "hanning", "hamming", "bartlett", "blackman"):
        raise ValueError("Window is on of 'flat', 'hanning', 'hamming', 'bartlett', 'blackman'")
    s = np.r_[x[window_len - 1:0:-1], x, x[-2:-window_len - 1:-1]]
    w = np.ones(window_len) if window == "flat" else getattr(np, window)(window_len)
    full = _E.fir_filter(w / w.sum(), s).astype(np.float64)
    return full[window_len - 1:]          # == np.convolve(w/w.sum(), s, mode='valid')
