"""Second-order IIR notch / peak design (reference: notch_filter.py:19-241, Orfanidis 11.3.x) and its
application.  The design is ten host flops and returns (b, a) like the reference; the reference never applies
the filter -- `apply_notch` does, as a truncated-impulse-response FIR through the GPU overlap-save kernel."""
import numpy as np

from . import engine as _E


def _design_notch_peak_filter(w0, Q, ftype):
    w0 = float(w0)
    Q = float(Q)
    if w0 > 1.0 or w0 < 0.0:
        raise ValueError("w0 should be such that 0 < w0 < 1")
    bw = np.pi * w0 / Q
    wc = np.pi * w0
    gb = 1 / np.sqrt(2)
    if ftype == "notch":
        beta = (np.sqrt(1.0 - gb ** 2.0) / gb) * np.tan(bw / 2.0)
    elif ftype == "peak":
        beta = (gb / np.sqrt(1.0 - gb ** 2.0)) * np.tan(bw / 2.0)
    else:
        raise ValueError("Unknown ftype.")
    gain = 1.0 / (1.0 + beta)
    if ftype == "notch":
        b = gain * np.array([1.0, -2.0 * np.cos(wc), 1.0])
    else:
        b = (1.0 - gain) * np.array([1.0, 0.0, -1.0])
    a = np.array([1.0, -2.0 * gain * np.cos(wc), (2.0 * gain - 1.0)])
    return b, a


def iirnotch(w0, Q):
    return _design_notch_peak_filter(w0, Q, "notch")


def iirpeak(w0, Q):
    return _design_notch_peak_filter(w0, Q, "peak")


def impulse_response(b, a, ntaps):
    """First ntaps samples of the impulse response of the biquad b/a (host recursion, float64)."""
    h = np.zeros(int(ntaps))
    for n in range(int(ntaps)):
        acc = b[n] if n < len(b) else 0.0
        for k in range(1, len(a)):
            if n - k >= 0:
                acc -= a[k] * h[n - k]
        h[n] = acc / a[0]
    return h


def apply_notch(x, w0, Q, ntaps=513, ftype="notch", nfft=0):
    """Filter x with the designed notch/peak biquad realised as an ntaps FIR (GPU overlap-save).  The truncation
    error against the exact recursion is bounded by the tail of the impulse response (|pole|^ntaps)."""
    b, a = _design_notch_peak_filter(w0, Q, ftype)
    return _E.fir_filter(impulse_response(b, a, ntaps), x, nfft=nfft)
