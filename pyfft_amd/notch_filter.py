"""Second-order IIR notch / peak design (reference: notch_filter.py:19-241, Orfanidis 11.3.x) and its
application.  The design is ten host flops and returns (b, a) like the reference; the reference never applies
the filter -- `apply_notch` does: by default the biquad recurrence itself, evaluated exactly on the GPU (sp_biquad:
float64 state, blocked scan), or on request as a truncated-impulse-response FIR through the overlap-save kernel, with
the truncation error bounded from the pole radius."""
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


def pole_radius(a):
    """largest pole magnitude of 1/A(z), a = [a0, a1, a2]"""
    r = np.roots(np.asarray(a, dtype=np.float64))
    return float(np.max(np.abs(r))) if r.size else 0.0


def fir_taps_for(a, tol=1e-6, limit=4096):
    """taps needed so that the dropped tail of the biquad's impulse response is below `tol` of its start: the tail decays
    like |p|^n, so n >= log(tol) / log|p|.  Raises when that exceeds `limit` (the overlap-save kernel takes
    2 (ntaps - 1) <= 8192): such a notch is what the exact recurrence (apply_notch's default) is for."""
    r = pole_radius(a)
    if r >= 1.0:
        raise ValueError("unstable section (pole radius %.6f): no FIR truncation exists" % r)
    n = 3 if r == 0.0 else int(np.ceil(np.log(tol) / np.log(r))) + 2
    if n > limit:
        raise ValueError("the impulse response decays like %.6f^n: %d taps are needed for a tail below %g, more than the "
                         "FIR kernel's %d -- use the exact recurrence (ntaps=None)" % (r, n, tol, limit))
    return max(n, 3)


def apply_notch(x, w0, Q, ntaps=None, ftype="notch", nfft=0, tol=1e-6):
    """Filter x with the designed notch/peak biquad: y = lfilter(b, a, x) (float32 samples).
    ntaps=None (default): the recurrence itself, exact, on the GPU (engine.biquad_filter).
    ntaps='auto' or an integer: the biquad as an ntaps-tap FIR through the overlap-save kernel; 'auto' sizes the taps so
    that the dropped tail of the impulse response (|pole|^ntaps) stays below `tol`, an explicit count whose tail exceeds
    `tol` raises ValueError instead of returning a silently different filter (w0=0.01, Q=30: |p|^513 = 0.77)."""
    b, a = _design_notch_peak_filter(w0, Q, ftype)
    if ntaps is None:
        return _E.biquad_filter(b, a, x)
    if ntaps == "auto":
        ntaps = fir_taps_for(a, tol)
    else:
        ntaps = int(ntaps)
        tail = pole_radius(a) ** ntaps
        if tail > tol:
            raise ValueError("a %d-tap FIR drops an impulse-response tail of relative size %.3g (pole radius %.6f) > tol=%g; "
                             "use ntaps=None (exact recurrence), ntaps='auto', or pass a larger tol on purpose"
                             % (ntaps, tail, pole_radius(a), tol))
    return _E.fir_filter(impulse_response(b, a, ntaps), x, nfft=nfft)
