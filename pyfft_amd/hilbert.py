"""Drop-in for the reference's hilbert.py: analytic signal u + j H[u] by FFT -> one-sided mask -> IFFT
(hilbert.py:22-112), fused in one kernel per row on the MI355X (complex64 math)."""
import numpy as np

from . import engine as _E


def _out_dtype(u):
    return np.complex64 if np.asarray(u).dtype == np.float32 else np.complex128


def hilbert(uin, nfft=None, axes=-1):
    """Analytic signal along `axes`; transform length nfft (default: the axis length).  `.squeeze()`d like the
    reference.  float32 input -> complex64 result (numpy>=2 behaviour of the reference), otherwise complex128."""
    u = np.atleast_1d(np.asarray(uin))
    if np.iscomplexobj(u):
        # FFT -> mask -> IFFT is linear: the complex case is the real kernel on both parts
        out_c = np.complex64 if u.dtype == np.complex64 else np.complex128
        return (hilbert(u.real, nfft, axes) + 1j * hilbert(u.imag, nfft, axes)).astype(out_c)
    if nfft is None:
        nfft = u.shape[axes]
    nfft = int(nfft)
    moved = np.moveaxis(u, axes, -1)
    rows = np.ascontiguousarray(moved.reshape(-1, moved.shape[-1]))
    z = _E.hilbert_rows(rows, nfft)
    z = z.reshape(moved.shape[:-1] + (nfft,))
    z = np.moveaxis(z, -1, axes)
    return z.astype(_out_dtype(u)).squeeze()


def hilbert_1d(uin, nfft=None):
    """1-D variant (hilbert.py:70-112); same mask, no squeeze."""
    u = np.atleast_1d(np.asarray(uin))
    if np.iscomplexobj(u):
        out_c = np.complex64 if u.dtype == np.complex64 else np.complex128
        return (hilbert_1d(u.real, nfft) + 1j * hilbert_1d(u.imag, nfft)).astype(out_c)
    if nfft is None:
        nfft = len(u)
    return _E.hilbert_rows(np.ascontiguousarray(u[None, :]), int(nfft))[0].astype(_out_dtype(u))
