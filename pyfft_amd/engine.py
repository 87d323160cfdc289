"""Array-level entry points over the C ABI (libspectral.so).

Two calling modes, chosen by the type of the sample array:
  * numpy arrays  -> `mem=0`: the library stages host buffers through its own device scratch and returns
                     numpy results (synchronous).  This is what the drop-in modules use.
  * torch CUDA tensors -> `mem=1`: device pointers are passed straight through, work is enqueued on
                     torch's current stream, results are torch tensors on the same device (asynchronous).
                     PyTorch is only the allocator / stream owner here.
There is no CPU implementation behind these functions.
"""
import numpy as np

from . import _ffi
from ._ffi import SIDED_ONE, SIDED_TWO, SIDED_RAW, SIDED_HALF, check, lib, ptr

try:                                    # torch is optional plumbing (device memory + streams)
    import torch
except Exception:                       # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _bind_stream(x):
    """mem=1 prologue: make the library launch on torch's current stream for x's device."""
    if not x.is_cuda:
        raise TypeError("torch tensors passed to pyfft_amd.engine must live on the GPU")
    _ffi.init(x.device.index if x.device.index is not None else torch.cuda.current_device())
    check(lib().sp_set_stream(torch.cuda.current_stream(x.device).cuda_stream))


def _torch_samples(x):
    if x.dtype not in (torch.float32, torch.complex64):
        raise TypeError("device path takes float32 or complex64 samples, got %s" % x.dtype)
    return x.contiguous()


def _tcode(x):
    return _ffi.DTYPE_C64 if x.dtype == torch.complex64 else _ffi.DTYPE_F32


def _win32(win):
    return np.ascontiguousarray(np.asarray(win), dtype=np.float32)


def nbins(nfft, sided):
    """Nnyquist bins for the reference's one-sided crop (fft_analysis.py:2471-2484), else nfft."""
    if sided == SIDED_ONE:
        return (nfft + 1) // 2 if nfft % 2 else nfft // 2
    if sided == SIDED_HALF:
        return nfft // 2 + 1
    return nfft


def _detrend_args(detrend, mean_value):
    """detrend: False/0/None none, True/1/'mean' mean, 2/'linear' least-squares line, 3/'segmean' every segment's own
    mean (welch_psd / welch_csd only: the matplotlib.mlab convention); an explicit mean_value
    (with detrend truthy) is subtracted as a constant instead of being computed."""
    if detrend in (None, False, 0, "none"):
        return _ffi.DETREND_CONST, 0j
    if detrend in (2, "linear"):
        return _ffi.DETREND_LINEAR, 0j
    if detrend in (3, "segmean"):
        return _ffi.DETREND_SEGMEAN, 0j
    if detrend in (4, "seglinear"):
        return _ffi.DETREND_SEGLINEAR, 0j
    if mean_value is not None:
        return _ffi.DETREND_CONST, complex(mean_value)
    return _ffi.DETREND_MEAN, 0j


def max_wg_fft():
    return int(lib().sp_max_wg_fft())


# ------------------------------------------------------------------------------------------ A7
def fft(x, n=None, axis=-1, inverse=False):
    """np.fft.fft / ifft semantics (forward unnormalised, inverse 1/n) on the GPU, complex64 math."""
    if _is_torch(x):
        _bind_stream(x)
        if axis not in (-1, x.dim() - 1) or (n is not None and n != x.shape[-1]):
            raise NotImplementedError("device-tensor fft: last axis, n == length")
        xc = x.to(torch.complex64).contiguous()
        out = torch.empty_like(xc)
        nn = xc.shape[-1]
        check(lib().sp_fft_c2c(ptr(xc.data_ptr()), ptr(out.data_ptr()), nn, xc.numel() // nn, 1 if inverse else -1, 1))
        return out
    a = np.asarray(x)
    a = np.moveaxis(a, axis, -1)
    if n is not None and n != a.shape[-1]:
        if n < a.shape[-1]:
            a = a[..., :n]
        else:
            pad = [(0, 0)] * (a.ndim - 1) + [(0, n - a.shape[-1])]
            a = np.pad(a, pad)
    a = np.ascontiguousarray(a, dtype=np.complex64)
    out = np.empty_like(a)
    nn = a.shape[-1]
    _ffi.init()
    check(lib().sp_fft_c2c(ptr(a), ptr(out), nn, a.size // nn if nn else 0, 1 if inverse else -1, 0))
    return np.moveaxis(out, -1, axis)


def ifft(x, n=None, axis=-1):
    return fft(x, n=n, axis=axis, inverse=True)


# ------------------------------------------------------------------------------------------ A13
def mean(x):
    """Mean of a float32/complex64 vector accumulated in double on the device (python float / complex)."""
    out = (_ffi.C.c_double * 2)()
    if _is_torch(x):
        _bind_stream(x)
        xs = _torch_samples(x)
        cplx = xs.is_complex()
        check(lib().sp_mean(ptr(xs.data_ptr()), _tcode(xs), xs.numel(), out, 1))
    else:
        xs = _ffi.as_samples(x)
        cplx = xs.dtype == np.complex64
        _ffi.init()
        check(lib().sp_mean(ptr(xs), _ffi.dtype_code(xs.dtype), xs.size, out, 0))
    return complex(out[0], out[1]) if cplx else float(out[0])


def profile_enable(on=True):
    _ffi.init()
    check(lib().sp_profile_enable(1 if on else 0))


def profile_last_ms():
    ms = _ffi.C.c_double(0.0)
    check(lib().sp_profile_last_ms(_ffi.C.byref(ms)))
    return ms.value


def profile_last_kernel():
    return (lib().sp_profile_last_kernel() or b"").decode()


# ------------------------------------------------------------------------------------------ A3+A4
def welch_psd(x, win, hop, nframes, detrend=True, sided=SIDED_TWO, scale=1.0, mean_value=None):
    """Fused Welch PSD: scale/nframes * sum_g |FFT(win*(x_g - mean))|^2, float64 [nbins].
    detrend=True subtracts the mean of the whole of x (computed on the device) unless mean_value is given."""
    w = _win32(win)
    nfft = w.size
    nb = nbins(nfft, sided)
    want, mv = _detrend_args(detrend, mean_value)
    if _is_torch(x):
        _bind_stream(x)
        xs = _torch_samples(x)
        out = torch.empty(nb, dtype=torch.float64, device=xs.device)
        check(lib().sp_welch_psd(ptr(xs.data_ptr()), _tcode(xs), xs.numel(), ptr(w), nfft, int(hop), int(nframes), want,
                                 mv.real, mv.imag, sided, float(scale), ptr(out.data_ptr()), 1))
        return out
    xs = _ffi.as_samples(x)
    out = np.empty(nb, dtype=np.float64)
    _ffi.init()
    check(lib().sp_welch_psd(ptr(xs), _ffi.dtype_code(xs.dtype), xs.size, ptr(w), nfft, int(hop), int(nframes), want,
                             mv.real, mv.imag, sided, float(scale), ptr(out), 0))
    return out


def welch_accum(x, win, hop, nframes, nmean=None):
    """First half of the sharded Welch PSD (see include/spectral.h: sp_welch_accum): accumulates this shard's frames
    against a local mean estimate.  Returns sum(x[0:nmean]) as a (2,) float64 (torch tensor on the device for device
    input, numpy otherwise) for the cross-shard all-reduce."""
    w = _win32(win)
    if _is_torch(x):
        _bind_stream(x)
        xs = _torch_samples(x)
        nm = xs.numel() if nmean is None else int(nmean)
        out = torch.empty(2, dtype=torch.float64, device=xs.device)
        check(lib().sp_welch_accum(ptr(xs.data_ptr()), _tcode(xs), xs.numel(), ptr(w), w.size, int(hop), int(nframes), nm,
                                   ptr(out.data_ptr()), 1))
        return out
    xs = _ffi.as_samples(x)
    nm = xs.size if nmean is None else int(nmean)
    out = np.empty(2, dtype=np.float64)
    _ffi.init()
    check(lib().sp_welch_accum(ptr(xs), _ffi.dtype_code(xs.dtype), xs.size, ptr(w), w.size, int(hop), int(nframes), nm,
                               ptr(out), 0))
    return out


def welch_export(x, win, hop, nframes, nmean=None):
    """One-collective form of the sharded Welch PSD (include/spectral.h: sp_welch_export): this shard's additive state,
    float64[5*nfft + 8] (torch tensor on the device for device input).  Sum the states of all shards, then welch_apply."""
    w = _win32(win)
    n = 5 * w.size + 8
    if _is_torch(x):
        _bind_stream(x)
        xs = _torch_samples(x)
        nm = xs.numel() if nmean is None else int(nmean)
        out = torch.empty(n, dtype=torch.float64, device=xs.device)
        check(lib().sp_welch_export(ptr(xs.data_ptr()), _tcode(xs), xs.numel(), ptr(w), w.size, int(hop), int(nframes), nm,
                                    ptr(out.data_ptr()), 1))
        return out
    xs = _ffi.as_samples(x)
    nm = xs.size if nmean is None else int(nmean)
    out = np.empty(n, dtype=np.float64)
    _ffi.init()
    check(lib().sp_welch_export(ptr(xs), _ffi.dtype_code(xs.dtype), xs.size, ptr(w), w.size, int(hop), int(nframes), nm,
                                ptr(out), 0))
    return out


def welch_apply(state, win, frames_total, sided=SIDED_TWO, scale=1.0):
    """PSD of the whole stream (global-mean detrend) from the summed shard states: float64[nbins]."""
    w = _win32(win)
    nb = nbins(w.size, sided)
    if _is_torch(state):
        _bind_stream(state)
        st = state.to(torch.float64).contiguous()
        out = torch.empty(nb, dtype=torch.float64, device=st.device)
        check(lib().sp_welch_apply(ptr(st.data_ptr()), ptr(w), w.size, int(frames_total), sided, float(scale),
                                   ptr(out.data_ptr()), 1))
        return out
    st = np.ascontiguousarray(state, dtype=np.float64)
    out = np.empty(nb, dtype=np.float64)
    _ffi.init()
    check(lib().sp_welch_apply(ptr(st), ptr(w), w.size, int(frames_total), sided, float(scale), ptr(out), 0))
    return out


# ---- multi-GPU inside the library (include/spectral.h: sp_comm_*, sp_welch_dist_*) ------------------------------------
def comm_unique_id():
    """rank 0: the RCCL unique id (bytes) every rank passes to comm_init"""
    buf = (_ffi.C.c_char * 128)()
    check(lib().sp_comm_unique_id(_ffi.C.cast(buf, _ffi.C.c_void_p)))
    return bytes(buf.raw)


def comm_init(uid, world, rank, device=None):
    """join the library's own RCCL communicator (collective over all ranks; one process per GPU)"""
    _ffi.init(-1 if device is None else int(device))
    buf = (_ffi.C.c_char * 128).from_buffer_copy(uid)
    check(lib().sp_comm_init(_ffi.C.cast(buf, _ffi.C.c_void_p), int(world), int(rank)))


def comm_info():
    out = (_ffi.C.c_int * 2)()
    check(lib().sp_comm_info(out))
    return int(out[0]), int(out[1])


def comm_destroy():
    check(lib().sp_comm_destroy())


def welch_dist_submit(x, win, hop, nframes, nmean, frames_total, sided=SIDED_TWO, scale=1.0):
    """One step of the streaming Welch PSD (sp_welch_dist_submit): the main kernel on torch's current stream, the epilogue (and
    with a communicator the RCCL all-reduce of the shard's state) on the library's own stream beside the NEXT step's main kernel.
    x: device tensor (this rank's shard).  Returns (out, ndone): `out` = the tensor that will hold THIS step's PSD of the whole
    stream (float64 [nbins], device), `ndone` = how many earlier steps' tensors became valid with this call.  Keep x and out
    alive until reported (NativeWelchPipeline does the bookkeeping)."""
    w = _win32(win)
    _bind_stream(x)
    xs = _torch_samples(x)
    out = torch.empty(nbins(w.size, sided), dtype=torch.float64, device=xs.device)
    nd = _ffi.C.c_int(0)
    check(lib().sp_welch_dist_submit(ptr(xs.data_ptr()), _tcode(xs), xs.numel(), ptr(w), w.size, int(hop), int(nframes),
                                     int(nmean), int(frames_total), sided, float(scale), ptr(out.data_ptr()), _ffi.C.byref(nd)))
    return out, xs, nd.value


def welch_dist_flush():
    """finish every step in flight; returns how many outputs became valid"""
    nd = _ffi.C.c_int(0)
    check(lib().sp_welch_dist_flush(_ffi.C.byref(nd)))
    return nd.value


def welch_finish(nfft, mean, frames_total, sided=SIDED_TWO, scale=1.0, like=None):
    """Second half: apply the (global) mean -- (2,) float64 [re, im], same kind of array welch_accum returned, or None
    for the shard's own mean -- and return scale/frames_total * sum_{local frames} |X|^2 as float64 [nbins]."""
    nb = nbins(nfft, sided)
    if _is_torch(like) or _is_torch(mean):
        dev = (mean if _is_torch(mean) else like).device
        out = torch.empty(nb, dtype=torch.float64, device=dev)
        mp = None
        if mean is not None:
            mean = mean.to(torch.float64).contiguous()
            mp = ptr(mean.data_ptr())
        check(lib().sp_welch_finish(mp, int(frames_total), sided, float(scale), ptr(out.data_ptr()), 1))
        return out
    out = np.empty(nb, dtype=np.float64)
    m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
    check(lib().sp_welch_finish(ptr(m), int(frames_total), sided, float(scale), ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ A5
def welch_csd(x, y, win, hop, nframes, detrend=True, sided=SIDED_ONE, scale=1.0):
    """Reference signal x[nsig] against channels y[nch, nsig] (channel-major).
    Returns (Pxx[nb], Pyy[nch, nb], Pxy[nch, nb] complex128 = Y conj(X)), all float64 based."""
    w = _win32(win)
    nfft = w.size
    nb = nbins(nfft, sided)
    if _is_torch(x):
        _bind_stream(x)
        xs = _torch_samples(x)
        if not _is_torch(y):
            raise TypeError("welch_csd: x is a device tensor, y must be one too")
        ys = _torch_samples(y)
        if ys.dim() == 1:
            ys = ys[None, :]
        if ys.dtype != xs.dtype:
            raise TypeError("welch_csd: x is %s but y is %s (both float32 or both complex64)" % (xs.dtype, ys.dtype))
        if ys.device != xs.device:
            raise ValueError("welch_csd: x and y live on different devices")
        if xs.dim() != 1 or ys.dim() != 2 or ys.shape[1] < xs.numel():
            raise ValueError("welch_csd: x[nsig] against y[nch, >= nsig]")
        nch, ld = ys.shape
        pxx = torch.empty(nb, dtype=torch.float64, device=xs.device)
        pyy = torch.empty((nch, nb), dtype=torch.float64, device=xs.device)
        pxy = torch.empty((nch, nb), dtype=torch.complex128, device=xs.device)
        check(lib().sp_welch_csd(ptr(xs.data_ptr()), ptr(ys.data_ptr()), _tcode(xs), xs.numel(), nch, ld, ptr(w), nfft,
                                 int(hop), int(nframes), _detrend_args(detrend, None)[0], None, None, sided,
                                 float(scale), ptr(pxx.data_ptr()), ptr(pyy.data_ptr()), ptr(pxy.data_ptr()), 1))
        return pxx, pyy, pxy
    xs = _ffi.as_samples(x)
    ys = np.asarray(y)
    if ys.ndim == 1:
        ys = ys[None, :]
    ys = np.ascontiguousarray(ys, dtype=xs.dtype)
    if np.iscomplexobj(y) and xs.dtype != np.complex64:
        xs = xs.astype(np.complex64)
        ys = np.ascontiguousarray(y, dtype=np.complex64).reshape(ys.shape)
    nch, ld = ys.shape
    pxx = np.empty(nb, dtype=np.float64)
    pyy = np.empty((nch, nb), dtype=np.float64)
    pxy = np.empty((nch, nb), dtype=np.complex128)
    _ffi.init()
    check(lib().sp_welch_csd(ptr(xs), ptr(ys), _ffi.dtype_code(xs.dtype), xs.size, nch, ld, ptr(w), nfft, int(hop),
                             int(nframes), _detrend_args(detrend, None)[0], None, None, sided, float(scale), ptr(pxx),
                             ptr(pyy), ptr(pxy), 0))
    return pxx, pyy, pxy


def csd_matrix(x, win, hop, nframes, detrend=True, scale=1.0, means=None):
    """Full cross-spectral-density matrix of real channels x[nch, nsig]:
    G[k, i, j] = scale/nframes * sum_g X_i[g,k] conj(X_j[g,k]),  k = 0..nfft/2 (rfft bins, no doubling), complex128.
    means (nch values): remove these constants instead of each channel's own mean (a frame shard of a longer record
    passes the means of the WHOLE record, see dist.csd_matrix_sharded)."""
    w = _win32(win)
    nfft = w.size
    nb = nfft // 2 + 1
    want = _detrend_args(detrend, None)[0]
    mh = None
    if means is not None:
        mh = np.ascontiguousarray(means.detach().cpu().numpy() if _is_torch(means) else means, dtype=np.float64).ravel()
    if _is_torch(x):
        _bind_stream(x)
        xs = x.to(torch.float32).contiguous()
        nch, ld = xs.shape
        out = torch.empty((nb, nch, nch), dtype=torch.complex128, device=xs.device)
        if mh is not None:
            if mh.size != nch:
                raise ValueError("means must have one value per channel")
            check(lib().sp_csd_matrix_means(ptr(xs.data_ptr()), nch, ld, ld, ptr(w), nfft, int(hop), int(nframes), ptr(mh),
                                            float(scale), ptr(out.data_ptr()), 1))
        else:
            check(lib().sp_csd_matrix(ptr(xs.data_ptr()), nch, ld, ld, ptr(w), nfft, int(hop), int(nframes), want,
                                      float(scale), ptr(out.data_ptr()), 1))
        return out
    xs = np.ascontiguousarray(x, dtype=np.float32)
    nch, ld = xs.shape
    out = np.empty((nb, nch, nch), dtype=np.complex128)
    _ffi.init()
    if mh is not None:
        if mh.size != nch:
            raise ValueError("means must have one value per channel")
        check(lib().sp_csd_matrix_means(ptr(xs), nch, ld, ld, ptr(w), nfft, int(hop), int(nframes), ptr(mh), float(scale),
                                        ptr(out), 0))
    else:
        check(lib().sp_csd_matrix(ptr(xs), nch, ld, ld, ptr(w), nfft, int(hop), int(nframes), want, float(scale), ptr(out), 0))
    return out


def channel_means(x, nsamples=None):
    """Mean of the first `nsamples` samples of every row of x[nch, nsig] (float32 channels), float64[nch]."""
    if _is_torch(x):
        _bind_stream(x)
        xs = x.to(torch.float32).contiguous()
        nch, ld = xs.shape
        n = ld if nsamples is None else int(nsamples)
        out = torch.empty(nch, dtype=torch.float64, device=xs.device)
        check(lib().sp_channel_means(ptr(xs.data_ptr()), nch, n, ld, ptr(out.data_ptr()), 1))
        return out
    xs = np.ascontiguousarray(x, dtype=np.float32)
    nch, ld = xs.shape
    n = ld if nsamples is None else int(nsamples)
    out = np.empty(nch, dtype=np.float64)
    _ffi.init()
    check(lib().sp_channel_means(ptr(xs), nch, n, ld, ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ A8/A9
def stft_frames(x, win, hop, nframes, detrend=True, sided=SIDED_ONE, amp_scale=1.0, power=False, bin_major=False,
                want_pseg=False, mean_value=None):
    """Per-frame spectra.  complex64 [nframes, nbins] (or float32 power); bin_major -> [nbins, nframes].
    Returns (out, pseg or None); pseg[g] = trapz(|win*(x_g-mean)|^2), unit sample spacing, float64."""
    w = _win32(win)
    nfft = w.size
    nb = nbins(nfft, sided)
    want, mv = _detrend_args(detrend, mean_value)
    shape = (nb, int(nframes)) if bin_major else (int(nframes), nb)
    if _is_torch(x):
        _bind_stream(x)
        xs = _torch_samples(x)
        out = torch.empty(shape, dtype=torch.float32 if power else torch.complex64, device=xs.device)
        pseg = torch.empty(int(nframes), dtype=torch.float64, device=xs.device) if want_pseg else None
        check(lib().sp_stft(ptr(xs.data_ptr()), _tcode(xs), xs.numel(), ptr(w), nfft, int(hop), int(nframes), want,
                            mv.real, mv.imag, sided, float(amp_scale), 1 if power else 0, 1 if bin_major else 0,
                            ptr(out.data_ptr()), ptr(pseg.data_ptr()) if want_pseg else None, 1))
        return out, pseg
    xs = _ffi.as_samples(x)
    out = np.empty(shape, dtype=np.float32 if power else np.complex64)
    pseg = np.empty(int(nframes), dtype=np.float64) if want_pseg else None
    _ffi.init()
    check(lib().sp_stft(ptr(xs), _ffi.dtype_code(xs.dtype), xs.size, ptr(w), nfft, int(hop), int(nframes), want,
                        mv.real, mv.imag, sided, float(amp_scale), 1 if power else 0, 1 if bin_major else 0, ptr(out),
                        ptr(pseg), 0))
    return out, pseg


# ------------------------------------------------------------------------------------------ N3
def stft_cog(x, win, hop, nframes, fs, fmin=0.0, fmax=None, detrend=False, mean_value=None):
    """Centre of gravity (power-weighted mean frequency, Doppler.py:43-58) of every frame's two-sided spectrum, reduced
    inside the transform kernel.  float64 [nframes]; band fmin <= |f| <= fmax (default: every bin)."""
    w = _win32(win)
    nfft = w.size
    want, mv = _detrend_args(detrend, mean_value)
    fmax = float(fs) if fmax is None else float(fmax)
    if _is_torch(x):
        _bind_stream(x)
        xs = _torch_samples(x)
        out = torch.empty(int(nframes), dtype=torch.float64, device=xs.device)
        check(lib().sp_stft_cog(ptr(xs.data_ptr()), _tcode(xs), xs.numel(), ptr(w), nfft, int(hop), int(nframes), want,
                                mv.real, mv.imag, float(fs), float(fmin), fmax, ptr(out.data_ptr()), 1))
        return out
    xs = _ffi.as_samples(x)
    out = np.empty(int(nframes), dtype=np.float64)
    _ffi.init()
    check(lib().sp_stft_cog(ptr(xs), _ffi.dtype_code(xs.dtype), xs.size, ptr(w), nfft, int(hop), int(nframes), want,
                            mv.real, mv.imag, float(fs), float(fmin), fmax, ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ A10
def hilbert_rows(x2d, nfft):
    """Analytic signal of each row of a real [batch, n_in] array, transform length nfft -> complex64 [batch, nfft]."""
    if _is_torch(x2d):
        _bind_stream(x2d)
        xs = x2d.to(torch.float32).contiguous()
        batch, n_in = xs.shape
        out = torch.empty((batch, nfft), dtype=torch.complex64, device=xs.device)
        check(lib().sp_hilbert(ptr(xs.data_ptr()), n_in, n_in, int(nfft), batch, ptr(out.data_ptr()), 1))
        return out
    xs = np.ascontiguousarray(x2d, dtype=np.float32)
    batch, n_in = xs.shape
    out = np.empty((batch, nfft), dtype=np.complex64)
    _ffi.init()
    check(lib().sp_hilbert(ptr(xs), n_in, n_in, int(nfft), batch, ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ A5 (nT-model)
def frame_sum(y2d, nfft, hop, nframes, detrend=True):
    """c[ch][n] = sum_g detrended(y[ch][g*hop + n]), n < nfft, for each row of [nch, nsig]; complex128 (float64 for real
    input) [nch, nfft].  sum_g FFT(win*frame_g) = FFT(win*c): the mean spectrum without writing the frames' spectra."""
    want, _ = _detrend_args(detrend, None)
    if want not in (_ffi.DETREND_CONST, _ffi.DETREND_MEAN, _ffi.DETREND_LINEAR):
        raise ValueError("frame_sum: detrend must be none, mean or linear")
    if _is_torch(y2d):
        _bind_stream(y2d)
        ys = y2d.contiguous()
        ys = ys.to(torch.complex64) if ys.is_complex() else ys.to(torch.float32)
        nch, nsig = ys.shape
        out = torch.empty((nch, int(nfft), 2), dtype=torch.float64, device=ys.device)
        check(lib().sp_frame_sum(ptr(ys.data_ptr()), _tcode(ys), nsig, nch, nsig, int(nfft), int(hop), int(nframes), want,
                                 ptr(out.data_ptr()), 1))
        return torch.view_as_complex(out) if ys.is_complex() else out[..., 0]
    a = np.asarray(y2d)
    ys = np.ascontiguousarray(a, dtype=np.complex64 if np.iscomplexobj(a) else np.float32)
    nch, nsig = ys.shape
    out = np.empty((nch, int(nfft), 2), dtype=np.float64)
    _ffi.init()
    check(lib().sp_frame_sum(ptr(ys), _ffi.dtype_code(ys.dtype), nsig, nch, nsig, int(nfft), int(hop), int(nframes), want,
                             ptr(out), 0))
    return out.view(np.complex128)[..., 0] if np.iscomplexobj(ys) else out[..., 0].copy()


# ------------------------------------------------------------------------------------------ N4
def spectral_filter_rows(x2d, H):
    """IFFT(H * FFT(row)) for each real row of [batch, n]; H complex [n] (host table).  complex64 [batch, n]."""
    Hc = np.ascontiguousarray(H, dtype=np.complex64)
    nfft = Hc.size
    if _is_torch(x2d):
        _bind_stream(x2d)
        xs = x2d.to(torch.float32).contiguous()
        batch, n_in = xs.shape
        out = torch.empty((batch, nfft), dtype=torch.complex64, device=xs.device)
        check(lib().sp_spectral_filter(ptr(xs.data_ptr()), n_in, n_in, nfft, batch, ptr(Hc), ptr(out.data_ptr()), 1))
        return out
    xs = np.ascontiguousarray(x2d, dtype=np.float32)
    batch, n_in = xs.shape
    out = np.empty((batch, nfft), dtype=np.complex64)
    _ffi.init()
    check(lib().sp_spectral_filter(ptr(xs), n_in, n_in, nfft, batch, ptr(Hc), ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ A11
def xcorr_normalised(x1, x2):
    """co[2n-1] = correlate(x1-m1, x2-m2, 'full') / (n std1 std2), float32."""
    if _is_torch(x1):
        _bind_stream(x1)
        if not _is_torch(x2) or x2.device != x1.device:
            raise ValueError("xcorr: both signals must be tensors on the same device")
        if x1.dim() != 1 or x2.shape != x1.shape:
            raise ValueError("xcorr: two 1-D signals of equal length")
        if x1.is_complex() or x2.is_complex():
            raise TypeError("xcorr: real signals")
        a = x1.to(torch.float32).contiguous()
        b = x2.to(torch.float32).contiguous()
        n = a.numel()
        out = torch.empty(2 * n - 1, dtype=torch.float32, device=a.device)
        check(lib().sp_xcorr(ptr(a.data_ptr()), ptr(b.data_ptr()), n, ptr(out.data_ptr()), 1))
        return out
    if np.iscomplexobj(x1) or np.iscomplexobj(x2):
        raise TypeError("xcorr: real signals")
    a = np.ascontiguousarray(x1, dtype=np.float32)
    b = np.ascontiguousarray(x2, dtype=np.float32)
    if a.shape != b.shape or a.ndim != 1:
        raise ValueError("xcorr: two 1-D signals of equal length")
    n = a.size
    out = np.empty(2 * n - 1, dtype=np.float32)
    _ffi.init()
    check(lib().sp_xcorr(ptr(a), ptr(b), n, ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ F1
def fir_filter(h, x, nfft=0):
    """Causal FIR y = lfilter(h, 1, x) (float32) by overlap-save on the GPU."""
    taps = np.ascontiguousarray(h, dtype=np.float32)
    if _is_torch(x):
        _bind_stream(x)
        xs = x.to(torch.float32).contiguous()
        out = torch.empty_like(xs)
        check(lib().sp_fftfilt(ptr(taps), taps.size, ptr(xs.data_ptr()), xs.numel(), int(nfft), ptr(out.data_ptr()), 1))
        return out
    xs = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(xs)
    _ffi.init()
    check(lib().sp_fftfilt(ptr(taps), taps.size, ptr(xs), xs.size, int(nfft), ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ F2
def biquad_filter(b, a, x):
    """y = scipy.signal.lfilter(b, a, x) for one second-order section (b[3], a[3]), float32 samples, evaluated exactly
    on the GPU (float64 recurrence, blocked scan of the state maps; include/spectral.h: sp_biquad)."""
    bb = np.ascontiguousarray(b, dtype=np.float64).ravel()
    aa = np.ascontiguousarray(a, dtype=np.float64).ravel()
    if bb.size > 3 or aa.size > 3 or aa.size < 1 or bb.size < 1:
        raise ValueError("biquad_filter: b and a hold at most 3 coefficients")
    bb = np.concatenate([bb, np.zeros(3 - bb.size)])
    aa = np.concatenate([aa, np.zeros(3 - aa.size)])
    if aa[0] == 0.0:
        raise ValueError("biquad_filter: a[0] must not be zero")
    rts = np.roots(aa)
    if rts.size and float(np.max(np.abs(rts))) > 1.0 + 1e-12:
        # (the device scan raises the state map to powers up to len(x): an unstable section would overflow them to inf / NaN)
        raise ValueError("biquad_filter: unstable section (pole radius %.9g > 1)" % float(np.max(np.abs(rts))))
    if _is_torch(x):
        _bind_stream(x)
        if x.is_complex() or x.dim() != 1:
            raise TypeError("biquad_filter: one real 1-D signal")
        xs = x.to(torch.float32).contiguous()
        out = torch.empty_like(xs)
        check(lib().sp_biquad(ptr(bb), ptr(aa), ptr(xs.data_ptr()), xs.numel(), ptr(out.data_ptr()), 1))
        return out
    if np.iscomplexobj(x) or np.ndim(x) != 1:
        raise TypeError("biquad_filter: one real 1-D signal")
    xs = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(xs)
    _ffi.init()
    check(lib().sp_biquad(ptr(bb), ptr(aa), ptr(xs), xs.size, ptr(out), 0))
    return out


# ------------------------------------------------------------------------------------------ A6 / N1
def csd_epilogue(pxx, pyy, pxy, nfft, onesided, enbw):
    """The fft_pwelch epilogue (fft_analysis.py:489-648) on the averaged spectra, on the device: pxx[nb], pyy[nch, nb],
    pxy[nch, nb] complex as welch_csd returns them (numpy, or torch tensors that stay on the GPU).  Returns a dict of
    channel-major arrays: Cxy[nch, nb] complex, Cxy2, phi, Lyy, Lxy [nch, nb], Lxx[nb], and the fftshifted correlations
    Rxx[nfft], Ryy / Rxy / iCxy / corrcoef [nch, nfft] (real for one-sided input, complex otherwise), Ex, Ey[nch]."""
    nfft, onesided = int(nfft), bool(onesided)
    dev = _is_torch(pxx)
    if dev:
        _bind_stream(pxx)
        a = pxx.to(torch.float64).contiguous() if not pxx.is_complex() else pxx.real.to(torch.float64).contiguous()
        b = (pyy if not pyy.is_complex() else pyy.real).to(torch.float64).contiguous()
        c = pxy.to(torch.complex128).contiguous()
        if b.dim() == 1:
            b, c = b[None, :], c[None, :]
        nch, nb = b.shape
        n_out = int(lib().sp_csd_epilogue_doubles(nch, nb, nfft))
        out = torch.empty(n_out, dtype=torch.float64, device=a.device)
        check(lib().sp_csd_epilogue(ptr(a.data_ptr()), ptr(b.data_ptr()), ptr(c.data_ptr()), nch, nb, nfft, 1 if onesided else 0,
                                    float(enbw), ptr(out.data_ptr()), 1))
        cplx = lambda v: torch.complex(v[..., 0], v[..., 1])          # noqa: E731  (slices of `out` may start at odd offsets)
    else:
        a = np.ascontiguousarray(np.real(pxx), dtype=np.float64)
        b = np.ascontiguousarray(np.real(pyy), dtype=np.float64)
        c = np.ascontiguousarray(pxy, dtype=np.complex128)
        if b.ndim == 1:
            b, c = b[None, :], c[None, :]
        nch, nb = b.shape
        _ffi.init()
        n_out = int(lib().sp_csd_epilogue_doubles(nch, nb, nfft))
        out = np.empty(n_out, dtype=np.float64)
        check(lib().sp_csd_epilogue(ptr(a), ptr(b), ptr(c), nch, nb, nfft, 1 if onesided else 0, float(enbw), ptr(out), 0))
        cplx = lambda v: v[..., 0] + 1j * v[..., 1]                     # noqa: E731
    pos = [0]

    def take(*shape):
        n = 1
        for d in shape:
            n *= d
        v = out[pos[0]:pos[0] + n].reshape(shape)
        pos[0] += n
        return v
    r = {}
    r["Cxy"] = cplx(take(nch, nb, 2))
    r["Cxy2"] = take(nch, nb)
    r["phi"] = take(nch, nb)
    r["Lxx"] = take(nb)
    r["Lyy"] = take(nch, nb)
    r["Lxy"] = take(nch, nb)
    corr = {"Rxx": take(nfft, 2), "Ryy": take(nch, nfft, 2), "Rxy": take(nch, nfft, 2), "iCxy": take(nch, nfft, 2),
            "corrcoef": take(nch, nfft, 2)}
    e = take(1 + nch, 2)
    for k, v in corr.items():
        r[k] = v[..., 0] if onesided else cplx(v)
    r["Ex"] = e[0, 0] if onesided else cplx(e[0])
    r["Ey"] = e[1:, 0] if onesided else cplx(e[1:])
    return r
