"""ctypes binding of libspectral.so (include/spectral.h).

The HIP library is the product: there is no CPU fallback.  Importing this module never touches
the GPU; the first call does (sp_init), and fails loudly if the shared object is missing, was not
built for gfx950, or no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SP_LIB_PATH") or os.path.join(_HERE, "lib", "libspectral.so")   # SP_LIB_PATH: diagnostic builds

DTYPE_F32, DTYPE_C64 = 0, 1
SIDED_ONE, SIDED_TWO, SIDED_RAW, SIDED_HALF = 1, 2, 3, 4
DETREND_CONST, DETREND_MEAN, DETREND_LINEAR, DETREND_SEGMEAN, DETREND_SEGLINEAR = 0, 1, 2, 3, 4

_lib = None

_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double

# name -> (restype, argtypes); must list every symbol include/spectral.h declares
SIGNATURES = {
    "sp_init": (_i, [_i]),
    "sp_shutdown": (None, []),
    "sp_last_error": (C.c_char_p, []),
    "sp_set_stream": (_i, [_vp]),
    "sp_synchronize": (_i, []),
    "sp_version": (_i, []),
    "sp_max_wg_fft": (_i, []),
    "sp_device_info": (_i, [C.POINTER(_i64)]),
    "sp_profile_enable": (_i, [_i]),
    "sp_profile_last_ms": (_i, [C.POINTER(_d)]),
    "sp_profile_last_kernel": (C.c_char_p, []),
    "sp_fft_c2c": (_i, [_vp, _vp, _i64, _i64, _i, _i]),
    "sp_welch_psd": (_i, [_vp, _i, _i64, _vp, _i, _i, _i64, _i, _d, _d, _i, _d, _vp, _i]),
    "sp_welch_accum": (_i, [_vp, _i, _i64, _vp, _i, _i, _i64, _i64, _vp, _i]),
    "sp_welch_finish": (_i, [_vp, _i64, _i, _d, _vp, _i]),
    "sp_welch_export": (_i, [_vp, _i, _i64, _vp, _i, _i, _i64, _i64, _vp, _i]),
    "sp_welch_apply": (_i, [_vp, _vp, _i, _i64, _i, _d, _vp, _i]),
    "sp_comm_unique_id": (_i, [_vp]),
    "sp_comm_init": (_i, [_vp, _i, _i]),
    "sp_comm_info": (_i, [C.POINTER(_i)]),
    "sp_comm_destroy": (_i, []),
    "sp_welch_dist_submit": (_i, [_vp, _i, _i64, _vp, _i, _i, _i64, _i64, _i64, _i, _d, _vp, C.POINTER(_i)]),
    "sp_welch_dist_flush": (_i, [C.POINTER(_i)]),
    "sp_welch_psd_dist": (_i, [_vp, _i, _i64, _vp, _i, _i, _i64, _i64, _i64, _i, _d, _vp]),
    "sp_welch_csd": (_i, [_vp, _vp, _i, _i64, _i, _i64, _vp, _i, _i, _i64, _i, _vp, _vp, _i, _d, _vp, _vp, _vp, _i]),
    "sp_csd_matrix": (_i, [_vp, _i, _i64, _i64, _vp, _i, _i, _i64, _i, _d, _vp, _i]),
    "sp_csd_matrix_means": (_i, [_vp, _i, _i64, _i64, _vp, _i, _i, _i64, _vp, _d, _vp, _i]),
    "sp_channel_means": (_i, [_vp, _i, _i64, _i64, _vp, _i]),
    "sp_stft": (_i, [_vp, _i, _i64, _vp, _i, _i, _i64, _i, _d, _d, _i, _d, _i, _i, _vp, _vp, _i]),
    "sp_stft_cog": (_i, [_vp, _i, _i64, _vp, _i, _i, _i64, _i, _d, _d, _d, _d, _d, _vp, _i]),
    "sp_hilbert": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _i]),
    "sp_frame_sum": (_i, [_vp, _i, _i64, _i, _i64, _i, _i, _i64, _i, _vp, _i]),
    "sp_spectral_filter": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i]),
    "sp_xcorr": (_i, [_vp, _vp, _i64, _vp, _i]),
    "sp_fftfilt": (_i, [_vp, _i, _vp, _i64, _i, _vp, _i]),
    "sp_biquad": (_i, [_vp, _vp, _vp, _i64, _vp, _i]),
    "sp_csd_epilogue_doubles": (_i64, [_i, _i, _i]),
    "sp_csd_epilogue": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _d, _vp, _i]),
    "sp_mean": (_i, [_vp, _i, _i64, C.POINTER(_d), _i]),
}


class SpectralError(RuntimeError):
    pass


def load_library():
    """dlopen libspectral.so and bind every declared symbol (no GPU work)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpectralError("libspectral.so not found at %s -- run `make` (or __graft_entry__.build()); "
                            "pyfft_amd has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lib():
    l = load_library()
    return l


def check(rc):
    if rc != 0:
        raise SpectralError((lib().sp_last_error() or b"unknown error").decode())


def init(device=-1):
    check(lib().sp_init(int(device)))


def ptr(a):
    """void* of a C-contiguous numpy array or a raw device address (int) / None."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def dtype_code(arr_dtype):
    if arr_dtype == np.float32:
        return DTYPE_F32
    if arr_dtype == np.complex64:
        return DTYPE_C64
    raise TypeError("device path takes float32 or complex64 samples, got %s" % arr_dtype)


def as_samples(x):
    """Cast like the reference's device boundary would: real -> float32, complex -> complex64, contiguous."""
    x = np.asarray(x)
    if np.iscomplexobj(x):
        return np.ascontiguousarray(x, dtype=np.complex64)
    return np.ascontiguousarray(x, dtype=np.float32)
