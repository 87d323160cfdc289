"""pyfft_amd -- MI355X-native spectral-analysis engine behind gmweir/PYFFT's function signatures.

Mirrors the reference package's export list (__init__.py:11-31) for the hot path:
    fft_analysis (alias `fft`), fft_pwelch, fftanal, windows, specgram, stft, hilbert, hilbert_1d, ccf,
    iirnotch, iirpeak, plus the build-defined fftfilt / apply_notch.
Importing the package loads nothing from the GPU; the first kernel call initialises the HIP library
(pyfft_amd/lib/libspectral.so) and fails loudly if it is missing -- there is no CPU fallback.
"""
from . import _ffi                      # noqa: F401
from . import engine                    # noqa: F401
from . import fft_analysis              # noqa: F401
from . import fft_analysis as fft       # noqa: F401   (reference: `import fft_analysis as fft`, __init__.py:21)
from . import spectrogram, hilbert as _hilbert_mod, ccf as _ccf_mod, filters, notch_filter   # noqa: F401
from .windows import windows            # noqa: F401
from .fft_analysis import fft_pwelch, fftanal, Cxy_Cxy2, psd, csd, coh, coh2     # noqa: F401
from .fft_analysis import detrend_none, detrend_mean, detrend_linear, unwrap_tol, fft_deriv   # noqa: F401  (__init__.py:22-23)
from .fft_analysis import integratespectra, varcoh, varphi, mean_angle                       # noqa: F401  (fft_analysis.py:835, :1218-1376)
from .spectrogram import specgram, stft                      # noqa: F401
from .hilbert import hilbert, hilbert_1d                     # noqa: F401
from .ccf import ccf                                         # noqa: F401
from .notch_filter import iirnotch, iirpeak, apply_notch     # noqa: F401
from .filters import fftfilt                                 # noqa: F401
from . import doppler                                        # noqa: F401   (Doppler.cog / cogspec window loop)
from .doppler import cog, cog_frames                         # noqa: F401
from . import heatpulse                                      # noqa: F401   (HeatPulse_Funcs._PWELCH_chloop as one call)
from .heatpulse import pwelch_chloop                         # noqa: F401
