"""pyfft_amd -- MI355X-native spectral-analysis engine behind gmweir/PYFFT's function signatures.

Importing the package loads nothing from the GPU; the first call into a kernel initialises the HIP
library (pyfft_amd/lib/libspectral.so) and fails loudly if it is missing -- there is no CPU fallback.
"""
from . import _ffi            # noqa: F401
from . import engine          # noqa: F401
from .windows import windows  # noqa: F401

__all__ = ["engine", "windows"]
