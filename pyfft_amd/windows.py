"""Window catalogue: host-side table generation for the fused first butterfly stage.

Mirrors `windows(windowfunction, nwins=, periodic=True, verbose=True, msgout=False, beta=)` of the
reference (windows.py:57-297): name matching by substring, the Heinzel cosine-sum coefficient sets with
their recommended overlaps (ROV), "periodic" = generator(N+1)[:-1].  The table is computed in float64
(like the reference) and handed to the device as float32.
"""
import numpy as np

# (family label, description) strings are informational only (reference prints them when verbose)
_HEINZEL = {
    # key substring : (label, coefficients, recommended overlap)
    "3f": ("Fast-decaying Flattop", (0.26526, -0.5, 0.23474), 0.667),
    "4f": ("Fast-decaying Flattop", (0.21706, -0.42103, 0.28294, -0.07897), 0.75),
    "5f": ("Fast-decaying Flattop", (0.1881, -0.36923, 0.28702, -0.13077, 0.02488), 0.785),
    "3m": ("Minimum sidelobe Flattop", (0.28235, -0.52105, 0.19659), 0.655),
    "4m": ("Minimum sidelobe Flattop", (0.241906, -0.460841, 0.255381, -0.041872), 0.721),
    "5m": ("Minimum sidelobe Flattop", (0.209671, -0.407331, 0.281225, -0.092669, 0.0091036), 0.760),
    "3a": ("3-term Blackman-Harris type", (0.40897, -0.5, 0.09103), 0.612),
    "3b": ("3-term Blackman-Harris type", (0.4243801, -0.4973406, 0.0782793), 0.598),
    "3": ("3-term Blackman-Harris type", (0.375, -0.5, 0.125), 0.647),
    "4a": ("4-term Blackman-Harris type", (0.338946, -0.481973, 0.161054, -0.018027), 0.68),
    "4b": ("4-term Blackman-Harris type", (0.355768, -0.487396, 0.144232, -0.012604), 0.663),
    "4c": ("4-term Blackman-Harris type", (0.3635819, -0.4891775, 0.1365995, -0.0106411), 0.656),
    "4": ("4-term Blackman-Harris type", (0.3125, -0.46875, 0.1875, -0.03125), 0.705),
}
# the reference tests the substrings in this order (windows.py:101-207)
_HEINZEL_ORDER = ("3f", "4f", "5f", "3m", "4m", "5m", "3a", "3b", "3", "4a", "4b", "4c", "4")


def cosine_sum(coeffs, n):
    """sum_i c_i cos(2 pi i k / n), k = 0..n-1 (the divisor is the generated length: windows.py:222-232)."""
    k = np.arange(n, dtype=np.float64)
    z = 2.0 * np.pi * k / n
    w = np.full(n, coeffs[0], dtype=np.float64)
    for i in range(1, len(coeffs)):
        w = w + coeffs[i] * np.cos(i * z)
    return w


def _select(name, beta):
    s = name.lower()
    if "hann" in s:
        return "Hanning", np.hanning, 0.50
    if "hamm" in s:
        return "Hamming", np.hamming, 0.50
    if "black" in s:
        return "Blackman-Harris type", (lambda n: cosine_sum((0.35875, -0.48829, 0.14128, -0.01168), n)), 0.661
    if ("nut" in s) or ("flat" in s) or ("sft" in s):
        for key in _HEINZEL_ORDER:
            if key in s:
                label, cc, rov = _HEINZEL[key]
                return label, (lambda n, cc=cc: cosine_sum(cc, n)), rov
        raise NameError("window %r: no Nuttall/flat-top coefficient set matches" % name)
    if "kaiser" in s:
        if beta is None:
            raise KeyError("beta")
        return "Kaiser type", (lambda n: np.kaiser(n, beta)), 2.0 / 3.0
    if "welch" in s:
        def parabola(n):
            z = 2.0 * np.arange(n, dtype=np.float64) / n
            return 1.0 - (z - 1.0) ** 2
        return "Welch", parabola, 0.293
    if "bart" in s:
        return "Bartlett", np.bartlett, 0.50
    return "Rectangular", (lambda n: np.ones(n, dtype=np.float64)), 0.0


def windows(windowfunction, **kwargs):
    """Drop-in for the reference's `windows()`: returns the float64 window table when `nwins` is given,
    otherwise the recommended overlap fraction.  With msgout=True returns (value, (label, ''))."""
    verbose = kwargs.get("verbose", True)
    periodic = kwargs.get("periodic", True)
    msgout = kwargs.get("msgout", False)
    label, gen, rov = _select(windowfunction, kwargs.get("beta"))
    if "nwins" in kwargs:
        n = int(kwargs["nwins"])
        val = gen(n + 1)[:-1] if periodic else gen(n)
        msg = "Using a %s %s window function" % ("periodic" if periodic else "aperiodic", label)
    else:
        val = rov
        msg = "Getting recommended overlap for a %s window function" % label
    if verbose:
        print(msg)
    if msgout:
        return val, (label, "ROV=%4.1f%%" % (100.0 * rov))
    return val
