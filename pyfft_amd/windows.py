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


# ======================================================================================================== #
# Named window generators (reference windows.py:301-2425, itself the scipy.signal.windows catalogue).
# Host-side float64 tables: `sym=True` gives the filter-design (symmetric) form, `sym=False` the DFT-even
# ("periodic") form = generator(M+1)[:-1].  M <= 1 returns ones(M); a negative or fractional M raises.
# ======================================================================================================== #
def _len_guards(M):
    if int(M) != M or M < 0:
        raise ValueError("Window length M must be a non-negative integer")
    return M <= 1


def _extend(M, sym):
    return (M, False) if sym else (M + 1, True)


def _truncate(w, needed):
    return w[:-1] if needed else w


def _table(shape_fn):
    """Wrap `shape_fn(m, *params)` (symmetric table of m > 1 points) with the length guards and the DFT-even rule."""
    def make(M, *params, sym=True):
        if _len_guards(M):
            return np.ones(int(M))
        m, cut = _extend(int(M), sym)
        return _truncate(np.asarray(shape_fn(m, *params), dtype=np.float64), cut)
    make.__name__ = shape_fn.__name__.lstrip("_")
    make.__doc__ = shape_fn.__doc__
    return make


def _general_cosine(m, a):
    """sum_k (-1)^k a_k cos(k t), t in [-pi, pi] over m points (windows.py:301-380)."""
    t = np.linspace(-np.pi, np.pi, m)
    w = np.zeros(m)
    for k, ak in enumerate(a):
        w += ak * np.cos(k * t)
    return w


general_cosine = _table(_general_cosine)


def general_hamming(M, alpha, sym=True):
    """alpha - (1 - alpha) cos(2 pi n / (M - 1))  (windows.py:1195-1279)."""
    return general_cosine(M, [alpha, 1.0 - alpha], sym=sym)


# fixed cosine-sum members of the catalogue: name -> coefficients a_k of general_cosine
_COSINE_SUMS = {
    "blackman": (0.42, 0.50, 0.08),
    "nuttall": (0.3635819, 0.4891775, 0.1365995, 0.0106411),
    "blackmanharris": (0.35875, 0.48829, 0.14128, 0.01168),
    "flattop": (0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368),
}


def blackman(M, sym=True):
    return general_cosine(M, _COSINE_SUMS["blackman"], sym=sym)


def nuttall(M, sym=True):
    return general_cosine(M, _COSINE_SUMS["nuttall"], sym=sym)


def blackmanharris(M, sym=True):
    return general_cosine(M, _COSINE_SUMS["blackmanharris"], sym=sym)


def flattop(M, sym=True):
    return general_cosine(M, _COSINE_SUMS["flattop"], sym=sym)


def hann(M, sym=True):
    return general_hamming(M, 0.5, sym=sym)


def hanning(*args, **kwargs):
    return hann(*args, **kwargs)


def hamming(M, sym=True):
    return general_hamming(M, 0.54, sym=sym)


@_table
def _boxcar(m):
    return np.ones(m)


boxcar = _boxcar


@_table
def _triang(m):
    """Triangle that does not touch zero (windows.py:436-497)."""
    half = np.arange(1, (m + 1) // 2 + 1)
    if m % 2 == 0:
        up = (2 * half - 1.0) / m
        return np.concatenate((up, up[::-1]))
    up = 2 * half / (m + 1.0)
    return np.concatenate((up, up[-2::-1]))


triang = _triang


@_table
def _parzen(m):
    """Piecewise-cubic B-spline window (windows.py:500-561)."""
    n = np.arange(-(m - 1) / 2.0, (m - 1) / 2.0 + 0.5, 1.0)
    outer = n[n < -(m - 1) / 4.0]
    inner = n[np.abs(n) <= (m - 1) / 4.0]
    h = m / 2.0
    wo = 2 * (1 - np.abs(outer) / h) ** 3.0
    wi = 1 - 6 * (np.abs(inner) / h) ** 2.0 + 6 * (np.abs(inner) / h) ** 3.0
    return np.concatenate((wo, wi, wo[::-1]))


parzen = _parzen


@_table
def _bohman(m):
    """(1-|x|) cos(pi|x|) + sin(pi|x|)/pi, x in [-1, 1], end points exactly zero (windows.py:564-616)."""
    x = np.abs(np.linspace(-1, 1, m)[1:-1])
    core = (1 - x) * np.cos(np.pi * x) + 1.0 / np.pi * np.sin(np.pi * x)
    return np.concatenate(([0.0], core, [0.0]))


bohman = _bohman


@_table
def _bartlett(m):
    """Triangle with zero end points (windows.py:872-967)."""
    n = np.arange(0, m)
    return np.where(n <= (m - 1) / 2.0, 2.0 * n / (m - 1), 2.0 - 2.0 * n / (m - 1))


bartlett = _bartlett


@_table
def _barthann(m):
    """Modified Bartlett-Hann (windows.py:1140-1192)."""
    x = np.abs(np.arange(0, m) / (m - 1.0) - 0.5)
    return 0.62 - 0.48 * x + 0.38 * np.cos(2 * np.pi * x)


barthann = _barthann


@_table
def _gaussian(m, std):
    """exp(-n^2 / (2 std^2)) centred (windows.py:1360-1420)."""
    n = np.arange(0, m) - (m - 1.0) / 2.0
    return np.exp(-n ** 2 / (2 * std * std))


gaussian = _gaussian


@_table
def _general_gaussian(m, p, sig):
    """exp(-|n/sig|^(2p) / 2) centred (windows.py:1423-1490)."""
    n = np.arange(0, m) - (m - 1.0) / 2.0
    return np.exp(-0.5 * np.abs(n / sig) ** (2 * p))


general_gaussian = _general_gaussian


@_table
def _cosine(m):
    """Half-period sine lobe sampled at bin centres (windows.py:1493-1549)."""
    return np.sin(np.pi / m * (np.arange(0, m) + 0.5))


cosine = _cosine


def tukey(M, alpha=0.5, sym=True):
    """Tapered cosine: flat top with cosine lobes over a fraction alpha (windows.py:1057-1137)."""
    if _len_guards(M):
        return np.ones(int(M))
    if alpha <= 0:
        return np.ones(int(M), "d")
    if alpha >= 1.0:
        return hann(M, sym=sym)
    m, cut = _extend(int(M), sym)
    n = np.arange(0, m)
    edge = int(np.floor(alpha * (m - 1) / 2.0))
    rise = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n[: edge + 1] / alpha / (m - 1))))
    fall = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n[m - edge - 1:] / alpha / (m - 1))))
    w = np.concatenate((rise, np.ones(max(m - 2 * edge - 2, 0)), fall))
    return _truncate(w, cut)


def exponential(M, center=None, tau=1.0, sym=True):
    """exp(-|n - center| / tau); an off-centre window only exists in the DFT-even form (windows.py:1552-1640)."""
    if sym and center is not None:
        raise ValueError("If sym==True, center must be None.")
    if _len_guards(M):
        return np.ones(int(M))
    m, cut = _extend(int(M), sym)
    if center is None:
        center = (m - 1) / 2
    return _truncate(np.exp(-np.abs(np.arange(0, m) - center) / tau), cut)


@_table
def _kaiser(m, beta):
    """I0(beta sqrt(1 - x^2)) / I0(beta), x in [-1, 1] (windows.py:1647-1762)."""
    half = (m - 1) / 2.0
    x = (np.arange(0, m) - half) / half
    return np.i0(beta * np.sqrt(1 - x ** 2.0)) / np.i0(float(beta))


kaiser = _kaiser


@_table
def _chebwin(m, at):
    """Dolph-Chebyshev window with `at` dB sidelobes, via the DFT of the Chebyshev polynomial (windows.py:1765-1894)."""
    order = m - 1.0
    beta = np.cosh(np.arccosh(10 ** (np.abs(at) / 20.0)) / order)
    x = beta * np.cos(np.pi * np.arange(m) / m)
    p = np.empty(m)
    hi, lo, mid = x > 1, x < -1, np.abs(x) <= 1
    p[hi] = np.cosh(order * np.arccosh(x[hi]))
    p[lo] = (2 * (m % 2) - 1) * np.cosh(order * np.arccosh(-x[lo]))
    p[mid] = np.cos(order * np.arccos(x[mid]))
    if m % 2:
        spec = np.real(np.fft.fft(p))
        h = (m + 1) // 2
        w = np.concatenate((spec[h - 1:0:-1], spec[:h]))
    else:
        spec = np.real(np.fft.fft(p * np.exp(1j * np.pi / m * np.arange(m))))
        h = m // 2 + 1
        w = np.concatenate((spec[h - 1:0:-1], spec[1:h]))
    return w / w.max()


chebwin = _chebwin


def dpss(M, NW, Kmax=None, sym=True, norm=None, return_ratios=False):
    """Discrete prolate spheroidal sequences (windows.py:1986-2244).  The reference carries scipy's routine and defines
    it only when scipy imports; here the installed scipy's own routine is called, and its absence raises."""
    from scipy.signal import windows as _sw
    return _sw.dpss(M, NW, Kmax=Kmax, sym=sym, norm=norm, return_ratios=return_ratios)


def slepian(M, width, sym=True):
    """First Slepian sequence from the banded eigenproblem (windows.py:1897-1983); needs scipy.linalg like the reference."""
    from scipy import linalg
    if _len_guards(M):
        return np.ones(int(M))
    m, cut = _extend(int(M), sym)
    q = width / 4.0
    k = np.arange(m, dtype="d")
    band = np.zeros((2, m))
    band[0, 1:] = k[1:] * (m - k[1:]) / 2
    band[1, :] = ((m - 1 - 2 * k) / 2) ** 2 * np.cos(2 * np.pi * q)
    _, vec = linalg.eig_banded(band, select="i", select_range=(m - 1, m - 1))
    vec = vec.ravel()
    return _truncate(vec / vec.max(), cut)


_WINDOW_ALIASES = {
    barthann: ("barthann", "brthan", "bth"),
    bartlett: ("bartlett", "bart", "brt"),
    blackman: ("blackman", "black", "blk"),
    blackmanharris: ("blackmanharris", "blackharr", "bkh"),
    bohman: ("bohman", "bman", "bmn"),
    boxcar: ("boxcar", "box", "ones", "rect", "rectangular"),
    chebwin: ("chebwin", "cheb"),
    cosine: ("cosine", "halfcosine"),
    exponential: ("exponential", "poisson"),
    flattop: ("flattop", "flat", "flt"),
    gaussian: ("gaussian", "gauss", "gss"),
    general_gaussian: ("general gaussian", "general_gaussian", "general gauss", "general_gauss", "ggs"),
    hamming: ("hamming", "hamm", "ham"),
    hann: ("hanning", "hann", "han"),
    kaiser: ("kaiser", "ksr"),
    nuttall: ("nuttall", "nutl", "nut"),
    parzen: ("parzen", "parz", "par"),
    slepian: ("slepian", "slep", "optimal"),
    dpss: ("dpss", "dss"),
    triang: ("triangle", "triang", "tri"),
    tukey: ("tukey", "tuk"),
}
_win_equiv = {name: fn for fn, names in _WINDOW_ALIASES.items() for name in names}
_needs_param = {name for fn in (chebwin, exponential, gaussian, general_gaussian, kaiser, slepian, dpss, tukey)
                for name in _WINDOW_ALIASES[fn]}


def get_window(window, Nx, fftbins=True):
    """Window by name, ('name', param, ...) tuple, or a bare float = Kaiser beta (windows.py:2325-2425).
    fftbins=True gives the DFT-even form."""
    sym = not fftbins
    try:
        beta = float(window)
    except (TypeError, ValueError):
        params = ()
        if isinstance(window, tuple):
            name, params = window[0], tuple(window[1:])
        elif isinstance(window, str):
            if window in _needs_param:
                raise ValueError("The '" + window + "' window needs one or more parameters -- pass a tuple.")
            name = window
        else:
            raise ValueError("%s as window type is not supported." % str(type(window)))
        try:
            fn = _win_equiv[name]
        except KeyError:
            raise ValueError("Unknown window type.")
        return fn(Nx, *params, sym=sym)
    return kaiser(Nx, beta, sym=sym)
