"""CPU-only checks of the product's host side: window tables and geometry against the golden fixtures, the notch
design, that libspectral.so loads and exports every symbol include/spectral.h declares, and that the product
fails loudly (no CPU fallback) when no GPU is present.  No compute calls into the library here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, have_gpu

import pyfft_amd
from pyfft_amd import _ffi
from pyfft_amd import fft_analysis as FA
from pyfft_amd.windows import windows

WNAMES = ["Hanning", "Hamming", "Blackman", "SFT3F", "SFT4F", "SFT5F", "SFT3M", "SFT4M", "SFT5M", "Nuttall3a",
          "Nuttall3b", "Nuttall3", "Nuttall4a", "Nuttall4b", "Nuttall4c", "Nuttall4", "Kaiser", "Welch", "Bartlett",
          "box"]


@pytest.mark.parametrize("name", WNAMES)
def test_product_windows_match_reference(name):
    g = load_golden("windows")
    kw = {"beta": 8.6} if name == "Kaiser" else {}
    for N in (16, 255, 1024):
        np.testing.assert_allclose(windows(name, nwins=N, verbose=False, **kw), g["%s_per_%d" % (name, N)], rtol=1e-13,
                                   atol=1e-15)
        np.testing.assert_allclose(windows(name, nwins=N, periodic=False, verbose=False, **kw),
                                   g["%s_sym_%d" % (name, N)], rtol=1e-13, atol=1e-15)
    assert windows(name, verbose=False, **kw) == float(g["%s_rov" % name])
    val, (label, _) = windows(name, nwins=8, verbose=False, msgout=True, **kw)
    assert len(val) == 8 and isinstance(label, str)


def test_product_geometry_matches_reference():
    g = load_golden("geometry")
    for nsig, Navr, ov, nw, no, na, nyq in g["table"]:
        nwins = FA._nwins(int(nsig), int(Navr), ov)
        assert nwins == int(nw)
        assert FA._noverlap(nwins, ov) == int(no)
        assert FA._navr(int(nsig), nwins, int(no)) == int(na)
        assert FA._nnyquist(nwins) == int(nyq)
        assert FA.fftanal._getNwins(int(nsig), int(Navr), ov) == int(nw)
    np.testing.assert_allclose(np.array(FA._norms(windows("Hanning", nwins=4096, verbose=False), 2048, 1.0)),
                               g["hann4096_norms"], rtol=1e-13)


def test_fftanal_init_geometry_without_gpu():
    g = load_golden("welch_class_c64_2e16_n4096")
    x = g["x"]
    t = np.arange(x.size, dtype=np.float64)
    ft = pyfft_amd.fftanal(t, x, None, tbounds=[t[0], t[-1]], nwins=4096, windowfunction="Hanning", windowoverlap=0.5,
                           verbose=False)
    assert (ft.nwins, ft.noverlap, ft.Navr, ft.Nnyquist) == (4096, 2048, int(g["Navr"]), 2048)
    assert not ft.onesided
    for k in ("S1", "S2", "ENBW", "NENBW", "Fs"):
        np.testing.assert_allclose(getattr(ft, k), g[k], rtol=1e-13)
    # tper path truncates like the reference (Q5)
    ft2 = pyfft_amd.fftanal(t, x, None, tper=4096.0 * (1 - 1e-12), verbose=False)
    assert ft2.nwins == 4095


def test_notch_design_matches_reference():
    g = load_golden("notch")
    b, a = pyfft_amd.iirnotch(60.0 / 100.0, 30.0)
    np.testing.assert_allclose(b, g["b_doc"], rtol=1e-14)
    np.testing.assert_allclose(a, g["a_doc"], rtol=1e-14)
    for row in g["sweep"]:
        bn, an = pyfft_amd.iirnotch(row[0], row[1])
        bp, ap = pyfft_amd.iirpeak(row[0], row[1])
        np.testing.assert_allclose(np.concatenate([bn, an, bp, ap]), row[2:], rtol=1e-13, atol=1e-16)
    with pytest.raises(ValueError):
        pyfft_amd.iirnotch(-0.1, 3.0)
    from pyfft_amd.notch_filter import impulse_response
    import scipy.signal as ss
    h = impulse_response(b, a, 40)
    np.testing.assert_allclose(h, ss.lfilter(b, a, np.r_[1.0, np.zeros(39)]), rtol=1e-11, atol=1e-14)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "spectral.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    assert declared == set(_ffi.SIGNATURES), (declared ^ set(_ffi.SIGNATURES))
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _ffi.load_library().sp_version() >= 100
    assert _ffi.lib().sp_max_wg_fft() == 8192


def test_export_list_mirrors_reference_package():
    for name in ("fft", "fft_pwelch", "fftanal", "windows", "specgram", "stft", "hilbert", "hilbert_1d", "ccf",
                 "iirnotch", "iirpeak", "fftfilt", "apply_notch", "Cxy_Cxy2"):
        assert hasattr(pyfft_amd, name), name
    assert pyfft_amd.fft is pyfft_amd.fft_analysis


@pytest.mark.skipif(have_gpu(), reason="needs a box without a GPU")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(_ffi.SpectralError, match="no HIP device"):
        pyfft_amd.engine.fft(np.zeros(8, dtype=np.complex64))
    with pytest.raises(_ffi.SpectralError):
        pyfft_amd.hilbert(np.zeros(16))
    with pytest.raises(_ffi.SpectralError):
        pyfft_amd.ccf(np.zeros(16), np.ones(16), 1.0)


def test_ntmodel_argument_errors_mirror_reference():
    """nT-model branch (sigx one window long): the reference only runs it with Navr=None and tbounds inside the record
    (tests/golden/pwelch_ntmodel.npz "errors"); the same exception types come back before any device work"""
    t = np.arange(100.0)
    with pytest.raises(ValueError):
        pyfft_amd.fft_pwelch(t, np.zeros(50), np.zeros(100))       # full record: the model would be reflected too
    with pytest.raises(UnboundLocalError):
        pyfft_amd.fft_pwelch(t, np.zeros(50), np.zeros(100), [t[2], t[-2]], Navr=4)
    with pytest.raises(NotImplementedError):
        pyfft_amd.fft_pwelch(t, np.zeros(50), np.zeros(100), [t[2], t[-2]], useMLAB=True)


def test_named_window_catalogue_matches_reference():
    """Named generators + get_window (reference windows.py:301-2425) against tables captured from the reference
    (tests/golden/make_golden_windows.py).  Host table generation only: these feed the device as window tables."""
    import ast
    import importlib
    W = importlib.import_module("pyfft_amd.windows")
    g = load_golden("named_windows")
    get_window_args = ["hann", "tri", "flt", "box", "bkh", ("tukey", 0.3), ("ggs", 1.5, 3.0), ("ksr", 5.0), 4.0,
                       ("poisson", None, 2.0)]
    n = 0
    for k in g.files:
        parts = k.split("|")
        if parts[0] == "get_window":
            got = W.get_window(get_window_args[int(parts[1])], 16, bool(int(parts[2])))
        elif parts[0] == "dpss":
            got = W.dpss(64, 2.5, 3)
        else:
            params = ast.literal_eval(parts[1])
            got = getattr(W, parts[0])(int(parts[2]), *params, sym=bool(int(parts[3])))
        assert got.shape == g[k].shape, k
        np.testing.assert_allclose(got, g[k], rtol=0, atol=2e-15, err_msg=k)
        n += 1
    assert n > 250
    with pytest.raises(ValueError):
        W.hann(-1)
    with pytest.raises(ValueError):
        W.hann(2.5)
    with pytest.raises(ValueError):
        W.get_window("kaiser", 8)           # parametrised window given by bare name
    with pytest.raises(ValueError):
        W.get_window("nope", 8)
    with pytest.raises(ValueError):
        W.exponential(8, center=2, sym=True)
    assert W.hanning is not None and set(W._win_equiv) >= {"hann", "tuk", "optimal", "dss"}


def test_table_cache_policy(tmp_path):
    """the device-table cache's eviction policy (pyfft_amd/csrc/table_cache.h) is host-only logic: built with g++ and run
    here -- LRU, never an entry the current call obtained, never one a pending sp_welch_accum holds"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cache_policy_test")
    # with AddressSanitizer + UndefinedBehaviorSanitizer on the CPU build (sanitizers; GPU ASan is not
    # available on this pool): any out-of-bounds / use-after-free / UB in the policy code fails the run
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", "-I" + os.path.join(root, "pyfft_amd", "csrc"),
                    os.path.join(root, "tests", "cache_policy_test.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cache policy ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_uncertainty_and_band_integration_match_reference():
    """varcoh / varphi / mean_angle / integratespectra (fft_analysis.py:835-937, :1218-1376; the harmonic-band integration
    of HeatPulse_Funcs.py:498-530): fixture produced by the reference's own functions (make_golden_integrate.py).
    trapz_var / reshapech are absent from the reference checkout -> stated stand-ins, PARITY UNPINNED at that boundary."""
    g = load_golden("integrate")
    freq, Pxx, Pyy, Pxy = g["freq"], g["Pxx"], g["Pyy"], g["Pxy"]
    vPxx, vPyy, vPxy = g["vPxx"], g["vPyy"], g["vPxy"]
    nch = Pyy.shape[1]
    ones = np.ones((1, nch))
    for ms in (True, False):
        Coh, vCoh = FA.varcoh(Pxy, vPxy, Pxx[:, None] * ones, vPxx[:, None] * ones, Pyy, vPyy, meansquared=ms)
        np.testing.assert_allclose(Coh, g["coh_ms%d" % ms], rtol=1e-12)
        np.testing.assert_allclose(vCoh, g["vcoh_ms%d" % ms], rtol=1e-12)
    for ar, tag in ((np.pi, "pi"), (0.25 * np.pi, "qpi")):
        ph, vph = FA.varphi(Pxy.real, Pxy.imag, vPxy.real, vPxy.imag, angle_range=ar)
        np.testing.assert_allclose(ph, g["ph_" + tag], rtol=1e-12)
        np.testing.assert_allclose(vph, g["vph_" + tag], rtol=1e-12)
    phi = np.angle(Pxy)
    mph, vmph = FA.mean_angle(phi, vphi=0.01 * np.ones_like(phi), dim=0, angle_range=np.pi, vsyst=0.001 * np.ones_like(phi))
    np.testing.assert_allclose(mph, g["mean_phi"], rtol=1e-12)
    np.testing.assert_allclose(vmph, g["var_mean_phi"], rtol=1e-12)
    Pxy_i, Pxx_i, Pyy_i, Cxy_i, ph_i, info = pyfft_amd.integratespectra(freq, Pxy, Pxx, Pyy, list(g["frange"]), vPxy, vPxx, vPyy)
    for got, key in ((Pxy_i, "Pxy_i"), (Pxx_i, "Pxx_i"), (Pyy_i, "Pyy_i"), (Cxy_i, "Cxy_i"), (ph_i, "ph_i"),
                     (info.varPxy_i, "varPxy_i"), (info.varPxx_i, "varPxx_i"), (info.varPyy_i, "varPyy_i"),
                     (info.varCxy_i, "varCxy_i"), (info.varph_i, "varph_i"), (info.fweighted, "fweighted")):
        np.testing.assert_allclose(np.asarray(got).reshape(g[key].shape), g[key], rtol=1e-12, err_msg=key)
    assert list(info.ifrange) == list(g["ifrange"])
    # defaults: the reference's own default branch cannot run (numpy.size_like does not exist); zeros are used
    r = pyfft_amd.integratespectra(freq, Pxy[:, 0], Pxx, Pyy[:, 0], list(g["frange"]))
    np.testing.assert_allclose(np.asarray(r[0]).ravel()[0], g["Pxy_i"].ravel()[0], rtol=1e-12)


def test_heatpulse_harmonic_search_matches_reference_bookkeeping():
    """pyfft_amd.heatpulse.harmonic_indices (HeatPulse_Funcs.py:412-441) on the reference's own averaged Pxx (chloop.npz)"""
    from pyfft_amd.heatpulse import harmonic_indices
    g = load_golden("chloop")
    for tag in ("n8", "n64"):
        ifk, ifw = harmonic_indices(g["freq_" + tag], g["Pxx_" + tag], float(g["fmod"]), list(g["harms"]), float(g["fwid"]))
        assert list(ifk) == list(g["ifk_" + tag]) and ifw == int(g["ifw_" + tag])
