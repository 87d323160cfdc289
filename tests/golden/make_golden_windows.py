#!/usr/bin/env python3
"""Golden vectors for the reference's named window generators and get_window (windows.py:301-2425).

TEST INFRASTRUCTURE, build container only (needs /root/reference).  Calls the reference functions unmodified and stores
each table in tests/golden/named_windows.npz under the key "<name>|<params>|<M>|<sym>".
Usage:  python tests/golden/make_golden_windows.py
"""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
from make_golden import _install_shims, _load, save

CASES = [
    ("boxcar", ()), ("triang", ()), ("parzen", ()), ("bohman", ()), ("blackman", ()), ("nuttall", ()),
    ("blackmanharris", ()), ("flattop", ()), ("bartlett", ()), ("hann", ()), ("hamming", ()), ("barthann", ()),
    ("cosine", ()), ("tukey", (0.3,)), ("tukey", (1.0,)), ("gaussian", (2.5,)), ("general_gaussian", (1.5, 3.0)),
    ("general_hamming", (0.6,)), ("general_cosine", ((0.4, 0.3, 0.2, 0.1),)), ("kaiser", (8.6,)), ("chebwin", (80,)),
    ("slepian", (0.3,)), ("exponential", (None, 3.0)),
]
LENGTHS = (0, 1, 2, 7, 8, 33)
GET_WINDOW = ["hann", "tri", "flt", "box", "bkh", ("tukey", 0.3), ("ggs", 1.5, 3.0), ("ksr", 5.0), 4.0, ("poisson", None, 2.0)]


def key(name, params, M, sym):
    return "%s|%r|%d|%d" % (name, params, M, int(sym))


def main():
    _install_shims()
    W = _load("windows")
    d = {}
    with contextlib.redirect_stdout(io.StringIO()):       # slepian / chebwin print notices
        for name, params in CASES:
            for M in LENGTHS:
                for sym in (True, False):
                    d[key(name, params, M, sym)] = getattr(W, name)(M, *params, sym=sym)
        d["dpss|(2.5, 3)|64|1"] = W.dpss(64, 2.5, 3)
        for i, w in enumerate(GET_WINDOW):
            for fb in (True, False):
                d["get_window|%d|%d" % (i, int(fb))] = W.get_window(w, 16, fb)
    save("named_windows", **d)


if __name__ == "__main__":
    main()
