"""ilqr::sincos_fast (csrc/ilqr_math.hpp) compiled for the host with FMA, against a
200-bit reference: the device code is the same source."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("math")
    src = d / "mt.cpp"
    src.write_text('#include "ilqr_math.hpp"\nextern "C" void sc(const double* x, double* s, double* c, int n) {'
                   ' for (int i = 0; i < n; ++i) ilqr::sincos_fast(x[i], s[i], c[i]); }\n')
    so = d / "mt.so"
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC",
                           "-I", os.path.join(ROOT, "iterativelqr.jl_amd", "csrc"), str(src), "-o", str(so)])
    return ctypes.CDLL(str(so))


def _run(lib, xs):
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    s = np.zeros_like(xs); c = np.zeros_like(xs)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    lib.sc(p(xs), p(s), p(c), len(xs))
    return s, c


def test_sincos_fast_accuracy(lib):
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-10, 10, 4000), rng.uniform(-1e3, 1e3, 1000), rng.uniform(-9e4, 9e4, 500),
                         np.arange(-40, 41) * np.pi / 2, np.arange(-40, 41) * np.pi / 4, rng.uniform(-1e-3, 1e-3, 300)])
    s, c = _run(lib, xs)
    worst = 0.0
    for x, si, ci in zip(xs, s, c):
        for val, ref in ((si, mp.sin(mp.mpf(x))), (ci, mp.cos(mp.mpf(x)))):
            u = np.spacing(abs(float(ref))) or 5e-324
            worst = max(worst, abs(float((mp.mpf(val) - ref) / mp.mpf(u))))
    assert worst < 1.6, worst      # ulp
    xs = rng.uniform(-1e8, 1e8, 2000)
    s, c = _run(lib, xs)
    err = max(abs(float(mp.sin(mp.mpf(x)) - mp.mpf(si))) for x, si in zip(xs, s))
    assert err < 3e-16             # absolute accuracy holds far beyond the 1-ulp range


def test_sincos_fast_edge_cases(lib):
    s, c = _run(lib, [0.0, -0.0, np.inf, -np.inf, np.nan, 1e300, 2.0 ** 30, 5e9])
    assert s[0] == 0.0 and c[0] == 1.0 and s[1] == 0.0
    assert np.isnan(s[2:5]).all() and np.isnan(c[2:5]).all()
    assert np.isfinite(s[5:]).all() and (np.abs(s[5:]) <= 1).all() and (np.abs(c[5:]) <= 1).all()
    assert np.allclose(s[5:] ** 2 + c[5:] ** 2, 1.0, atol=1e-15)
