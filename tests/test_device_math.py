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
                   ' for (int i = 0; i < n; ++i) ilqr::sincos_fast(x[i], s[i], c[i]); }\n'
                   # the pair-of-lanes form of the cooperative rollout code: even lane = sine kernel, odd lane = cosine kernel,
                   # the quadrant fix-up takes the partner's value (the device exchanges them with one DPP quad_perm)
                   'extern "C" void sc_pair(const double* x, double* s, double* c, int n) {'
                   ' const ilqr::TrigPair te = ilqr::trig_pair_constants(false), to = ilqr::trig_pair_constants(true);'
                   ' for (int i = 0; i < n; ++i) { int qe, qo; const double oe = ilqr::trig_pair_own(x[i], te, qe), oo = ilqr::trig_pair_own(x[i], to, qo);'
                   ' s[i] = ilqr::trig_pair_fix(oe, oo, qe, te); c[i] = ilqr::trig_pair_fix(oo, oe, qo, to); } }\n')
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


def test_sincos_pair_is_bitwise_sincos_fast(lib):
    """ilqr::sincos_pair (one Horner chain per lane, per-lane coefficients) returns the values of sincos_fast bit for bit
    (up to the sign of a zero): the cooperative rollout code and the per-lane model code agree exactly."""
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-10, 10, 20000), rng.uniform(-1e6, 1e6, 5000), np.arange(-64, 65) * np.pi / 4,
                         rng.uniform(-1e-6, 1e-6, 500), [0.0, -0.0, 1e300, 5e9, 2.0 ** 30, np.inf, np.nan]])
    xs = np.ascontiguousarray(xs)
    s, c = _run(lib, xs)
    s2 = np.zeros_like(xs); c2 = np.zeros_like(xs)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    lib.sc_pair(p(xs), p(s2), p(c2), len(xs))
    assert np.array_equal(s, s2, equal_nan=True) and np.array_equal(c, c2, equal_nan=True)
