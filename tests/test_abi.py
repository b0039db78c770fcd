"""C-ABI checks that need no GPU: the library loads, exports every symbol the header
declares, mirrors the reference's option defaults, and refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

from ilqr_amd_loader import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    g.build()
    return load_package()


def test_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "ilqr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ilqr_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = pkg._ffi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libilqr_hip.so does not export %s" % name
    assert declared == set(pkg._ffi.SYMBOLS), declared ^ set(pkg._ffi.SYMBOLS)


def test_default_options_match_reference(pkg):
    o = pkg.Options()   # src/options.jl:1-15
    assert (o.line_search, o.max_iterations, o.max_dual_updates) == (1, 100, 10)
    assert (o.min_step_size, o.objective_tolerance, o.lagrangian_gradient_tolerance) == (1e-5, 1e-3, 1e-3)
    assert (o.constraint_tolerance, o.initial_constraint_penalty, o.scaling_penalty, o.max_penalty) == (5e-3, 1.0, 10.0, 1e8)
    assert o.constraint_norm == float("inf") and o.reset_cache == 0
    assert pkg.Options(line_search="none").line_search == 0
    with pytest.raises(TypeError):
        pkg.Options(no_such_field=1)


def test_builtin_models_registered(pkg):
    L = pkg._ffi.lib()
    names = {L.ilqr_model_name(i).decode() for i in range(L.ilqr_model_count())}
    assert {"particle", "acrobot", "car", "car_goal", "pendulum_euler"} <= names


def test_no_cpu_fallback(pkg):
    L = pkg._ffi.lib()
    if L.ilqr_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg._ffi.IlqrError, match="no HIP device"):
        pkg.Solver(model="acrobot", horizon=101, batch=4)
    desc = pkg._ffi.ProblemDesc(b"nope", None, 11, 1, 0, 1)
    h = C.c_void_p()
    assert L.ilqr_create(C.byref(desc), C.byref(h)) < 0
    assert b"unknown model" in L.ilqr_last_error()
    assert L.ilqr_solve(None) < 0 and L.ilqr_get_stats(None, None) < 0
