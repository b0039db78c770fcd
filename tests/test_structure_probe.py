"""ilqr_compile_model on a large model: the constant / zero entries of the callables' Jacobians and Hessians are found by
running the callables on the HOST (a second compilation of the C source with the host compiler) — the role Symbolics' sparse
expressions play for the reference (/root/reference/src/dynamics.jl:16-34, src/costs.jl:17-44). hipcc cross-compiles the model
module without a GPU, so the tables can be checked here; the solve against the oracle is tests/test_gpu_parity.py."""
import ctypes as C
import os

import numpy as np

from ilqr_amd_loader import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Src(C.Structure):
    _fields_ = [("name", C.c_char_p), ("nx", C.c_int32), ("nu", C.c_int32), ("nw", C.c_int32), ("nc_stage", C.c_int32),
                ("nc_term", C.c_int32), ("ineq_stage", C.c_uint64), ("ineq_term", C.c_uint64), ("source", C.c_char_p), ("flags", C.c_int32)]


def _compile(L, name, text, dims, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        ms = Src(name, *dims, text)
        reg = C.create_string_buffer(128); path = C.create_string_buffer(1024)
        rc = L.ilqr_compile_model(C.byref(ms), reg, 128, path, 1024)
        assert rc == 0, L.ilqr_last_error().decode()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    jv, hs = C.c_int32(-1), C.c_int32(-1)
    assert L.ilqr_model_compact_sizes(reg.value, C.byref(jv), C.byref(hs)) == 0
    return reg.value.decode(), jv.value, hs.value


def test_constant_and_zero_entries_of_c_callables_are_found_by_probing():
    pkg = load_package()
    L = pkg._ffi.lib()
    text = open(os.path.join(ROOT, "examples", "synth12_model.c"), "rb").read()
    dims = (12, 5, 0, 10, 3, (1 << 10) - 1, 0)
    name, jv, hs = _compile(L, b"synth12_probe", text, dims)
    # synth12: fx = I + h (A + diag(0.1 cos x_i + 0.02 u_{i mod 5})) -> 12 state-dependent entries (the diagonal);
    # fu = h (B + 0.02 x_i at (i, i mod 5)) -> 12 more. Hessians: gxx diagonal (cost) + the terminal goal rows' Gauss-Newton terms
    # (x_1..x_3: diagonal already there), guu diagonal (cost and the action box), gux nothing
    assert jv == 24, jv
    assert hs == 12 + 5, hs
    # without the probe every entry is listed
    name_d, jv_d, hs_d = _compile(L, b"synth12_probe", text, dims, env={"ILQR_NO_STRUCTURE_PROBE": "1"})
    assert name_d != name and jv_d == 12 * 12 + 12 * 5 and hs_d == 12 * 12 + 5 * 5 + 5 * 12
    # a source that is not host C++ (a device-only intrinsic) falls back to the dense tables instead of failing
    bad = text.replace(b"acc += 0.1 * sin(x[i]);", b"acc += 0.1 * __builtin_amdgcn_sin(x[i] * 0.15915494309189535) * 1.0;") \
        if False else text + b"\n/* device-only */ static ILQR_MODEL_FN double only_on_device(double v) { return __builtin_amdgcn_rcp(v); }\n"
    name_b, jv_b, hs_b = _compile(L, b"synth12_probe_dev", bad, dims)
    assert jv_b == 12 * 12 + 12 * 5 and hs_b == 12 * 12 + 5 * 5 + 5 * 12


def test_the_c_example_is_the_generators_output():
    """examples/synth32_model.c is models.synth_c_source(32, 8) (literal coefficient tables, dense Jacobian loops): the file a C or
    Julia host reads and the text the size-limit GPU test generates for 64 x 16 cannot drift apart."""
    pkg = load_package()
    assert open(os.path.join(ROOT, "examples", "synth32_model.c")).read() == pkg.models.synth_c_source(32, 8)
