"""C-ABI checks that need no GPU: the library loads, exports every symbol the header
declares, mirrors the reference's option defaults, and refuses to run without a device."""
import ctypes as C
import os
import re

import numpy as np
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
    # the sharded constructor refuses the same way (and validates its device list first)
    with pytest.raises(pkg._ffi.IlqrError, match="no HIP device"):
        pkg.Solver(model="acrobot", horizon=101, batch=4, devices=[0, 0])
    good = pkg._ffi.ProblemDesc(b"acrobot", None, 11, 2, 0, 1)
    devs = (C.c_int32 * 3)(0, 0, 0)
    assert L.ilqr_create_sharded(C.byref(good), devs, 0, C.byref(h)) < 0 and b"empty device list" in L.ilqr_last_error()
    assert L.ilqr_create_sharded(C.byref(good), devs, 3, C.byref(h)) < 0 and b"more devices than instances" in L.ilqr_last_error()
    assert L.ilqr_create_sharded(C.byref(good), None, 1, C.byref(h)) < 0


class _ModelSource(C.Structure):
    _fields_ = [("name", C.c_char_p), ("nx", C.c_int32), ("nu", C.c_int32), ("nw", C.c_int32), ("nc_stage", C.c_int32),
                ("nc_term", C.c_int32), ("ineq_stage", C.c_uint64), ("ineq_term", C.c_uint64), ("source", C.c_char_p), ("flags", C.c_int32)]


def test_compile_model_from_c_source_without_python_codegen(pkg):
    """ilqr_compile_model: the reference's user-supplied callables (src/dynamics.jl:55-60, src/costs.jl:1-15,
    src/constraints.jl:54-64) as C source -> adapter -> hipcc child process -> registered model (cross-compiles here)."""
    L = pkg._ffi.lib()
    fn = "ILQR_MODEL_FN void %s(double* o, const double* x, const double* u, const double* w) { %s }\n"
    src = "".join(fn % kv for kv in [
        ("dynamics", "o[0] = x[0] + 0.1 * x[1]; o[1] = x[1] + 0.1 * (u[0] - sin(x[0]));"),
        ("dynamics_jacobian_state", "o[0] = 1.0; o[1] = -0.1 * cos(x[0]); o[2] = 0.1; o[3] = 1.0;"),
        ("dynamics_jacobian_action", "o[1] = 0.1;"),
        ("cost_stage", "o[0] = x[0] * x[0] + x[1] * x[1] + 0.1 * u[0] * u[0];"),
        ("cost_stage_gradient_state", "o[0] = 2.0 * x[0]; o[1] = 2.0 * x[1];"),
        ("cost_stage_gradient_action", "o[0] = 0.2 * u[0];"),
        ("cost_stage_hessian_state_state", "o[0] = 2.0; o[3] = 2.0;"),
        ("cost_stage_hessian_action_action", "o[0] = 0.2;"),
        ("cost_stage_hessian_action_state", ""),
        ("cost_terminal", "o[0] = 10.0 * (x[0] * x[0] + x[1] * x[1]);"),
        ("cost_terminal_gradient_state", "o[0] = 20.0 * x[0]; o[1] = 20.0 * x[1];"),
        ("cost_terminal_hessian_state_state", "o[0] = 20.0; o[3] = 20.0;"),
        ("constraint_stage", "o[0] = u[0] - 2.0; o[1] = -2.0 - u[0];"),
        ("constraint_stage_jacobian_state", ""),
        ("constraint_stage_jacobian_action", "o[0] = 1.0; o[1] = -1.0;"),
    ])
    ms = _ModelSource(b"abi_pendulum", 2, 1, 0, 2, 0, 3, 0, src.encode())
    name = C.create_string_buffer(128); path = C.create_string_buffer(1024)
    rc = L.ilqr_compile_model(C.byref(ms), name, 128, path, 1024)
    assert rc == 0, L.ilqr_last_error().decode()
    assert name.value.startswith(b"abi_pendulum_c") and os.path.exists(path.value.decode())
    names = {L.ilqr_model_name(i) for i in range(L.ilqr_model_count())}
    assert name.value in names
    # a second call with the same source is served from the cache and registers under the same name
    name2 = C.create_string_buffer(128)
    assert L.ilqr_compile_model(C.byref(ms), name2, 128, path, 1024) == 0 and name2.value == name.value
    # errors come back as messages, not crashes
    bad = _ModelSource(b"abi_bad", 2, 1, 0, 0, 0, 0, 0, b"this is not C")
    assert L.ilqr_compile_model(C.byref(bad), name2, 128, path, 1024) < 0 and b"hipcc failed" in L.ilqr_last_error()
    big = _ModelSource(b"abi_big", 9, 1, 0, 0, 0, 0, 0, b"")
    assert L.ilqr_compile_model(C.byref(big), name2, 128, path, 1024) < 0


def _rows70_source():
    """pendulum (2, 1) with 70 stage rows: 34 nested action boxes (68 inequality rows), row 68 an EQUALITY that is identically
    zero, row 69 an INEQUALITY that is identically -1 (inactive as an inequality, a violation of 1 if its bit were lost)."""
    fn = "ILQR_MODEL_FN void %s(double* o, const double* x, const double* u, const double* w) { %s }\n"
    return "".join(fn % kv for kv in [
        ("dynamics", "o[0] = x[0] + 0.1 * x[1]; o[1] = x[1] + 0.1 * (u[0] - sin(x[0]));"),
        ("dynamics_jacobian_state", "o[0] = 1.0; o[1] = -0.1 * cos(x[0]); o[2] = 0.1; o[3] = 1.0;"),
        ("dynamics_jacobian_action", "o[1] = 0.1;"),
        ("cost_stage", "o[0] = x[0] * x[0] + x[1] * x[1] + 0.1 * u[0] * u[0];"),
        ("cost_stage_gradient_state", "o[0] = 2.0 * x[0]; o[1] = 2.0 * x[1];"),
        ("cost_stage_gradient_action", "o[0] = 0.2 * u[0];"),
        ("cost_stage_hessian_state_state", "o[0] = 2.0; o[3] = 2.0;"),
        ("cost_stage_hessian_action_action", "o[0] = 0.2;"),
        ("cost_stage_hessian_action_state", ""),
        ("cost_terminal", "o[0] = 10.0 * ((x[0] - 1.0) * (x[0] - 1.0) + x[1] * x[1]);"),
        ("cost_terminal_gradient_state", "o[0] = 20.0 * (x[0] - 1.0); o[1] = 20.0 * x[1];"),
        ("cost_terminal_hessian_state_state", "o[0] = 20.0; o[3] = 20.0;"),
        ("constraint_stage", "for (int i = 0; i < 34; ++i) { o[2 * i] = u[0] - (0.5 + 0.01 * i); o[2 * i + 1] = -(0.5 + 0.01 * i) - u[0]; } o[68] = 0.0; o[69] = -1.0;"),
        ("constraint_stage_jacobian_state", ""),
        ("constraint_stage_jacobian_action", "for (int i = 0; i < 34; ++i) { o[2 * i] = 1.0; o[2 * i + 1] = -1.0; }"),
    ])


ROWS70_WORDS = [(1 << 64) - 1, 0b101111]        # rows 0..67 and 69 inequalities, row 68 an equality


def test_compile_model_with_more_than_64_constraint_rows(pkg):
    """The reference's Constraint has no row limit (src/constraints.jl:54-64). Up to 64 rows the inequality set is a 64-bit mask in
    ilqr_model_source; beyond, ilqr_compile_model_rows takes it as words and the generated wrapper carries INEQ_WORDS /
    INEQ_S_W / INEQ_T_W for ilqr::IneqMask. ilqr_compile_model itself refuses such a model with a message that names the other
    entry point; the Python generator emits the same members."""
    L = pkg._ffi.lib()
    ms = _ModelSource(b"abi_rows70", 2, 1, 0, 70, 0, 0, 0, _rows70_source().encode())
    name = C.create_string_buffer(128); path = C.create_string_buffer(1024)
    assert L.ilqr_compile_model(C.byref(ms), name, 128, path, 1024) < 0 and b"ilqr_compile_model_rows" in L.ilqr_last_error()
    assert L.ilqr_compile_model_rows(C.byref(ms), None, None, name, 128, path, 1024) < 0       # 70 rows and no words
    words = (C.c_uint64 * 2)(*ROWS70_WORDS)
    rc = L.ilqr_compile_model_rows(C.byref(ms), words, None, name, 128, path, 1024)
    assert rc == 0, L.ilqr_last_error().decode()
    assert name.value.startswith(b"abi_rows70_c") and os.path.exists(path.value.decode())
    # other words, other module
    words2 = (C.c_uint64 * 2)((1 << 64) - 1, 0b111111)
    name2 = C.create_string_buffer(128)
    assert L.ilqr_compile_model_rows(C.byref(ms), words2, None, name2, 128, path, 1024) == 0 and name2.value != name.value
    too_many = _ModelSource(b"abi_rows300", 2, 1, 0, 300, 0, 0, 0, b"")
    assert L.ilqr_compile_model_rows(C.byref(too_many), (C.c_uint64 * 5)(), None, name2, 128, path, 1024) < 0
    # the symbolic generator: same members, from indices_inequality
    mdl = pkg.models.synth_box(30, 5)
    _, src = pkg.codegen.generate_model_source("box30", mdl["dynamics"], mdl["cost_stage"], mdl["cost_term"], mdl["con_stage"], mdl["con_term"])
    assert "INEQ_WORDS = 2;" in src and "INEQ_S_W[2] = {0xffffffffffffffffull, 0x3full}" in src
    _, src12 = pkg.codegen.generate_model_source("s12", **pkg.models.synth12())
    assert "INEQ_WORDS" not in src12           # models with <= 64 rows keep their header


def test_issue_model_is_a_product_of_the_build_and_goes_stale_with_the_sources(pkg, tmp_path, monkeypatch):
    """csrc/Makefile writes lib/issue_model.json from the assembly of the library's own compilation, stamped with the hash of
    the device sources; the loader refuses it once the sources differ (bench.py then reports why instead of comparing the
    measurement with another build's instruction lists)."""
    import importlib.util
    im, why = pkg._ffi.issue_model("acrobot")
    assert im is not None, why
    assert im["rollout_step_instr"] > 50 and im["riccati_step_instr"] > 30 and "save-temps" in im["built_from"]
    spec = importlib.util.spec_from_file_location("issue_model_tool", os.path.join(ROOT, "tools", "issue_model.py"))
    tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
    assert tool.device_source_hash() == pkg._ffi.device_source_hash() == im["source_hash"]
    assert pkg._ffi.issue_model("no_such_model")[0] is None
    monkeypatch.setattr(pkg._ffi, "device_source_hash", lambda: "0" * 16)      # as if a kernel header had been edited
    im2, why2 = pkg._ffi.issue_model("acrobot")
    assert im2 is None and "stale" in why2


def test_synthetic_inputs_are_a_pure_function_of_seed_instance_timestep_component(pkg):
    """ilqr_synthetic_inputs (splitmix64 -> U(0,1) -> Box-Muller; no device needed): shards of a larger batch are slices
    of it, the values are standard normals at the reference's scales, the car's instance 0 is test/car.jl:24-29 exactly."""
    W = pkg.workloads
    model, T, x1, ub = W.make_inputs("acrobot", 2048, generator="splitmix64")
    assert not x1.any() and abs(ub.mean()) < 0.01 and abs(ub.std() - 1.0) < 0.01 and np.isfinite(ub).all()
    _, _, x1s, ubs = W.make_inputs("acrobot", 32, offset=1000, generator="splitmix64")
    assert np.array_equal(ubs, ub[1000:1032])
    _, _, _, ub2 = W.make_inputs("acrobot", 8, seed=1, generator="splitmix64")
    assert not np.array_equal(ub2, ub[:8])
    _, _, x1c, ubc = W.make_inputs("car", 64, generator="splitmix64")
    assert not x1c[0].any() and np.array_equal(ubc[0], np.tile([1.0e-2, 1.0e-3], (50, 1)))
    sc = ubc[1:, 0, 0] / 1.0e-2
    assert (sc > 0.5).all() and (sc < 1.5).all() and np.allclose(ubc[1:, :, 1], 0.1 * ubc[1:, :, 0]) and (x1c[1:, 2] == 0).all()
    _, _, x1y, uby = W.make_inputs("synth32", 512, generator="splitmix64")
    assert not uby.any() and abs(x1y.std() - 0.5) < 0.01 and abs(np.corrcoef(x1y[:, 0], x1y[:, 16])[0, 1]) < 0.15
    L = pkg._ffi.lib()
    assert L.ilqr_synthetic_inputs(b"no_such_model", 11, 1, 0, 1, pkg._ffi.c_double_p(), pkg._ffi.c_double_p()) < 0
