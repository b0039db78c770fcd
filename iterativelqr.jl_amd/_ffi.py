"""ctypes binding of the C-ABI (include/ilqr_hip.h). No torch types cross it."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# ILQR_LIB: an alternative build of the same library (e.g. the -DILQR_PROFILE build of tools/phase_cycles.py); tools only
LIB_PATH = os.environ.get("ILQR_LIB") or os.path.join(LIB_DIR, "libilqr_hip.so")
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

c_double_p = C.POINTER(C.c_double)
ALLREDUCE_SUM_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int32, C.c_void_p)       # ilqr_allreduce_sum_fn


class Options(C.Structure):
    """Options{T} — src/options.jl:1-15 (same field names and defaults)."""
    _fields_ = [
        ("line_search", C.c_int32), ("max_iterations", C.c_int32), ("max_dual_updates", C.c_int32),
        ("min_step_size", C.c_double), ("objective_tolerance", C.c_double),
        ("lagrangian_gradient_tolerance", C.c_double), ("constraint_tolerance", C.c_double),
        ("constraint_norm", C.c_double), ("initial_constraint_penalty", C.c_double),
        ("scaling_penalty", C.c_double), ("max_penalty", C.c_double),
        ("reset_cache", C.c_int32), ("verbose", C.c_int32),
    ]


class ProblemDesc(C.Structure):
    _fields_ = [("model", C.c_char_p), ("model_library", C.c_char_p), ("horizon", C.c_int32),
                ("batch", C.c_int32), ("device", C.c_int32), ("constrained", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("objective", C.c_double), ("gradient_norm", C.c_double), ("max_violation", C.c_double),
                ("step_size", C.c_double), ("iterations", C.c_int32), ("outer_iterations", C.c_int32),
                ("status", C.c_int32), ("potrf_info", C.c_int32), ("rollouts", C.c_int32), ("reserved", C.c_int32)]


class ModelSource(C.Structure):
    """ilqr_model_source: the reference's user-supplied callables as C source (ilqr_compile_model). flags: 0 or MODEL_DENSE_TABLES."""
    _fields_ = [("name", C.c_char_p), ("nx", C.c_int32), ("nu", C.c_int32), ("nw", C.c_int32), ("nc_stage", C.c_int32),
                ("nc_term", C.c_int32), ("ineq_stage", C.c_uint64), ("ineq_term", C.c_uint64), ("source", C.c_char_p),
                ("flags", C.c_int32)]


MODEL_DENSE_TABLES = 1
MAX_STAGE_KINDS = 16
c_int32_p = C.POINTER(C.c_int32)


class StageKinds(C.Structure):
    """ilqr_stage_kinds: the distinct per-step objects of a problem and which one acts at every step."""
    _fields_ = [("horizon", C.c_int32), ("num_parameter", C.c_int32),
                ("n_dynamics", C.c_int32), ("dynamics_nx", c_int32_p), ("dynamics_nu", c_int32_p), ("dynamics_nx_next", c_int32_p),
                ("dynamics_of_step", c_int32_p),
                ("n_costs", C.c_int32), ("cost_nx", c_int32_p), ("cost_nu", c_int32_p), ("cost_of_step", c_int32_p),
                ("n_constraints", C.c_int32), ("constraint_nc", c_int32_p), ("constraint_nx", c_int32_p), ("constraint_nu", c_int32_p),
                ("constraint_ineq", C.POINTER(C.c_uint64)), ("constraint_of_step", c_int32_p),
                ("nx_term", C.c_int32), ("nc_term", C.c_int32), ("ineq_term", C.c_uint64 * 4)]


class StagePlan(C.Structure):
    """ilqr_stage_plan: the one-stage template the kinds are lowered onto."""
    _fields_ = [("nx", C.c_int32), ("nu", C.c_int32), ("nw", C.c_int32), ("nc_stage", C.c_int32), ("nc_term", C.c_int32),
                ("n_selectors", C.c_int32), ("sel_dynamics", C.c_int32), ("sel_cost", C.c_int32), ("sel_constraint", C.c_int32),
                ("constraint_row0", C.c_int32 * MAX_STAGE_KINDS), ("ineq_stage_words", C.c_uint64 * 4)]


def stage_kinds(T, num_parameter, dynamics, dynamics_of_step, costs, cost_of_step, constraints, constraint_of_step,
                nx_term, nc_term, ineq_term):
    """StageKinds from plain lists: dynamics = [(nx, nu, nx_next)], costs = [(nx, nu)], constraints = [(nc, nx, nu, ineq_mask as a Python int of up to 256 bits)]
    (empty = no stage constraints). The returned struct keeps its arrays alive."""
    def arr(vals, typ=C.c_int32):
        return (typ * max(len(vals), 1))(*vals)
    k = StageKinds()
    keep = dict(dnx=arr([d[0] for d in dynamics]), dnu=arr([d[1] for d in dynamics]), dnn=arr([d[2] for d in dynamics]),
                dof=arr(dynamics_of_step), cnx=arr([c[0] for c in costs]), cnu=arr([c[1] for c in costs]), cof=arr(cost_of_step),
                knc=arr([q[0] for q in constraints]), knx=arr([q[1] for q in constraints]), knu=arr([q[2] for q in constraints]),
                kin=arr([(q[3] >> (64 * j)) & (2 ** 64 - 1) for q in constraints for j in range(4)], C.c_uint64),
                kof=arr(constraint_of_step if constraints else []))
    k.horizon, k.num_parameter = T, num_parameter
    k.n_dynamics, k.dynamics_nx, k.dynamics_nu, k.dynamics_nx_next, k.dynamics_of_step = len(dynamics), keep["dnx"], keep["dnu"], keep["dnn"], keep["dof"]
    k.n_costs, k.cost_nx, k.cost_nu, k.cost_of_step = len(costs), keep["cnx"], keep["cnu"], keep["cof"]
    k.n_constraints, k.constraint_nc, k.constraint_nx, k.constraint_nu = len(constraints), keep["knc"], keep["knx"], keep["knu"]
    k.constraint_ineq, k.constraint_of_step = keep["kin"], keep["kof"]
    k.nx_term, k.nc_term = nx_term, nc_term
    for j in range(4):
        k.ineq_term[j] = (ineq_term >> (64 * j)) & (2 ** 64 - 1)
    k._keep = keep
    return k


def plan_stages(kinds):
    """ilqr_plan_stages -> (StagePlan, selectors[T][S] as nested lists, state_dims, action_dims). Needs no device."""
    T = kinds.horizon
    cap = T * (kinds.n_dynamics + kinds.n_costs + kinds.n_constraints)
    plan, sel = StagePlan(), (C.c_double * max(cap, 1))()
    sd, ad = (C.c_int32 * T)(), (C.c_int32 * max(T - 1, 1))()
    check(lib().ilqr_plan_stages(C.byref(kinds), C.byref(plan), sel, cap, sd, ad))
    S = plan.n_selectors
    return plan, [[sel[t * S + j] for j in range(S)] for t in range(T)], list(sd), list(ad)[:T - 1]


# every symbol include/ilqr_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ilqr_last_error": (C.c_char_p, []),
    "ilqr_device_count": (C.c_int, []),
    "ilqr_default_options": (C.c_int, [C.POINTER(Options)]),
    "ilqr_create": (C.c_int, [C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]),
    "ilqr_create_sharded": (C.c_int, [C.POINTER(ProblemDesc), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]),
    "ilqr_destroy": (C.c_int, [C.c_void_p]),
    "ilqr_set_options": (C.c_int, [C.c_void_p, C.POINTER(Options)]),
    "ilqr_get_dims": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int32)] * 7),
    "ilqr_set_parameters": (C.c_int, [C.c_void_p, c_double_p]),
    "ilqr_reset": (C.c_int, [C.c_void_p]),
    "ilqr_initialize_controls": (C.c_int, [C.c_void_p, c_double_p]),
    "ilqr_initialize_states": (C.c_int, [C.c_void_p, c_double_p]),
    "ilqr_initialize_rollout": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "ilqr_initialize_rollout_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ilqr_initialize_rollout_resident": (C.c_int, [C.c_void_p]),
    "ilqr_solve": (C.c_int, [C.c_void_p]),
    "ilqr_synchronize": (C.c_int, [C.c_void_p]),
    "ilqr_solve_shared_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_double_p, C.c_int32, C.POINTER(C.c_int32)]),
    "ilqr_run_stage": (C.c_int, [C.c_void_p, C.c_int32]),
    "ilqr_run_stage_param": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32]),
    "ilqr_get_trajectory": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "ilqr_get_policy": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "ilqr_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "ilqr_buffer_len": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]),
    "ilqr_get_buffer": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p]),
    "ilqr_set_buffer": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p]),
    "ilqr_enable_action_value_buffers": (C.c_int, [C.c_void_p]),
    "ilqr_scalar_slot": (C.c_int, [C.c_char_p]),
    "ilqr_set_kernel_variant": (C.c_int, [C.c_void_p, C.c_int32]),
    "ilqr_resolved_kernel_variant": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "ilqr_set_handover": (C.c_int, [C.c_void_p, C.c_int32]),
    "ilqr_set_handover_live": (C.c_int, [C.c_void_p, C.c_int32]),
    "ilqr_set_handover_mark": (C.c_int, [C.c_void_p, C.c_int32]),
    "ilqr_get_handover_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ilqr_enable_trace": (C.c_int, [C.c_void_p, C.c_int32]),
    "ilqr_get_trace": (C.c_int, [C.c_void_p, c_double_p]),
    "ilqr_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ilqr_timing_reset": (C.c_int, [C.c_void_p]),
    "ilqr_timing_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "ilqr_compile_model": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "ilqr_compile_model_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "ilqr_plan_stages": (C.c_int, [C.POINTER(StageKinds), C.POINTER(StagePlan), c_double_p, C.c_size_t, c_int32_p, c_int32_p]),
    "ilqr_compile_model_stages": (C.c_int, [C.c_char_p, C.POINTER(StageKinds), C.c_char_p, C.POINTER(StagePlan), c_double_p, C.c_size_t,
                                            c_int32_p, c_int32_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "ilqr_set_stage_selectors": (C.c_int, [C.c_void_p, c_double_p, C.c_int32]),
    "ilqr_synthetic_inputs": (C.c_int, [C.c_char_p, C.c_int32, C.c_uint64, C.c_int64, C.c_int32, c_double_p, c_double_p]),
    "ilqr_device_math": (C.c_int, [C.c_char_p, c_double_p, c_double_p, C.c_int32]),
    "ilqr_register_model": (C.c_int, [C.c_void_p]),
    "ilqr_model_count": (C.c_int, []),
    "ilqr_model_name": (C.c_char_p, [C.c_int32]),
    "ilqr_model_compact_sizes": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

STAGES = dict(cost_nominal=0, gradients=1, backward_pass=2, forward_pass=3, reset_model_objective=4,
              ilqr_solve=5, al_update=6, al_begin=7, al_outer=8, ss_inner_begin=9, ss_trial=10, ss_finish=11, ss_outer=12)

_lib = None


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


def device_source_hash():
    """sha256 over the sources lib/builtin_models.o is made from (kernel headers, builtin_models.hip, model headers, the C header; names and
    contents, sorted; ilqr_api.hip holds no kernel the model prices) — the stamp
    tools/issue_model.py puts on lib/issue_model.json at build time (the same function there; tests/test_abi.py compares them)."""
    import hashlib
    root = os.path.dirname(_HERE)
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp") or f == "builtin_models.hip"] +
                   [os.path.join(CSRC, "models", f) for f in os.listdir(os.path.join(CSRC, "models")) if f.endswith(".h")] +
                   [os.path.join(INCLUDE, "ilqr_hip.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def issue_model(config):
    """The per-loop instruction lists of the library's own compilation (lib/issue_model.json, written by csrc/Makefile), or
    (None, reason) when there is none for this model or it was made from other sources than the ones in the tree."""
    import json
    path = os.path.join(os.path.dirname(LIB_PATH), "issue_model.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError) as e:
        return None, "no issue model next to the library (%s)" % e
    stamp = d.get("_build", {}).get("source_hash")
    if stamp != device_source_hash():
        return None, "stale: built from sources %s, the tree is %s — rebuild (make -C iterativelqr.jl_amd/csrc)" % (stamp, device_source_hash())
    if config not in d:
        return None, "no serial-loop model for %r" % config
    return dict(d[config], source_hash=stamp, built_from=d["_build"].get("from")), None


def lib():
    """Load libilqr_hip.so; the product path has no fallback if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP extension %s is not built; run __graft_entry__.build() "
                               "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


class IlqrError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise IlqrError("ilqr error %d: %s" % (rc, lib().ilqr_last_error().decode()))
