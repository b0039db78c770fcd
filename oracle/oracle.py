"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)


class OrcOptions(C.Structure):
    # src/options.jl:1-15
    _fields_ = [
        ("line_search", C.c_int), ("max_iterations", C.c_int), ("max_dual_updates", C.c_int),
        ("min_step_size", C.c_double), ("objective_tolerance", C.c_double),
        ("lagrangian_gradient_tolerance", C.c_double), ("constraint_tolerance", C.c_double),
        ("constraint_norm", C.c_double), ("initial_constraint_penalty", C.c_double),
        ("scaling_penalty", C.c_double), ("max_penalty", C.c_double),
        ("reset_cache", C.c_int), ("verbose", C.c_int),
    ]


class OrcStats(C.Structure):
    _fields_ = [
        ("objective", C.c_double), ("gradient_norm", C.c_double), ("max_violation", C.c_double),
        ("step_size", C.c_double), ("iterations", C.c_int), ("outer_iterations", C.c_int),
        ("status", C.c_int), ("potrf_info", C.c_int), ("rollouts", C.c_int),
    ]


class OrcTrace(C.Structure):
    _fields_ = [
        ("outer", C.c_int), ("inner", C.c_int), ("objective", C.c_double),
        ("gradient_norm", C.c_double), ("max_violation", C.c_double), ("step_size", C.c_double),
        ("status", C.c_int),
    ]


class OrcProblem(C.Structure):
    _fields_ = [
        ("T", C.c_int), ("nx", C.c_int), ("nu", C.c_int), ("nw", C.c_int),
        ("dynamics", C.c_void_p), ("costs", C.c_void_p), ("constraints", C.c_void_p),
        ("owner", C.c_void_p),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ilqr_oracle.cpp", "models.cpp", "ilqr_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_default_options.argtypes = [C.POINTER(OrcOptions)]
        L.orc_problem_builtin.argtypes = [C.c_char_p, C.c_int, C.POINTER(OrcProblem)]
        L.orc_problem_builtin.restype = C.c_int
        L.orc_problem_free.argtypes = [C.POINTER(OrcProblem)]
        L.orc_problem_dims.argtypes = [C.POINTER(OrcProblem), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_problem_dims.restype = None
        L.orc_solver_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, c_double_p, C.POINTER(OrcOptions)]
        L.orc_solver_create.restype = C.c_void_p
        L.orc_solver_destroy.argtypes = [C.c_void_p]
        L.orc_initialize_controls.argtypes = [C.c_void_p, c_double_p]
        L.orc_initialize_states.argtypes = [C.c_void_p, c_double_p]
        L.orc_rollout.argtypes = [C.c_int, C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]
        for f in ("orc_solve", "orc_gradients", "orc_backward_pass", "orc_forward_pass", "orc_lagrangian_gradient",
                  "orc_reset_model_objective", "orc_ilqr_solve", "orc_augmented_lagrangian_update"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = None
        L.orc_cost_bang.argtypes = [C.c_void_p, C.c_int]
        L.orc_cost_bang.restype = C.c_double
        L.orc_rollout_bang.argtypes = [C.c_void_p, C.c_double]
        L.orc_buffer.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.orc_buffer.restype = c_double_p
        L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(OrcStats)]
        L.orc_set_trace.argtypes = [C.c_void_p, C.POINTER(OrcTrace), C.c_int]
        L.orc_trace_len.argtypes = [C.c_void_p]
        L.orc_trace_len.restype = C.c_int
        L.orc_last_delta.argtypes = [C.c_void_p]
        L.orc_last_delta.restype = C.c_double
        L.orc_set_active_set.argtypes = [C.c_void_p, c_double_p]
        L.orc_set_active_set.restype = None
        L.orc_set_scalars.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int]
        L.orc_set_scalars.restype = None
        L.orc_potrf_U.argtypes = [c_double_p, C.c_int]
        L.orc_potrf_U.restype = C.c_int
        L.orc_potrs_U.argtypes = [c_double_p, C.c_int, c_double_p, C.c_int]
        L.orc_potrs_U.restype = None
        L.orc_solve_batch.argtypes = [C.c_char_p, C.c_int, C.c_int, c_double_p, c_double_p, C.POINTER(OrcOptions),
                                      C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(OrcStats)]
        L.orc_solve_batch.restype = C.c_int
        L.orc_solve_batch_w.argtypes = [C.c_char_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                        C.POINTER(OrcOptions), C.c_int, c_double_p, c_double_p, c_double_p,
                                        c_double_p, C.POINTER(OrcStats)]
        L.orc_solve_batch_w.restype = C.c_int
        _LIB = L
    return _LIB


def default_options(**kw):
    o = OrcOptions()
    lib().orc_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _p(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


class Problem:
    """Built-in oracle problem (model zoo of oracle/models.cpp)."""

    def __init__(self, name, T):
        self.name, self.T = name, T
        self.c = OrcProblem()
        if lib().orc_problem_builtin(name.encode(), T, C.byref(self.c)) != 0:
            raise ValueError("unknown oracle model %r" % name)
        self.nx, self.nu = self.c.nx, self.c.nu          # the LARGEST dimensions of the horizon
        nxt = (C.c_int * T)(); nut = (C.c_int * max(T - 1, 1))()
        lib().orc_problem_dims(C.byref(self.c), nxt, nut)
        self.state_dims, self.action_dims = list(nxt), list(nut)[:T - 1]
        self.uniform = set(self.state_dims) == {self.nx} and set(self.action_dims) == {self.nu}

    # Problems whose dimensions vary along the horizon: the solver's buffers are ragged concatenations of per-timestep blocks
    # (as the reference's Vectors of Vectors are); the helpers below convert to and from arrays padded to (nx, nu).
    def pack_states(self, x):
        x = np.asarray(x, dtype=np.float64).reshape(self.T, self.nx)
        return np.ascontiguousarray(np.concatenate([x[t, :n] for t, n in enumerate(self.state_dims)]))

    def pack_actions(self, u):
        u = np.asarray(u, dtype=np.float64).reshape(self.T - 1, self.nu)
        return np.ascontiguousarray(np.concatenate([u[t, :m] for t, m in enumerate(self.action_dims)]))

    def unpack(self, flat, rows, cols=None):
        """Ragged per-timestep blocks -> zero-padded array. rows / cols: "x" | "u" | "x+" (next state) | None, column-major blocks;
        result [t][col][row] padded (what the device's getters return), or [t][row] for vectors."""
        flat = np.asarray(flat)
        dim = {"x": (self.state_dims, self.nx), "u": (self.action_dims, self.nu), "x+": (self.state_dims[1:], self.nx)}
        rd, rmax = dim[rows]
        steps, o = 0, 0          # as many leading timesteps as the buffer holds (Qx, Qxx stop one short of gx, gxx)
        while steps < len(rd) and (cols is None or steps < len(dim[cols][0])):
            o += rd[steps] * (1 if cols is None else dim[cols][0][steps])
            if o > flat.size:
                break
            steps += 1
        if cols is None:
            out = np.zeros((steps, rmax)); o = 0
            for t in range(steps):
                out[t, :rd[t]] = flat[o:o + rd[t]]; o += rd[t]
            return out
        cd, cmax = dim[cols]
        out = np.zeros((steps, cmax, rmax)); o = 0
        for t in range(steps):
            out[t, :cd[t], :rd[t]] = flat[o:o + rd[t] * cd[t]].reshape(cd[t], rd[t]); o += rd[t] * cd[t]
        return out

    def rollout(self, x1, u, w=None):
        """rollout(dynamics, x1, ū[, parameters]) — src/rollout.jl:33-42. Padded arrays in and out."""
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64) if w is not None else None
        if self.uniform:
            x = np.zeros((self.T, self.nx))
            lib().orc_rollout(self.T, self.c.dynamics, _p(x1), _p(u), _p(w), _p(x))
            return x
        ur = self.pack_actions(u)
        x = np.zeros(sum(self.state_dims))
        lib().orc_rollout(self.T, self.c.dynamics, _p(x1), _p(ur), _p(w), _p(x))
        return self.unpack(x, "x")

    def __del__(self):
        try:
            lib().orc_problem_free(C.byref(self.c))
        except Exception:
            pass


class Solver:
    """Solver(dynamics, costs, constraints) of the oracle — src/solver.jl:28-46."""

    def __init__(self, problem, options=None, w=None):
        self.problem = problem
        self.opt = options if options is not None else default_options()
        w = np.ascontiguousarray(w, dtype=np.float64) if w is not None else None
        self.h = lib().orc_solver_create(problem.T, problem.c.dynamics, problem.c.costs, problem.c.constraints,
                                         _p(w), C.byref(self.opt))
        if not self.h:
            raise RuntimeError("orc_solver_create failed")
        self._trace = None

    def initialize_controls(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64) if self.problem.uniform else self.problem.pack_actions(u)
        lib().orc_initialize_controls(self.h, _p(u))

    def initialize_states(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64) if self.problem.uniform else self.problem.pack_states(x)
        lib().orc_initialize_states(self.h, _p(x))

    def enable_trace(self, capacity=4096):
        self._trace = (OrcTrace * capacity)()
        lib().orc_set_trace(self.h, self._trace, capacity)

    def trace(self):
        n = lib().orc_trace_len(self.h)
        return [self._trace[i] for i in range(n)]

    def solve(self):
        lib().orc_solve(self.h)

    def call(self, name, *args):
        return getattr(lib(), "orc_" + name)(self.h, *args)

    def buffer(self, name):
        n = C.c_int(0)
        p = lib().orc_buffer(self.h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def set_buffer(self, name, values):
        if name == "active_set":       # Vector{Int} in the reference: its own setter (orc_buffer hands out a copy)
            v = np.ascontiguousarray(values, dtype=np.float64).ravel()
            lib().orc_set_active_set(self.h, _p(v))
            return
        n = C.c_int(0)
        p = lib().orc_buffer(self.h, name.encode(), C.byref(n))
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        assert v.size == n.value, (name, v.size, n.value)
        np.ctypeslib.as_array(p, shape=(n.value,))[:] = v

    def stats(self):
        st = OrcStats()
        lib().orc_get_stats(self.h, C.byref(st))
        return st

    def get_trajectory(self):
        T, n, m = self.problem.T, self.problem.nx, self.problem.nu
        if not self.problem.uniform:
            return (self.problem.unpack(self.buffer("nominal_states"), "x"), self.problem.unpack(self.buffer("nominal_actions"), "u"))
        return (self.buffer("nominal_states").reshape(T, n), self.buffer("nominal_actions").reshape(T - 1, m))

    # reference field name -> (rows, cols) of its per-timestep blocks, for Problem.unpack
    BLOCKS = {"nominal_states": ("x", None), "states": ("x", None), "nominal_actions": ("u", None), "actions": ("u", None),
              "jacobian_state": ("x+", "x"), "jacobian_action": ("x+", "u"), "gradient_state": ("x", None),
              "gradient_action": ("u", None), "hessian_state_state": ("x", "x"), "hessian_action_action": ("u", "u"),
              "hessian_action_state": ("u", "x"), "K": ("u", "x"), "k": ("u", None), "P": ("x", "x"), "p": ("x", None),
              "Qx": ("x", None), "Qu": ("u", None), "Qxx": ("x", "x"), "Quu": ("u", "u"), "Qux": ("u", "x")}

    def padded(self, name):
        """A workspace buffer as the zero-padded array the device path keeps: [t][col][row] (or [t][row])."""
        rows, cols = self.BLOCKS[name]
        return self.problem.unpack(self.buffer(name), rows, cols)

    def __del__(self):
        try:
            if self.h:
                lib().orc_solver_destroy(self.h)
                self.h = None
        except Exception:
            pass


def solve_batch(model, T, x1, ubar, options=None, nthreads=1, want_policy=True, w=None):
    """CPU baseline / batch oracle: returns dict(x,u,K,k,stats)."""
    pr = Problem(model, T)
    n, m = pr.nx, pr.nu
    x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, n)
    B = x1.shape[0]
    ubar = np.ascontiguousarray(ubar, dtype=np.float64).reshape(B, T - 1, m)
    opt = options if options is not None else default_options()
    x = np.zeros((B, T, n)); u = np.zeros((B, T - 1, m))
    K = np.zeros((B, T - 1, n, m)) if want_policy else None   # column-major m×n per step → [n][m]
    k = np.zeros((B, T - 1, m)) if want_policy else None
    st = (OrcStats * B)()
    if w is not None:
        w = np.ascontiguousarray(w, dtype=np.float64).reshape(B, T, pr.c.nw)
    rc = lib().orc_solve_batch_w(model.encode(), T, B, _p(x1), _p(ubar), _p(w), C.byref(opt), nthreads,
                                 _p(x), _p(u), _p(K), _p(k), st)
    if rc != 0:
        raise RuntimeError("orc_solve_batch rc=%d" % rc)
    stats = {f: np.array([getattr(s, f) for s in st]) for f, _ in OrcStats._fields_}
    return dict(x=x, u=u, K=K, k=k, stats=stats)
