"""Host-side mirror of IterativeLQR.jl's user API over the C-ABI.

Names follow src/IterativeLQR.jl:30-45: Dynamics, Cost, Constraint (codegen.py),
Solver, Options, initialize_controls!, initialize_states!, solve!, get_trajectory,
rollout. Julia's `f!` becomes `f_`. One Solver here owns a BATCH of B independent
instances; every array gains a leading batch axis.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _ffi, lowering, codegen
from ._ffi import Options as _COptions


def Options(**kw):
    """Options(; kwargs...) — src/options.jl:1-15. line_search: "armijo" | "none"."""
    o = _COptions()
    _ffi.check(_ffi.lib().ilqr_default_options(C.byref(o)))
    for k, v in kw.items():
        if k == "line_search" and isinstance(v, str):
            v = {"armijo": 1, "none": 0}[v]
        if not hasattr(o, k):
            raise TypeError("Options has no field %r" % k)
        setattr(o, k, v)
    return o


def _p(a):
    return a.ctypes.data_as(_ffi.c_double_p)


MODEL_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-mllvm", "-amdgpu-mfma-vgpr-form",   # MFMA results straight into VGPRs (no AGPR copies)
               "-Wno-unused-parameter"]


def _toolchain_hash():
    """Everything besides the model source that ends up in a model module: the kernel templates it instantiates
    (csrc/*.hpp), the C-ABI header and the compile flags. A cached module built against other headers would be
    launched with a KArgs / Layout it misreads."""
    import hashlib
    h = hashlib.sha256(" ".join(MODEL_FLAGS).encode())
    for d, pat in ((_ffi.CSRC, ".hpp"), (_ffi.INCLUDE, ".h")):
        for f in sorted(os.listdir(d)):
            if f.endswith(pat):
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def compile_model(name, dynamics, cost_stage, cost_term, con_stage=None, con_term=None):
    """Symbolic model objects -> generated device code -> model module (.so).

    Returns (registered_name, path). The registered name carries a hash of the generated source AND of the kernel
    headers / flags, so two different problems traced under the same user name never share a registry entry or a
    cached module, and a header change invalidates every cached module."""
    import hashlib
    sname, src = codegen.generate_model_source(name, dynamics, cost_stage, cost_term, con_stage, con_term)
    tag = hashlib.sha256((src + _toolchain_hash()).encode()).hexdigest()[:16]
    uname = "%s_%s" % (name, tag)
    src = src.replace("struct %s {" % sname, "struct %s_%s {" % (sname, tag), 1)
    src = src.replace('NAME = "%s";' % name, 'NAME = "%s";' % uname, 1)
    cache = os.path.join(_ffi.LIB_DIR, "models")
    os.makedirs(cache, exist_ok=True)
    so = os.path.join(cache, "libilqr_model_%s.so" % uname)
    if not os.path.exists(so):
        hip = os.path.join(cache, "model_%s.hip" % uname)
        with open(hip, "w") as f:
            f.write('#include "ilqr_device.hpp"\n' + src + "ILQR_DEFINE_MODEL(%s_%s)\n" % (sname, tag))
        tmp = so + ".tmp%d" % os.getpid()
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + MODEL_FLAGS + ["-I", _ffi.CSRC, hip, "-o", tmp,
                               "-L", _ffi.LIB_DIR, "-lilqr_hip", "-Wl,-rpath," + _ffi.LIB_DIR])
        os.replace(tmp, so)
    return uname, so


class Solver:
    """Solver(dynamics, costs[, constraints]; options) — src/solver.jl:11-46 — for a batch.

    `dynamics` / `costs` / `constraints` are either lists of codegen objects laid
    out like the reference's (T-1 dynamics, T costs, T constraints; stage objects
    identical, terminal object last) or omitted when `model` names a built-in.
    """

    def __init__(self, dynamics=None, costs=None, constraints=None, *, model=None, horizon=None, batch=1,
                 options=None, device=0, devices=None, name="user", stage_sources=None):
        """devices: list of HIP ordinals — the batch is split into contiguous ranges over them (ilqr_create_sharded: one handle,
        every GPU of the node); device: a single ordinal (ilqr_create).
        stage_sources = (StageKinds, C source): distinct per-step objects given as C callables per kind — the route of a host
        without the symbolic generator (ilqr_compile_model_stages; the Julia wrapper's Solver(dynamics, costs, constraints) with
        differing objects); dynamics / costs / constraints stay None."""
        L = _ffi.lib()
        model_library = None
        self._selectors = None
        if stage_sources is not None:
            kinds, source = stage_sources
            T = kinds.horizon
            cap = T * (kinds.n_dynamics + kinds.n_costs + kinds.n_constraints)
            plan, sel = _ffi.StagePlan(), (C.c_double * max(cap, 1))()
            sd, ad = (C.c_int32 * T)(), (C.c_int32 * max(T - 1, 1))()
            reg, path = C.create_string_buffer(160), C.create_string_buffer(1024)
            _ffi.check(L.ilqr_compile_model_stages(name.encode(), C.byref(kinds), source if isinstance(source, bytes) else source.encode(),
                                                   C.byref(plan), sel, cap, sd, ad, reg, 160, path, 1024))
            S = plan.n_selectors
            self._selectors = np.array([sel[i] for i in range(T * S)], dtype=np.float64).reshape(T, S)
            self.num_user_parameter = kinds.num_parameter
            self.state_dims, self.action_dims = list(sd), list(ad)[:T - 1]
            self.constraint_rows = [[plan.constraint_row0[kinds.constraint_of_step[t]] + i
                                     for i in range(kinds.constraint_nc[kinds.constraint_of_step[t]])] if kinds.n_constraints else []
                                    for t in range(T - 1)]
            model, model_library = reg.value.decode(), path.value.decode()
            horizon = T
            constrained = True if constraints is None else bool(constraints)
        elif model is None:
            T = len(costs)
            low = lowering.lower(dynamics, costs, constraints)       # distinct per-step objects -> one stage template
            self._selectors = low["selectors"]
            self.num_user_parameter = low["num_user_parameter"]
            self.constraint_rows = low["constraint_rows"]
            self.state_dims, self.action_dims = low["state_dims"], low["action_dims"]
            model, model_library = compile_model(name, low["dynamics"], low["cost_stage"], low["cost_term"],
                                                 low["con_stage"], low["con_term"])
            horizon = T
            constrained = constraints is not None
        else:
            constrained = True if constraints is None else bool(constraints)
        self.model, self.T, self.B = model, int(horizon), int(batch)
        desc = _ffi.ProblemDesc(model.encode(), model_library.encode() if model_library else None,
                                self.T, self.B, device, 1 if constrained else 0)
        h = C.c_void_p()
        self.devices = None if devices is None else [int(d) for d in devices]
        if self.devices is None:
            _ffi.check(L.ilqr_create(C.byref(desc), C.byref(h)))
        else:
            arr = (C.c_int32 * len(self.devices))(*self.devices)
            _ffi.check(L.ilqr_create_sharded(C.byref(desc), arr, len(self.devices), C.byref(h)))
        self._h = h
        d = [C.c_int32() for _ in range(7)]
        _ffi.check(L.ilqr_get_dims(self._h, *[C.byref(v) for v in d]))
        self.nx, self.nu, self.nw, self.nc_stage, self.nc_term = [v.value for v in d[:5]]
        self.options = options if options is not None else Options()
        _ffi.check(L.ilqr_set_options(self._h, C.byref(self.options)))
        if self._selectors is None or self._selectors.shape[1] == 0:
            self._selectors, self.num_user_parameter = None, self.nw
        else:      # time-varying stage objects: the handle keeps the selector table and writes it behind the user's parameters in θ_t
            sel = np.ascontiguousarray(self._selectors, dtype=np.float64)
            _ffi.check(L.ilqr_set_stage_selectors(self._h, _p(sel), sel.shape[1]))
            self.nw -= sel.shape[1]             # what ilqr_get_dims reports from here on: the user's parameters per timestep

    # -- src/solver.jl:56-66
    def initialize_controls_(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.B, self.T - 1, self.nu)
        _ffi.check(_ffi.lib().ilqr_initialize_controls(self._h, _p(u)))

    def initialize_states_(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.B, self.T, self.nx)
        _ffi.check(_ffi.lib().ilqr_initialize_states(self._h, _p(x)))

    def initialize_rollout_(self, x1, u):
        """x̄ = rollout(dynamics, x1, ū); initialize_controls!; initialize_states! on the device."""
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(self.B, self.nx)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.B, self.T - 1, self.nu)
        _ffi.check(_ffi.lib().ilqr_initialize_rollout(self._h, _p(x1), _p(u)))

    def initialize_rollout_device_(self, d_x1_ptr, d_u_ptr):
        _ffi.check(_ffi.lib().ilqr_initialize_rollout_device(self._h, C.c_void_p(d_x1_ptr), C.c_void_p(d_u_ptr)))

    def initialize_rollout_resident_(self):
        """initialize_rollout_ again from the inputs the handle keeps in HBM since the last call with host arrays (works on a
        handle that spans several devices, where device pointers of one device make no sense)."""
        _ffi.check(_ffi.lib().ilqr_initialize_rollout_resident(self._h))

    def set_parameters_(self, w):
        """Solver(...; parameters=θ): w[b, t] is the parameter vector of timestep t of instance b."""
        w = np.ascontiguousarray(w, dtype=np.float64).reshape(self.B, self.T, self.num_user_parameter)
        _ffi.check(_ffi.lib().ilqr_set_parameters(self._h, _p(w)))      # (selector columns of a lowered problem are the library's)

    def reset_(self):
        _ffi.check(_ffi.lib().ilqr_reset(self._h))

    # -- src/solve.jl:137-143
    def solve_(self, sync=True, augmented_lagrangian_callback_=None):
        """solve!(solver; augmented_lagrangian_callback!) — src/solve.jl:88-129,137-143.

        Without a callback the whole AL/iLQR solve is ONE kernel launch. With a callback the outer AL
        loop is stepped from the host (one launch per outer iteration) and `callback(solver)` runs after
        each dual update exactly where the reference calls it (src/solve.jl:125), e.g. for continuation
        on the parameters."""
        _ffi.check(_ffi.lib().ilqr_set_options(self._h, C.byref(self.options)))
        verbose = bool(self.options.verbose) and augmented_lagrangian_callback_ is None
        if verbose and not getattr(self, "_trace_cap", 0):     # options.verbose: the per-iteration report of src/solve.jl:39-44
            self.enable_trace_(int(self.options.max_iterations) * max(1, int(self.options.max_dual_updates)))
        if augmented_lagrangian_callback_ is None:
            _ffi.check(_ffi.lib().ilqr_solve(self._h))
            if sync or verbose:
                self.synchronize()
            if verbose:
                self.print_trace()
            return
        self.run_stage_("al_begin")
        for _ in range(int(self.options.max_dual_updates)):
            self.run_stage_("al_outer")
            if bool((self.buffer("_scalars")[:, _ffi.lib().ilqr_scalar_slot(b"done")] != 0.0).all()):   # every instance met the tolerance
                break
            augmented_lagrangian_callback_(self)

    def run_stage_param_(self, stage, param, flag):
        _ffi.check(_ffi.lib().ilqr_set_options(self._h, C.byref(self.options)))
        _ffi.check(_ffi.lib().ilqr_run_stage_param(self._h, _ffi.STAGES[stage], float(param), int(flag)))

    def solve_shared_step_(self, allreduce_sum=None):
        """solve! with ONE step size per inner iteration for the whole batch (all ranks): the optional mode of the north
        star, not a reference behaviour — the reference line-searches every Solver on its own. forward_pass!'s Armijo loop
        (src/forward_pass.jl:26-52) runs on the host over the SUMMED merit: per trial one all-reduce of three doubles
        (Σ J(α), Σ J_prev, Σ ∇Lᵀ·Δz over the instances still in their inner loop) — `allreduce_sum(np.ndarray) -> np.ndarray`,
        e.g. distributed.torch_allreduce_sum(dist, device) for RCCL; identity on one rank. Everything else (linearisation,
        Riccati pass, convergence tests, dual updates) stays per instance on the device. With a batch of one instance it
        reproduces solve_ exactly. Constrained solvers only. Returns the list of accepted step sizes.
        The loop itself lives in the library (ilqr_solve_shared_step: a Julia or C host passes its RCCL / MPI reducer as a C
        callback); this method only wraps `allreduce_sum` into that callback."""
        L = _ffi.lib()
        _ffi.check(L.ilqr_set_options(self._h, C.byref(self.options)))
        cb = None
        if allreduce_sum is not None:
            def reduce(values, n, _ctx):
                try:
                    v = np.ctypeslib.as_array(values, shape=(n,))
                    v[:] = np.asarray(allreduce_sum(v.copy()), dtype=np.float64)
                    return 0
                except Exception:                  # nothing may propagate through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = _ffi.ALLREDUCE_SUM_FN(reduce)
        cap = int(self.options.max_dual_updates) * int(self.options.max_iterations)
        steps = np.zeros(max(cap, 1)); n = C.c_int32(0)
        _ffi.check(L.ilqr_solve_shared_step(self._h, C.cast(cb, C.c_void_p) if cb is not None else None, None, _p(steps), cap, C.byref(n)))
        return [float(a) for a in steps[:n.value]]

    def synchronize(self):
        _ffi.check(_ffi.lib().ilqr_synchronize(self._h))

    def run_stage_(self, stage):
        _ffi.check(_ffi.lib().ilqr_set_options(self._h, C.byref(self.options)))
        _ffi.check(_ffi.lib().ilqr_run_stage(self._h, _ffi.STAGES[stage]))

    # -- src/solver.jl:48-50
    def get_trajectory(self):
        x = np.empty((self.B, self.T, self.nx)); u = np.empty((self.B, self.T - 1, self.nu))
        _ffi.check(_ffi.lib().ilqr_get_trajectory(self._h, _p(x), _p(u)))
        return x, u

    def get_policy(self):
        """(K, k): K[b, t] is the nu×nx gain as a [nx][nu] column-major block."""
        K = np.empty((self.B, self.T - 1, self.nx, self.nu)); k = np.empty((self.B, self.T - 1, self.nu))
        _ffi.check(_ffi.lib().ilqr_get_policy(self._h, _p(K), _p(k)))
        return K, k

    def stats(self):
        st = (_ffi.Stats * self.B)()
        _ffi.check(_ffi.lib().ilqr_get_stats(self._h, st))
        return {f: np.array([getattr(s, f) for s in st]) for f, _ in _ffi.Stats._fields_ if f != "reserved"}

    def enable_action_value_buffers_(self):
        """Have the backward-pass stage also store policy.action_value.* (Qx, Qu, Qxx, Quu, Qux) to HBM."""
        _ffi.check(_ffi.lib().ilqr_enable_action_value_buffers(self._h))

    def scalar(self, name):
        """One named SolverData scalar per instance (see ilqr_scalar_slot), e.g. "delta_grad_product"."""
        i = _ffi.lib().ilqr_scalar_slot(name.encode())
        if i < 0:
            raise KeyError(name)
        return self.buffer("_scalars")[:, i]

    def buffer(self, name):
        n = C.c_size_t(0)
        _ffi.check(_ffi.lib().ilqr_buffer_len(self._h, name.encode(), C.byref(n)))
        out = np.empty((self.B, n.value))
        if n.value:
            _ffi.check(_ffi.lib().ilqr_get_buffer(self._h, name.encode(), _p(out)))
        return out

    def set_buffer(self, name, values):
        n = C.c_size_t(0)
        _ffi.check(_ffi.lib().ilqr_buffer_len(self._h, name.encode(), C.byref(n)))
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(self.B, n.value)
        if n.value:
            _ffi.check(_ffi.lib().ilqr_set_buffer(self._h, name.encode(), _p(v)))

    def set_kernel_variant_(self, variant):
        """"auto" | "latency" | "throughput" | "packed" | "mid" | "packed1" | "packed2" (see ilqr_set_kernel_variant)."""
        v = {"auto": 0, "latency": 1, "throughput": 2, "packed": 3, "mid": 4, "packed1": 5, "packed2": 6}.get(variant, variant)
        _ffi.check(_ffi.lib().ilqr_set_kernel_variant(self._h, int(v)))

    def resolved_kernel_variant(self):
        """The kernel solve_ launches for this handle as it stands: "latency" | "throughput" | "mid" | "packed1" | "packed2"."""
        v = C.c_int32(0)
        _ffi.check(_ffi.lib().ilqr_resolved_kernel_variant(self._h, C.byref(v)))
        return {1: "latency", 2: "throughput", 4: "mid", 5: "packed1", 6: "packed2"}[v.value]

    def set_handover_(self, outer):
        """Straggler hand-over of the packed kernel (see ilqr_set_handover): -1 auto (by head count), 0 off, k >= 2 = instances entering outer iteration k."""
        _ffi.check(_ffi.lib().ilqr_set_handover(self._h, int(outer)))

    def set_handover_live_(self, live):
        """Hand-over by head count (see ilqr_set_handover_live): -1 auto, 0 off, n = survivors of the batch at which they all leave."""
        _ffi.check(_ffi.lib().ilqr_set_handover_live(self._h, int(live)))

    def set_handover_mark_(self, rejected):
        """Early leave of stragglers (see ilqr_set_handover_mark): -1 auto, 0 never, n = rejected line-search trials above the batch's mean."""
        _ffi.check(_ffi.lib().ilqr_set_handover_mark(self._h, int(rejected)))

    def handover_stats(self):
        """(instances the packed kernel's workgroups took from their queue, instances marked as stragglers) of the last solve_."""
        q = C.c_int32(0); m = C.c_int32(0)
        _ffi.check(_ffi.lib().ilqr_get_handover_stats(self._h, C.byref(q), C.byref(m)))
        return q.value, m.value

    def enable_trace_(self, capacity):
        """Record per-iteration rows (what `verbose` prints in the reference) during solve_."""
        self._trace_cap = int(capacity)
        _ffi.check(_ffi.lib().ilqr_enable_trace(self._h, self._trace_cap))

    def trace(self):
        """[B, capacity, 8]: outer, inner, objective, gradient_norm, max_violation, step_size, status, rollouts."""
        out = np.zeros((self.B, self._trace_cap, 8))
        _ffi.check(_ffi.lib().ilqr_get_trace(self._h, _p(out)))
        return out

    def print_trace(self, instance=0):
        """What the reference prints per inner iteration when options.verbose (src/solve.jl:39-44), for one instance."""
        rows = self.trace()[instance][:int(self.scalar("trace_len")[instance])]
        for outer, inner, J, g, v, a, status, _ in rows:
            print("iter:                  %d\n"
                  "             cost:                  %r\n"
                  "\t\t\t gradient_norm:         %r\n"
                  "\t\t\t max_violation:         %r\n"
                  "\t\t\t step_size:             %r" % (int(inner), float(J), float(g), float(v), float(a)))

    def timing(self):
        ms = C.c_double(0); nl = C.c_int32(0)
        _ffi.check(_ffi.lib().ilqr_timing_get(self._h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def timing_reset(self):
        _ffi.check(_ffi.lib().ilqr_timing_reset(self._h))

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().ilqr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# free-function spellings of the reference's exports (src/IterativeLQR.jl:40-45)
def initialize_controls_(solver, u):
    solver.initialize_controls_(u)


def initialize_states_(solver, x):
    solver.initialize_states_(x)


def solve_(solver, *args, augmented_lagrangian_callback_=None):
    """solve!(solver[, states, actions]; augmented_lagrangian_callback!) — src/solve.jl:56-60,88,131-135."""
    if args:
        states, actions = args
        solver.initialize_controls_(actions)
        solver.initialize_states_(states)
    solver.solve_(augmented_lagrangian_callback_=augmented_lagrangian_callback_)


def get_trajectory(solver):
    return solver.get_trajectory()
