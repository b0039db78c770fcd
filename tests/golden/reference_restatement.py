"""SECOND, INDEPENDENT restatement of the reference's hot path (TEST INFRASTRUCTURE — fixture generator).

Written from the Julia sources of thowell/IterativeLQR.jl v0.2.3 alone (paths below are relative to
/root/reference); it imports neither `oracle/` nor the product package, and shares no code with either:

    derivatives   sympy (the role Symbolics.jl plays: src/dynamics.jl:16-34, src/costs.jl:17-44,
                  src/constraints.jl:17-43) — exact symbolic Jacobians / gradients / Hessians
    mul!          numpy matmul (OpenBLAS dgemm/dgemv — what Julia's LinearAlgebra.mul! dispatches to)
    potrf!/potrs! scipy.linalg.lapack.dpotrf / dpotrs — the REAL LAPACK the reference calls at
                  src/backward_pass.jl:69-73 (return code ignored there, and here)

The reference itself cannot run in this image (no Julia), so these are not reference OUTPUTS; they are what
a second, differently-built reading of the same sources computes. The C++ oracle (CPU tests) and the HIP
path (GPU tests) must both agree with the fixtures this module generates (tests/golden/*.npz,
tests/golden/make_reference_fixtures.py).

Arrays are numpy row-major [row][col]; Julia's column-major buffers are the transposes (the tests convert).
"""
import math

import numpy as np
import sympy as sp
from scipy.linalg import lapack


# --------------------------------------------------------------------------------------------------
# src/dynamics.jl:1-34, src/costs.jl:1-44, src/constraints.jl:1-52 — symbolic constructors
# --------------------------------------------------------------------------------------------------
def _symbols(num_state, num_action, num_parameter):
    x = [sp.Symbol("x%d" % i, real=True) for i in range(num_state)]
    u = [sp.Symbol("u%d" % i, real=True) for i in range(num_action)]
    w = [sp.Symbol("w%d" % i, real=True) for i in range(num_parameter)]
    return x, u, w


def _build(exprs, x, u, w, shape):
    """eval(Symbolics.build_function(expr, x, u, w)[2]) — a callable (x, u, w) -> ndarray of `shape`."""
    flat = [sp.sympify(e) for e in exprs]
    size = int(np.prod(shape)) if len(shape) else 1
    assert len(flat) == size
    if size == 0:
        return lambda xv, uv, wv: np.zeros(shape)
    f = sp.lambdify([x + u + w], flat, modules="math", cse=True)

    def call(xv, uv, wv):
        args = list(xv) + list(uv) + list(wv)
        return np.array(f(args), dtype=np.float64).reshape(shape)
    return call


class Dynamics:
    """Dynamics(f, num_state, num_action; num_parameter) — src/dynamics.jl:16-34"""

    def __init__(self, f, num_state, num_action, num_parameter=0):
        x, u, w = _symbols(num_state, num_action, num_parameter)
        y = f(x, u, w) if num_parameter > 0 else f(x, u)                       # :23
        y = [sp.sympify(e) for e in y]
        self.num_next_state, self.num_state, self.num_action, self.num_parameter = len(y), num_state, num_action, num_parameter
        jx = [sp.diff(yi, xj) for yi in y for xj in x]                         # Symbolics.jacobian(y, x)  :24
        ju = [sp.diff(yi, uj) for yi in y for uj in u]                         # :25
        self.evaluate = _build(y, x, u, w, (len(y),))
        self.jacobian_state = _build(jx, x, u, w, (len(y), num_state))
        self.jacobian_action = _build(ju, x, u, w, (len(y), num_action))


class Cost:
    """Cost(f, num_state, num_action; num_parameter) — src/costs.jl:17-44"""

    def __init__(self, f, num_state, num_action, num_parameter=0):
        x, u, w = _symbols(num_state, num_action, num_parameter)
        ev = sp.sympify(f(x, u, w) if num_parameter > 0 else f(x, u))          # :25
        gx = [sp.diff(ev, xi) for xi in x]                                     # :26
        gu = [sp.diff(ev, ui) for ui in u]                                     # :27
        self.num_state, self.num_action = num_state, num_action
        self.evaluate = _build([ev], x, u, w, ())
        self.gradient_state = _build(gx, x, u, w, (num_state,))
        self.gradient_action = _build(gu, x, u, w, (num_action,))
        self.hessian_state_state = _build([sp.diff(g, xj) for g in gx for xj in x], x, u, w, (num_state, num_state))      # :28
        self.hessian_action_action = _build([sp.diff(g, uj) for g in gu for uj in u], x, u, w, (num_action, num_action))  # :29
        self.hessian_action_state = _build([sp.diff(g, xj) for g in gu for xj in x], x, u, w, (num_action, num_state))    # :30


class Constraint:
    """Constraint(f, num_state, num_action; indices_inequality, num_parameter) — src/constraints.jl:17-43;
    Constraint() (no arguments) — src/constraints.jl:45-52. indices_inequality is 1-based as in Julia."""

    def __init__(self, f=None, num_state=0, num_action=0, indices_inequality=(), num_parameter=0):
        self.indices_inequality = [i - 1 for i in indices_inequality]
        self.num_state, self.num_action = num_state, num_action
        if f is None:
            self.num_constraint = 0
            return
        x, u, w = _symbols(num_state, num_action, num_parameter)
        ev = [sp.sympify(e) for e in (f(x, u, w) if num_parameter > 0 else f(x, u))]
        self.num_constraint = len(ev)                                          # :35
        self.evaluate = _build(ev, x, u, w, (len(ev),))
        self.jacobian_state = _build([sp.diff(c, xj) for c in ev for xj in x], x, u, w, (len(ev), num_state))
        self.jacobian_action = _build([sp.diff(c, uj) for c in ev for uj in u], x, u, w, (len(ev), num_action))


class Options:
    """src/options.jl:1-15"""

    def __init__(self, **kw):
        self.line_search = "armijo"
        self.max_iterations = 100
        self.max_dual_updates = 10
        self.min_step_size = 1.0e-5
        self.objective_tolerance = 1.0e-3
        self.lagrangian_gradient_tolerance = 1.0e-3
        self.constraint_tolerance = 5.0e-3
        self.constraint_norm = math.inf
        self.initial_constraint_penalty = 1.0
        self.scaling_penalty = 10.0
        self.max_penalty = 1.0e8
        self.reset_cache = False
        self.verbose = False
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def rollout(dynamics, initial_state, actions, parameters=None):
    """rollout(dynamics, x1, ū) — src/rollout.jl:33-42"""
    if parameters is None:
        parameters = [np.zeros(d.num_parameter) for d in dynamics]
    x_history = [np.array(initial_state, dtype=np.float64)]
    for t, d in enumerate(dynamics):
        x_history.append(d.evaluate(x_history[-1], actions[t], parameters[t]).copy())
    return x_history


# --------------------------------------------------------------------------------------------------
# Solver: src/solver.jl:11-46 with all the workspaces of src/data/*.jl and src/augmented_lagrangian.jl:13-37
# --------------------------------------------------------------------------------------------------
class Solver:
    def __init__(self, dynamics, costs, constraints=None, parameters=None, options=None):
        H = len(dynamics) + 1
        assert len(costs) == H
        self.dynamics, self.costs, self.constraints = dynamics, costs, constraints
        self.options = options if options is not None else Options()
        self.H = H
        nxs = [d.num_state for d in dynamics] + [dynamics[-1].num_next_state]
        nus = [d.num_action for d in dynamics]
        self.nxs, self.nus = nxs, nus
        # problem_data — src/data/problem.jl:25-46
        if parameters is None:
            parameters = [np.zeros(d.num_parameter) for d in dynamics] + [np.zeros(0)]
        if len(parameters) == len(dynamics):
            parameters = list(parameters) + [np.zeros(0)]
        self.parameters = [np.asarray(p, dtype=np.float64) for p in parameters]
        self.states = [np.zeros(n) for n in nxs]
        self.actions = [np.zeros(m) for m in nus] + [np.zeros(0)]
        self.nominal_states = [np.zeros(n) for n in nxs]
        self.nominal_actions = [np.zeros(m) for m in nus] + [np.zeros(0)]
        # model_data / objective_data — src/data/model.jl:12-17, src/data/objective.jl:10-21
        self.fx = [np.zeros((d.num_next_state, d.num_state)) for d in dynamics]
        self.fu = [np.zeros((d.num_next_state, d.num_action)) for d in dynamics]
        self.gx = [np.zeros(n) for n in nxs]
        self.gu = [np.zeros(m) for m in nus]
        self.gxx = [np.zeros((n, n)) for n in nxs]
        self.guu = [np.zeros((m, m)) for m in nus]
        self.gux = [np.zeros((m, n)) for m, n in zip(nus, nxs)]
        # policy_data — src/data/policy.jl:44-78
        self.K = [np.zeros((m, n)) for m, n in zip(nus, nxs)]
        self.k = [np.zeros(m) for m in nus]
        self.P = [np.zeros((n, n)) for n in nxs]
        self.p = [np.zeros(n) for n in nxs]
        self.Qx = [np.zeros(n) for n in nxs[:-1]]
        self.Qu = [np.zeros(m) for m in nus]
        self.Qxx = [np.zeros((n, n)) for n in nxs[:-1]]
        self.Quu = [np.zeros((m, m)) for m in nus]
        self.Qux = [np.zeros((m, n)) for m, n in zip(nus, nxs)]
        # solver_data — src/data/solver.jl:20-47: z = (x1..xT, u1..uT-1)
        n_total = sum(nxs)
        self.indices_state, self.indices_action = [], []
        ns, ms = 0, 0
        for t in range(H - 1):
            self.indices_state.append(slice(ns, ns + nxs[t]))
            self.indices_action.append(slice(n_total + ms, n_total + ms + nus[t]))
            ns += nxs[t]
            ms += nus[t]
        self.indices_state.append(slice(ns, ns + nxs[-1]))
        self.num_trajectory = n_total + sum(nus)
        self.objective = math.inf                                              # :37
        self.max_violation = 0.0
        self.step_size = 1.0
        self.gradient = np.zeros(self.num_trajectory)
        self.status = False
        self.iterations = 0
        self.trajectory = np.zeros(self.num_trajectory)                        # src/data/problem.jl:43
        # augmented_lagrangian — src/augmented_lagrangian.jl:13-37
        if constraints is not None:
            assert len(constraints) == H
            ncs = [c.num_constraint for c in constraints]
            self.constraint_penalty = [np.ones(nc) for nc in ncs]
            self.constraint_dual = [np.zeros(nc) for nc in ncs]
            self.active_set = [np.ones(nc, dtype=np.int64) for nc in ncs]
            self.violations = [np.zeros(nc) for nc in ncs]                     # src/data/constraints.jl:12-17
            self.cx = [np.zeros((nc, n)) for nc, n in zip(ncs, nxs)]
            self.cu = [np.zeros((nc, m)) for nc, m in zip(ncs[:-1], nus)]
        # bookkeeping that is NOT in the reference (for the fixtures only)
        self.rollouts = 0
        self.outer_iterations = 0
        self.potrf_info = 0
        self.trace = []                 # rows (outer, inner, objective, gradient_norm, max_violation, step_size, status, rollouts)
        self.hooks = {}                 # name -> callable(solver, **info)
        self.last_forward = {}

    # -- src/solver.jl:48-66
    def initialize_controls(self, actions):
        for t, ut in enumerate(actions):
            self.nominal_actions[t][:] = ut

    def initialize_states(self, states):
        for t, xt in enumerate(states):
            self.nominal_states[t][:] = xt

    def get_trajectory(self):
        return self.nominal_states, self.nominal_actions[:-1]

    def _hook(self, name, **info):
        h = self.hooks.get(name)
        if h is not None:
            h(self, **info)

    # ------------------------------------------------------------------ src/data/methods.jl:56-62
    def trajectories(self, mode):
        if mode == "nominal":
            return self.nominal_states, self.nominal_actions, self.parameters
        return self.states, self.actions, self.parameters

    # ------------------------------------------------------------------ src/costs.jl:48-55
    def _cost_objective(self, states, actions, parameters):
        J = 0.0
        for t, cost in enumerate(self.costs):
            J += float(cost.evaluate(states[t], actions[t], parameters[t]))
        return J

    # ------------------------------------------------------------------ src/constraints.jl:66-73
    def _constraint_bang(self, states, actions, parameters):
        for t, con in enumerate(self.constraints):
            if con.num_constraint == 0:
                continue
            self.violations[t][:] = con.evaluate(states[t], actions[t], parameters[t])

    # ------------------------------------------------------------------ src/augmented_lagrangian.jl:68-85
    def _active_set_bang(self):
        for t in range(self.H):
            self.active_set[t][:] = 1
            for i in self.constraints[t].indices_inequality:
                if self.violations[t][i] < 0.0 and self.constraint_dual[t][i] == 0.0:
                    self.active_set[t][i] = 0

    # ------------------------------------------------------------------ src/augmented_lagrangian.jl:39-66
    def _cost_al(self, states, actions, parameters):
        J = self._cost_objective(states, actions, parameters)
        self._constraint_bang(states, actions, parameters)                     # :52
        self._active_set_bang()                                                # :53
        for t in range(self.H):
            J += float(self.constraint_dual[t] @ self.violations[t])           # :56
            for i in range(self.constraints[t].num_constraint):
                if self.active_set[t][i] == 1:
                    J += 0.5 * self.constraint_penalty[t][i] * self.violations[t][i] ** 2.0       # :60
        return J

    # ------------------------------------------------------------------ src/data/constraints.jl:23-46
    def _constraint_violation(self, states, actions, parameters):
        self._constraint_bang(states, actions, parameters)                     # :43
        max_violation = 0.0
        for t in range(self.H):
            ineq = self.constraints[t].indices_inequality
            for i in range(self.constraints[t].num_constraint):
                c = self.violations[t][i]
                cti = max(0.0, c) if i in ineq else abs(c)
                max_violation = _julia_max(max_violation, cti)
        return max_violation

    # ------------------------------------------------------------------ src/data/methods.jl:13-30
    def cost_bang(self, mode):
        x, u, w = self.trajectories(mode)
        if self.constraints is not None:
            self.objective = self._cost_al(x, u, w)
            # ALWAYS at problem.states / problem.actions (:22-27)
            self.max_violation = self._constraint_violation(self.states, self.actions, self.parameters)
        else:
            self.objective = self._cost_objective(x, u, w)
        return self.objective

    # ------------------------------------------------------------------ src/gradients.jl:1-98
    def gradients_bang(self, mode="nominal"):
        x, u, w = self.trajectories(mode)
        H = self.H
        # gradients!(dynamics) :1-8 -> jacobian! src/dynamics.jl:41-50
        for t, d in enumerate(self.dynamics):
            self.fx[t][:] = d.jacobian_state(x[t], u[t], w[t])
            self.fu[t][:] = d.jacobian_action(x[t], u[t], w[t])
        # gradients!(objective) :10-21 -> cost_gradient! (`.=`) src/costs.jl:57-68, cost_hessian! (`.+=`) src/costs.jl:70-84
        for t, cost in enumerate(self.costs):
            self.gx[t][:] = cost.gradient_state(x[t], u[t], w[t])
            if t == H - 1:
                continue
            self.gu[t][:] = cost.gradient_action(x[t], u[t], w[t])
        for t, cost in enumerate(self.costs):
            self.gxx[t] += cost.hessian_state_state(x[t], u[t], w[t])
            if t == H - 1:
                continue
            self.guu[t] += cost.hessian_action_action(x[t], u[t], w[t])
            self.gux[t] += cost.hessian_action_state(x[t], u[t], w[t])
        if self.constraints is None:
            return
        # gradients!(constraint_data) :83-90 -> jacobian! src/constraints.jl:75-87
        for t, con in enumerate(self.constraints):
            if con.num_constraint == 0:
                continue
            self.cx[t][:] = con.jacobian_state(x[t], u[t], w[t])
            if t == H - 1:
                continue
            self.cu[t][:] = con.jacobian_action(x[t], u[t], w[t])
        # :54-80 — uses the violations BUFFER and the active set as they stand
        for t in range(H):
            nc = self.constraints[t].num_constraint
            Irho = np.diag(self.constraint_penalty[t] * self.active_set[t]) if nc else np.zeros((0, 0))   # :56-58
            c_tmp = self.constraint_dual[t].copy()                             # :59
            c_tmp += Irho @ self.violations[t]                                 # :62
            self.gx[t] += self.cx[t].T @ c_tmp                                 # :63
            cx_tmp = Irho @ self.cx[t]                                         # :66
            self.gxx[t] += self.cx[t].T @ cx_tmp                               # :67
            if t == H - 1:
                continue
            self.gu[t] += self.cu[t].T @ c_tmp                                 # :72
            cu_tmp = Irho @ self.cu[t]                                         # :75
            self.guu[t] += self.cu[t].T @ cu_tmp                               # :76
            self.gux[t] += self.cu[t].T @ cx_tmp                               # :79

    # ------------------------------------------------------------------ src/backward_pass.jl:1-91
    def backward_pass_bang(self):
        H = self.H
        fx, fu, gx, gu, gxx, guu, gux = self.fx, self.fu, self.gx, self.gu, self.gxx, self.guu, self.gux
        K, k, P, p, Qx, Qu, Qxx, Quu, Qux = self.K, self.k, self.P, self.p, self.Qx, self.Qu, self.Qxx, self.Quu, self.Qux
        P[H - 1][:] = gxx[H - 1]                                               # :39
        p[H - 1][:] = gx[H - 1]                                                # :40
        for t in range(H - 2, -1, -1):                                         # :42
            Qx[t][:] = fx[t].T @ p[t + 1]                                      # :44
            Qx[t] += gx[t]
            Qu[t][:] = fu[t].T @ p[t + 1]                                      # :48
            Qu[t] += gu[t]
            xx_tmp = fx[t].T @ P[t + 1]                                        # :52
            Qxx[t][:] = xx_tmp @ fx[t]
            Qxx[t] += gxx[t]
            ux_tmp_hat = fu[t].T @ P[t + 1]                                    # :57
            Quu[t][:] = ux_tmp_hat @ fu[t]
            Quu[t] += guu[t]
            ux_tmp_hat = fu[t].T @ P[t + 1]                                    # :62
            Qux[t][:] = ux_tmp_hat @ fx[t]
            Qux[t] += gux[t]
            # LAPACK.potrf!('U', uu_tmp) — return code ignored (:68-69); potrs!('U', ...) (:72-73)
            uu_tmp = np.asfortranarray(Quu[t].copy())
            U, info = lapack.dpotrf(uu_tmp, lower=0, clean=0, overwrite_a=0)
            if info != 0 and self.potrf_info == 0:
                self.potrf_info = int(info)
            Kt, _ = lapack.dpotrs(U, np.asfortranarray(Qux[t].copy()), lower=0)
            kt, _ = lapack.dpotrs(U, np.asfortranarray(Qu[t].copy().reshape(-1, 1)), lower=0)
            K[t][:] = Kt
            k[t][:] = kt[:, 0]
            K[t] *= -1.0                                                       # :74
            k[t] *= -1.0                                                       # :75
            ux_tmp = Quu[t] @ K[t]                                             # :79
            P[t][:] = K[t].T @ ux_tmp                                          # :81
            P[t] += K[t].T @ Qux[t]                                            # :82
            P[t] += Qux[t].T @ K[t]                                            # :83
            P[t] += Qxx[t]                                                     # :84
            p[t][:] = ux_tmp.T @ k[t]                                          # :86
            p[t] += K[t].T @ Qu[t]                                             # :87
            p[t] += Qux[t].T @ k[t]                                            # :88
            p[t] += Qx[t]                                                      # :89

    # ------------------------------------------------------------------ src/solve.jl:67-83
    def lagrangian_gradient_bang(self):
        for t in range(self.H - 1):
            self.gradient[self.indices_state[t]] = self.Qx[t] - self.p[t]
            self.gradient[self.indices_action[t]] = self.Qu[t]

    # ------------------------------------------------------------------ src/data/methods.jl:42-54
    def trajectory_sensitivities(self):
        self.trajectory[:] = 0.0
        for t in range(self.H - 1):
            zx = self.trajectory[self.indices_state[t]]
            zu = self.k[t].copy()
            zu += self.K[t] @ zx
            self.trajectory[self.indices_action[t]] = zu
            zy = self.fu[t] @ zu
            zy += self.fx[t] @ zx
            self.trajectory[self.indices_state[t + 1]] = zy

    # ------------------------------------------------------------------ src/rollout.jl:1-31
    def rollout_bang(self, step_size=1.0):
        x, u, w = self.states, self.actions, self.parameters
        xb, ub = self.nominal_states, self.nominal_actions
        x[0][:] = xb[0]                                                        # :19
        for t, d in enumerate(self.dynamics):
            ut = self.k[t].copy()                                              # :24
            ut *= step_size                                                    # :25
            ut += ub[t]                                                        # :26
            ut += self.K[t] @ x[t]                                             # :27
            ut += -1.0 * (self.K[t] @ xb[t])                                   # :28
            u[t][:] = ut
            x[t + 1][:] = d.evaluate(x[t], u[t], w[t])                         # :29
        self.rollouts += 1

    # ------------------------------------------------------------------ src/data/methods.jl:32-39
    def update_nominal_trajectory(self):
        for t in range(self.H):
            self.nominal_states[t][:] = self.states[t]
            if t == self.H - 1:
                continue
            self.nominal_actions[t][:] = self.actions[t]

    # ------------------------------------------------------------------ src/forward_pass.jl:1-56
    def forward_pass_bang(self, c1=1.0e-4, max_iterations=25):
        opt = self.options
        self.status = False                                                    # :10
        J_prev = self.objective                                                # :13
        self.lagrangian_gradient_bang()                                        # :16
        if opt.line_search == "armijo":
            self.trajectory_sensitivities()                                    # :19
            delta_grad_product = float(self.gradient @ self.trajectory)        # :20
        else:
            delta_grad_product = 0.0
        self.step_size = 1.0                                                   # :26
        iteration = 1
        trial_J = []
        while self.step_size >= opt.min_step_size:                             # :28
            if iteration > max_iterations:
                break                                                          # :29
            self.rollout_bang(step_size=self.step_size)                        # :34
            J = self.cost_bang("current")                                      # :36
            trial_J.append(J)
            if J <= J_prev + c1 * self.step_size * delta_grad_product:         # :44 (NaN compares false)
                self.update_nominal_trajectory()                               # :46
                self.objective = J
                self.status = True
                break
            else:
                self.step_size *= 0.5                                          # :51
                iteration += 1
        self.last_forward = dict(delta_grad_product=delta_grad_product, trial_objectives=trial_J, J_prev=J_prev)

    # ------------------------------------------------------------------ src/solve.jl:1-54
    def ilqr_solve(self, outer=0):
        opt = self.options
        # reset!(problem.model) src/data/model.jl:19-26; reset!(problem.objective) src/data/objective.jl:23-33
        for t in range(self.H - 1):
            self.fx[t][:] = 0.0
            self.fu[t][:] = 0.0
        for t in range(self.H):
            self.gx[t][:] = 0.0
            self.gxx[t][:] = 0.0
            if t == self.H - 1:
                continue
            self.gu[t][:] = 0.0
            self.guu[t][:] = 0.0
            self.gux[t][:] = 0.0
        if opt.reset_cache:
            self.reset_data()
        self.cost_bang("nominal")                                              # :14
        self._hook("before_gradients", outer=outer, inner=0)
        self.gradients_bang("nominal")                                         # :16
        self.backward_pass_bang()                                              # :18
        self._hook("after_backward", outer=outer, inner=0)
        obj_prev = self.objective                                              # :21
        for i in range(1, opt.max_iterations + 1):                             # :22
            self._hook("before_forward", outer=outer, inner=i)
            self.forward_pass_bang()                                           # :23
            self._hook("after_forward", outer=outer, inner=i)
            if opt.line_search != "none":
                self._hook("before_gradients", outer=outer, inner=i)
                self.gradients_bang("nominal")                                 # :28
                self.backward_pass_bang()                                      # :30
                self.lagrangian_gradient_bang()                                # :32
                self._hook("after_backward", outer=outer, inner=i)
            gradient_norm = _julia_norm_inf(self.gradient)                     # :36
            self.gradient_norm = gradient_norm
            self.iterations += 1                                               # :39
            self.trace.append((outer, i, self.objective, gradient_norm, self.max_violation, self.step_size,
                               1.0 if self.status else 0.0, float(self.rollouts)))
            if gradient_norm < opt.lagrangian_gradient_tolerance:              # :48
                break
            if abs(self.objective - obj_prev) < opt.objective_tolerance:       # :49
                break
            obj_prev = self.objective
            if not self.status:                                                # :50
                break

    # reset!(data::SolverData) — src/data/solver.jl:49-59
    def reset_data(self):
        self.objective = 0.0
        self.gradient[:] = 0.0
        self.max_violation = 0.0
        self.status = False
        self.iterations = 0

    # ------------------------------------------------------------------ src/augmented_lagrangian.jl:87-110
    def augmented_lagrangian_update(self, scaling_penalty=10.0, max_penalty=1.0e12):
        for t in range(self.H):
            ineq = self.constraints[t].indices_inequality
            for i in range(self.constraints[t].num_constraint):
                self.constraint_dual[t][i] += self.constraint_penalty[t][i] * self.violations[t][i]
                if i in ineq:
                    self.constraint_dual[t][i] = _julia_max(0.0, self.constraint_dual[t][i])
                self.constraint_penalty[t][i] = _julia_min(scaling_penalty * self.constraint_penalty[t][i], max_penalty)

    # ------------------------------------------------------------------ src/solve.jl:88-129
    def constrained_ilqr_solve(self, augmented_lagrangian_callback=None):
        opt = self.options
        self.reset_data()                                                      # :93
        for lam in self.constraint_dual:
            lam[:] = 0.0                                                       # :96-98
        for rho in self.constraint_penalty:
            rho[:] = opt.initial_constraint_penalty                            # :101-103
        for i in range(1, opt.max_dual_updates + 1):                           # :105
            self.outer_iterations = i
            self.ilqr_solve(outer=i)                                           # :109
            self.cost_bang("nominal")                                          # :113
            if self.max_violation <= opt.constraint_tolerance:                 # :117
                break
            self.augmented_lagrangian_update(scaling_penalty=opt.scaling_penalty, max_penalty=opt.max_penalty)   # :120
            if augmented_lagrangian_callback is not None:
                augmented_lagrangian_callback(self)                            # :125

    # ------------------------------------------------------------------ src/solve.jl:137-143
    def solve(self, **kw):
        if self.constraints is not None:
            self.constrained_ilqr_solve(**kw)
        else:
            self.ilqr_solve()


def _julia_max(a, b):
    """Base.max on Float64: NaN-propagating."""
    if a != a or b != b:
        return math.nan
    return a if a > b else b


def _julia_min(a, b):
    if a != a or b != b:
        return math.nan
    return a if a < b else b


def _julia_norm_inf(v):
    """norm(v, Inf): max |v_i|, NaN-propagating."""
    m = 0.0
    for e in v:
        m = _julia_max(m, abs(float(e)))
    return m


# --------------------------------------------------------------------------------------------------
# The problems, written the way the reference's tests/examples write them
# --------------------------------------------------------------------------------------------------
def _dot(a, b):
    return sum(ai * bi for ai, bi in zip(a, b))


def _vadd(a, b):
    return [ai + bi for ai, bi in zip(a, b)]


def _vscale(s, a):
    return [s * ai for ai in a]


def particle_problem(T=11):
    """examples/particle.jl:17-43"""
    def particle_discrete(x, u):
        return [1.0 * x[0] + 1.0 * x[1] + 0.0 * u[0], 0.0 * x[0] + 1.0 * x[1] + 1.0 * u[0]]
    particle = Dynamics(particle_discrete, 2, 1)
    xT = [1.0, 0.0]
    stage = Cost(lambda x, u: 0.1 * _dot(x, x) + 0.1 * _dot(u, u), 2, 1)
    term = Cost(lambda x, u: 0.1 * _dot(x, x), 2, 0)
    cT = Constraint(lambda x, u: [x[i] - xT[i] for i in range(2)], 2, 0)
    return [particle] * (T - 1), [stage] * (T - 1) + [term], [Constraint() for _ in range(T - 1)] + [cT]


def acrobot_problem(T=51):
    """test/acrobot.jl:9-101 (T = 51 there; BASELINE uses T = 101)"""
    def acrobot_continuous(x, u):
        mass1, inertia1, length1, lengthcom1 = 1.0, 0.33, 1.0, 0.5
        mass2, inertia2, length2, lengthcom2 = 1.0, 0.33, 1.0, 0.5
        gravity, friction1, friction2 = 9.81, 0.1, 0.1

        def M(q):
            a = inertia1 + inertia2 + mass2 * length1 * length1 + 2.0 * mass2 * length1 * lengthcom2 * sp.cos(q[1])
            b = inertia2 + mass2 * length1 * lengthcom2 * sp.cos(q[1])
            c = inertia2
            return [[a, b], [b, c]]

        def Minv(q):
            m = M(q)
            a, b, c, d = m[0][0], m[0][1], m[1][0], m[1][1]
            s = 1.0 / (a * d - b * c)
            return [[s * d, s * (-b)], [s * (-c), s * a]]

        def tau(q):
            a = (-1.0 * mass1 * gravity * lengthcom1 * sp.sin(q[0])
                 - mass2 * gravity * (length1 * sp.sin(q[0]) + lengthcom2 * sp.sin(q[0] + q[1])))
            b = -1.0 * mass2 * gravity * lengthcom2 * sp.sin(q[0] + q[1])
            return [a, b]

        def Cm(x):
            a = -2.0 * mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[3]
            b = -1.0 * mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[3]
            c = mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[2]
            return [[a, b], [c, 0.0]]

        q, v = x[0:2], x[2:4]
        Cv = [Cm(x)[i][0] * v[0] + Cm(x)[i][1] * v[1] for i in range(2)]
        Bq = [0.0, 1.0]
        fr = [friction1, friction2]
        rhs = [-1.0 * Cv[i] + tau(q)[i] + Bq[i] * u[0] - fr[i] * v[i] for i in range(2)]
        Mi = Minv(q)
        qdd = [Mi[i][0] * rhs[0] + Mi[i][1] * rhs[1] for i in range(2)]
        return [x[2], x[3], qdd[0], qdd[1]]

    def acrobot_discrete(x, u):
        h = 0.1
        xm = _vadd(x, _vscale(0.5 * h, acrobot_continuous(x, u)))
        return _vadd(x, _vscale(h, acrobot_continuous(xm, u)))

    acrobot = Dynamics(acrobot_discrete, 4, 1)
    xT = [math.pi, 0.0, 0.0, 0.0]
    stage = Cost(lambda x, u: 0.1 * _dot(x[2:4], x[2:4]) + 0.1 * _dot(u, u), 4, 1)
    term = Cost(lambda x, u: 0.1 * _dot(x[2:4], x[2:4]), 4, 0)
    cT = Constraint(lambda x, u: [x[i] - xT[i] for i in range(4)], 4, 0)
    return [acrobot] * (T - 1), [stage] * (T - 1) + [term], [Constraint() for _ in range(T - 1)] + [cT]


def car_problem(T=51, goal_only=False):
    """test/car.jl:10-61; goal_only: only the terminal goal equality (BASELINE config 3 wording)."""
    def car_continuous(x, u):
        return [u[0] * sp.cos(x[2]), u[0] * sp.sin(x[2]), u[1]]

    def car_discrete(x, u):
        h = 0.1
        xm = _vadd(x, _vscale(0.5 * h, car_continuous(x, u)))
        return _vadd(x, _vscale(h, car_continuous(xm, u)))

    car = Dynamics(car_discrete, 3, 2)
    xT = [1.0, 1.0, 0.0]

    def d(x):
        return [x[i] - xT[i] for i in range(3)]
    stage = Cost(lambda x, u: 1.0 * _dot(d(x), d(x)) + 1.0e-2 * _dot(u, u), 3, 2)
    term = Cost(lambda x, u: 1000.0 * _dot(d(x), d(x)), 3, 0)
    ul, uu = [-5.0, -5.0], [5.0, 5.0]
    p_obs, r_obs = [0.5, 0.5], 0.1

    def con_stage(x, u):
        e = [x[0] - p_obs[0], x[1] - p_obs[1]]
        return [ul[0] - u[0], ul[1] - u[1], u[0] - uu[0], u[1] - uu[1], r_obs ** 2.0 - _dot(e, e)]

    def con_term(x, u):
        e = [x[0] - p_obs[0], x[1] - p_obs[1]]
        return d(x) + [r_obs ** 2.0 - _dot(e, e)]
    if goal_only:
        cons = [Constraint() for _ in range(T - 1)] + [Constraint(lambda x, u: d(x), 3, 0)]
    else:
        cs = Constraint(con_stage, 3, 2, indices_inequality=[1, 2, 3, 4, 5])
        cons = [cs] * (T - 1) + [Constraint(con_term, 3, 0, indices_inequality=[4])]
    return [car] * (T - 1), [stage] * (T - 1) + [term], cons


def synth32_problem(T=101, n=32, m=8):
    """x+ = x + h(Ax + Bu + 0.1 sin x), A_ij = -δ_ij + 0.3 cos(i + 2j)/n, B_ij = sin(3i + j)/√n
    (1-based), ℓ = 0.1‖x − x_g‖² + 0.01‖u‖², ℓ_T = 10‖x − x_g‖², x_g = 0.5·1, stage inequalities [−1 − u; u − 1]."""
    h = 0.05
    A = [[(-1.0 if i == j else 0.0) + 0.3 * math.cos(float((i + 1) + 2 * (j + 1))) / float(n) for j in range(n)] for i in range(n)]
    Bm = [[math.sin(float(3 * (i + 1) + (j + 1))) / math.sqrt(float(n)) for j in range(m)] for i in range(n)]

    def f(x, u):
        out = []
        for i in range(n):
            acc = 0.0
            for j in range(n):
                acc = acc + A[i][j] * x[j]
            for j in range(m):
                acc = acc + Bm[i][j] * u[j]
            acc = acc + 0.1 * sp.sin(x[i])
            out.append(x[i] + h * acc)
        return out
    dyn = Dynamics(f, n, m)
    xg = 0.5
    stage = Cost(lambda x, u: 0.1 * sum((xi - xg) * (xi - xg) for xi in x) + 0.01 * _dot(u, u), n, m)
    term = Cost(lambda x, u: 10.0 * sum((xi - xg) * (xi - xg) for xi in x), n, 0)
    box = Constraint(lambda x, u: [-1.0 - u[j] for j in range(m)] + [u[j] - 1.0 for j in range(m)], n, m,
                     indices_inequality=list(range(1, 2 * m + 1)))
    return [dyn] * (T - 1), [stage] * (T - 1) + [term], [box] * (T - 1) + [Constraint()]


def synth_box_problem(T, n, m, xmax=1.2):
    """synth32_problem with a state box |x_i| <= xmax on top of the action box: 2m + 2n stage inequalities."""
    dyn, costs, cons = synth32_problem(T, n, m)
    rows = lambda x, u: ([-1.0 - u[j] for j in range(m)] + [u[j] - 1.0 for j in range(m)] +
                         [x[i] - xmax for i in range(n)] + [-xmax - x[i] for i in range(n)])
    box = Constraint(rows, n, m, indices_inequality=list(range(1, 2 * m + 2 * n + 1)))
    return dyn, costs, [box] * (T - 1) + [Constraint()]


RAGGED_N, RAGGED_M = [3, 3, 4, 4, 2, 2, 3, 3], [2, 1, 2, 1, 1, 2, 2, 1]


def ragged_problem(T=9):
    """Time-varying DIMENSIONS (src/dynamics.jl:5-7, README.md:26): num_state = 3,3,4,4,2,2,3,3 | ..., num_action = 2,1,2,1,1,2,2,1 | ...
    (period 8), a linear-plus-sine map per step, quadratic costs, a terminal equality on the first two states. Same definition as
    tests/test_codegen.py::_ragged_problem (product) and oracle/models.cpp "ragged" (oracle)."""
    n_t = [RAGGED_N[t % 8] for t in range(T)]
    m_t = [RAGGED_M[t % 8] for t in range(T - 1)]

    def dyn(n0, m0, n1):
        A = [[(0.9 if i == j else 0.0) + 0.1 * math.cos(1.0 + i + 2 * j + n0) for j in range(n0)] for i in range(n1)]
        Bm = [[0.3 * math.sin(2.0 + 3 * i + j + m0) for j in range(m0)] for i in range(n1)]
        return Dynamics(lambda x, u: [sum(A[i][j] * x[j] for j in range(n0)) + sum(Bm[i][j] * u[j] for j in range(m0))
                                      + (0.1 * sp.sin(x[0]) if i == 0 else 0.0) for i in range(n1)], n0, m0)

    def cost(n0, m0):
        return Cost(lambda x, u: 0.5 * sum((1.0 + 0.1 * i) * x[i] * x[i] for i in range(n0))
                    + 0.05 * sum((1.0 + j) * u[j] * u[j] for j in range(m0)), n0, m0)

    dcache, ccache = {}, {}
    dynamics, costs = [], []
    for t in range(T - 1):
        kd, kc = (n_t[t], m_t[t], n_t[t + 1]), (n_t[t], m_t[t])
        if kd not in dcache:
            dcache[kd] = dyn(*kd)
        if kc not in ccache:
            ccache[kc] = cost(*kc)
        dynamics.append(dcache[kd]); costs.append(ccache[kc])
    nT = n_t[-1]
    costs.append(Cost(lambda x, u: 5.0 * sum(x[i] * x[i] for i in range(nT)), nT, 0))
    constraints = [Constraint() for _ in range(T - 1)] + [Constraint(lambda x, u: [x[0] - 0.2, x[1] + 0.1], nT, 0)]
    return dynamics, costs, constraints


PROBLEMS = {
    "ragged": ragged_problem,
    "particle": particle_problem,
    "acrobot": acrobot_problem,
    "car": car_problem,
    "car_goal": lambda T=51: car_problem(T, goal_only=True),
    "synth32": synth32_problem,
}
