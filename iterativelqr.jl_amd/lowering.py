"""Time-varying stage objects (README.md:26 of the reference: "costs, constraints and dynamics can differ
at every timestep") on a kernel that is compiled for ONE stage template.

The reference keeps a Vector of objects over t (src/solver.jl:11-46, src/data/*.jl) and simply calls object t at
step t. The device kernel is specialised at compile time on one Dynamics / stage Cost / stage Constraint, so
distinct objects are LOWERED here, exactly, onto that template:

  * the K distinct stage objects of a category (structurally distinct traced expressions, in order of first appearance)
    become one combined object that takes K extra per-timestep parameters s_k(t) ∈ {0, 1} (one-hot "selectors"):
        ℓ(x,u,w)  = Σ_k [s_k ? ℓ_k(x,u,w) : 0]     f(x,u,w) = Σ_k [s_k ? f_k(x,u,w) : 0]
        c(x,u,w)  = [s_1 ? c_1 : 0; …; s_K ? c_K : 0]     (rows concatenated, inequality indices shifted)
  * the selection is a ternary in the generated code (sympy Piecewise, carried through every derivative), sums with 0 are
    exact in IEEE arithmetic, so values, gradients, Jacobians and Hessians of step t are those of object t — also where a
    switched-off kind would evaluate to NaN / Inf (0 · NaN would poison a product form);
  * the rows of constraint kinds that are switched off at step t read c = 0 with zero Jacobian: their
    multipliers stay 0 (λ ← max(0, λ + ρ·0) or λ + ρ·0, src/augmented_lagrangian.jl:100-108), they add nothing
    to the AL cost, its gradient and Gauss-Newton Hessian (src/augmented_lagrangian.jl:39-66,
    src/gradients.jl:23-81) and nothing to max_violation (src/data/constraints.jl:23-46).

The selectors ride in the parameter trajectory θ_t behind the user's own parameters. Time-varying DIMENSIONS
(num_next_state ≠ num_state, src/dynamics.jl:5-7) are lowered by zero-padding every kind to the largest
(num_state, num_action) of the horizon — see the comment at the padding site for why that is exact; the host
arrays (x1, ū, trajectories, gains) are then the padded ones, with `state_dims[t]` / `action_dims[t]` real entries.
"""
import numpy as np
import sympy as sp

from .codegen import Constraint, Cost, Dynamics, MAX_CONSTRAINT_ROWS


def _key(o):
    """Structural identity of a traced object: two objects built from the same function (say, `[Constraint(f, ...) for t in
    range(T - 1)]`) are ONE kind, and every zero-row constraint is the same empty kind — otherwise each would get its own
    selector column in θ_t."""
    if isinstance(o, Constraint):
        if o.num_constraint == 0:
            return ("constraint", 0)
        return ("constraint", o.num_constraint, o.num_state, o.num_action, o.num_parameter, tuple(o.evaluate),
                tuple(o.indices_inequality))
    if isinstance(o, Cost):
        return ("cost", o.num_state, o.num_action, o.num_parameter, o.evaluate)
    return ("dynamics", o.num_next_state, o.num_state, o.num_action, o.num_parameter, tuple(o.evaluate))


def _kinds(objs):
    kinds, keys, index = [], [], []
    for o in objs:
        key = _key(o)
        for k, q in enumerate(keys):
            if kinds[k] is o or q == key:
                index.append(k)
                break
        else:
            kinds.append(o)
            keys.append(key)
            index.append(len(kinds) - 1)
    return kinds, index


def stage_kinds(dynamics, costs, constraints=None):
    """The problem's objects as ilqr_stage_kinds (include/ilqr_hip.h): distinct kinds per category and the kind of every step."""
    from . import _ffi
    T = len(costs)
    assert len(dynamics) == T - 1, "need T-1 dynamics and T costs"                    # src/data/problem.jl:30
    assert constraints is None or len(constraints) == T
    dk, di = _kinds(dynamics)
    ck, ci = _kinds(costs[:-1])
    kk, ki = _kinds(constraints[:-1]) if constraints is not None else ([], [])
    cost_term = costs[-1]
    con_term = constraints[-1] if constraints is not None else None
    everything = dk + ck + kk + [cost_term] + ([con_term] if con_term is not None else [])
    nwu = max(o.num_parameter for o in everything)
    mask = lambda c: sum(1 << (i - 1) for i in c.indices_inequality)          # indices_inequality is 1-based like the reference
    nct = con_term.num_constraint if con_term is not None else 0
    k = _ffi.stage_kinds(T, nwu, [(d.num_state, d.num_action, d.num_next_state) for d in dk], di,
                         [(c.num_state, c.num_action) for c in ck], ci,
                         [(c.num_constraint, c.num_state, c.num_action, mask(c)) for c in kk], ki,
                         cost_term.num_state, nct, mask(con_term) if nct else 0)
    return k, (dk, ck, kk, ki, cost_term, con_term, nwu)


def lower(dynamics, costs, constraints=None):
    """-> dict(dynamics, cost_stage, cost_term, con_stage, con_term, num_user_parameter, selectors[T, S],
    constraint_rows[t] = device rows of the stage constraint that belong to step t).

    The PLAN — template dimensions, selector columns and table, row offsets of the stacked constraint kinds, consistency of the
    chain of dimensions — comes from the library (ilqr_plan_stages, the same code a Julia or C host reaches through
    ilqr_compile_model_stages); what is done here is only what needs the symbolic objects: finding the distinct kinds and
    gating their expressions by the plan's selector columns."""
    from . import _ffi
    T = len(costs)
    kinds, (dk, ck, kk, ki, cost_term, con_term, nwu) = stage_kinds(dynamics, costs, constraints)
    plan, sel_rows, n_t, m_t = _ffi.plan_stages(kinds)         # raises IlqrError on an inconsistent chain of dimensions
    n, m, nw = plan.nx, plan.nu, plan.nw
    if plan.n_selectors == 0 and len(dk) == 1 and len(ck) == 1 and len(kk) <= 1:
        return dict(dynamics=dk[0], cost_stage=ck[0], cost_term=cost_term, con_stage=kk[0] if kk else None,
                    con_term=con_term, num_user_parameter=nwu, selectors=np.zeros((T, 0)),
                    constraint_rows=[list(range(kk[0].num_constraint if kk else 0))] * (T - 1),
                    state_dims=n_t, action_dims=m_t)
    blocks = {name: col for name, col in (("dynamics", plan.sel_dynamics), ("cost", plan.sel_cost), ("constraint", plan.sel_constraint))
              if col >= 0}
    sel = np.array(sel_rows, dtype=np.float64).reshape(T, plan.n_selectors)

    def gate(name, k, w, expr):
        """expr where kind k of the category is selected at this step, 0 elsewhere — a real SELECT (ternary in the generated
        code, carried through every derivative), not a product with the selector: 0 · NaN = NaN, and a kind that is
        switched off may well be outside its domain (sqrt, log, 1/x) at a step where the reference never calls it."""
        if name not in blocks:
            return expr
        return sp.Piecewise((expr, w[blocks[name] + k] > 0.5), (0, True))

    # the traced objects all use the same symbols x0.., u0.., w0.. (codegen._variables), so their expressions can be
    # combined directly; the lambdas only pick up the selector symbols
    # time-varying DIMENSIONS: every kind is zero-padded to (n, m) = (max n_t, max m_t). Padded next-state rows are 0;
    # padded actions get the cost u²/2 so that Quu stays positive definite — block-diagonal with the real block, so
    # its Cholesky and solves leave the real block untouched and return K = 0, k = 0 for the padding (u stays 0);
    # padded states never enter any function, so their rows/columns of fx, Qxx, Qux, P are exactly zero.
    dyn = Dynamics(lambda x, u, w: [sum(gate("dynamics", k, w, d.evaluate[i] if i < d.num_next_state else 0)
                                        for k, d in enumerate(dk)) for i in range(n)], n, m, nw)
    cost_stage = Cost(lambda x, u, w: sum(gate("cost", k, w, c.evaluate + sum(u[j] * u[j] for j in range(c.num_action, m)) / 2)
                                          for k, c in enumerate(ck)), n, m, nw)
    cost_term_l = Cost(lambda x, u, w: cost_term.evaluate, n, 0, nw)
    con_stage = con_term_l = None
    rows = [[] for _ in range(T - 1)]
    if constraints is not None:
        row0 = [plan.constraint_row0[k] for k in range(len(kk))]
        total = plan.nc_stage
        ineq = [i + 1 for i in range(total) if (plan.ineq_stage_words[i // 64] >> (i % 64)) & 1]
        assert total <= MAX_CONSTRAINT_ROWS, "at most %d stage constraint rows over all kinds" % MAX_CONSTRAINT_ROWS
        if total:
            con_stage = Constraint(lambda x, u, w: [gate("constraint", k, w, e) for k, c in enumerate(kk) for e in c.evaluate],
                                   n, m, indices_inequality=ineq, num_parameter=nw)
        else:
            con_stage = Constraint()
        for t in range(T - 1):
            rows[t] = [row0[ki[t]] + i for i in range(kk[ki[t]].num_constraint)]
        if con_term.num_constraint:
            con_term_l = Constraint(lambda x, u, w: list(con_term.evaluate), n, 0,
                                    indices_inequality=con_term.indices_inequality, num_parameter=nw)
        else:
            con_term_l = Constraint()
    return dict(dynamics=dyn, cost_stage=cost_stage, cost_term=cost_term_l, con_stage=con_stage, con_term=con_term_l,
                num_user_parameter=nwu, selectors=sel, constraint_rows=rows, state_dims=n_t, action_dims=m_t)


def c_stage_sources(dynamics, costs, constraints=None):
    """The problem's per-step objects as (StageKinds, C source of the distinct kinds' callables) for ilqr_compile_model_stages — the
    route of a host without the symbolic device-code generator (the Julia wrapper does the same with Symbolics' C target): the
    LIBRARY lowers the kinds onto its one-stage template. Solver(stage_sources = c_stage_sources(...))."""
    from . import codegen
    kinds, (dk, ck, kk, ki, cost_term, con_term, nwu) = stage_kinds(dynamics, costs, constraints)
    src = "".join(codegen.c_dynamics("dynamics_%d" % q, d) for q, d in enumerate(dk))
    src += "".join(codegen.c_cost("cost_stage_%d" % q, c) for q, c in enumerate(ck))
    src += "".join(codegen.c_constraint("constraint_stage_%d" % q, c) for q, c in enumerate(kk) if c.num_constraint > 0)
    src += codegen.c_cost("cost_terminal", cost_term, terminal=True)
    if con_term is not None and con_term.num_constraint > 0:
        src += codegen.c_constraint("constraint_terminal", con_term, terminal=True)
    return kinds, src
