"""Generates tests/golden/ref_*.npz from the INDEPENDENT restatement (reference_restatement.py: numpy + sympy +
scipy's real LAPACK dpotrf/dpotrs), never from `oracle/` and never from the product package.

    python tests/golden/make_reference_fixtures.py            # all cases (about two minutes)
    python tests/golden/make_reference_fixtures.py car_i0     # one case

A fixture = inputs + expected outputs of one problem instance:
    x1, ubar                       inputs (x̄ is rollout(dynamics, x1, ū), src/rollout.jl:33-42)
    trace[n_iter, 8]               per inner iteration: outer, inner, objective, gradient_norm, max_violation,
                                   step_size, status, rollouts-so-far  (what `verbose` prints, src/solve.jl:40-45)
    x, u, K, k, stats[9]           the solution: nominal trajectory, gains, {objective, gradient_norm, max_violation,
                                   step_size, iterations, outer_iterations, status, rollouts, potrf_info}
    s<j>_*                         stage snapshots at chosen (outer, inner) linearisation points j:
        pre_*    solver state right BEFORE gradients! (nominal / current trajectories, accumulated Hessians,
                 violations buffer, duals, penalties, active set, objective)
        fx fu gx gu gxx guu gux    after gradients!                         (src/gradients.jl)
        Qx Qu Qxx Quu Qux K k P p Lx Lu   after backward_pass! + lagrangian_gradient!   (src/backward_pass.jl, src/solve.jl:67-83)
        fwd_*    the forward_pass! that follows: delta (∇Lᵀ·Δz, src/forward_pass.jl:20), trial objectives, accepted
                 step size, status, objective, max_violation, new nominal / current trajectories, violations
Matrices are stored [t][row][col] (numpy); Julia's column-major buffers are the per-timestep transposes.
The instance inputs use the SAME seeded draws as iterativelqr.jl_amd/workloads.py (np.random.default_rng([20240607, b])),
re-stated here so that this script imports nothing from the product (tests/test_reference_fixtures.py checks they agree).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import reference_restatement as R  # noqa: E402

SEED = 20240607


def instance_inputs(model, T, b):
    """Same draws as workloads.make_inputs(config, B)[b] (instance index b, offset 0)."""
    n, m = {"particle": (2, 1), "acrobot": (4, 1), "car": (3, 2), "car_goal": (3, 2), "synth32": (32, 8)}[model]
    x1 = np.zeros(n)
    ub = np.zeros((T - 1, m))
    rng = np.random.default_rng([SEED, b])
    if model == "particle":
        ub[:] = 0.1 * rng.standard_normal((T - 1, m))
    elif model == "acrobot":
        ub[:] = 1.0 * rng.standard_normal((T - 1, m))
    elif model == "synth32":
        x1[:] = 0.5 * rng.standard_normal(n)
    else:
        ub[:] = 1.0e-2 * np.array([1.0, 0.1])
        if b > 0:
            ub *= 1.0 + 0.5 * rng.uniform(-1.0, 1.0)
            x1[:2] = 0.05 * rng.standard_normal(2)
    return x1, ub


# case name -> (model, T, instance index, input tweak, snapshot points [(outer, inner)])
CASES = {
    "particle_sin": ("particle", 11, None, "sin", [(1, 0), (1, 1), (1, 4), (2, 0)]),
    "particle_i0": ("particle", 11, 0, None, [(1, 0), (1, 3), (2, 0)]),
    "particle_i1": ("particle", 11, 1, None, [(1, 0)]),
    "car_i0": ("car", 51, 0, None, [(1, 0), (1, 1), (1, 40), (2, 0), (2, 2)]),
    "car_i1": ("car", 51, 1, None, [(1, 0), (1, 20), (2, 0)]),
    "car_i2": ("car", 51, 2, None, [(1, 0), (2, 1)]),
    "car_goal_i0": ("car_goal", 51, 0, None, [(1, 0), (1, 5), (2, 0)]),
    "car_goal_i1": ("car_goal", 51, 1, None, [(1, 0), (2, 0)]),
    "acrobot51_i0": ("acrobot", 51, 0, None, [(1, 0), (1, 3), (2, 0), (3, 50)]),
    "acrobot_i0": ("acrobot", 101, 0, None, [(1, 0), (1, 2), (2, 0), (3, 0), (3, 60)]),
    "acrobot_i1": ("acrobot", 101, 1, None, [(1, 0), (1, 10), (4, 30)]),
    "acrobot_i2": ("acrobot", 101, 2, None, [(1, 0), (2, 5), (5, 99)]),
    "synth32_i0": ("synth32", 101, 0, "box", []),
    "synth32_i1": ("synth32", 101, 1, "box", []),
    "synth32_t11_i0": ("synth32", 11, 0, "box", [(1, 0), (1, 2), (2, 0)]),
}


def stack(lst):
    return np.stack([np.asarray(a, dtype=np.float64) for a in lst]) if len(lst) else np.zeros((0,))


def cat(lst):
    return np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in lst]) if len(lst) else np.zeros(0)


def generate(case):
    model, T, b, tweak, points = CASES[case]
    dyn, costs, cons = R.PROBLEMS[model](T)
    if tweak == "sin":                                   # SURVEY.md Appendix C's particle input
        x1 = np.zeros(2)
        ub = (0.1 * np.sin(np.arange(1, T))).reshape(T - 1, 1)
    else:
        x1, ub = instance_inputs(model, T, b)
        if tweak == "box":                               # start outside the action box so that the inequalities act
            ub = ub + 1.5 * np.sin(0.37 * np.arange(ub.size).reshape(ub.shape))
    xb = R.rollout(dyn, x1, ub)
    s = R.Solver(dyn, costs, cons)
    s.initialize_controls(ub)
    s.initialize_states(xb)
    out = {"x1": x1, "ubar": ub, "xbar": stack(xb), "horizon": np.array([T]), "points": np.array(points, dtype=np.int64).reshape(-1, 2)}
    pending = {}

    def before_gradients(sol, outer, inner):
        if (outer, inner) not in points:
            return
        j = points.index((outer, inner))
        p = "s%d_" % j
        out[p + "pre_nominal_states"] = stack(sol.nominal_states)
        out[p + "pre_nominal_actions"] = stack(sol.nominal_actions[:-1])
        out[p + "pre_states"] = stack(sol.states)
        out[p + "pre_actions"] = stack(sol.actions[:-1])
        out[p + "pre_gxx"] = stack(sol.gxx)
        out[p + "pre_guu"] = stack(sol.guu)
        out[p + "pre_gux"] = stack(sol.gux)
        out[p + "pre_violations"] = cat(sol.violations)
        out[p + "pre_dual"] = cat(sol.constraint_dual)
        out[p + "pre_penalty"] = cat(sol.constraint_penalty)
        out[p + "pre_active_set"] = cat(sol.active_set)
        out[p + "pre_scalars"] = np.array([sol.objective, sol.max_violation, sol.step_size, float(sol.status)])

    def after_backward(sol, outer, inner):
        if (outer, inner) not in points:
            return
        j = points.index((outer, inner))
        p = "s%d_" % j
        for name in ("fx", "fu", "gx", "gu", "gxx", "guu", "gux", "Qx", "Qu", "Qxx", "Quu", "Qux", "K", "k", "P", "p"):
            out[p + name] = stack(getattr(sol, name))
        sol.lagrangian_gradient_bang()      # at inner == 0 the reference computes it at the top of forward_pass! (:16); same values
        out[p + "Lx"] = stack([sol.gradient[i] for i in sol.indices_state[:-1]])
        out[p + "Lu"] = stack([sol.gradient[i] for i in sol.indices_action])
        pending[(outer, inner + 1)] = j

    def after_forward(sol, outer, inner):
        j = pending.pop((outer, inner), None)
        if j is None:
            return
        p = "s%d_fwd_" % j
        lf = sol.last_forward
        out[p + "delta"] = np.array([lf["delta_grad_product"]])
        out[p + "trial_objectives"] = np.array(lf["trial_objectives"])
        out[p + "scalars"] = np.array([sol.objective, sol.max_violation, sol.step_size, float(sol.status)])
        out[p + "nominal_states"] = stack(sol.nominal_states)
        out[p + "nominal_actions"] = stack(sol.nominal_actions[:-1])
        out[p + "states"] = stack(sol.states)
        out[p + "actions"] = stack(sol.actions[:-1])
        out[p + "violations"] = cat(sol.violations)
        out[p + "active_set"] = cat(sol.active_set)
        out[p + "trajectory"] = sol.trajectory.copy()

    s.hooks = {"before_gradients": before_gradients, "after_backward": after_backward, "after_forward": after_forward}
    t0 = time.time()
    s.solve()
    dt = time.time() - t0
    # points the solve never reached (it converged earlier) are dropped and the rest renumbered
    taken = sorted(int(k[1:k.index("_")]) for k in out if k.endswith("_fx") and ("s%s_fwd_delta" % k[1:k.index("_")]) in out)
    renum = {j: i for i, j in enumerate(taken)}
    for key in [k for k in out if k[0] == "s" and k[1].isdigit()]:
        j = int(key[1:key.index("_")])
        val = out.pop(key)
        if j in renum:
            out["S%d%s" % (renum[j], key[key.index("_"):])] = val
    for key in [k for k in out if k[0] == "S"]:
        out["s" + key[1:]] = out.pop(key)
    out["points"] = np.array([points[j] for j in taken], dtype=np.int64).reshape(-1, 2)
    out["trace"] = np.array(s.trace, dtype=np.float64).reshape(-1, 8)
    out["x"] = stack(s.nominal_states)
    out["u"] = stack(s.nominal_actions[:-1])
    if model == "synth32" and T > 11:                    # gains of a few timesteps only (size)
        sel = [0, 1, T // 2, T - 2]
        out["K_steps"] = np.array(sel)
        out["K"] = stack([s.K[t] for t in sel])
    else:
        out["K"] = stack(s.K)
    out["k"] = stack(s.k)
    out["stats"] = np.array([s.objective, s.gradient_norm, s.max_violation, s.step_size, s.iterations,
                             s.outer_iterations, float(s.status), s.rollouts, s.potrf_info], dtype=np.float64)
    path = os.path.join(HERE, "ref_%s.npz" % case)
    np.savez_compressed(path, **out)
    print("%-16s iterations %4d  outer %2d  rollouts %4d  J %.12g  max_violation %.3e  (%.1f s, %d KB)"
          % (case, s.iterations, s.outer_iterations, s.rollouts, s.objective, s.max_violation, dt, os.path.getsize(path) // 1024))


if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES)):
        generate(c)
