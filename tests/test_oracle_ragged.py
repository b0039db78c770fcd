"""The oracle with PER-TIMESTEP dimensions (num_next_state != num_state, src/dynamics.jl:5-7; every buffer sized per step,
src/data/{model,objective,policy,problem}.jl) against the independent numpy / scipy-LAPACK restatement
(tests/golden/reference_restatement.py, which keeps the reference's Vectors of per-step arrays natively): whole solves with
identical control flow, and every stage buffer of one linearisation + Riccati pass + forward pass on identical inputs.
Both are restatements (the reference cannot run here); they share no code."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _inputs(pr, B, seed=5):
    rng = np.random.default_rng(seed)
    T = pr.T
    x1 = np.zeros((B, pr.nx)); x1[:, :pr.state_dims[0]] = 0.5 * rng.standard_normal((B, pr.state_dims[0]))
    ub = np.zeros((B, T - 1, pr.nu))
    for t in range(T - 1):
        ub[:, t, :pr.action_dims[t]] = 0.2 * rng.standard_normal((B, pr.action_dims[t]))
    return x1, ub


def _restatement_solver(R, T, pr, x1, ub):
    dynamics, costs, constraints = R.ragged_problem(T)
    s = R.Solver(dynamics, costs, constraints, options=R.Options())
    u = [ub[t, :pr.action_dims[t]] for t in range(T - 1)]
    xbar = R.rollout(dynamics, x1[:pr.state_dims[0]], u)
    s.initialize_controls(u); s.initialize_states(xbar)
    return s, xbar


@pytest.mark.parametrize("T", [9, 41])
def test_ragged_whole_solve_matches_the_independent_restatement(oracle, T):
    import reference_restatement as R
    pr = oracle.Problem("ragged", T)
    assert pr.state_dims == [R.RAGGED_N[t % 8] for t in range(T)] and pr.action_dims == [R.RAGGED_M[t % 8] for t in range(T - 1)]
    assert (pr.nx, pr.nu, pr.uniform) == (4, 2, False)
    B = 3
    x1, ub = _inputs(pr, B)
    got = oracle.solve_batch("ragged", T, x1, ub, nthreads=2)
    for b in range(B):
        s, xbar = _restatement_solver(R, T, pr, x1[b], ub[b])
        assert np.abs(pr.rollout(x1[b], ub[b]) - pr.unpack(np.concatenate(xbar), "x")).max() < 1e-13
        s.solve()
        st = got["stats"]
        assert (st["iterations"][b], st["outer_iterations"][b], st["rollouts"][b]) == (s.iterations, s.outer_iterations, s.rollouts)
        x, u = s.get_trajectory()
        for t in range(T):
            n = pr.state_dims[t]
            assert np.abs(got["x"][b, t, :n] - x[t]).max() < 1e-9 and (got["x"][b, t, n:] == 0).all()
        for t in range(T - 1):
            n, m = pr.state_dims[t], pr.action_dims[t]
            assert np.abs(got["u"][b, t, :m] - u[t]).max() < 1e-9 and (got["u"][b, t, m:] == 0).all()
            Kt = got["K"][b, t].T                      # [n][m] column-major -> m x n
            assert np.abs(Kt[:m, :n] - s.K[t]).max() <= 1e-8 * max(1.0, np.abs(s.K[t]).max())
            assert (Kt[m:, :] == 0).all() and (Kt[:, n:] == 0).all()
        assert abs(st["objective"][b] - s.objective) <= 1e-9 * max(1.0, abs(s.objective))
        assert abs(st["max_violation"][b] - s.max_violation) <= 1e-10


def test_ragged_stages_match_the_independent_restatement(oracle):
    """cost! -> gradients! -> backward_pass! -> forward_pass! from identical inputs: every per-timestep block."""
    import reference_restatement as R
    T = 17
    pr = oracle.Problem("ragged", T)
    x1, ub = _inputs(pr, 1, seed=9)
    s, xbar = _restatement_solver(R, T, pr, x1[0], ub[0])
    o = oracle.Solver(pr)
    o.initialize_controls(ub[0]); o.initialize_states(pr.rollout(x1[0], ub[0]))
    for _ in range(2):                                     # twice: the second linearisation adds to the accumulated Hessians (Q1)
        s.cost_bang("nominal"); s.gradients_bang(); s.backward_pass_bang()
        o.call("cost_bang", 0); o.call("gradients"); o.call("backward_pass")
        for name, want in (("jacobian_state", s.fx), ("jacobian_action", s.fu), ("hessian_state_state", s.gxx),
                           ("hessian_action_action", s.guu), ("hessian_action_state", s.gux), ("K", s.K), ("P", s.P),
                           ("Qxx", s.Qxx), ("Quu", s.Quu), ("Qux", s.Qux)):
            got = o.padded(name)
            for t, w in enumerate(want):
                r, c = w.shape
                blk = got[t].T                              # [col][row] -> [row][col]
                assert np.abs(blk[:r, :c] - w).max() <= 1e-10 * max(1.0, np.abs(w).max()), (name, t)
                assert (blk[r:, :] == 0).all() and (blk[:, c:] == 0).all()
        for name, want in (("gradient_state", s.gx), ("gradient_action", s.gu), ("k", s.k), ("p", s.p), ("Qx", s.Qx), ("Qu", s.Qu)):
            got = o.padded(name)
            for t, w in enumerate(want):
                assert np.abs(got[t, :len(w)] - w).max() <= 1e-10 * max(1.0, np.abs(w).max()), (name, t)
        s.forward_pass_bang(); o.call("forward_pass")
        assert o.stats().step_size == s.step_size and bool(o.stats().status) == s.status
        assert np.abs(o.buffer("gradient") - s.gradient).max() <= 1e-10 * max(1.0, np.abs(s.gradient).max())
        assert np.abs(o.buffer("trajectory") - s.trajectory).max() <= 1e-10 * max(1.0, np.abs(s.trajectory).max())
        x, u = o.get_trajectory()
        for t in range(T):
            assert np.abs(x[t, :pr.state_dims[t]] - s.nominal_states[t]).max() < 1e-11


@pytest.mark.parametrize("case", ["ragged_T9", "ragged_T41", "car_tv_T21"])
def test_oracle_against_reference_produced_f3_fixtures(oracle, case, tmp_path):
    """SURVEY §8 f3 on the route to a reference-produced pin: tests/golden/dump_fixture_inputs.py writes the inputs of three
    whole solves with per-step objects / dimensions, tests/golden/make_julia_fixtures.jl (the REAL package; no Julia in this
    image) solves them, julia_to_npz.py turns its output into ref_<case>_julia.npz. Always checked here: the dumped inputs are
    what the Julia script's reader expects (padded to the largest dimensions) and the oracle solves them; compared with the
    reference's own numbers when such a file exists, skipped (with the reason) while it does not."""
    import dump_fixture_inputs as D
    model, T, x1, ub, n_t, m_t = D.f3_inputs(case)
    ind = D.main(str(tmp_path / "julia_in"))
    assert open(os.path.join(ind, case + ".txt")).read().split("\n")[:3] == [model, str(T), "0"]
    assert np.array_equal(np.fromfile(os.path.join(ind, case + ".x1.f64")), x1)
    assert np.array_equal(np.fromfile(os.path.join(ind, case + ".u.f64")).reshape(T - 1, -1), ub)
    pr = oracle.Problem(model, T)
    assert pr.state_dims == n_t and pr.action_dims == m_t and (pr.nx, pr.nu) == (max(n_t), max(m_t))
    got = oracle.solve_batch(model, T, x1[None], ub[None], nthreads=1)
    st = got["stats"]
    assert np.isfinite(got["x"]).all() and st["max_violation"][0] <= 5e-3 and st["iterations"][0] > 0
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_%s_julia.npz" % case)
    if not os.path.exists(path):
        pytest.skip("no reference-produced fixture for %s: the image has no Julia (tests/golden/make_julia_fixtures.jl is the route)" % case)
    d = np.load(path)
    iters, outer, rollouts = int(d["stats"][4]), int(d["stats"][5]), int(d["stats"][7])
    assert (st["iterations"][0], st["outer_iterations"][0], st["rollouts"][0]) == (iters, outer, rollouts)
    assert np.abs(got["x"][0] - d["x"]).max() <= 1e-8 and np.abs(got["u"][0] - d["u"]).max() <= 1e-8
    K = np.asarray(d["K"])                                   # [t][m][n] padded
    Kmax = max(1.0, np.abs(K).max())
    assert np.abs(got["K"][0].transpose(0, 2, 1) - K).max() <= 1e-6 * Kmax
    assert abs(st["objective"][0] - d["stats"][0]) <= 1e-8 * max(1.0, abs(d["stats"][0]))
