"""Pin the C++ oracle against the fixtures of the SECOND, independent restatement (tests/golden/ref_*.npz —
numpy + sympy derivatives + scipy's real LAPACK dpotrf/dpotrs, tests/golden/reference_restatement.py).

The two restatements share no code: different language, different derivative source (forward-mode duals in the
oracle, symbolic differentiation there), different linear algebra (hand loops vs OpenBLAS). Agreement pins:
    * every stage (gradients!, backward_pass! incl. Qx/Qu/Qxx/Quu/Qux, forward_pass! incl. Δ = ∇Lᵀ·Δz,
      per-trial J, accepted α) on identical inputs — stage tolerance 1e-11 relative (rounding only),
    * the per-iteration trace of whole solves: control flow EXACT (iteration / outer / rollout counts,
      step sizes, status), objective to 1e-9 relative,
    * solutions x, u to 1e-8, gains K to 1e-6·max|K| (SURVEY Appendix C tolerance; observed ≤ 1e-9 / 1e-8).
Nothing here reads /root/reference.
"""
import os
import sys

import numpy as np
import pytest

import refdata
from refdata import CASES, FIELD, colmajor, load, npoints, rel

STAGE_TOL = 1e-11
RICCATI_TOL = 1e-9       # the recursion amplifies rounding over 50-100 steps (observed ≤ 3e-11)


def _oracle_solver(oracle, d):
    pr = oracle.Problem(d["model"], d["T"])
    s = oracle.Solver(pr, oracle.default_options())
    return pr, s


def test_fixture_inputs_are_the_workload_instances():
    """The fixture instances are the first instances of the seeded bench/test workloads (same draws)."""
    from ilqr_amd_loader import load_package
    wl = load_package().workloads
    for case in CASES:
        d = load(case)
        name, tag = case.rsplit("_", 1)
        if not tag.startswith("i"):
            continue
        b = int(tag[1:])
        cfg = {"acrobot": "acrobot", "acrobot51": "acrobot51", "car": "car", "car_goal": "car_goal", "particle": "particle",
               "synth32": "synth32", "synth32_t11": None}[name]
        if cfg is None:
            continue
        model, T, x1, ub = wl.make_inputs(cfg, b + 1)
        assert (model, T) == (d["model"], d["T"])
        if model == "synth32":
            ub = ub + 1.5 * np.sin(0.37 * np.arange(ub[b].size).reshape(ub[b].shape))
        assert np.array_equal(x1[b], d["x1"]) and np.array_equal(ub[b], d["ubar"]), case


@pytest.mark.parametrize("case", CASES)
def test_oracle_whole_solve_matches_reference_fixture(oracle, case):
    d = load(case)
    T = d["T"]
    pr, s = _oracle_solver(oracle, d)
    xb = pr.rollout(d["x1"], d["ubar"])
    assert rel(xb, d["xbar"]) < 1e-13                                        # rollout(), src/rollout.jl:33-42
    s.initialize_controls(d["ubar"]); s.initialize_states(xb)
    s.enable_trace(); s.solve()
    tr = s.trace()
    ref = d["trace"]
    assert len(tr) == ref.shape[0], (len(tr), ref.shape[0])
    got = np.array([[r.outer, r.inner, r.objective, r.gradient_norm, r.max_violation, r.step_size, r.status] for r in tr])
    assert np.array_equal(got[:, [0, 1, 5, 6]], ref[:, [0, 1, 5, 6]])        # outer, inner, step_size, status: exact
    assert np.allclose(got[:, 2], ref[:, 2], rtol=1e-9, atol=1e-12)          # objective
    assert np.allclose(got[:, 4], ref[:, 4], rtol=1e-6, atol=1e-12)          # max_violation
    # ‖∇L‖∞ — tiny near convergence, so relative to the trace's scale
    assert np.abs(got[:, 3] - ref[:, 3]).max() <= 1e-7 * max(1.0, np.abs(ref[:, 3]).max())
    st, rs = s.stats(), d["stats"]
    assert (st.iterations, st.outer_iterations, st.status, st.rollouts, st.potrf_info) == tuple(int(v) for v in rs[4:9])
    x, u = s.get_trajectory()
    assert np.abs(x - d["x"]).max() <= 1e-8 and np.abs(u - d["u"]).max() <= 1e-8
    n, m = pr.nx, pr.nu
    K = s.buffer("K").reshape(T - 1, n, m).transpose(0, 2, 1)
    if "K_steps" in d:
        K = K[d["K_steps"]]
    assert np.abs(K - d["K"]).max() <= 1e-6 * max(1.0, np.abs(d["K"]).max())
    assert np.abs(s.buffer("k").reshape(T - 1, m) - d["k"]).max() <= 1e-7 * max(1.0, np.abs(d["k"]).max())


def _load_pre_state(s, d, j):
    p = "s%d_" % j
    s.set_buffer("nominal_states", colmajor(d[p + "pre_nominal_states"]))
    s.set_buffer("nominal_actions", colmajor(d[p + "pre_nominal_actions"]))
    s.set_buffer("states", colmajor(d[p + "pre_states"]))
    s.set_buffer("actions", colmajor(d[p + "pre_actions"]))
    s.set_buffer("hessian_state_state", colmajor(d[p + "pre_gxx"]))
    s.set_buffer("hessian_action_action", colmajor(d[p + "pre_guu"]))
    s.set_buffer("hessian_action_state", colmajor(d[p + "pre_gux"]))
    s.set_buffer("violations", d[p + "pre_violations"])
    s.set_buffer("constraint_dual", d[p + "pre_dual"])
    s.set_buffer("constraint_penalty", d[p + "pre_penalty"])
    s.set_buffer("active_set", d[p + "pre_active_set"])
    sc = d[p + "pre_scalars"]
    s.call("set_scalars", float(sc[0]), float(sc[1]), float(sc[2]), int(sc[3]))


@pytest.mark.parametrize("case", [c for c in CASES if npoints(load(c)) > 0])
def test_oracle_stages_match_reference_fixture(oracle, case):
    """gradients! -> backward_pass! -> forward_pass! from the fixture's pre-state, each stage restarted from the
    fixture's exact values so that every comparison is on identical inputs."""
    d = load(case)
    T = d["T"]
    for j in range(npoints(d)):
        p = "s%d_" % j
        pr, s = _oracle_solver(oracle, d)
        n, m = pr.nx, pr.nu
        _load_pre_state(s, d, j)
        s.call("gradients")
        for key in ("fx", "fu", "gx", "gu", "gxx", "guu", "gux"):
            assert rel(s.buffer(FIELD[key]), colmajor(d[p + key])) < STAGE_TOL, (case, j, key)
        # backward pass from the fixture's linearisation
        for key in ("fx", "fu", "gx", "gu", "gxx", "guu", "gux"):
            s.set_buffer(FIELD[key], colmajor(d[p + key]))
        s.call("backward_pass"); s.call("lagrangian_gradient")
        for key in ("Qx", "Qu", "Qxx", "Quu", "Qux", "K", "k", "P", "p"):
            assert rel(s.buffer(FIELD[key]), colmajor(d[p + key])) < RICCATI_TOL, (case, j, key)
        g = s.buffer("gradient")
        assert rel(g[:(T - 1) * n], d[p + "Lx"]) < RICCATI_TOL and rel(g[T * n:], d[p + "Lu"]) < RICCATI_TOL
        # forward pass from the fixture's policy
        for key in ("Qx", "Qu", "K", "k", "P", "p"):
            s.set_buffer(FIELD[key], colmajor(d[p + key]))
        r0 = s.stats().rollouts
        s.call("forward_pass")
        f = p + "fwd_"
        delta = s.call("last_delta")
        assert delta == pytest.approx(float(d[f + "delta"][0]), rel=1e-10, abs=1e-12), (case, j)
        st = s.stats()
        J, viol, alpha, status = d[f + "scalars"]
        assert st.step_size == alpha and st.status == int(status), (case, j)
        assert st.rollouts - r0 == d[f + "trial_objectives"].size                       # number of line-search trials
        assert st.objective == pytest.approx(J, rel=1e-11)
        assert st.max_violation == pytest.approx(viol, rel=1e-9, abs=1e-13)
        assert rel(s.buffer("trajectory"), d[f + "trajectory"]) < 1e-10                 # Δz (src/data/methods.jl:42-54)
        for key, name in (("nominal_states", "nominal_states"), ("nominal_actions", "nominal_actions"),
                          ("states", "states"), ("actions", "actions")):
            assert rel(s.buffer(name), colmajor(d[f + key])) < 1e-11, (case, j, key)
        assert rel(s.buffer("violations"), d[f + "violations"]) < 1e-11
        assert np.array_equal(s.buffer("active_set"), d[f + "active_set"])


def test_restatement_reproduces_committed_fixtures():
    """The committed .npz files are what tests/golden/reference_restatement.py computes today (small cases, live)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_reference_fixtures as G
    import reference_restatement as R
    for case in ("particle_sin", "particle_i1", "car_i0"):
        d = load(case)
        model, T, b, tweak, points = G.CASES[case]
        dyn, costs, cons = R.PROBLEMS[model](T)
        s = R.Solver(dyn, costs, cons)
        s.initialize_controls(d["ubar"]); s.initialize_states(R.rollout(dyn, d["x1"], d["ubar"]))
        s.solve()
        tr = np.array(s.trace).reshape(-1, 8)
        assert tr.shape == d["trace"].shape and np.allclose(tr, d["trace"], rtol=1e-12, atol=1e-14)
        assert np.allclose(np.stack(s.nominal_states), d["x"], rtol=0, atol=1e-12)
        assert np.allclose(np.stack(s.K), d["K"], rtol=1e-10, atol=1e-12)


def test_reference_fingerprints_in_fixtures():
    """Facts the reference's own tests assert, evaluated on the fixtures (test/car.jl:74-79, test/acrobot.jl:114),
    plus SURVEY Appendix C's fingerprints of a third, surveyor-side restatement."""
    d = load("car_i0")
    x, u = d["x"], d["u"]
    e = x[:-1, :2] - 0.5
    assert (np.c_[-5.0 - u, u - 5.0, 0.01 - (e * e).sum(1)] <= 5e-3).all()              # test/car.jl:74
    assert 0.01 - ((x[-1, :2] - 0.5) ** 2).sum() <= 5e-3                                # :78
    assert (np.abs(x[-1] - [1.0, 1.0, 0.0]) <= 5e-3).all()                              # :79
    assert int(d["stats"][4]) == 92 and int(d["stats"][5]) == 2
    assert d["stats"][0] == pytest.approx(7.250360358, abs=1e-8)
    assert np.allclose(d["K"][0], [[-0.079278963437552, -0.028131011408227, 0.003989384779559],
                                   [0.018558107298948, -0.052789270448899, -0.082443394459734]], atol=1e-9)
    for case in ("acrobot51_i0", "acrobot_i0", "acrobot_i1", "acrobot_i2"):
        assert np.abs(load(case)["x"][-1] - [np.pi, 0, 0, 0]).max() < 5e-3              # test/acrobot.jl:114
    g = load("particle_sin")["trace"]
    g1 = g[g[:, 0] == 1][:, 3]
    assert np.allclose(g1 * np.arange(1, g1.size + 1), g1[0], rtol=1e-6)                # Q1 fingerprint: ‖∇L‖∞ ∝ 1/k


def test_julia_fixture_route_round_trips(tmp_path):
    """The route to a reference-produced pin (tests/golden/make_julia_fixtures.jl runs the REAL package; no Julia here): its
    input dump and its output converter are exercised by writing an existing fixture in the Julia script's on-disk format
    (column-major arrays + manifest) and converting it back — every key must come back bit for bit, and refdata must then
    prefer the *_julia file."""
    import sys
    sys.path.insert(0, refdata.GOLDEN)
    import dump_fixture_inputs
    import julia_to_npz
    ind = dump_fixture_inputs.main(str(tmp_path / "julia_in"))
    case = "car_i1"
    d = dict(np.load(os.path.join(refdata.GOLDEN, "ref_%s.npz" % case)))
    lines = open(os.path.join(ind, case + ".txt")).read().split("\n")
    assert lines[0] == "car" and int(lines[1]) == 51 and int(lines[2]) == 3
    assert np.array_equal(np.fromfile(os.path.join(ind, case + ".u.f64")).reshape(50, 2), d["ubar"])
    src = tmp_path / "julia_out" / case
    src.mkdir(parents=True)
    with open(src / "manifest.txt", "w") as man:
        for key, a in d.items():
            a = np.asarray(a, dtype=np.float64)
            if a.ndim == 3:
                jl = a.transpose(1, 2, 0)              # [t][row][col] -> Julia (rows, cols, T)
            elif a.ndim == 2:
                jl = a.T                               # [t][i] -> Julia (n, T)
            else:
                jl = a
            jl.ravel(order="F").astype("<f8").tofile(src / (key + ".f64"))
            man.write("%s %s\n" % (key, " ".join(str(v) for v in jl.shape)))
    assert julia_to_npz.main(str(tmp_path / "julia_out"), str(tmp_path)) == [case]
    back = dict(np.load(tmp_path / ("ref_%s_julia.npz" % case)))
    assert set(back) == set(d)
    for key in d:
        assert back[key].shape == d[key].shape and np.array_equal(back[key], np.asarray(d[key], dtype=back[key].dtype)), key
    assert refdata.source_of(case) == "restatement"     # nothing of the kind is committed: the image has no Julia
