"""Pin the CPU oracle against the reference's own known-answer / property tests.

Each test re-expresses one reference test file (paths relative to
/root/reference); nothing here reads the reference at run time.
"""
import numpy as np
import pytest


def _fd_jac(f, z, eps=1e-6):
    z = np.asarray(z, dtype=float)
    y0 = f(z)
    J = np.zeros((y0.size, z.size))
    for i in range(z.size):
        zp, zm = z.copy(), z.copy()
        zp[i] += eps
        zm[i] -= eps
        J[:, i] = (f(zp) - f(zm)) / (2 * eps)
    return J


# ---------------------------------------------------------------- test/objective.jl
def test_objective_kat(oracle):
    T, n, m = 3, 2, 1
    pr = oracle.Problem("kat_objective", T)
    s = oracle.Solver(pr)
    X = np.ones((T, n)); U = np.ones((T - 1, m))
    s.initialize_states(X); s.initialize_controls(U)
    # cost sum over the horizon — test/objective.jl:35
    J = s.call("cost_bang", 0)
    ot = lambda x, u: x @ x + 0.1 * u @ u
    oT = lambda x: 10.0 * x @ x
    assert J == pytest.approx(sum(ot(X[t], U[t]) for t in range(T - 1)) + oT(X[-1]), abs=1e-12)
    # gradients — test/objective.jl:27-28,33,38-40
    s.call("reset_model_objective"); s.call("gradients")
    gx = s.buffer("gradient_state").reshape(T, n); gu = s.buffer("gradient_action").reshape(T - 1, m)
    assert np.abs(gx[:-1] - 2.0 * X[:-1]).max() < 1e-8
    assert np.abs(gu - 0.2 * U).max() < 1e-8
    assert np.abs(gx[-1] - 20.0 * X[-1]).max() < 1e-8
    # Hessians are unpinned by the reference; check the analytic value + Q1 accumulation
    gxx = s.buffer("hessian_state_state").reshape(T, n, n)
    assert np.allclose(gxx[0], 2.0 * np.eye(n)) and np.allclose(gxx[-1], 20.0 * np.eye(n))
    s.call("gradients")   # second linearisation without reset: `.+=` (src/costs.jl:74)
    gxx2 = s.buffer("hessian_state_state").reshape(T, n, n)
    assert np.allclose(gxx2, 2.0 * gxx)
    assert np.allclose(s.buffer("gradient_state").reshape(T, n), gx)   # gradients are `.=`


# ----------------------------------------------------------------- test/dynamics.jl
def test_dynamics_pendulum_kat(oracle):
    T, n, m = 3, 2, 1
    pr = oracle.Problem("pendulum_euler", T)

    def euler(z):
        x, u = z[:2], z[2:]
        f = np.array([x[1], u[0] / 1.0 - 9.81 * np.sin(x[0]) / 1.0 - 0.1 * x[1] / 1.0])
        return x + 0.1 * f

    x1 = np.ones(n); u1 = np.ones(m)
    U = np.ones((T - 1, m))
    X = pr.rollout(x1, U)
    assert np.linalg.norm(X[1] - euler(np.r_[x1, u1])) < 1e-8        # :32
    s = oracle.Solver(pr)
    s.initialize_states(np.ones((T, n))); s.initialize_controls(U)
    s.call("reset_model_objective"); s.call("gradients")
    fx = s.buffer("jacobian_state").reshape(T - 1, n, n).transpose(0, 2, 1)   # column-major → [row][col]
    fu = s.buffer("jacobian_action").reshape(T - 1, m, n).transpose(0, 2, 1)
    jac_fd = _fd_jac(euler, np.r_[x1, u1])
    for t in range(T - 1):                                           # :37, :45-50
        assert np.linalg.norm(np.c_[fx[t], fu[t]] - jac_fd) < 1e-8


# -------------------------------------------------------------- test/constraints.jl
def test_constraints_kat(oracle):
    T, n, m = 5, 2, 1
    rng = np.random.default_rng(3)
    x = rng.random((T, n)); u = rng.random((T - 1, m))
    pr = oracle.Problem("kat_constraints", T)
    s = oracle.Solver(pr)
    s.initialize_states(x); s.initialize_controls(u)
    s.set_buffer("states", x); s.set_buffer("actions", u)
    s.call("cost_bang", 0)
    c = s.buffer("violations")
    ref = np.concatenate([np.r_[-1.0 - x[t], x[t] - 1.0] for t in range(T - 1)] + [x[-1]])   # :27-33
    assert np.linalg.norm(c - ref) < 1e-8
    # the AL cost with λ=0, ρ=1 only counts active (violated) inequalities; all c<0 here except terminal x>0
    st = s.stats()
    assert st.max_violation == pytest.approx(max(0.0, ref.max()))


# ---------------------------------------------------------------------- test/car.jl
def test_car_solve_property(oracle):
    T = 51
    pr = oracle.Problem("car", T)
    ubar = np.tile(1.0e-2 * np.array([1.0, 0.1]), (T - 1, 1))       # :28
    xbar = pr.rollout(np.zeros(3), ubar)
    s = oracle.Solver(pr); s.initialize_controls(ubar); s.initialize_states(xbar); s.solve()
    x, u = s.get_trajectory()
    tol = 5.0e-3
    for t in range(T - 1):                                          # :74
        e = x[t, :2] - 0.5
        ct = np.r_[-5.0 - u[t], u[t] - 5.0, 0.01 - e @ e]
        assert (ct <= tol).all()
    e = x[-1, :2] - 0.5
    assert 0.01 - e @ e <= tol                                      # :78
    assert (np.abs(x[-1] - [1.0, 1.0, 0.0]) <= tol).all()           # :79
    # fingerprints of an independent restatement
    st = s.stats()
    assert (st.iterations, st.outer_iterations) == (92, 2)
    assert st.objective == pytest.approx(7.250360358, abs=1e-8)
    assert st.max_violation == pytest.approx(1.826e-3, abs=1e-6)
    assert np.allclose(x[-1], [1.000001377963824, 0.9981735209489987, -4.190788e-06], atol=1e-9)
    K = s.buffer("K").reshape(T - 1, 3, 2)
    assert np.allclose(K[0].T, [[-0.079278963437552, -0.028131011408227, 0.003989384779559],
                                [0.018558107298948, -0.052789270448899, -0.082443394459734]], atol=1e-9)


# ------------------------------------------------------------------ test/acrobot.jl
@pytest.mark.parametrize("T,seed", [(51, 0), (51, 1), (101, 0), (101, 7)])
def test_acrobot_solve_property(oracle, T, seed):
    pr = oracle.Problem("acrobot", T)
    ubar = np.random.default_rng(seed).standard_normal((T - 1, 1))  # :88 (unseeded randn in the reference)
    xbar = pr.rollout(np.zeros(4), ubar)
    s = oracle.Solver(pr); s.initialize_controls(ubar); s.initialize_states(xbar); s.solve()
    x, _ = s.get_trajectory()
    assert np.abs(x[-1] - [np.pi, 0, 0, 0]).max() < 5.0e-3          # :114
    assert s.stats().potrf_info == 0


# ------------------------------------------------- examples/particle.jl + Q1 fingerprint
def test_particle_q1_fingerprint(oracle):
    T = 11
    pr = oracle.Problem("particle", T)
    ubar = (0.1 * np.sin(np.arange(1, T))).reshape(T - 1, 1)
    xbar = pr.rollout(np.zeros(2), ubar)
    s = oracle.Solver(pr); s.initialize_controls(ubar); s.initialize_states(xbar); s.enable_trace(); s.solve()
    tr = s.trace()
    g = np.array([r.gradient_norm for r in tr if r.outer == 1])
    # Hessian accumulation (src/costs.jl:74 `.+=`) ⇒ ‖∇L‖∞ ∝ 1/k on an LQ problem
    assert np.allclose(g * np.arange(1, g.size + 1), g[0], rtol=1e-6)
    assert s.stats().objective == pytest.approx(0.1947121908, abs=1e-9)
    x, _ = s.get_trajectory()
    assert np.abs(x[-1] - [1.0, 0.0]).max() < 5e-3


# -------------------------------------------- derivative self-checks of the model zoo
@pytest.mark.parametrize("model,n,m", [("acrobot", 4, 1), ("car", 3, 2), ("particle", 2, 1), ("synth32", 32, 8)])
def test_model_jacobians_fd(oracle, model, n, m):
    T = 3
    pr = oracle.Problem(model, T)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((T, n)) * 0.7; u = rng.standard_normal((T - 1, m)) * 0.7
    s = oracle.Solver(pr); s.initialize_states(x); s.initialize_controls(u)
    s.call("reset_model_objective"); s.call("gradients")
    fx = s.buffer("jacobian_state").reshape(T - 1, n, n).transpose(0, 2, 1)
    fu = s.buffer("jacobian_action").reshape(T - 1, m, n).transpose(0, 2, 1)

    def step(z):
        return pr.rollout(z[:n], np.tile(z[n:], (T - 1, 1)))[1]

    for t in range(T - 1):
        J = _fd_jac(step, np.r_[x[t], u[t]])
        assert np.abs(np.c_[fx[t], fu[t]] - J).max() < 1e-6


def test_backward_pass_vs_dense_riccati(oracle):
    """K,k,P,p of one backward pass vs an independent numpy Riccati on the same fx,fu,g*."""
    T, n, m = 21, 3, 2
    pr = oracle.Problem("car", T)
    rng = np.random.default_rng(5)
    ubar = 0.3 * rng.standard_normal((T - 1, m))
    xbar = pr.rollout(np.array([0.1, -0.2, 0.3]), ubar)
    s = oracle.Solver(pr); s.initialize_controls(ubar); s.initialize_states(xbar)
    s.call("reset_model_objective"); s.call("cost_bang", 0); s.call("gradients"); s.call("backward_pass")
    cm = lambda name, r, c, cnt: s.buffer(name).reshape(cnt, c, r).transpose(0, 2, 1)
    fx, fu = cm("jacobian_state", n, n, T - 1), cm("jacobian_action", n, m, T - 1)
    gx = s.buffer("gradient_state").reshape(T, n); gu = s.buffer("gradient_action").reshape(T - 1, m)
    gxx, guu, gux = cm("hessian_state_state", n, n, T), cm("hessian_action_action", m, m, T - 1), cm("hessian_action_state", m, n, T - 1)
    K, k = cm("K", m, n, T - 1), s.buffer("k").reshape(T - 1, m)
    P, p = gxx[-1].copy(), gx[-1].copy()
    for t in range(T - 2, -1, -1):
        Qx = gx[t] + fx[t].T @ p; Qu = gu[t] + fu[t].T @ p
        Qxx = gxx[t] + fx[t].T @ P @ fx[t]; Quu = guu[t] + fu[t].T @ P @ fu[t]; Qux = gux[t] + fu[t].T @ P @ fx[t]
        Kt = -np.linalg.solve(Quu, Qux); kt = -np.linalg.solve(Quu, Qu)
        assert np.allclose(K[t], Kt, rtol=1e-9, atol=1e-11) and np.allclose(k[t], kt, rtol=1e-9, atol=1e-11)
        P = Qxx + Kt.T @ Quu @ Kt + Kt.T @ Qux + Qux.T @ Kt
        p = Qx + Kt.T @ Quu @ kt + Kt.T @ Qu + Qux.T @ kt
    assert np.allclose(cm("P", n, n, T)[0], P, rtol=1e-8)
