"""Built-in model zoo: the reference's example problems written the way a user
of the reference writes them (plain functions of x, u on symbolic vectors).

particle  examples/particle.jl:17-43      acrobot  test/acrobot.jl:9-101
car       test/car.jl:10-61               pendulum test/dynamics.jl:8-19 (KAT)
synth32   SURVEY.md §8(d) C5 (synthetic nx=32, nu=8 with action box)
"""
import math

import sympy as sp

from .codegen import Constraint, Cost, Dynamics, generate_model_source


def _dot(a, b):
    return sum(ai * bi for ai, bi in zip(a, b))


# ---------------------------------------------------------------- particle
def particle_discrete(x, u):
    A = [[1.0, 1.0], [0.0, 1.0]]
    Bm = [0.0, 1.0]
    return [A[i][0] * x[0] + A[i][1] * x[1] + Bm[i] * u[0] for i in range(2)]


def particle():
    xT = [1.0, 0.0]
    return dict(
        dynamics=Dynamics(particle_discrete, 2, 1),
        cost_stage=Cost(lambda x, u: 0.1 * _dot(x, x) + 0.1 * _dot(u, u), 2, 1),
        cost_term=Cost(lambda x, u: 0.1 * _dot(x, x), 2, 0),
        con_stage=Constraint(),
        con_term=Constraint(lambda x, u: [x[i] - xT[i] for i in range(2)], 2, 0),
    )


# ---------------------------------------------------------------- pendulum (explicit Euler)
def pendulum_euler_discrete(x, u):
    mass, lc, gravity, damping, h = 1.0, 1.0, 9.81, 0.1, 0.1
    f = [x[1], u[0] / (mass * lc * lc) - gravity * sp.sin(x[0]) / lc - damping * x[1] / (mass * lc * lc)]
    return [x[i] + h * f[i] for i in range(2)]


def pendulum_euler():
    return dict(
        dynamics=Dynamics(pendulum_euler_discrete, 2, 1),
        cost_stage=Cost(lambda x, u: _dot(x, x) + 0.1 * _dot(u, u), 2, 1),
        cost_term=Cost(lambda x, u: 10.0 * _dot(x, x), 2, 0),
        con_stage=Constraint(), con_term=Constraint(),
    )


# ---------------------------------------------------------------- acrobot
def acrobot_continuous(x, u):
    mass1, inertia1, length1, lengthcom1 = 1.0, 0.33, 1.0, 0.5
    mass2, inertia2, length2, lengthcom2 = 1.0, 0.33, 1.0, 0.5
    gravity, friction1, friction2 = 9.81, 0.1, 0.1
    q = x[0:2]
    v = x[2:4]
    a = inertia1 + inertia2 + mass2 * length1 * length1 + 2.0 * mass2 * length1 * lengthcom2 * sp.cos(q[1])
    b = inertia2 + mass2 * length1 * lengthcom2 * sp.cos(q[1])
    c = inertia2
    det = a * c - b * b
    Minv = [[c / det, -b / det], [-b / det, a / det]]
    tau = [-1.0 * mass1 * gravity * lengthcom1 * sp.sin(q[0])
           - mass2 * gravity * (length1 * sp.sin(q[0]) + lengthcom2 * sp.sin(q[0] + q[1])),
           -1.0 * mass2 * gravity * lengthcom2 * sp.sin(q[0] + q[1])]
    Cm = [[-2.0 * mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[3], -1.0 * mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[3]],
          [mass2 * length1 * lengthcom2 * sp.sin(x[1]) * x[2], 0.0]]
    Bv = [0.0, 1.0]
    fr = [friction1, friction2]
    rhs = [-1.0 * (Cm[i][0] * v[0] + Cm[i][1] * v[1]) + tau[i] + Bv[i] * u[0] - fr[i] * v[i] for i in range(2)]
    qdd = [Minv[i][0] * rhs[0] + Minv[i][1] * rhs[1] for i in range(2)]
    return [x[2], x[3], qdd[0], qdd[1]]


def _midpoint(fc, n, h):
    def f(x, u):
        k1 = fc(x, u)
        xm = [x[i] + 0.5 * h * k1[i] for i in range(n)]
        k2 = fc(xm, u)
        return [x[i] + h * k2[i] for i in range(n)]
    return f


def acrobot():
    xT = [math.pi, 0.0, 0.0, 0.0]
    return dict(
        dynamics=Dynamics(_midpoint(acrobot_continuous, 4, 0.1), 4, 1),
        cost_stage=Cost(lambda x, u: 0.1 * _dot(x[2:4], x[2:4]) + 0.1 * _dot(u, u), 4, 1),
        cost_term=Cost(lambda x, u: 0.1 * _dot(x[2:4], x[2:4]), 4, 0),
        con_stage=Constraint(),
        con_term=Constraint(lambda x, u: [x[i] - xT[i] for i in range(4)], 4, 0),
    )


# ---------------------------------------------------------------- car
def car_continuous(x, u):
    return [u[0] * sp.cos(x[2]), u[0] * sp.sin(x[2]), u[1]]


def car(goal_only=False):
    xT = [1.0, 1.0, 0.0]
    ul, uu = [-5.0, -5.0], [5.0, 5.0]
    p_obs, r_obs = [0.5, 0.5], 0.1

    def e(x):
        return [x[0] - p_obs[0], x[1] - p_obs[1]]

    def stage(x, u):
        ee = e(x)
        return [ul[0] - u[0], ul[1] - u[1], u[0] - uu[0], u[1] - uu[1], r_obs ** 2.0 - _dot(ee, ee)]

    def term(x, u):
        ee = e(x)
        return [x[0] - xT[0], x[1] - xT[1], x[2] - xT[2], r_obs ** 2.0 - _dot(ee, ee)]

    d = dict(
        dynamics=Dynamics(_midpoint(car_continuous, 3, 0.1), 3, 2),
        cost_stage=Cost(lambda x, u: 1.0 * _dot([x[i] - xT[i] for i in range(3)], [x[i] - xT[i] for i in range(3)])
                        + 1.0e-2 * _dot(u, u), 3, 2),
        cost_term=Cost(lambda x, u: 1000.0 * _dot([x[i] - xT[i] for i in range(3)], [x[i] - xT[i] for i in range(3)]), 3, 0),
    )
    if goal_only:
        d["con_stage"] = Constraint()
        d["con_term"] = Constraint(lambda x, u: [x[i] - xT[i] for i in range(3)], 3, 0)
    else:
        d["con_stage"] = Constraint(stage, 3, 2, indices_inequality=[1, 2, 3, 4, 5])
        d["con_term"] = Constraint(term, 3, 0, indices_inequality=[4])
    return d


def car_obs():
    """car with the obstacle centre as a per-timestep parameter θ_t = (p_x, p_y) (README.md:19,28 of the
    reference: "parameters"); everything else as test/car.jl."""
    xT = [1.0, 1.0, 0.0]
    ul, uu, r_obs = [-5.0, -5.0], [5.0, 5.0], 0.1

    def e(x, w):
        return [x[0] - w[0], x[1] - w[1]]

    def stage(x, u, w):
        ee = e(x, w)
        return [ul[0] - u[0], ul[1] - u[1], u[0] - uu[0], u[1] - uu[1], r_obs ** 2.0 - _dot(ee, ee)]

    def term(x, u, w):
        ee = e(x, w)
        return [x[0] - xT[0], x[1] - xT[1], x[2] - xT[2], r_obs ** 2.0 - _dot(ee, ee)]

    fm = _midpoint(car_continuous, 3, 0.1)
    return dict(
        dynamics=Dynamics(lambda x, u, w: fm(x, u), 3, 2, num_parameter=2),
        cost_stage=Cost(lambda x, u, w: 1.0 * _dot([x[i] - xT[i] for i in range(3)], [x[i] - xT[i] for i in range(3)])
                        + 1.0e-2 * _dot(u, u), 3, 2, num_parameter=2),
        cost_term=Cost(lambda x, u, w: 1000.0 * _dot([x[i] - xT[i] for i in range(3)], [x[i] - xT[i] for i in range(3)]),
                       3, 0, num_parameter=2),
        con_stage=Constraint(stage, 3, 2, indices_inequality=[1, 2, 3, 4, 5], num_parameter=2),
        con_term=Constraint(term, 3, 0, indices_inequality=[4], num_parameter=2),
    )


def car_tv(T):
    """Time-varying stage objects over the car (uniform dimensions), the way the reference is given Vectors of
    objects (README.md:26): dynamics kind by t % 3 (midpoint h = 0.1 / explicit Euler h = 0.05), stage cost by
    halves of the horizon, stage constraint by t % 4 (test/car.jl's five inequalities / none / one equality / none).
    Returns (dynamics[T-1], costs[T], constraints[T]) for Solver(dynamics, costs, constraints)."""
    base = car()
    xT = [1.0, 1.0, 0.0]
    dyn_a = base["dynamics"]
    dyn_b = Dynamics(lambda x, u: [x[i] + 0.05 * car_continuous(x, u)[i] for i in range(3)], 3, 2)
    cost_a = base["cost_stage"]
    q, xg, r = [5.0, 2.0, 0.5], [0.9, 1.1, 0.2], [0.05, 0.02]
    cost_b = Cost(lambda x, u: sum(q[i] * (x[i] - xg[i]) ** 2 for i in range(3)) + sum(r[j] * u[j] ** 2 for j in range(2)), 3, 2)
    con_a = base["con_stage"]
    con_none = Constraint()
    con_eq = Constraint(lambda x, u: [u[1] - 0.3 * x[2] - 0.05], 3, 2)
    dynamics = [dyn_b if t % 3 == 2 else dyn_a for t in range(T - 1)]
    costs = [cost_b if 2 * t >= T - 1 else cost_a for t in range(T - 1)] + [base["cost_term"]]
    constraints = [con_a if t % 4 == 0 else (con_eq if t % 4 == 2 else con_none) for t in range(T - 1)] + [base["con_term"]]
    return dynamics, costs, constraints


# ---------------------------------------------------------------- ragged: time-varying DIMENSIONS (src/dynamics.jl:5-7, README.md:26)
RAGGED_N, RAGGED_M = [3, 3, 4, 4, 2, 2, 3, 3], [2, 1, 2, 1, 1, 2, 2, 1]


def _ragged_tables(n0, m0, n1):
    A = [[(0.9 if i == j else 0.0) + 0.1 * math.cos(1.0 + i + 2 * j + n0) for j in range(n0)] for i in range(n1)]
    Bm = [[0.3 * math.sin(2.0 + 3 * i + j + m0) for j in range(m0)] for i in range(n1)]
    return A, Bm


def ragged(T):
    """num_state = 3,3,4,4,2,2,3,3 | ..., num_action = 2,1,2,1,1,2,2,1 | ... (period 8): y_i = sum_j A_ij x_j + sum_j B_ij u_j +
    [i == 0] 0.1 sin x_0 per step, quadratic costs, a terminal equality on the first two states. Twins: oracle/models.cpp "ragged",
    tests/golden/reference_restatement.py:ragged_problem. Returns (dynamics[T-1], costs[T], constraints[T], state_dims, action_dims)."""
    n_t = [RAGGED_N[t % 8] for t in range(T)]
    m_t = [RAGGED_M[t % 8] for t in range(T - 1)]
    dyn_cache, cost_cache = {}, {}

    def dyn(n0, m0, n1):
        if (n0, m0, n1) not in dyn_cache:
            A, Bm = _ragged_tables(n0, m0, n1)
            dyn_cache[(n0, m0, n1)] = Dynamics(
                lambda x, u: [sum(A[i][j] * x[j] for j in range(n0)) + sum(Bm[i][j] * u[j] for j in range(m0))
                              + (0.1 * sp.sin(x[0]) if i == 0 else 0.0) for i in range(n1)], n0, m0)
        return dyn_cache[(n0, m0, n1)]

    def cost(n0, m0):
        if (n0, m0) not in cost_cache:
            cost_cache[(n0, m0)] = Cost(lambda x, u: 0.5 * sum((1.0 + 0.1 * i) * x[i] * x[i] for i in range(n0))
                                        + 0.05 * sum((1.0 + j) * u[j] * u[j] for j in range(m0)), n0, m0)
        return cost_cache[(n0, m0)]

    dynamics = [dyn(n_t[t], m_t[t], n_t[t + 1]) for t in range(T - 1)]
    costs = [cost(n_t[t], m_t[t]) for t in range(T - 1)] + [Cost(lambda x, u: 5.0 * sum(x[i] * x[i] for i in range(n_t[-1])), n_t[-1], 0)]
    none = Constraint()
    goal = Constraint(lambda x, u: [x[0] - 0.2, x[1] + 0.1], n_t[-1], 0)
    return dynamics, costs, [none] * (T - 1) + [goal], n_t, m_t


def ragged_c_stages(T):
    """The same problem as C source of the reference's callables PER KIND, for ilqr_compile_model_stages — what a Julia or C host
    hands over when the objects of a Solver differ along the horizon. Returns (StageKinds, source)."""
    from . import _ffi
    n_t = [RAGGED_N[t % 8] for t in range(T)]
    m_t = [RAGGED_M[t % 8] for t in range(T - 1)]
    dk, ck, di, ci = [], [], [], []
    for t in range(T - 1):
        kd, kc = (n_t[t], m_t[t], n_t[t + 1]), (n_t[t], m_t[t])
        if kd not in dk:
            dk.append(kd)
        if kc not in ck:
            ck.append(kc)
        di.append(dk.index(kd)); ci.append(ck.index(kc))
    src = ["/* GENERATED by iterativelqr.jl_amd/models.py:ragged_c_stages: per-kind callables, each in its own dimensions, `out` column-major and zeroed */"]
    g = lambda v: "%.17g" % v
    for q, (n0, m0, n1) in enumerate(dk):
        A, Bm = _ragged_tables(n0, m0, n1)
        body = []
        for i in range(n1):
            terms = ["%s * x[%d]" % (g(A[i][j]), j) for j in range(n0)] + ["%s * u[%d]" % (g(Bm[i][j]), j) for j in range(m0)]
            body.append("    y[%d] = %s%s;" % (i, " + ".join(terms), " + 0.1 * sin(x[0])" if i == 0 else ""))
        src.append("ILQR_MODEL_FN void dynamics_%d(double* y, const double* x, const double* u, const double* w) {\n%s\n}" % (q, "\n".join(body)))
        jx = ["    fx[%d] = %s%s;" % (j * n1 + i, g(A[i][j]), " + 0.1 * cos(x[0])" if (i == 0 and j == 0) else "") for j in range(n0) for i in range(n1)]
        src.append("ILQR_MODEL_FN void dynamics_%d_jacobian_state(double* fx, const double* x, const double* u, const double* w) {\n%s\n}" % (q, "\n".join(jx)))
        ju = ["    fu[%d] = %s;" % (j * n1 + i, g(Bm[i][j])) for j in range(m0) for i in range(n1)]
        src.append("ILQR_MODEL_FN void dynamics_%d_jacobian_action(double* fu, const double* x, const double* u, const double* w) {\n%s\n}" % (q, "\n".join(ju)))
    sig = "(double* o, const double* x, const double* u, const double* w)"
    for q, (n0, m0) in enumerate(ck):
        qx = [0.5 * (1.0 + 0.1 * i) for i in range(n0)]
        ru = [0.05 * (1.0 + j) for j in range(m0)]
        ell = " + ".join(["%s * x[%d] * x[%d]" % (g(qx[i]), i, i) for i in range(n0)] + ["%s * u[%d] * u[%d]" % (g(ru[j]), j, j) for j in range(m0)])
        src.append("ILQR_MODEL_FN void cost_stage_%d%s { o[0] = %s; }" % (q, sig, ell))
        src.append("ILQR_MODEL_FN void cost_stage_%d_gradient_state%s { %s }" % (q, sig, " ".join("o[%d] = %s * x[%d];" % (i, g(2.0 * qx[i]), i) for i in range(n0))))
        src.append("ILQR_MODEL_FN void cost_stage_%d_gradient_action%s { %s }" % (q, sig, " ".join("o[%d] = %s * u[%d];" % (j, g(2.0 * ru[j]), j) for j in range(m0))))
        src.append("ILQR_MODEL_FN void cost_stage_%d_hessian_state_state%s { %s }" % (q, sig, " ".join("o[%d] = %s;" % (i * n0 + i, g(2.0 * qx[i])) for i in range(n0))))
        src.append("ILQR_MODEL_FN void cost_stage_%d_hessian_action_action%s { %s }" % (q, sig, " ".join("o[%d] = %s;" % (j * m0 + j, g(2.0 * ru[j])) for j in range(m0))))
        src.append("ILQR_MODEL_FN void cost_stage_%d_hessian_action_state%s { }" % (q, sig))
    nT = n_t[-1]
    src.append("ILQR_MODEL_FN void cost_terminal%s { o[0] = %s; }" % (sig, " + ".join("5.0 * x[%d] * x[%d]" % (i, i) for i in range(nT))))
    src.append("ILQR_MODEL_FN void cost_terminal_gradient_state%s { %s }" % (sig, " ".join("o[%d] = 10.0 * x[%d];" % (i, i) for i in range(nT))))
    src.append("ILQR_MODEL_FN void cost_terminal_hessian_state_state%s { %s }" % (sig, " ".join("o[%d] = 10.0;" % (i * nT + i) for i in range(nT))))
    src.append("ILQR_MODEL_FN void constraint_terminal%s { o[0] = x[0] - 0.2; o[1] = x[1] + 0.1; }" % sig)
    src.append("ILQR_MODEL_FN void constraint_terminal_jacobian_state%s { o[0] = 1.0; o[3] = 1.0; }" % sig)     # 2 x nT column-major: (0,0), (1,1)
    kinds = _ffi.stage_kinds(T, 0, dk, di, ck, ci, [], [], nT, 2, 0)
    return kinds, "\n".join(src) + "\n"


# ---------------------------------------------------------------- synth32
def synth32():
    """x⁺ = x + h(Ax + Bu + 0.1 sin x), nx = 32, nu = 8, action box as 16 stage inequalities."""
    n, m, h = 32, 8, 0.05
    A = [[(-1.0 if i == j else 0.0) + 0.3 * math.cos(float((i + 1) + 2 * (j + 1))) / 32.0 for j in range(n)] for i in range(n)]
    Bm = [[math.sin(float(3 * (i + 1) + (j + 1))) / math.sqrt(32.0) for j in range(m)] for i in range(n)]

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

    xg = 0.5
    return dict(
        dynamics=Dynamics(f, n, m),
        cost_stage=Cost(lambda x, u: 0.1 * sum((xi - xg) * (xi - xg) for xi in x) + 0.01 * _dot(u, u), n, m),
        cost_term=Cost(lambda x, u: 10.0 * sum((xi - xg) * (xi - xg) for xi in x), n, 0),
        con_stage=Constraint(lambda x, u: [-1.0 - u[j] for j in range(m)] + [u[j] - 1.0 for j in range(m)], n, m,
                             indices_inequality=list(range(1, 2 * m + 1))),
        con_term=Constraint(),
    )


def synth_nm(n, m):
    """The synth32 family at other sizes (x⁺ = x + h(Ax + Bu + 0.1 sin x), action box as 2 nu stage inequalities): the large
    path takes nx <= 48, nu <= 16."""
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

    xg = 0.5
    return dict(
        dynamics=Dynamics(f, n, m),
        cost_stage=Cost(lambda x, u: 0.1 * sum((xi - xg) * (xi - xg) for xi in x) + 0.01 * _dot(u, u), n, m),
        cost_term=Cost(lambda x, u: 10.0 * sum((xi - xg) * (xi - xg) for xi in x), n, 0),
        con_stage=Constraint(lambda x, u: [-1.0 - u[j] for j in range(m)] + [u[j] - 1.0 for j in range(m)], n, m,
                             indices_inequality=list(range(1, 2 * m + 1))),
        con_term=Constraint(),
    )


def synth_box(n, m, xmax=1.2):
    """synth_nm with a state box on top of the action box: 2 nu + 2 nx stage inequalities — more than 64 rows from nx + nu > 32
    (the reference has no row limit, src/constraints.jl:54-64; here the inequality masks become word arrays)."""
    base = synth_nm(n, m)
    rows = lambda x, u: ([-1.0 - u[j] for j in range(m)] + [u[j] - 1.0 for j in range(m)] +
                         [x[i] - xmax for i in range(n)] + [-xmax - x[i] for i in range(n)])
    base["con_stage"] = Constraint(rows, n, m, indices_inequality=list(range(1, 2 * m + 2 * n + 1)))
    return base


def synth_c_source(n=32, m=8):
    """The synth family (synth_nm) as C source of the reference's callables (src/dynamics.jl:55-60, src/costs.jl:1-15,
    src/constraints.jl:54-64) for ilqr_compile_model — what a host without the symbolic generator hands over. The coefficient
    matrices are literal tables, as Symbolics' build_function would print them (an earlier version computed every entry with
    cos / sin at run time: 1280 libm calls per dynamics evaluation, 15 ms per rollout of BASELINE config 5's shard). The Jacobian
    callables stay dense loops: finding the 2 nx constant-free entries is the library's job (probe_model_structure).
    examples/synth32_model.c is synth_c_source(32, 8) (tests/test_structure_probe.py keeps it so)."""
    h = 0.05
    A = [[(-1.0 if i == j else 0.0) + 0.3 * math.cos(float((i + 1) + 2 * (j + 1))) / float(n) for j in range(n)] for i in range(n)]
    Bm = [[math.sin(float(3 * (i + 1) + (j + 1))) / math.sqrt(float(n)) for j in range(m)] for i in range(n)]
    tab = lambda rows: ",\n".join("    " + ", ".join("%.17g" % v for v in r) for r in rows)
    return """/* GENERATED by iterativelqr.jl_amd/models.py:synth_c_source(%(n)d, %(m)d) - the reference's callable contract (src/dynamics.jl:55-60,
 * src/costs.jl:1-15, src/constraints.jl:54-64) for the synth family: x+ = x + h (A x + B u + 0.1 sin x), l = 0.1 |x - 0.5|^2 + 0.01 |u|^2,
 * l_T = 10 |x - 0.5|^2, action box as 2 nu stage inequalities; nx = %(n)d, nu = %(m)d is BASELINE config 5's model for (32, 8) - the twin
 * of models.py:synth_nm and of oracle/models.cpp "synth32". Handed to ilqr_compile_model as text: the Jacobians are dense loops, the
 * library finds the constant entries and the structurally non-zero Hessian entries by probing them on the host. `out` arrives zeroed. */
#define SN %(n)d
#define SM %(m)d
#define SH %(h)r
static const double S_A[SN * SN] = {   /* row-major */
%(A)s};
static const double S_B[SN * SM] = {   /* row-major */
%(B)s};
ILQR_MODEL_FN void dynamics(double* y, const double* x, const double* u, const double* w) {
    for (int i = 0; i < SN; ++i) {
        double acc = 0.0;
        for (int j = 0; j < SN; ++j) acc += S_A[i * SN + j] * x[j];
        for (int j = 0; j < SM; ++j) acc += S_B[i * SM + j] * u[j];
        acc += 0.1 * sin(x[i]);
        y[i] = x[i] + SH * acc;
    }
}
ILQR_MODEL_FN void dynamics_jacobian_state(double* fx, const double* x, const double* u, const double* w) {   /* column-major n x n */
    for (int j = 0; j < SN; ++j)
        for (int i = 0; i < SN; ++i)
            fx[j * SN + i] = (i == j ? 1.0 : 0.0) + SH * (S_A[i * SN + j] + (i == j ? 0.1 * cos(x[i]) : 0.0));
}
ILQR_MODEL_FN void dynamics_jacobian_action(double* fu, const double* x, const double* u, const double* w) {  /* column-major n x m */
    for (int j = 0; j < SM; ++j)
        for (int i = 0; i < SN; ++i) fu[j * SN + i] = SH * S_B[i * SM + j];
}
ILQR_MODEL_FN void cost_stage(double* l, const double* x, const double* u, const double* w) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < SN; ++i) a += (x[i] - 0.5) * (x[i] - 0.5);
    for (int j = 0; j < SM; ++j) b += u[j] * u[j];
    l[0] = 0.1 * a + 0.01 * b;
}
ILQR_MODEL_FN void cost_stage_gradient_state(double* g, const double* x, const double* u, const double* w) { for (int i = 0; i < SN; ++i) g[i] = 0.2 * (x[i] - 0.5); }
ILQR_MODEL_FN void cost_stage_gradient_action(double* g, const double* x, const double* u, const double* w) { for (int j = 0; j < SM; ++j) g[j] = 0.02 * u[j]; }
ILQR_MODEL_FN void cost_stage_hessian_state_state(double* h, const double* x, const double* u, const double* w) { for (int i = 0; i < SN; ++i) h[i * SN + i] = 0.2; }
ILQR_MODEL_FN void cost_stage_hessian_action_action(double* h, const double* x, const double* u, const double* w) { for (int j = 0; j < SM; ++j) h[j * SM + j] = 0.02; }
ILQR_MODEL_FN void cost_stage_hessian_action_state(double* h, const double* x, const double* u, const double* w) { }
ILQR_MODEL_FN void cost_terminal(double* l, const double* x, const double* u, const double* w) {
    double a = 0.0;
    for (int i = 0; i < SN; ++i) a += (x[i] - 0.5) * (x[i] - 0.5);
    l[0] = 10.0 * a;
}
ILQR_MODEL_FN void cost_terminal_gradient_state(double* g, const double* x, const double* u, const double* w) { for (int i = 0; i < SN; ++i) g[i] = 20.0 * (x[i] - 0.5); }
ILQR_MODEL_FN void cost_terminal_hessian_state_state(double* h, const double* x, const double* u, const double* w) { for (int i = 0; i < SN; ++i) h[i * SN + i] = 20.0; }
/* stage: the action box -1 <= u <= 1 as 2 m inequalities */
ILQR_MODEL_FN void constraint_stage(double* c, const double* x, const double* u, const double* w) {
    for (int j = 0; j < SM; ++j) { c[j] = -1.0 - u[j]; c[SM + j] = u[j] - 1.0; }
}
ILQR_MODEL_FN void constraint_stage_jacobian_state(double* cx, const double* x, const double* u, const double* w) { }
ILQR_MODEL_FN void constraint_stage_jacobian_action(double* cu, const double* x, const double* u, const double* w) {   /* column-major 2m x m */
    for (int j = 0; j < SM; ++j) { cu[j * 2 * SM + j] = -1.0; cu[j * 2 * SM + SM + j] = 1.0; }
}
""" % dict(n=n, m=m, h=h, A=tab(A), B=tab(Bm))


def synth12():
    """A second large-path model whose dimensions are NOT multiples of the MFMA tile (nx = 12, nu = 5), with a
    bilinear term (state-dependent fu entries) and a terminal equality; twin of oracle/models.cpp "synth12"."""
    n, m, h = 12, 5, 0.05
    A = [[(-1.0 if i == j else 0.0) + 0.3 * math.cos(float((i + 1) + 2 * (j + 1))) / 12.0 for j in range(n)] for i in range(n)]
    Bm = [[math.sin(float(3 * (i + 1) + (j + 1))) / math.sqrt(12.0) for j in range(m)] for i in range(n)]

    def f(x, u):
        out = []
        for i in range(n):
            acc = 0.0
            for j in range(n):
                acc = acc + A[i][j] * x[j]
            for j in range(m):
                acc = acc + Bm[i][j] * u[j]
            acc = acc + 0.1 * sp.sin(x[i])
            acc = acc + 0.02 * (x[i] * u[i % m])
            out.append(x[i] + h * acc)
        return out

    xg = 0.5
    return dict(
        dynamics=Dynamics(f, n, m),
        cost_stage=Cost(lambda x, u: 0.1 * sum((xi - xg) * (xi - xg) for xi in x) + 0.01 * _dot(u, u), n, m),
        cost_term=Cost(lambda x, u: 10.0 * sum((xi - xg) * (xi - xg) for xi in x), n, 0),
        con_stage=Constraint(lambda x, u: [-1.0 - u[j] for j in range(m)] + [u[j] - 1.0 for j in range(m)], n, m,
                             indices_inequality=list(range(1, 2 * m + 1))),
        con_term=Constraint(lambda x, u: [x[i] - 0.1 for i in range(3)], n, 0),
    )


BUILTIN = {
    "particle": particle,
    "pendulum_euler": pendulum_euler,
    "acrobot": acrobot,
    "car": car,
    "car_goal": lambda: car(goal_only=True),
    "car_obs": car_obs,
    "synth32": synth32,
    "synth12": synth12,        # nx = 12, nu = 5: every matrix ONE 16x16 tile (the one-wave variant of the large path exists for it)
}


def builtin_source(name):
    return generate_model_source(name, **BUILTIN[name]())
