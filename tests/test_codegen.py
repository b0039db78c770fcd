"""The model code generator (sympy twin of the reference's Symbolics constructors,
src/dynamics.jl:16-34, src/costs.jl:17-44, src/constraints.jl:17-43): the generated device
structs are compiled for the HOST and compared with the oracle's independently derived
model zoo (dual-number Jacobians) on random points."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from ilqr_amd_loader import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r'''
#include <cmath>
#define __device__
#define __forceinline__ inline
using std::sin; using std::cos; using std::fma; using std::fabs; using std::rint;
#include "ilqr_math.hpp"
#include "%(header)s"
typedef %(struct)s M;
template <int N> struct cd { static const int v = N > 0 ? N : 1; };
static double g_w[16] = {0};
#define LOADXU double xx[M::NX], uu[cd<M::NU>::v], w[cd<M::NW>::v]; \
  for (int i=0;i<M::NX;++i) xx[i]=x[i]; for (int i=0;i<M::NU;++i) uu[i]=u[i]; for (int i=0;i<cd<M::NW>::v;++i) w[i]=g_w[i];
#define LOADX double xx[M::NX], w[cd<M::NW>::v]; \
  for (int i=0;i<M::NX;++i) xx[i]=x[i]; for (int i=0;i<cd<M::NW>::v;++i) w[i]=g_w[i];
extern "C" {
void set_w(const double* w, int n) { for (int i = 0; i < n; ++i) g_w[i] = w[i]; }
int dims(int* o) { o[0]=M::NX; o[1]=M::NU; o[2]=M::NCS; o[3]=M::NCT; o[4]=M::NW; return 0; }
unsigned long long ineq(int term) { return term ? M::INEQ_T : M::INEQ_S; }
void dyn(const double* x, const double* u, double* y) { LOADXU
  double yy[M::NX]; M::dyn(xx,uu,w,yy); for (int i=0;i<M::NX;++i) y[i]=yy[i]; }
void dyn_jac(const double* x, const double* u, double* fx, double* fu) { LOADXU
  double a[M::NX*M::NX], b[M::NX*cd<M::NU>::v];
  M::dyn_jac(xx,uu,w,a,b); for (int i=0;i<M::NX*M::NX;++i) fx[i]=a[i]; for (int i=0;i<M::NX*M::NU;++i) fu[i]=b[i]; }
double cost_s(const double* x, const double* u) { LOADXU return M::cost_s(xx,uu,w); }
double cost_t(const double* x) { LOADX return M::cost_t(xx,w); }
void cost_s_grad(const double* x, const double* u, double* gx, double* gu) { LOADXU
  double a[M::NX], b[cd<M::NU>::v];
  M::cost_s_grad(xx,uu,w,a,b); for (int i=0;i<M::NX;++i) gx[i]=a[i]; for (int i=0;i<M::NU;++i) gu[i]=b[i]; }
void cost_s_hess(const double* x, const double* u, double* gxx, double* guu, double* gux) { LOADXU
  double a[M::NX*M::NX], b[cd<M::NU*M::NU>::v], c[cd<M::NU*M::NX>::v];
  M::cost_s_hess(xx,uu,w,a,b,c); for (int i=0;i<M::NX*M::NX;++i) gxx[i]=a[i];
  for (int i=0;i<M::NU*M::NU;++i) guu[i]=b[i]; for (int i=0;i<M::NU*M::NX;++i) gux[i]=c[i]; }
void con_s(const double* x, const double* u, double* c) { LOADXU
  double cc[cd<M::NCS>::v]; M::con_s(xx,uu,w,cc); for (int i=0;i<M::NCS;++i) c[i]=cc[i]; }
void con_s_jac(const double* x, const double* u, double* cx, double* cu) { LOADXU
  double a[cd<M::NCS*M::NX>::v], b[cd<M::NCS*M::NU>::v];
  M::con_s_jac(xx,uu,w,a,b); for (int i=0;i<M::NCS*M::NX;++i) cx[i]=a[i]; for (int i=0;i<M::NCS*M::NU;++i) cu[i]=b[i]; }
void con_t(const double* x, double* c) { LOADX
  double cc[cd<M::NCT>::v]; M::con_t(xx,w,cc); for (int i=0;i<M::NCT;++i) c[i]=cc[i]; }
void con_t_jac(const double* x, double* cx) { LOADX
  double a[cd<M::NCT*M::NX>::v]; M::con_t_jac(xx,w,a); for (int i=0;i<M::NCT*M::NX;++i) cx[i]=a[i]; }
}
'''


def _build(tmp, header, struct):
    src = tmp / ("h_%s.cpp" % struct)
    src.write_text(HARNESS % dict(header=header, struct=struct))
    so = tmp / ("h_%s.so" % struct)
    subprocess.check_call(["g++", "-O1", "-mfma", "-std=c++17", "-shared", "-fPIC", "-w",
                           "-I", os.path.join(ROOT, "iterativelqr.jl_amd", "csrc"), str(src), "-o", str(so)])
    L = ctypes.CDLL(str(so))
    L.cost_s.restype = ctypes.c_double
    L.cost_t.restype = ctypes.c_double
    L.ineq.restype = ctypes.c_ulonglong
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@pytest.mark.parametrize("model", ["particle", "pendulum_euler", "acrobot", "car", "car_goal", "car_obs", "synth32"])
def test_generated_builtin_matches_oracle(tmp_path, oracle, model):
    hdr = os.path.join(ROOT, "iterativelqr.jl_amd", "csrc", "models", "model_%s.h" % model)
    L = _build(tmp_path, hdr, "Model_" + model)
    d = (ctypes.c_int * 5)(); L.dims(d)
    n, m, ncs, nct, nw = list(d)
    T = 3
    pr = oracle.Problem(model, T)
    assert (pr.nx, pr.nu, pr.c.nw) == (n, m, nw)
    rng = np.random.default_rng(2)
    for trial in range(5):
        x = rng.standard_normal((T, n)) * (1.0 if trial else 3.0); u = rng.standard_normal((T - 1, m))
        w = np.tile(rng.standard_normal((1, nw)), (T, 1)) if nw else None     # same θ at every step for this check
        if nw:
            L.set_w(_p(np.ascontiguousarray(w[0])), nw)
        s = oracle.Solver(pr, w=w); s.initialize_states(x); s.initialize_controls(u)
        s.set_buffer("states", x); s.set_buffer("actions", u)
        s.call("reset_model_objective"); s.call("cost_bang", 0); s.call("gradients")
        # dynamics value and Jacobians
        y = np.zeros(n); L.dyn(_p(x[0]), _p(u[0]), _p(y))
        assert np.allclose(y, pr.rollout(x[0], u, w)[1], rtol=1e-13, atol=1e-14)
        fx = np.zeros(n * n); fu = np.zeros(n * m); L.dyn_jac(_p(x[0]), _p(u[0]), _p(fx), _p(fu))
        assert np.allclose(fx, s.buffer("jacobian_state")[:n * n], rtol=1e-11, atol=1e-12)
        assert np.allclose(fu, s.buffer("jacobian_action")[:n * m], rtol=1e-11, atol=1e-12)
        # costs
        J = sum(L.cost_s(_p(x[t]), _p(u[t])) for t in range(T - 1)) + L.cost_t(_p(x[-1]))
        gx = np.zeros(n); gu = np.zeros(m); L.cost_s_grad(_p(x[0]), _p(u[0]), _p(gx), _p(gu))
        gxx = np.zeros(n * n); guu = np.zeros(m * m); gux = np.zeros(m * n)
        L.cost_s_hess(_p(x[0]), _p(u[0]), _p(gxx), _p(guu), _p(gux))
        if pr.c.constraints is None:
            assert J == pytest.approx(s.stats().objective, rel=1e-13)
            assert np.allclose(gx, s.buffer("gradient_state")[:n]) and np.allclose(gu, s.buffer("gradient_action")[:m])
            assert np.allclose(gxx, s.buffer("hessian_state_state")[:n * n])
        # constraints (values via the violations buffer = c(states))
        if ncs:
            c = np.zeros(ncs); L.con_s(_p(x[0]), _p(u[0]), _p(c))
            assert np.allclose(c, s.buffer("violations")[:ncs], rtol=1e-13, atol=1e-14)
        if nct:
            c = np.zeros(nct); L.con_t(_p(x[-1]), _p(c))
            assert np.allclose(c, s.buffer("violations")[(T - 1) * ncs:], rtol=1e-13, atol=1e-14)


def test_user_model_codegen_roundtrip(tmp_path):
    """A user-defined model through the public constructors: values and derivatives vs finite differences."""
    import sympy as sp
    pkg = load_package()
    f = lambda x, u: [x[0] + 0.1 * x[1] * sp.cos(x[0]), x[1] + 0.1 * (u[0] - 9.81 * sp.sin(x[0]) - 0.1 * x[1] * u[1])]
    dyn = pkg.Dynamics(f, 2, 2)
    stage = pkg.Cost(lambda x, u: x[0] ** 2 * x[1] + 0.1 * u[0] ** 2 + sp.sin(u[1]) * x[1], 2, 2)
    term = pkg.Cost(lambda x, u: 10.0 * (x[0] ** 2 + x[1] ** 4), 2, 0)
    cs = pkg.Constraint(lambda x, u: [u[0] - 2.0, -2.0 - u[0], x[0] * u[1]], 2, 2, indices_inequality=[1, 2])
    ct = pkg.Constraint(lambda x, u: [x[0] - 3.0, x[1] * x[0]], 2, 0)
    sname, src = pkg.codegen.generate_model_source("usr", dyn, stage, term, cs, ct)
    hdr = tmp_path / "model_usr.h"
    hdr.write_text(src)
    L = _build(tmp_path, str(hdr), sname)
    assert L.ineq(0) == 0b011 and L.ineq(1) == 0
    rng = np.random.default_rng(4)
    x = rng.standard_normal(2); u = rng.standard_normal(2)

    def fd(fun, z, k, eps=1e-6):
        out = []
        for i in range(z.size):
            zp, zm = z.copy(), z.copy(); zp[i] += eps; zm[i] -= eps
            out.append((fun(zp) - fun(zm)) / (2 * eps))
        return np.array(out).T.reshape(k, z.size)

    def F(z):
        y = np.zeros(2); L.dyn(_p(np.ascontiguousarray(z[:2])), _p(np.ascontiguousarray(z[2:])), _p(y)); return y
    fx = np.zeros(4); fu = np.zeros(4); L.dyn_jac(_p(x), _p(u), _p(fx), _p(fu))
    Jfd = fd(F, np.r_[x, u], 2)
    assert np.allclose(np.c_[fx.reshape(2, 2).T, fu.reshape(2, 2).T], Jfd, atol=1e-7)
    assert np.allclose(F(np.r_[x, u]), [float(v) for v in f(list(x), list(u))], rtol=1e-14)

    def Cst(z):
        return np.array([L.cost_s(_p(np.ascontiguousarray(z[:2])), _p(np.ascontiguousarray(z[2:])))])
    gx = np.zeros(2); gu = np.zeros(2); L.cost_s_grad(_p(x), _p(u), _p(gx), _p(gu))
    assert np.allclose(np.r_[gx, gu], fd(Cst, np.r_[x, u], 1)[0], atol=1e-6)

    def G(z):
        a = np.zeros(2); b = np.zeros(2)
        L.cost_s_grad(_p(np.ascontiguousarray(z[:2])), _p(np.ascontiguousarray(z[2:])), _p(a), _p(b)); return np.r_[a, b]
    H = fd(G, np.r_[x, u], 4)
    gxx = np.zeros(4); guu = np.zeros(4); gux = np.zeros(4); L.cost_s_hess(_p(x), _p(u), _p(gxx), _p(guu), _p(gux))
    assert np.allclose(gxx.reshape(2, 2).T, H[:2, :2], atol=1e-6) and np.allclose(guu.reshape(2, 2).T, H[2:, 2:], atol=1e-6)
    assert np.allclose(gux.reshape(2, 2).T, H[2:, :2], atol=1e-6)      # hessian_action_state is nu×nx

    def Cn(z):
        c = np.zeros(3); L.con_s(_p(np.ascontiguousarray(z[:2])), _p(np.ascontiguousarray(z[2:])), _p(c)); return c
    cx = np.zeros(6); cu = np.zeros(6); L.con_s_jac(_p(x), _p(u), _p(cx), _p(cu))
    assert np.allclose(np.c_[cx.reshape(2, 3).T, cu.reshape(2, 3).T], fd(Cn, np.r_[x, u], 3), atol=1e-7)


def test_time_varying_objects_lowering():
    """lowering.lower: with the selectors of step t substituted, the combined stage objects ARE object t."""
    import sympy as sp
    from ilqr_amd_loader import load_package
    pkg = load_package()
    T = 13
    dynamics, costs, constraints = pkg.models.car_tv(T)
    low = pkg.lowering.lower(dynamics, costs, constraints)
    sel, nwu = low["selectors"], low["num_user_parameter"]
    assert nwu == 0 and sel.shape == (T, 2 + 2 + 3) and (sel[-1] == 0).all()
    assert (sel[:-1].sum(axis=1) == 3).all()                         # one-hot per category
    D, Cs, Ks = low["dynamics"], low["cost_stage"], low["con_stage"]
    assert Ks.num_constraint == 6 and sorted(Ks.indices_inequality) == [1, 2, 3, 4, 5]
    for t in range(T - 1):
        sub = {D.w[nwu + j]: sel[t, j] for j in range(sel.shape[1])}
        same = lambda a, b: sp.simplify(sp.sympify(a).subs(sub) - b) == 0
        assert all(same(a, b) for a, b in zip(D.evaluate, dynamics[t].evaluate))
        assert all(same(a, b) for ra, rb in zip(D.jacobian_state, dynamics[t].jacobian_state) for a, b in zip(ra, rb))
        assert same(Cs.evaluate, costs[t].evaluate)
        assert all(same(a, b) for ra, rb in zip(Cs.hessian_state_state, costs[t].hessian_state_state) for a, b in zip(ra, rb))
        rows = low["constraint_rows"][t]
        assert len(rows) == constraints[t].num_constraint
        for i in range(Ks.num_constraint):
            want = constraints[t].evaluate[rows.index(i)] if i in rows else 0
            assert same(Ks.evaluate[i], want)
    # uniform objects pass through untouched
    one = pkg.models.car()
    low1 = pkg.lowering.lower([one["dynamics"]] * 4, [one["cost_stage"]] * 4 + [one["cost_term"]],
                              [one["con_stage"]] * 4 + [one["con_term"]])
    assert low1["dynamics"] is one["dynamics"] and low1["selectors"].shape[1] == 0


def _ragged_problem(pkg, T=9):
    """Dimensions change along the horizon: n_t = 3,3,4,4,2,2,3,3 | 3.. and m_t = 2,1,2,1,1,2,2,1 | 2.. (models.ragged; the oracle's
    "ragged" problem and the independent restatement's ragged_problem are the same definition)."""
    return pkg.models.ragged(T)


def test_time_varying_dimensions_lowering():
    from ilqr_amd_loader import load_package
    pkg = load_package()
    dynamics, costs, constraints, n_t, m_t = _ragged_problem(pkg)
    low = pkg.lowering.lower(dynamics, costs, constraints)
    assert low["state_dims"] == n_t and low["action_dims"] == m_t
    D, Cs = low["dynamics"], low["cost_stage"]
    assert (D.num_state, D.num_action, D.num_next_state) == (4, 2, 4)
    sel = low["selectors"]
    for t in range(len(m_t)):
        sub = {D.w[j]: sel[t, j] for j in range(sel.shape[1])}
        y = [pkg.codegen.sp.sympify(e).subs(sub) for e in D.evaluate]
        assert all(pkg.codegen.sp.simplify(y[i] - dynamics[t].evaluate[i]) == 0 for i in range(n_t[t + 1]))
        assert all(y[i] == 0 for i in range(n_t[t + 1], 4))                       # padded next-state rows
        ell = pkg.codegen.sp.sympify(Cs.evaluate).subs(sub)
        pad = sum(Cs.u[j] ** 2 for j in range(m_t[t], 2)) / 2
        assert pkg.codegen.sp.simplify(ell - costs[t].evaluate - pad) == 0          # u²/2 on padded actions only


def test_lowering_plan_comes_from_the_library_and_refuses_an_inconsistent_chain():
    """lowering.lower asks ilqr_plan_stages (the C-ABI entry a Julia / C host uses through ilqr_compile_model_stages) for template
    dimensions, selector columns, row offsets and inequality masks; a chain of dimensions the reference would throw on
    (x[t+1] .= dynamics!(...), src/rollout.jl:29) is refused there."""
    import pytest
    from ilqr_amd_loader import load_package
    pkg = load_package()
    T = 13
    dynamics, costs, constraints = pkg.models.car_tv(T)
    kinds, _ = pkg.lowering.stage_kinds(dynamics, costs, constraints)
    plan, sel, n_t, m_t = pkg._ffi.plan_stages(kinds)
    assert (plan.nx, plan.nu, plan.nw, plan.nc_stage, plan.nc_term, plan.n_selectors) == (3, 2, 7, 6, 4, 7)
    assert (plan.sel_dynamics, plan.sel_cost, plan.sel_constraint) == (0, 2, 4) and list(plan.constraint_row0)[:3] == [0, 5, 5]
    assert plan.ineq_stage_words[0] == 0b011111 and n_t == [3] * T and m_t == [2] * (T - 1)
    low = pkg.lowering.lower(dynamics, costs, constraints)
    assert (low["selectors"] == np.array(sel)).all() and low["constraint_rows"][2] == [5] and low["constraint_rows"][1] == []
    d2 = pkg.Dynamics(lambda x, u: [x[0] + u[0], x[1]], 3, 2)              # 3 states in, 2 out
    with pytest.raises(pkg._ffi.IlqrError, match="does not produce the state"):
        pkg.lowering.lower([dynamics[0], d2] + dynamics[2:], costs, constraints)


def test_compile_model_stages_composes_the_template_from_c_sources_per_kind(tmp_path):
    """ilqr_compile_model_stages: per-kind C callables in their own dimensions -> combined, padded template callables (selector
    branches, re-strided matrices, u^2 / 2 on padded actions) -> hipcc. Compiles without a GPU; the GPU test solves with it."""
    import ctypes as C
    import os
    from ilqr_amd_loader import load_package
    pkg = load_package()
    F, L = pkg._ffi, pkg._ffi.lib()
    T = 17
    kinds, src = pkg.models.ragged_c_stages(T)
    cap = T * (kinds.n_dynamics + kinds.n_costs + kinds.n_constraints)
    plan, sel = F.StagePlan(), (C.c_double * cap)()
    sd, ad = (C.c_int32 * T)(), (C.c_int32 * (T - 1))()
    reg, path = C.create_string_buffer(160), C.create_string_buffer(1024)
    os.environ["ILQR_KEEP_MODEL_SOURCE"] = "1"
    try:
        rc = L.ilqr_compile_model_stages(b"ragged_c", C.byref(kinds), src.encode(), C.byref(plan), sel, cap, sd, ad, reg, 160, path, 1024)
    finally:
        del os.environ["ILQR_KEEP_MODEL_SOURCE"]
    assert rc == 0, L.ilqr_last_error().decode()
    _, _, _, n_t, m_t = pkg.models.ragged(T)
    assert (plan.nx, plan.nu, plan.nc_stage, plan.nc_term) == (4, 2, 0, 2) and list(sd) == n_t and list(ad) == m_t
    assert plan.n_selectors == kinds.n_dynamics + kinds.n_costs and plan.nw == plan.n_selectors
    S = plan.n_selectors
    rows = np.array([sel[i] for i in range(T * S)]).reshape(T, S)
    assert (rows[:-1].sum(axis=1) == 2).all() and (rows[-1] == 0).all()
    # registered under its name (not necessarily last: a module the process has loaded before is registered once)
    assert os.path.exists(path.value.decode()) and reg.value in [L.ilqr_model_name(i) for i in range(L.ilqr_model_count())]
    # the same objects through the symbolic lowering give the same plan
    dynamics, costs, constraints, _, _ = pkg.models.ragged(T)
    low = pkg.lowering.lower(dynamics, costs, constraints)
    assert (low["selectors"] == rows).all() and low["state_dims"] == n_t
    # inconsistent kinds are refused before anything is compiled
    bad = F.stage_kinds(3, 0, [(2, 1, 3), (2, 1, 2)], [0, 1], [(2, 1)], [0, 0], [], [], 2, 0, 0)
    assert L.ilqr_compile_model_stages(b"bad", C.byref(bad), b"", C.byref(plan), None, 0, None, None, reg, 160, path, 1024) < 0
    assert b"does not produce the state" in L.ilqr_last_error()


def test_lowering_selects_instead_of_multiplying_and_dedupes_kinds():
    """A kind that is switched off at a step may be outside its domain there (the reference never calls it): the lowered
    template must SELECT, not multiply by 0 (0 · NaN = NaN). Objects traced from the same function are one kind."""
    import sympy as sp
    from ilqr_amd_loader import load_package
    pkg = load_package()
    T = 5
    dyn = pkg.Dynamics(lambda x, u: [x[0] + 0.1 * u[0]], 1, 1)
    cost_a = pkg.Cost(lambda x, u: x[0] * x[0] + u[0] * u[0], 1, 1)
    cost_b = pkg.Cost(lambda x, u: sp.sqrt(x[0] - 5.0) + u[0] * u[0], 1, 1)        # NaN for x0 < 5
    term = pkg.Cost(lambda x, u: x[0] * x[0], 1, 0)
    cons = [pkg.Constraint(lambda x, u: [u[0] - 1.0], 1, 1, indices_inequality=[1]) for _ in range(T - 1)] + [pkg.Constraint()]
    low = pkg.lowering.lower([dyn] * (T - 1), [cost_a, cost_a, cost_b, cost_a, term], cons)
    assert low["selectors"].shape[1] == 2                            # two cost kinds; the T-1 identical constraints are ONE kind
    Cs = low["cost_stage"]
    nwu = low["num_user_parameter"]
    on_a = {Cs.w[nwu + 0]: 1.0, Cs.w[nwu + 1]: 0.0, Cs.x[0]: 0.5, Cs.u[0]: 2.0}
    for expr, want in ((Cs.evaluate, 0.25 + 4.0), (Cs.gradient_state[0], 1.0), (Cs.hessian_state_state[0][0], 2.0)):
        val = complex(sp.sympify(expr).subs(on_a).evalf())
        assert val.imag == 0.0 and abs(val.real - want) < 1e-14, (expr, val)
    src = pkg.codegen.generate_model_source("sel", low["dynamics"], low["cost_stage"], low["cost_term"], low["con_stage"], low["con_term"])[1]
    assert "?" in src                                                # the selection is a ternary in the device code


def test_cooperative_rollout_code_shape():
    """What the generator emits for the serial rollout path of small models (iterativelqr.jl_amd/codegen.py, the `coop` form):
    one ilqr::sincos_pair per dependency level of trig arguments, affine arguments as ONE FMA chain over per-lane coefficients
    of the WaveCtx (no per-angle selects), non-affine arguments through selects on the pair slot, results handed round by row
    broadcasts from the even (sine) / odd (cosine) lane of the pair, and no dead temporaries left behind."""
    import re
    import sympy as sp
    pkg = load_package()
    _, src = pkg.models.builtin_source("acrobot")
    body = src[src.index("static void dyn_wave("):src.index("static void dyn_jac(")]
    assert body.count("ilqr::sincos_pair(") == 1 and "sincos_fast" not in body           # six angles, one level, one evaluation
    assert re.search(r"const double ta0 = fma\(cx\.a\[\d\], x3, fma\(cx\.a\[\d\], x2, fma\(cx\.a\[\d\], x1, cx\.a\[\d\] \* x0\)\)\);", body)
    assert "(pq ==" not in body                                                               # no per-angle select for affine arguments
    assert len(re.findall(r"BC::template bcast<\d+>\(tr0\)", body)) == 8                       # sn0, sn1, cs1, sn2, sn3, sn4, cs4, sn5
    assert "BC::template bcast<3>(tr0)" in body and "BC::template bcast<2>(tr0)" in body      # cosine / sine lane of pair 1
    ctx = src[src.index("struct WaveCtx"):src.index("static void dyn_wave(")]
    assert "ilqr::TrigPair tp; double a[4]" in ctx and ctx.count("ILQR_OPAQUE(cx.a[") == 4
    for tmp in re.findall(r"const double (t\d+) = ", body):                                   # every temporary is used
        assert len(re.findall(r"\b%s\b" % tmp, body)) >= 2, tmp
    # a non-affine trig argument falls back to selects on the pair slot
    dyn = pkg.Dynamics(lambda x, u: [x[0] + 0.1 * sp.sin(x[0] * x[1]) + 0.1 * sp.cos(x[1]), x[1] + 0.1 * u[0]], 2, 1)
    cost = pkg.Cost(lambda x, u: x[0] * x[0] + u[0] * u[0], 2, 1)
    term = pkg.Cost(lambda x, u: x[0] * x[0], 2, 0)
    _, src2 = pkg.codegen.generate_model_source("nonaffine", dyn, cost, term, pkg.Constraint(), pkg.Constraint())
    body2 = src2[src2.index("static void dyn_wave("):src2.index("static void dyn_jac(")]
    assert "(pq == 1)" in body2 and body2.count("ilqr::sincos_pair(") == 1


def test_large_model_structure_detection():
    """What the generator finds in a large model (codegen.py, _emit_large_model_extras) and the kernels then exploit: synth32
    (x+ = A x + B u + c sin x) has 32 state-dependent Jacobian entries of 1280, each A_ii + c cos x_i — ONE function of ONE state
    component plus a constant (JAC_VAR_ELEMENTWISE: gradients! evaluates them one (timestep, entry) pair per thread) — and an
    elementwise remainder independent of u (dyn_rem_own); synth12's state-dependent entries sit in fu and mix components: no such
    structure, the generic forms are emitted."""
    import re
    pkg = load_package()
    _, s32 = pkg.models.builtin_source("synth32")
    assert "static constexpr int JAC_NVAR = 32;" in s32
    assert "static constexpr bool JAC_VAR_ELEMENTWISE = true;" in s32 and "static constexpr bool DYN_REM_ELEMENTWISE = true;" in s32
    src = [int(v) for v in re.search(r"JAC_VAR_SRC\[32\] = \{([^}]*)\}", s32).group(1).split(",")]
    assert src == list(range(32))                                                          # entry q is a function of x_q
    own = s32[s32.index("static double dyn_jac_var_own("):]
    own = own[:own.index("}")]
    assert own.count("cos_fast(xl)") == 1 and "x[" not in own                             # one cosine of the lane's own component
    add = [float(v) for v in re.search(r"JAC_VAR_ADD\[1\]\[32\] = \{\s*\{([^}]*)\}", s32).group(1).split(",")]
    assert len(add) == 32 and all(0.9 < v < 1.0 for v in add)                             # the constants A_ii
    m12 = pkg.models.synth12()
    _, s12 = pkg.codegen.generate_model_source("synth12_t", m12["dynamics"], m12["cost_stage"], m12["cost_term"], m12["con_stage"], m12["con_term"])
    assert "static constexpr bool JAC_VAR_ELEMENTWISE = false;" in s12 and "dyn_jac_var_own" not in s12 and "static void dyn_jac_var(" in s12


def test_c_stage_sources_of_distinct_objects_compile(tmp_path):
    """lowering.c_stage_sources: symbolic per-step objects -> C callables per kind -> ilqr_compile_model_stages (all three selector
    categories, stacked constraint kinds). Compiles without a GPU."""
    import ctypes as C
    from ilqr_amd_loader import load_package
    pkg = load_package()
    F, L = pkg._ffi, pkg._ffi.lib()
    T = 13
    kinds, src = pkg.lowering.c_stage_sources(*pkg.models.car_tv(T))
    assert "dynamics_1_jacobian_action" in src and "constraint_stage_2_jacobian_state" in src and "constraint_stage_1" not in src
    cap = T * (kinds.n_dynamics + kinds.n_costs + kinds.n_constraints)
    plan, sel = F.StagePlan(), (C.c_double * cap)()
    reg, path = C.create_string_buffer(160), C.create_string_buffer(1024)
    rc = L.ilqr_compile_model_stages(b"car_tv_c", C.byref(kinds), src.encode(), C.byref(plan), sel, cap, None, None, reg, 160, path, 1024)
    assert rc == 0, L.ilqr_last_error().decode()
    assert (plan.nx, plan.nu, plan.nw, plan.nc_stage, plan.nc_term, plan.n_selectors) == (3, 2, 7, 6, 4, 7)
