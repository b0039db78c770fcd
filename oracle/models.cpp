/*
 * models.cpp — CPU ORACLE model zoo (test infrastructure, NOT product code).
 *
 * Hand-written model functions following the reference's examples/tests, with
 * Jacobians by forward-mode dual numbers (exact to rounding, like the
 * Symbolics-generated derivatives of src/dynamics.jl:24-28). The product's
 * device code derives its Jacobians symbolically (sympy codegen), so the two
 * derivations are independent.
 *
 * Model definitions:
 *   particle   examples/particle.jl:17-43
 *   pendulum   test/dynamics.jl:8-19        (explicit Euler, KAT only)
 *   acrobot    test/acrobot.jl:9-101        (explicit midpoint, h = 0.1)
 *   car        test/car.jl:10-61            (explicit midpoint, h = 0.1)
 *   synth32    SURVEY.md §8(d) C5           (synthetic nx=32, nu=8)
 *   kat_*      test/objective.jl:6-10, test/constraints.jl:13-19
 */
#include "ilqr_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

// ------------------------------------------------------------ dual numbers
template <int ND>
struct Dual {
    double v;
    double d[ND];
    Dual() : v(0.0) { for (int i = 0; i < ND; ++i) d[i] = 0.0; }
    Dual(double a) : v(a) { for (int i = 0; i < ND; ++i) d[i] = 0.0; }
};
template <int ND> Dual<ND> operator+(const Dual<ND>& a, const Dual<ND>& b) { Dual<ND> r; r.v = a.v + b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int ND> Dual<ND> operator-(const Dual<ND>& a, const Dual<ND>& b) { Dual<ND> r; r.v = a.v - b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int ND> Dual<ND> operator-(const Dual<ND>& a) { Dual<ND> r; r.v = -a.v; for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i]; return r; }
template <int ND> Dual<ND> operator*(const Dual<ND>& a, const Dual<ND>& b) { Dual<ND> r; r.v = a.v * b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int ND> Dual<ND> operator/(const Dual<ND>& a, const Dual<ND>& b) { Dual<ND> r; r.v = a.v / b.v; for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
template <int ND> Dual<ND> operator+(double a, const Dual<ND>& b) { return Dual<ND>(a) + b; }
template <int ND> Dual<ND> operator+(const Dual<ND>& a, double b) { return a + Dual<ND>(b); }
template <int ND> Dual<ND> operator-(double a, const Dual<ND>& b) { return Dual<ND>(a) - b; }
template <int ND> Dual<ND> operator-(const Dual<ND>& a, double b) { return a - Dual<ND>(b); }
template <int ND> Dual<ND> operator*(double a, const Dual<ND>& b) { Dual<ND> r; r.v = a * b.v; for (int i = 0; i < ND; ++i) r.d[i] = a * b.d[i]; return r; }
template <int ND> Dual<ND> operator*(const Dual<ND>& a, double b) { return b * a; }
template <int ND> Dual<ND> operator/(double a, const Dual<ND>& b) { return Dual<ND>(a) / b; }
template <int ND> Dual<ND> operator/(const Dual<ND>& a, double b) { Dual<ND> r; r.v = a.v / b; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] / b; return r; }
template <int ND> Dual<ND> sin(const Dual<ND>& a) { Dual<ND> r; r.v = std::sin(a.v); double c = std::cos(a.v); for (int i = 0; i < ND; ++i) r.d[i] = c * a.d[i]; return r; }
template <int ND> Dual<ND> cos(const Dual<ND>& a) { Dual<ND> r; r.v = std::cos(a.v); double s = -std::sin(a.v); for (int i = 0; i < ND; ++i) r.d[i] = s * a.d[i]; return r; }
using std::sin;
using std::cos;

// Jacobian driver: F is a functor  template<class S> void operator()(const S* x, const S* u, S* y) const
template <int NX, int NU, class F>
void jac_state(double* out, const double* x, const double* u, const F& f) {
    typedef Dual<NX> S;
    S xs[NX], us[NU > 0 ? NU : 1], ys[NX];
    for (int i = 0; i < NX; ++i) { xs[i] = S(x[i]); xs[i].d[i] = 1.0; }
    for (int i = 0; i < NU; ++i) us[i] = S(u[i]);
    f(xs, us, ys);
    for (int j = 0; j < NX; ++j) for (int i = 0; i < NX; ++i) out[j * NX + i] = ys[i].d[j];
}
template <int NX, int NU, class F>
void jac_action(double* out, const double* x, const double* u, const F& f) {
    typedef Dual<(NU > 0 ? NU : 1)> S;
    S xs[NX], us[NU > 0 ? NU : 1], ys[NX];
    for (int i = 0; i < NX; ++i) xs[i] = S(x[i]);
    for (int i = 0; i < NU; ++i) { us[i] = S(u[i]); us[i].d[i] = 1.0; }
    f(xs, us, ys);
    for (int j = 0; j < NU; ++j) for (int i = 0; i < NX; ++i) out[j * NX + i] = ys[i].d[j];
}

// ---------------------------------------------------------------- particle
// examples/particle.jl:17-21
struct ParticleF {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        y[0] = 1.0 * x[0] + 1.0 * x[1] + 0.0 * u[0];
        y[1] = 0.0 * x[0] + 1.0 * x[1] + 1.0 * u[0];
    }
};

// ---------------------------------------------------------------- pendulum
// test/dynamics.jl:8-19
struct PendulumEulerF {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        const double mass = 1.0, lc = 1.0, gravity = 9.81, damping = 0.1, h = 0.1;
        S f0 = x[1];
        S f1 = u[0] / (mass * lc * lc) - gravity * sin(x[0]) / lc - damping * x[1] / (mass * lc * lc);
        y[0] = x[0] + h * f0;
        y[1] = x[1] + h * f1;
    }
};

// A pendulum with a POLE in its dynamics (no reference counterpart: a test model for what a division by zero, or by something
// tiny, inside f does to a solve — IEEE gives +-Inf there and the reference carries it on, src/rollout.jl:27-29)
struct PendulumPoleF {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        const double h = 0.1;
        S f0 = x[1];
        S f1 = u[0] - sin(x[0]) - 0.1 * x[1] + 0.001 / (x[0] - 0.3);
        y[0] = x[0] + h * f0;
        y[1] = x[1] + h * f1;
    }
};

// ----------------------------------------------------------------- acrobot
// test/acrobot.jl:9-74
template <class S>
void acrobot_continuous(const S* x, const S* u, S* dx) {
    const double mass1 = 1.0, inertia1 = 0.33, length1 = 1.0, lengthcom1 = 0.5;
    const double mass2 = 1.0, inertia2 = 0.33, length2 = 1.0, lengthcom2 = 0.5;
    const double gravity = 9.81, friction1 = 0.1, friction2 = 0.1;
    (void)length2;
    // M(q) with q = x[1:2] → the reference indexes x[2] of the view = q2   (:24-33)
    S c2 = cos(x[1]);
    S Ma = inertia1 + inertia2 + mass2 * length1 * length1 + 2.0 * mass2 * length1 * lengthcom2 * c2;
    S Mb = inertia2 + mass2 * length1 * lengthcom2 * c2;
    S Mc = S(inertia2);
    // Minv (:35-42): 1/(a d − b c) [d −b; −c a] with m = [a b; b c]
    S det = Ma * Mc - Mb * Mb;
    S idet = 1.0 / det;
    S i11 = idet * Mc, i12 = idet * (-Mb), i21 = idet * (-Mb), i22 = idet * Ma;
    // τ(q) (:44-52)
    S ta = -1.0 * mass1 * gravity * lengthcom1 * sin(x[0])
           - mass2 * gravity * (length1 * sin(x[0]) + lengthcom2 * sin(x[0] + x[1]));
    S tb = -1.0 * mass2 * gravity * lengthcom2 * sin(x[0] + x[1]);
    // C(x) (:54-61)
    S s2 = sin(x[1]);
    S Ca = -2.0 * mass2 * length1 * lengthcom2 * s2 * x[3];
    S Cb = -1.0 * mass2 * length1 * lengthcom2 * s2 * x[3];
    S Cc = mass2 * length1 * lengthcom2 * s2 * x[2];
    S Cd = S(0.0);
    // qdd = Minv * (−C v + τ + B u − friction .* v)   (:70-71)
    S r1 = -1.0 * (Ca * x[2] + Cb * x[3]) + ta + 0.0 * u[0] - friction1 * x[2];
    S r2 = -1.0 * (Cc * x[2] + Cd * x[3]) + tb + 1.0 * u[0] - friction2 * x[3];
    dx[0] = x[2];
    dx[1] = x[3];
    dx[2] = i11 * r1 + i12 * r2;
    dx[3] = i21 * r1 + i22 * r2;
}
// test/acrobot.jl:76-79 — explicit midpoint
struct AcrobotF {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        const double h = 0.1;
        S k1[4], xm[4], k2[4];
        acrobot_continuous(x, u, k1);
        for (int i = 0; i < 4; ++i) xm[i] = x[i] + 0.5 * h * k1[i];
        acrobot_continuous(xm, u, k2);
        for (int i = 0; i < 4; ++i) y[i] = x[i] + h * k2[i];
    }
};

// --------------------------------------------------------------------- car
// test/car.jl:10-17
template <class S>
void car_continuous(const S* x, const S* u, S* dx) {
    dx[0] = u[0] * cos(x[2]);
    dx[1] = u[0] * sin(x[2]);
    dx[2] = u[1];
}
struct CarF {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        const double h = 0.1;
        S k1[3], xm[3], k2[3];
        car_continuous(x, u, k1);
        for (int i = 0; i < 3; ++i) xm[i] = x[i] + 0.5 * h * k1[i];
        car_continuous(xm, u, k2);
        for (int i = 0; i < 3; ++i) y[i] = x[i] + h * k2[i];
    }
};

// car_tv (time-varying objects, README.md:26 of the reference): explicit Euler with h = 0.05 on some steps
struct CarEulerF {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        const double h = 0.05;
        S k1[3];
        car_continuous(x, u, k1);
        for (int i = 0; i < 3; ++i) y[i] = x[i] + h * k1[i];
    }
};

// ----------------------------------------------------------------- synth32
// x⁺ = x + h(Ax + Bu + 0.1 sin(x)), h = 0.05
struct Synth32Tables {
    double A[32][32], B[32][8];
    Synth32Tables() {
        for (int i = 0; i < 32; ++i) {
            for (int j = 0; j < 32; ++j)
                A[i][j] = (i == j ? -1.0 : 0.0) + 0.3 * std::cos((double)((i + 1) + 2 * (j + 1))) / 32.0;
            for (int j = 0; j < 8; ++j)
                B[i][j] = std::sin((double)(3 * (i + 1) + (j + 1))) / std::sqrt(32.0);
        }
    }
};
const Synth32Tables& synth_tables() { static Synth32Tables t; return t; }
struct Synth32F {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        const Synth32Tables& tb = synth_tables();
        const double h = 0.05;
        for (int i = 0; i < 32; ++i) {
            S acc = S(0.0);
            for (int j = 0; j < 32; ++j) acc = acc + tb.A[i][j] * x[j];
            for (int j = 0; j < 8; ++j) acc = acc + tb.B[i][j] * u[j];
            acc = acc + 0.1 * sin(x[i]);
            y[i] = x[i] + h * acc;
        }
    }
};

// ----------------------------------------------------------------- synth12
// A second large-path model with dimensions that are NOT multiples of the MFMA tile (nx = 12, nu = 5), a bilinear
// term (state-dependent fu entries) and a terminal equality: x⁺ = x + h(Ax + Bu + 0.1 sin x + 0.02 x∘u_{i mod 5}),
// h = 0.05, A_ij = −δ_ij + 0.3 cos(i + 2j)/12, B_ij = sin(3i + j)/√12 (1-based).
struct Synth12F {
    template <class S> void operator()(const S* x, const S* u, S* y) const {
        const double h = 0.05;
        for (int i = 0; i < 12; ++i) {
            S acc = S(0.0);
            for (int j = 0; j < 12; ++j)
                acc = acc + ((i == j ? -1.0 : 0.0) + 0.3 * std::cos((double)((i + 1) + 2 * (j + 1))) / 12.0) * x[j];
            for (int j = 0; j < 5; ++j) acc = acc + (std::sin((double)(3 * (i + 1) + (j + 1))) / std::sqrt(12.0)) * u[j];
            acc = acc + 0.1 * sin(x[i]);
            acc = acc + 0.02 * (x[i] * u[i % 5]);
            y[i] = x[i] + h * acc;
        }
    }
};
// terminal equality on the first three states: c = x[1:3] − 0.1
void synth12_term_eval(double* out, const double* x, const double*, const double*, const void*) {
    for (int i = 0; i < 3; ++i) out[i] = x[i] - 0.1;
}
void synth12_term_jx(double* out, const double*, const double*, const double*, const void*) {
    for (int i = 0; i < 3; ++i) out[i * 3 + i] = 1.0;
}

// ------------------------------------------------------------------ ragged
// Time-varying DIMENSIONS (src/dynamics.jl:5-7: num_next_state != num_state; README.md:26 of the reference): a chain of
// small maps (n0, m0) -> n1, y_i = sum_j A_ij x_j + sum_j B_ij u_j + [i == 0] 0.1 sin(x_0),
// A_ij = 0.9 [i == j] + 0.1 cos(1 + i + 2j + n0), B_ij = 0.3 sin(2 + 3i + j + m0) (0-based) — the twin of
// tests/test_codegen.py::_ragged_problem. Dimensions are run-time here (ctx), at most 4 states / 4 actions.
struct RaggedCtx { int n0, m0, n1; };
template <class S> void ragged_f(const RaggedCtx* c, const S* x, const S* u, S* y) {
    for (int i = 0; i < c->n1; ++i) {
        S acc = S(0.0);
        for (int j = 0; j < c->n0; ++j) acc = acc + ((i == j ? 0.9 : 0.0) + 0.1 * std::cos(1.0 + i + 2 * j + c->n0)) * x[j];
        for (int j = 0; j < c->m0; ++j) acc = acc + (0.3 * std::sin(2.0 + 3 * i + j + c->m0)) * u[j];
        if (i == 0) acc = acc + 0.1 * sin(x[0]);
        y[i] = acc;
    }
}
void ragged_eval(double* out, const double* x, const double* u, const double*, const void* ctx) {
    ragged_f((const RaggedCtx*)ctx, x, u, out);
}
void ragged_jx(double* out, const double* x, const double* u, const double*, const void* ctx) {
    const RaggedCtx* c = (const RaggedCtx*)ctx;
    typedef Dual<4> S;
    S xs[4], us[4], ys[4];
    for (int i = 0; i < c->n0; ++i) { xs[i] = S(x[i]); xs[i].d[i] = 1.0; }
    for (int i = 0; i < c->m0; ++i) us[i] = S(u[i]);
    ragged_f(c, xs, us, ys);
    for (int j = 0; j < c->n0; ++j) for (int i = 0; i < c->n1; ++i) out[j * c->n1 + i] = ys[i].d[j];     // n1 x n0, column-major
}
void ragged_ju(double* out, const double* x, const double* u, const double*, const void* ctx) {
    const RaggedCtx* c = (const RaggedCtx*)ctx;
    typedef Dual<4> S;
    S xs[4], us[4], ys[4];
    for (int i = 0; i < c->n0; ++i) xs[i] = S(x[i]);
    for (int i = 0; i < c->m0; ++i) { us[i] = S(u[i]); us[i].d[i] = 1.0; }
    ragged_f(c, xs, us, ys);
    for (int j = 0; j < c->m0; ++j) for (int i = 0; i < c->n1; ++i) out[j * c->n1 + i] = ys[i].d[j];     // n1 x m0
}
const int RAGGED_N[8] = {3, 3, 4, 4, 2, 2, 3, 3}, RAGGED_M[8] = {2, 1, 2, 1, 1, 2, 2, 1};

// generic wrappers: orc_fn adaptors for a dynamics functor
template <int NX, int NU, class F> void dyn_eval(double* out, const double* x, const double* u, const double*, const void*) {
    F f; f(x, u, out);
}
template <int NX, int NU, class F> void dyn_jx(double* out, const double* x, const double* u, const double*, const void*) {
    jac_state<NX, NU>(out, x, u, F());
}
template <int NX, int NU, class F> void dyn_ju(double* out, const double* x, const double* u, const double*, const void*) {
    jac_action<NX, NU>(out, x, u, F());
}
template <int NX, int NU, class F> OrcDynamics make_dynamics() {
    OrcDynamics d;
    d.evaluate = dyn_eval<NX, NU, F>; d.jacobian_state = dyn_jx<NX, NU, F>; d.jacobian_action = dyn_ju<NX, NU, F>;
    d.num_next_state = NX; d.num_state = NX; d.num_action = NU; d.num_parameter = 0; d.ctx = nullptr;
    return d;
}

// ------------------------------------------------------- quadratic costs
// ℓ(x,u) = Σ q_i (x_i − xg_i)² + Σ r_j u_j²  — covers every cost the reference's
// examples/tests use (examples/particle.jl:34-37, test/acrobot.jl:92-95,
// test/car.jl:32-35, test/objective.jl:6-7).
struct QuadCtx { int n, m; double q[32], xg[32], r[8]; };
void quad_eval(double* out, const double* x, const double* u, const double*, const void* ctx) {
    const QuadCtx* c = (const QuadCtx*)ctx;
    double J = 0.0;
    for (int i = 0; i < c->n; ++i) { double e = x[i] - c->xg[i]; J += c->q[i] * (e * e); }
    for (int j = 0; j < c->m; ++j) J += c->r[j] * (u[j] * u[j]);
    out[0] = J;
}
void quad_gx(double* out, const double* x, const double*, const double*, const void* ctx) {
    const QuadCtx* c = (const QuadCtx*)ctx;
    for (int i = 0; i < c->n; ++i) out[i] = 2.0 * c->q[i] * (x[i] - c->xg[i]);
}
void quad_gu(double* out, const double*, const double* u, const double*, const void* ctx) {
    const QuadCtx* c = (const QuadCtx*)ctx;
    for (int j = 0; j < c->m; ++j) out[j] = 2.0 * c->r[j] * u[j];
}
void quad_gxx(double* out, const double*, const double*, const double*, const void* ctx) {
    const QuadCtx* c = (const QuadCtx*)ctx;
    for (int i = 0; i < c->n; ++i) out[i * c->n + i] = 2.0 * c->q[i];
}
void quad_guu(double* out, const double*, const double*, const double*, const void* ctx) {
    const QuadCtx* c = (const QuadCtx*)ctx;
    for (int j = 0; j < c->m; ++j) out[j * c->m + j] = 2.0 * c->r[j];
}
void quad_gux(double*, const double*, const double*, const double*, const void*) {}
OrcCost make_quad_cost(const QuadCtx* c) {
    OrcCost k;
    k.evaluate = quad_eval; k.gradient_state = quad_gx; k.gradient_action = quad_gu;
    k.hessian_state_state = quad_gxx; k.hessian_action_action = quad_guu; k.hessian_action_state = quad_gux;
    k.num_state = c->n; k.num_action = c->m; k.ctx = c;
    return k;
}

// ------------------------------------------------------------ constraints
void con_nothing(double*, const double*, const double*, const double*, const void*) {}
OrcConstraint make_empty_constraint() {    // src/constraints.jl:45-52
    OrcConstraint c; std::memset(&c, 0, sizeof(c));
    c.evaluate = con_nothing; c.jacobian_state = con_nothing; c.jacobian_action = con_nothing;
    return c;
}
// goal equality: c = x − xT (examples/particle.jl:42, test/acrobot.jl:100)
struct GoalCtx { int n; double xT[32]; };
void goal_eval(double* out, const double* x, const double*, const double*, const void* ctx) {
    const GoalCtx* g = (const GoalCtx*)ctx;
    for (int i = 0; i < g->n; ++i) out[i] = x[i] - g->xT[i];
}
void goal_jx(double* out, const double*, const double*, const double*, const void* ctx) {
    const GoalCtx* g = (const GoalCtx*)ctx;
    for (int i = 0; i < g->n; ++i) out[i * g->n + i] = 1.0;
}
// car stage: [ul − u; u − uu; r² − ‖x[1:2] − p‖²], all inequality (test/car.jl:45-53)
const double CAR_UL = -5.0, CAR_UU = 5.0, CAR_PX = 0.5, CAR_PY = 0.5, CAR_R = 0.1;
void car_stage_eval(double* out, const double* x, const double* u, const double*, const void*) {
    double e0 = x[0] - CAR_PX, e1 = x[1] - CAR_PY;
    out[0] = CAR_UL - u[0]; out[1] = CAR_UL - u[1];
    out[2] = u[0] - CAR_UU; out[3] = u[1] - CAR_UU;
    out[4] = CAR_R * CAR_R - (e0 * e0 + e1 * e1);
}
void car_stage_jx(double* out, const double* x, const double*, const double*, const void*) {
    // 5×3 column-major
    out[0 * 5 + 4] = -2.0 * (x[0] - CAR_PX);
    out[1 * 5 + 4] = -2.0 * (x[1] - CAR_PY);
}
void car_stage_ju(double* out, const double*, const double*, const double*, const void*) {
    // 5×2 column-major
    out[0 * 5 + 0] = -1.0; out[1 * 5 + 1] = -1.0;
    out[0 * 5 + 2] = 1.0;  out[1 * 5 + 3] = 1.0;
}
// car_tv: on some steps a single equality on the steering rate, c = u₂ − 0.3·x₃ − 0.05
void cartv_eq_eval(double* out, const double* x, const double* u, const double*, const void*) {
    out[0] = u[1] - 0.3 * x[2] - 0.05;
}
void cartv_eq_jx(double* out, const double*, const double*, const double*, const void*) { out[2] = -0.3; }
void cartv_eq_ju(double* out, const double*, const double*, const double*, const void*) { out[1] = 1.0; }
// car terminal: [x − xT; obstacle], inequality index 4 (1-based) (test/car.jl:54-60)
const double CAR_XT[3] = {1.0, 1.0, 0.0};
void car_term_eval(double* out, const double* x, const double*, const double*, const void*) {
    double e0 = x[0] - CAR_PX, e1 = x[1] - CAR_PY;
    for (int i = 0; i < 3; ++i) out[i] = x[i] - CAR_XT[i];
    out[3] = CAR_R * CAR_R - (e0 * e0 + e1 * e1);
}
void car_term_jx(double* out, const double* x, const double*, const double*, const void*) {
    // 4×3 column-major
    out[0 * 4 + 0] = 1.0; out[1 * 4 + 1] = 1.0; out[2 * 4 + 2] = 1.0;
    out[0 * 4 + 3] = -2.0 * (x[0] - CAR_PX);
    out[1 * 4 + 3] = -2.0 * (x[1] - CAR_PY);
}
// car_obs: the obstacle centre is the per-timestep parameter w = (p_x, p_y)
void carobs_stage_eval(double* out, const double* x, const double* u, const double* w, const void*) {
    double e0 = x[0] - w[0], e1 = x[1] - w[1];
    out[0] = CAR_UL - u[0]; out[1] = CAR_UL - u[1];
    out[2] = u[0] - CAR_UU; out[3] = u[1] - CAR_UU;
    out[4] = CAR_R * CAR_R - (e0 * e0 + e1 * e1);
}
void carobs_stage_jx(double* out, const double* x, const double*, const double* w, const void*) {
    out[0 * 5 + 4] = -2.0 * (x[0] - w[0]);
    out[1 * 5 + 4] = -2.0 * (x[1] - w[1]);
}
void carobs_term_eval(double* out, const double* x, const double*, const double* w, const void*) {
    double e0 = x[0] - w[0], e1 = x[1] - w[1];
    for (int i = 0; i < 3; ++i) out[i] = x[i] - CAR_XT[i];
    out[3] = CAR_R * CAR_R - (e0 * e0 + e1 * e1);
}
void carobs_term_jx(double* out, const double* x, const double*, const double* w, const void*) {
    out[0 * 4 + 0] = 1.0; out[1 * 4 + 1] = 1.0; out[2 * 4 + 2] = 1.0;
    out[0 * 4 + 3] = -2.0 * (x[0] - w[0]);
    out[1 * 4 + 3] = -2.0 * (x[1] - w[1]);
}
// state box [−1 − x; x − 1] (test/constraints.jl:13), all inequality
struct BoxCtx { int n; };
void xbox_eval(double* out, const double* x, const double*, const double*, const void* ctx) {
    int n = ((const BoxCtx*)ctx)->n;
    for (int i = 0; i < n; ++i) { out[i] = -1.0 - x[i]; out[n + i] = x[i] - 1.0; }
}
void xbox_jx(double* out, const double*, const double*, const double*, const void* ctx) {
    int n = ((const BoxCtx*)ctx)->n;
    for (int i = 0; i < n; ++i) { out[i * 2 * n + i] = -1.0; out[i * 2 * n + n + i] = 1.0; }
}
// terminal c = x, all inequality (test/constraints.jl:14,17)
void xid_eval(double* out, const double* x, const double*, const double*, const void* ctx) {
    int n = ((const BoxCtx*)ctx)->n;
    for (int i = 0; i < n; ++i) out[i] = x[i];
}
void xid_jx(double* out, const double*, const double*, const double*, const void* ctx) {
    int n = ((const BoxCtx*)ctx)->n;
    for (int i = 0; i < n; ++i) out[i * n + i] = 1.0;
}
// action box [−1 − u; u − 1] (synth32), all inequality
void ubox_eval(double* out, const double*, const double* u, const double*, const void* ctx) {
    int m = ((const BoxCtx*)ctx)->n;
    for (int i = 0; i < m; ++i) { out[i] = -1.0 - u[i]; out[m + i] = u[i] - 1.0; }
}
void ubox_ju(double* out, const double*, const double*, const double*, const void* ctx) {
    int m = ((const BoxCtx*)ctx)->n;
    for (int i = 0; i < m; ++i) { out[i * 2 * m + i] = -1.0; out[i * 2 * m + m + i] = 1.0; }
}

// ------------------------------------------------------------ problem zoo
struct Zoo {
    OrcDynamics dyn;
    QuadCtx qs, qt;
    OrcCost cs, ct;
    GoalCtx goal; BoxCtx box;
    OrcConstraint ks, kt;
    OrcDynamics dyn2; QuadCtx qs2; OrcCost cs2; OrcConstraint ks2, ks3;   // car_tv: further stage kinds
    std::vector<OrcDynamics> rdyn; std::vector<RaggedCtx> rctx; std::vector<QuadCtx> rq; std::vector<OrcCost> rcost;   // ragged: one object per step
    std::vector<const OrcDynamics*> dptr;
    std::vector<const OrcCost*> cptr;
    std::vector<const OrcConstraint*> kptr;
};

void quad_init(QuadCtx* c, int n, int m) {
    std::memset(c, 0, sizeof(*c)); c->n = n; c->m = m;
}

}  // namespace

extern "C" int orc_problem_builtin(const char* name, int T, OrcProblem* out) {
    Zoo* z = new Zoo();
    bool constrained = true;
    z->ks = make_empty_constraint(); z->kt = make_empty_constraint();
    const double PI = 3.14159265358979323846;
    if (!std::strcmp(name, "particle")) {
        z->dyn = make_dynamics<2, 1, ParticleF>();
        quad_init(&z->qs, 2, 1); quad_init(&z->qt, 2, 0);
        for (int i = 0; i < 2; ++i) { z->qs.q[i] = 0.1; z->qt.q[i] = 0.1; }
        z->qs.r[0] = 0.1;
        z->goal.n = 2; z->goal.xT[0] = 1.0; z->goal.xT[1] = 0.0;
        z->kt.evaluate = goal_eval; z->kt.jacobian_state = goal_jx; z->kt.num_constraint = 2;
        z->kt.num_state = 2; z->kt.ctx = &z->goal;
    } else if (!std::strcmp(name, "pendulum_euler")) {
        z->dyn = make_dynamics<2, 1, PendulumEulerF>();
        quad_init(&z->qs, 2, 1); quad_init(&z->qt, 2, 0);
        for (int i = 0; i < 2; ++i) { z->qs.q[i] = 1.0; z->qt.q[i] = 10.0; }
        z->qs.r[0] = 0.1;
        constrained = false;
    } else if (!std::strcmp(name, "pendulum_pole")) {
        z->dyn = make_dynamics<2, 1, PendulumPoleF>();
        quad_init(&z->qs, 2, 1); quad_init(&z->qt, 2, 0);
        for (int i = 0; i < 2; ++i) { z->qs.q[i] = 1.0; z->qt.q[i] = 10.0; }
        z->qs.r[0] = 0.1;
        constrained = false;
    } else if (!std::strcmp(name, "acrobot") || !std::strcmp(name, "acrobot_unconstrained")) {
        z->dyn = make_dynamics<4, 1, AcrobotF>();
        quad_init(&z->qs, 4, 1); quad_init(&z->qt, 4, 0);
        z->qs.q[2] = z->qs.q[3] = 0.1; z->qt.q[2] = z->qt.q[3] = 0.1; z->qs.r[0] = 0.1;
        z->goal.n = 4; z->goal.xT[0] = PI; z->goal.xT[1] = z->goal.xT[2] = z->goal.xT[3] = 0.0;
        z->kt.evaluate = goal_eval; z->kt.jacobian_state = goal_jx; z->kt.num_constraint = 4;
        z->kt.num_state = 4; z->kt.ctx = &z->goal;
        if (!std::strcmp(name, "acrobot_unconstrained")) constrained = false;
    } else if (!std::strcmp(name, "car") || !std::strcmp(name, "car_goal") || !std::strcmp(name, "car_obs")) {
        z->dyn = make_dynamics<3, 2, CarF>();
        if (!std::strcmp(name, "car_obs")) z->dyn.num_parameter = 2;
        quad_init(&z->qs, 3, 2); quad_init(&z->qt, 3, 0);
        for (int i = 0; i < 3; ++i) { z->qs.q[i] = 1.0; z->qt.q[i] = 1000.0; z->qs.xg[i] = CAR_XT[i]; z->qt.xg[i] = CAR_XT[i]; }
        z->qs.r[0] = z->qs.r[1] = 1.0e-2;
        if (!std::strcmp(name, "car")) {
            z->ks.evaluate = car_stage_eval; z->ks.jacobian_state = car_stage_jx; z->ks.jacobian_action = car_stage_ju;
            z->ks.num_constraint = 5; z->ks.num_state = 3; z->ks.num_action = 2;
            z->ks.num_inequality = 5; for (int i = 0; i < 5; ++i) z->ks.indices_inequality[i] = i;
            z->kt.evaluate = car_term_eval; z->kt.jacobian_state = car_term_jx;
            z->kt.num_constraint = 4; z->kt.num_state = 3;
            z->kt.num_inequality = 1; z->kt.indices_inequality[0] = 3;
        } else if (!std::strcmp(name, "car_obs")) {
            z->ks.evaluate = carobs_stage_eval; z->ks.jacobian_state = carobs_stage_jx; z->ks.jacobian_action = car_stage_ju;
            z->ks.num_constraint = 5; z->ks.num_state = 3; z->ks.num_action = 2;
            z->ks.num_inequality = 5; for (int i = 0; i < 5; ++i) z->ks.indices_inequality[i] = i;
            z->kt.evaluate = carobs_term_eval; z->kt.jacobian_state = carobs_term_jx;
            z->kt.num_constraint = 4; z->kt.num_state = 3;
            z->kt.num_inequality = 1; z->kt.indices_inequality[0] = 3;
        } else {   // goal-only variant (BASELINE.json configs[2])
            z->goal.n = 3; for (int i = 0; i < 3; ++i) z->goal.xT[i] = CAR_XT[i];
            z->kt.evaluate = goal_eval; z->kt.jacobian_state = goal_jx; z->kt.num_constraint = 3;
            z->kt.num_state = 3; z->kt.ctx = &z->goal;
        }
    } else if (!std::strcmp(name, "car_tv")) {
        // time-varying stage objects over the car (uniform dimensions): dynamics kind by t % 3, cost kind by
        // halves of the horizon, constraint kind by t % 4 (5 inequalities / none / 1 equality / none)
        z->dyn = make_dynamics<3, 2, CarF>();
        z->dyn2 = make_dynamics<3, 2, CarEulerF>();
        quad_init(&z->qs, 3, 2); quad_init(&z->qt, 3, 0); quad_init(&z->qs2, 3, 2);
        for (int i = 0; i < 3; ++i) { z->qs.q[i] = 1.0; z->qt.q[i] = 1000.0; z->qs.xg[i] = CAR_XT[i]; z->qt.xg[i] = CAR_XT[i]; }
        z->qs.r[0] = z->qs.r[1] = 1.0e-2;
        z->qs2.q[0] = 5.0; z->qs2.q[1] = 2.0; z->qs2.q[2] = 0.5;
        z->qs2.xg[0] = 0.9; z->qs2.xg[1] = 1.1; z->qs2.xg[2] = 0.2;
        z->qs2.r[0] = 0.05; z->qs2.r[1] = 0.02;
        z->ks.evaluate = car_stage_eval; z->ks.jacobian_state = car_stage_jx; z->ks.jacobian_action = car_stage_ju;
        z->ks.num_constraint = 5; z->ks.num_state = 3; z->ks.num_action = 2;
        z->ks.num_inequality = 5; for (int i = 0; i < 5; ++i) z->ks.indices_inequality[i] = i;
        z->ks2 = make_empty_constraint();
        z->ks3 = make_empty_constraint();
        z->ks3.evaluate = cartv_eq_eval; z->ks3.jacobian_state = cartv_eq_jx; z->ks3.jacobian_action = cartv_eq_ju;
        z->ks3.num_constraint = 1; z->ks3.num_state = 3; z->ks3.num_action = 2;
        z->kt.evaluate = car_term_eval; z->kt.jacobian_state = car_term_jx;
        z->kt.num_constraint = 4; z->kt.num_state = 3;
        z->kt.num_inequality = 1; z->kt.indices_inequality[0] = 3;
    } else if (!std::strcmp(name, "ragged")) {
        // dimensions n_t = 3,3,4,4,2,2,3,3 | 3,..., m_t = 2,1,2,1,1,2,2,1 | 2,... (period 8); stage cost
        // 0.5 sum (1 + 0.1 i) x_i^2 + 0.05 sum (1 + j) u_j^2, terminal cost 5 |x|^2, terminal equality [x_0 - 0.2, x_1 + 0.1]
        const int N = T - 1;
        z->rdyn.resize(N); z->rctx.resize(N); z->rq.resize(N); z->rcost.resize(N);
        for (int t = 0; t < N; ++t) {
            RaggedCtx& c = z->rctx[t];
            c.n0 = RAGGED_N[t % 8]; c.m0 = RAGGED_M[t % 8]; c.n1 = RAGGED_N[(t + 1) % 8];
            OrcDynamics& d = z->rdyn[t];
            d.evaluate = ragged_eval; d.jacobian_state = ragged_jx; d.jacobian_action = ragged_ju;
            d.num_state = c.n0; d.num_action = c.m0; d.num_next_state = c.n1; d.num_parameter = 0; d.ctx = &c;
            quad_init(&z->rq[t], c.n0, c.m0);
            for (int i = 0; i < c.n0; ++i) z->rq[t].q[i] = 0.5 * (1.0 + 0.1 * i);
            for (int j = 0; j < c.m0; ++j) z->rq[t].r[j] = 0.05 * (1.0 + j);
        }
        const int nT = RAGGED_N[N % 8];
        quad_init(&z->qs, 1, 1);            // unused (placeholder for the shared tail below)
        quad_init(&z->qt, nT, 0);
        for (int i = 0; i < nT; ++i) z->qt.q[i] = 5.0;
        z->goal.n = 2; z->goal.xT[0] = 0.2; z->goal.xT[1] = -0.1;
        z->kt.evaluate = goal_eval; z->kt.jacobian_state = goal_jx; z->kt.num_constraint = 2;
        z->kt.num_state = nT; z->kt.ctx = &z->goal;
    } else if (!std::strcmp(name, "synth32")) {
        z->dyn = make_dynamics<32, 8, Synth32F>();
        quad_init(&z->qs, 32, 8); quad_init(&z->qt, 32, 0);
        for (int i = 0; i < 32; ++i) { z->qs.q[i] = 0.1; z->qt.q[i] = 10.0; z->qs.xg[i] = 0.5; z->qt.xg[i] = 0.5; }
        for (int j = 0; j < 8; ++j) z->qs.r[j] = 0.01;
        z->box.n = 8;
        z->ks.evaluate = ubox_eval; z->ks.jacobian_state = con_nothing; z->ks.jacobian_action = ubox_ju;
        z->ks.num_constraint = 16; z->ks.num_state = 32; z->ks.num_action = 8; z->ks.ctx = &z->box;
        z->ks.num_inequality = 16; for (int i = 0; i < 16; ++i) z->ks.indices_inequality[i] = i;
    } else if (!std::strcmp(name, "synth12")) {
        z->dyn = make_dynamics<12, 5, Synth12F>();
        quad_init(&z->qs, 12, 5); quad_init(&z->qt, 12, 0);
        for (int i = 0; i < 12; ++i) { z->qs.q[i] = 0.1; z->qt.q[i] = 10.0; z->qs.xg[i] = 0.5; z->qt.xg[i] = 0.5; }
        for (int j = 0; j < 5; ++j) z->qs.r[j] = 0.01;
        z->box.n = 5;
        z->ks.evaluate = ubox_eval; z->ks.jacobian_state = con_nothing; z->ks.jacobian_action = ubox_ju;
        z->ks.num_constraint = 10; z->ks.num_state = 12; z->ks.num_action = 5; z->ks.ctx = &z->box;
        z->ks.num_inequality = 10; for (int i = 0; i < 10; ++i) z->ks.indices_inequality[i] = i;
        z->kt.evaluate = synth12_term_eval; z->kt.jacobian_state = synth12_term_jx;
        z->kt.num_constraint = 3; z->kt.num_state = 12;
    } else if (!std::strcmp(name, "kat_objective")) {
        // test/objective.jl:6-10 (dynamics are a placeholder: particle)
        z->dyn = make_dynamics<2, 1, ParticleF>();
        quad_init(&z->qs, 2, 1); quad_init(&z->qt, 2, 0);
        for (int i = 0; i < 2; ++i) { z->qs.q[i] = 1.0; z->qt.q[i] = 10.0; }
        z->qs.r[0] = 0.1;
        constrained = false;
    } else if (!std::strcmp(name, "kat_constraints")) {
        // test/constraints.jl:13-19
        z->dyn = make_dynamics<2, 1, ParticleF>();
        quad_init(&z->qs, 2, 1); quad_init(&z->qt, 2, 0);
        z->box.n = 2;
        z->ks.evaluate = xbox_eval; z->ks.jacobian_state = xbox_jx; z->ks.jacobian_action = con_nothing;
        z->ks.num_constraint = 4; z->ks.num_state = 2; z->ks.num_action = 1; z->ks.ctx = &z->box;
        z->ks.num_inequality = 4; for (int i = 0; i < 4; ++i) z->ks.indices_inequality[i] = i;
        z->kt.evaluate = xid_eval; z->kt.jacobian_state = xid_jx; z->kt.num_constraint = 2; z->kt.num_state = 2;
        z->kt.ctx = &z->box; z->kt.num_inequality = 2; z->kt.indices_inequality[0] = 0; z->kt.indices_inequality[1] = 1;
    } else {
        delete z;
        return -1;
    }
    z->cs = make_quad_cost(&z->qs);
    z->ct = make_quad_cost(&z->qt);
    if (!std::strcmp(name, "car_tv")) {
        z->cs2 = make_quad_cost(&z->qs2);
        for (int t = 0; t < T - 1; ++t) {
            z->dptr.push_back(t % 3 == 2 ? &z->dyn2 : &z->dyn);
            z->cptr.push_back(2 * t >= T - 1 ? &z->cs2 : &z->cs);
            z->kptr.push_back(t % 4 == 0 ? &z->ks : (t % 4 == 2 ? &z->ks3 : &z->ks2));
        }
    } else if (!std::strcmp(name, "ragged")) {
        for (int t = 0; t < T - 1; ++t) {
            z->rcost[t] = make_quad_cost(&z->rq[t]);
            z->dptr.push_back(&z->rdyn[t]); z->cptr.push_back(&z->rcost[t]); z->kptr.push_back(&z->ks);
        }
        z->dyn = z->rdyn[0]; z->dyn.num_state = 4; z->dyn.num_action = 2;     // (out->nx, out->nu below: the LARGEST dimensions)
    } else
    for (int t = 0; t < T - 1; ++t) { z->dptr.push_back(&z->dyn); z->cptr.push_back(&z->cs); z->kptr.push_back(&z->ks); }
    z->cptr.push_back(&z->ct); z->kptr.push_back(&z->kt);
    out->T = T; out->nx = z->dyn.num_state; out->nu = z->dyn.num_action; out->nw = z->dyn.num_parameter;
    out->dynamics = z->dptr.data(); out->costs = z->cptr.data();
    out->constraints = constrained ? z->kptr.data() : nullptr;
    out->owner = z;
    return 0;
}

extern "C" void orc_problem_free(OrcProblem* p) {
    if (!p || !p->owner) return;
    delete (Zoo*)p->owner;
    p->owner = nullptr; p->dynamics = nullptr;
}
