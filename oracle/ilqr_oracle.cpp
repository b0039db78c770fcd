/*
 * ilqr_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
 * Literal single-trajectory restatement of IterativeLQR.jl's hot path.
 * See ilqr_oracle.h for the pinning statement. Citations: /root/reference.
 *
 * Matrices are column-major like Julia's Matrix{Float64}; per-timestep
 * buffers are stored back to back ([t][col][row]).
 */
#include "ilqr_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

// Julia's max/min propagate NaN (Base.max(NaN, x) == NaN), unlike C's fmax/fmin.
static inline double jl_max(double a, double b) { return (a != a || b != b) ? std::numeric_limits<double>::quiet_NaN() : (a > b ? a : b); }
static inline double jl_min(double a, double b) { return (a != a || b != b) ? std::numeric_limits<double>::quiet_NaN() : (a < b ? a : b); }

struct OrcSolver {
    int T, n, m, nw;                                // n, m: the LARGEST num_state / num_action of the horizon
    // per-timestep dimensions (src/dynamics.jl:5-7: num_next_state may differ from num_state; every buffer of
    // src/data/{model,objective,policy,problem,solver}.jl is sized per timestep) and the offsets of block t in the flat arrays:
    // xo — x_t in states / gx / p / Qx and in `gradient` / `trajectory` (data.indices_state, src/data/solver.jl:23-35);
    // uo — u_t in actions / gu / k / Qu (data.indices_action = X + uo); fxo, fuo — jacobian_state[t] (nx[t+1] x nx[t]),
    // jacobian_action[t] (nx[t+1] x nu[t]); xxo — nx[t] x nx[t] blocks (gxx, P, Qxx); uuo — nu[t] x nu[t]; uxo — nu[t] x nx[t] (gux, K, Qux)
    std::vector<int> nx, nu, xo, uo, fxo, fuo, xxo, uuo, uxo;
    bool constrained;
    std::vector<const OrcDynamics*> dynamics;     // T-1
    std::vector<const OrcCost*> costs;            // T
    std::vector<const OrcConstraint*> cons;       // T (constrained only)
    OrcOptions opt;

    // ProblemData — src/data/problem.jl:3-46
    std::vector<double> states, actions;            // current trajectory
    std::vector<double> parameters;                 // T*nw
    std::vector<double> nominal_states, nominal_actions;
    std::vector<double> trajectory;                 // Δz, T*n + (T-1)*m
    // ModelData — src/data/model.jl:5-17
    std::vector<double> fx, fu;
    // ObjectiveData — src/data/objective.jl:3-21
    std::vector<double> gx, gu, gxx, guu, gux;
    // PolicyData — src/data/policy.jl:23-78
    std::vector<double> K, k, P, p, Qx, Qu, Qxx, Quu, Qux;
    std::vector<double> xx_tmp, ux_hat_tmp, uu_tmp, ux_tmp;
    // SolverData — src/data/solver.jl:4-47
    double objective, max_violation, step_size;
    std::vector<double> gradient;
    bool status;
    int iterations;
    // AugmentedLagrangianCosts — src/augmented_lagrangian.jl:1-37
    std::vector<int> coff;                          // offset of c[t] in flat arrays, T+1
    std::vector<double> violations, rho, lambda, c_tmp;
    std::vector<int> active;
    std::vector<double> active_d;                   // mirror for orc_buffer
    std::vector<double> cx, cu, cx_tmp, cu_tmp;     // Jacobians + Iρ·J temporaries
    std::vector<int> cxoff, cuoff;
    // model-call caches (evaluate_cache etc.)
    std::vector<double> cache;
    // bookkeeping (not in the reference)
    int outer_iterations, potrf_info, rollouts;
    double gradient_norm;
    OrcTrace* trace; int trace_cap, trace_len; int cur_outer;
    double last_delta = 0.0;                        // ∇Lᵀ·Δz of the last forward_pass! (src/forward_pass.jl:20)
};

extern "C" void orc_default_options(OrcOptions* o) {
    // src/options.jl:1-15
    o->line_search = 1;
    o->max_iterations = 100;
    o->max_dual_updates = 10;
    o->min_step_size = 1.0e-5;
    o->objective_tolerance = 1.0e-3;
    o->lagrangian_gradient_tolerance = 1.0e-3;
    o->constraint_tolerance = 5.0e-3;
    o->constraint_norm = std::numeric_limits<double>::infinity();
    o->initial_constraint_penalty = 1.0;
    o->scaling_penalty = 10.0;
    o->max_penalty = 1.0e8;
    o->reset_cache = 0;
    o->verbose = 0;
}

static inline int nc_at(const OrcSolver* s, int t) { return s->constrained ? s->cons[t]->num_constraint : 0; }

extern "C" OrcSolver* orc_solver_create(int T, const OrcDynamics* const* dynamics,
                                        const OrcCost* const* costs,
                                        const OrcConstraint* const* constraints,
                                        const double* w, const OrcOptions* opts) {
    OrcSolver* s = new OrcSolver();
    s->T = T;
    s->nw = dynamics[0]->num_parameter;
    const int N = T - 1;
    // dimensions along the horizon — src/data/problem.jl:32-38: states[t] has dynamics[t].num_state entries, the last one
    // dynamics[end].num_next_state; the chain must be consistent (x[t+1] .= dynamics!(d, ...) would throw otherwise, src/rollout.jl:29)
    s->nx.assign(T, 0); s->nu.assign(N, 0);
    for (int t = 0; t < N; ++t) {
        s->nx[t] = dynamics[t]->num_state; s->nu[t] = dynamics[t]->num_action;
        if (t + 1 < N && dynamics[t]->num_next_state != dynamics[t + 1]->num_state) { delete s; return nullptr; }
        if (costs[t]->num_state != s->nx[t] || costs[t]->num_action != s->nu[t]) { delete s; return nullptr; }
        s->dynamics.push_back(dynamics[t]);
    }
    s->nx[N] = dynamics[N - 1]->num_next_state;
    if (costs[N]->num_state != s->nx[N]) { delete s; return nullptr; }
    s->n = 0; s->m = 0;
    for (int t = 0; t < T; ++t) if (s->nx[t] > s->n) s->n = s->nx[t];
    for (int t = 0; t < N; ++t) if (s->nu[t] > s->m) s->m = s->nu[t];
    s->xo.assign(T + 1, 0); s->uo.assign(N + 1, 0); s->fxo.assign(N + 1, 0); s->fuo.assign(N + 1, 0);
    s->xxo.assign(T + 1, 0); s->uuo.assign(N + 1, 0); s->uxo.assign(N + 1, 0);
    for (int t = 0; t < T; ++t) { s->xo[t + 1] = s->xo[t] + s->nx[t]; s->xxo[t + 1] = s->xxo[t] + s->nx[t] * s->nx[t]; }
    for (int t = 0; t < N; ++t) {
        s->uo[t + 1] = s->uo[t] + s->nu[t];
        s->fxo[t + 1] = s->fxo[t] + s->nx[t + 1] * s->nx[t];
        s->fuo[t + 1] = s->fuo[t] + s->nx[t + 1] * s->nu[t];
        s->uuo[t + 1] = s->uuo[t] + s->nu[t] * s->nu[t];
        s->uxo[t + 1] = s->uxo[t] + s->nu[t] * s->nx[t];
    }
    const int n = s->n, m = s->m, X = s->xo[T], U = s->uo[N];
    for (int t = 0; t < T; ++t) s->costs.push_back(costs[t]);
    s->constrained = constraints != nullptr;
    if (s->constrained) for (int t = 0; t < T; ++t) s->cons.push_back(constraints[t]);
    if (opts) s->opt = *opts; else orc_default_options(&s->opt);

    // src/data/problem.jl:32-43 — everything zero-initialised
    s->states.assign(X, 0.0); s->actions.assign(U, 0.0);
    s->nominal_states.assign(X, 0.0); s->nominal_actions.assign(U, 0.0);
    s->parameters.assign(T * (s->nw > 0 ? s->nw : 0) + 1, 0.0);
    if (w && s->nw > 0) std::memcpy(s->parameters.data(), w, sizeof(double) * T * s->nw);
    s->trajectory.assign(X + U, 0.0);                                   // num_trajectory, src/dynamics.jl:52
    // src/data/model.jl:11-16, src/data/objective.jl:12-19
    s->fx.assign(s->fxo[N], 0.0); s->fu.assign(s->fuo[N], 0.0);
    s->gx.assign(X, 0.0); s->gu.assign(U, 0.0);
    s->gxx.assign(s->xxo[T], 0.0); s->guu.assign(s->uuo[N], 0.0); s->gux.assign(s->uxo[N], 0.0);
    // src/data/policy.jl:44-78
    s->K.assign(s->uxo[N], 0.0); s->k.assign(U, 0.0);
    s->P.assign(s->xxo[T], 0.0); s->p.assign(X, 0.0);
    s->Qx.assign(s->xo[N], 0.0); s->Qu.assign(U, 0.0);
    s->Qxx.assign(s->xxo[N], 0.0); s->Quu.assign(s->uuo[N], 0.0); s->Qux.assign(s->uxo[N], 0.0);
    s->xx_tmp.assign(n * n, 0.0); s->ux_hat_tmp.assign(m * n, 0.0);    // (one buffer of the largest block each; the reference keeps one per t)
    s->uu_tmp.assign(m * m, 0.0); s->ux_tmp.assign(m * n, 0.0);
    // src/data/solver.jl:37-46
    s->objective = std::numeric_limits<double>::infinity();
    s->max_violation = 0.0; s->step_size = 1.0;
    s->gradient.assign(X + U, 0.0);
    s->status = false; s->iterations = 0;
    // src/augmented_lagrangian.jl:13-37 — ρ=1, λ=0, a=1; src/data/constraints.jl:11-17 — cx[t] is nc x nx[t], cu[t] nc x nu[t]
    s->coff.assign(T + 1, 0); s->cxoff.assign(T + 1, 0); s->cuoff.assign(T + 1, 0);
    int maxnc = 0;
    for (int t = 0; t < T; ++t) {
        int nc = nc_at(s, t);
        if (nc > ORC_MAX_NC) { delete s; return nullptr; }
        if (nc > 0 && (s->cons[t]->num_state != s->nx[t] || (t < N && s->cons[t]->num_action != s->nu[t]))) { delete s; return nullptr; }
        if (nc > maxnc) maxnc = nc;
        s->coff[t + 1] = s->coff[t] + nc;
        s->cxoff[t + 1] = s->cxoff[t] + nc * s->nx[t];
        s->cuoff[t + 1] = s->cuoff[t] + (t < N ? nc * s->nu[t] : 0);
    }
    int C = s->coff[T];
    s->violations.assign(C + 1, 0.0); s->rho.assign(C + 1, 1.0); s->lambda.assign(C + 1, 0.0);
    s->c_tmp.assign(C + 1, 0.0); s->active.assign(C + 1, 1); s->active_d.assign(C + 1, 1.0);
    s->cx.assign(s->cxoff[T] + 1, 0.0); s->cu.assign(s->cuoff[T] + 1, 0.0);
    s->cx_tmp.assign(maxnc * n + 1, 0.0); s->cu_tmp.assign(maxnc * m + 1, 0.0);
    int cs = n * n; if (n * m > cs) cs = n * m; if (m * m > cs) cs = m * m;
    if (maxnc * n > cs) cs = maxnc * n;
    if (maxnc * m > cs) cs = maxnc * m;
    s->cache.assign(cs + 1, 0.0);
    s->outer_iterations = 0; s->potrf_info = 0; s->rollouts = 0; s->gradient_norm = 0.0;
    s->trace = nullptr; s->trace_cap = 0; s->trace_len = 0; s->cur_outer = 0;
    return s;
}

extern "C" void orc_solver_destroy(OrcSolver* s) { delete s; }

extern "C" void orc_initialize_controls(OrcSolver* s, const double* u) {
    // src/solver.jl:56-60 — writes the NOMINAL buffer only
    std::memcpy(s->nominal_actions.data(), u, sizeof(double) * s->uo[s->T - 1]);
}
extern "C" void orc_initialize_states(OrcSolver* s, const double* x) {
    // src/solver.jl:62-66
    std::memcpy(s->nominal_states.data(), x, sizeof(double) * s->xo[s->T]);
}

extern "C" void orc_rollout(int T, const OrcDynamics* const* dynamics, const double* x1,
                            const double* u, const double* w, double* x_out) {
    // src/rollout.jl:33-42 — open-loop rollout
    // x and u are the ragged concatenations [x_1 | x_2 | ...], [u_1 | u_2 | ...] when the dimensions vary along the horizon
    const int nw = dynamics[0]->num_parameter;
    std::memcpy(x_out, x1, sizeof(double) * dynamics[0]->num_state);
    int xo = 0, uo = 0;
    for (int t = 0; t < T - 1; ++t) {
        const OrcDynamics* d = dynamics[t];
        d->evaluate(x_out + xo + d->num_state, x_out + xo, u + uo, w ? w + t * nw : nullptr, d->ctx);
        xo += d->num_state; uo += d->num_action;
    }
}

// ---------------------------------------------------------------- L1 drivers
// src/costs.jl:48-55
static double cost_objective(OrcSolver* s, const double* x, const double* u) {
    double J = 0.0;
    const int nw = s->nw, T = s->T;
    for (int t = 0; t < T; ++t) {
        double out = 0.0;
        const OrcCost* c = s->costs[t];
        c->evaluate(&out, x + s->xo[t], t < T - 1 ? u + s->uo[t] : nullptr, s->parameters.data() + t * nw, c->ctx);
        J += out;
    }
    return J;
}

// src/constraints.jl:66-73
static void constraint_bang(OrcSolver* s, const double* x, const double* u) {
    const int nw = s->nw, T = s->T;
    for (int t = 0; t < T; ++t) {
        const OrcConstraint* con = s->cons[t];
        if (con->num_constraint == 0) continue;
        double* cache = s->cache.data();
        for (int i = 0; i < con->num_constraint; ++i) cache[i] = 0.0;
        con->evaluate(cache, x + s->xo[t], t < T - 1 ? u + s->uo[t] : nullptr, s->parameters.data() + t * nw, con->ctx);
        for (int i = 0; i < con->num_constraint; ++i) s->violations[s->coff[t] + i] = cache[i];
    }
}

static bool is_ineq(const OrcConstraint* con, int i) {
    for (int j = 0; j < con->num_inequality; ++j) if (con->indices_inequality[j] == i) return true;
    return false;
}

// src/augmented_lagrangian.jl:68-85
static void active_set_bang(OrcSolver* s) {
    for (int t = 0; t < s->T; ++t) {
        const OrcConstraint* con = s->cons[t];
        int off = s->coff[t];
        for (int i = 0; i < con->num_constraint; ++i) s->active[off + i] = 1;
        for (int j = 0; j < con->num_inequality; ++j) {
            int i = con->indices_inequality[j];
            if (s->violations[off + i] < 0.0 && s->lambda[off + i] == 0.0) s->active[off + i] = 0;
        }
    }
}

// src/augmented_lagrangian.jl:39-66
static double cost_al(OrcSolver* s, const double* x, const double* u) {
    double J = cost_objective(s, x, u);
    constraint_bang(s, x, u);
    active_set_bang(s);
    for (int t = 0; t < s->T; ++t) {
        int off = s->coff[t], nc = s->cons[t]->num_constraint;
        double dot = 0.0;
        for (int i = 0; i < nc; ++i) dot += s->lambda[off + i] * s->violations[off + i];
        J += dot;
        for (int i = 0; i < nc; ++i)
            if (s->active[off + i] == 1) {
                double c = s->violations[off + i];
                J += 0.5 * s->rho[off + i] * (c * c);   // c^2.0
            }
    }
    return J;
}

// src/data/constraints.jl:23-46
static double constraint_violation(OrcSolver* s, const double* x, const double* u) {
    constraint_bang(s, x, u);
    double mv = 0.0;
    for (int t = 0; t < s->T; ++t) {
        const OrcConstraint* con = s->cons[t];
        for (int i = 0; i < con->num_constraint; ++i) {
            double c = s->violations[s->coff[t] + i];
            double cti = is_ineq(con, i) ? jl_max(0.0, c) : std::fabs(c);
            mv = jl_max(mv, cti);
        }
    }
    return mv;
}

// src/data/methods.jl:13-30
extern "C" double orc_cost_bang(OrcSolver* s, int mode_current) {
    const double* x = mode_current ? s->states.data() : s->nominal_states.data();
    const double* u = mode_current ? s->actions.data() : s->nominal_actions.data();
    s->objective = s->constrained ? cost_al(s, x, u) : cost_objective(s, x, u);
    if (s->constrained)   // ALWAYS at problem.states / problem.actions (Appendix A, Q2)
        s->max_violation = constraint_violation(s, s->states.data(), s->actions.data());
    return s->objective;
}

// ------------------------------------------------------------- gradients.jl
extern "C" void orc_gradients(OrcSolver* s) {
    const int nw = s->nw, T = s->T, N = T - 1;
    const double* x = s->nominal_states.data();
    const double* u = s->nominal_actions.data();
    double* cache = s->cache.data();
    // gradients!(dynamics) — src/gradients.jl:1-8 → src/dynamics.jl:41-50 (`.=`); jacobian_state[t] is num_next_state x num_state
    for (int t = 0; t < N; ++t) {
        const OrcDynamics* d = s->dynamics[t];
        const int n0 = s->nx[t], m0 = s->nu[t], n1 = s->nx[t + 1];
        const double* w = s->parameters.data() + t * nw;
        for (int i = 0; i < n1 * n0; ++i) cache[i] = 0.0;
        d->jacobian_state(cache, x + s->xo[t], u + s->uo[t], w, d->ctx);
        std::memcpy(&s->fx[s->fxo[t]], cache, sizeof(double) * n1 * n0);
        for (int i = 0; i < n1 * m0; ++i) cache[i] = 0.0;
        d->jacobian_action(cache, x + s->xo[t], u + s->uo[t], w, d->ctx);
        std::memcpy(&s->fu[s->fuo[t]], cache, sizeof(double) * n1 * m0);
    }
    // gradients!(objective) — src/gradients.jl:10-21
    // cost_gradient! — src/costs.jl:57-68 (`.=`)
    for (int t = 0; t < T; ++t) {
        const OrcCost* c = s->costs[t];
        const int n0 = s->nx[t], m0 = t < N ? s->nu[t] : 0;
        const double* w = s->parameters.data() + t * nw;
        const double* ut = t < N ? u + s->uo[t] : nullptr;
        for (int i = 0; i < n0; ++i) cache[i] = 0.0;
        c->gradient_state(cache, x + s->xo[t], ut, w, c->ctx);
        for (int i = 0; i < n0; ++i) s->gx[s->xo[t] + i] = cache[i];
        if (t == N) continue;
        for (int i = 0; i < m0; ++i) cache[i] = 0.0;
        c->gradient_action(cache, x + s->xo[t], ut, w, c->ctx);
        for (int i = 0; i < m0; ++i) s->gu[s->uo[t] + i] = cache[i];
    }
    // cost_hessian! — src/costs.jl:70-84 (`.+=` : ACCUMULATES, Appendix A Q1)
    for (int t = 0; t < T; ++t) {
        const OrcCost* c = s->costs[t];
        const int n0 = s->nx[t], m0 = t < N ? s->nu[t] : 0;
        const double* w = s->parameters.data() + t * nw;
        const double* ut = t < N ? u + s->uo[t] : nullptr;
        for (int i = 0; i < n0 * n0; ++i) cache[i] = 0.0;
        c->hessian_state_state(cache, x + s->xo[t], ut, w, c->ctx);
        for (int i = 0; i < n0 * n0; ++i) s->gxx[s->xxo[t] + i] += cache[i];
        if (t == N) continue;
        for (int i = 0; i < m0 * m0; ++i) cache[i] = 0.0;
        c->hessian_action_action(cache, x + s->xo[t], ut, w, c->ctx);
        for (int i = 0; i < m0 * m0; ++i) s->guu[s->uuo[t] + i] += cache[i];
        for (int i = 0; i < m0 * n0; ++i) cache[i] = 0.0;
        c->hessian_action_state(cache, x + s->xo[t], ut, w, c->ctx);
        for (int i = 0; i < m0 * n0; ++i) s->gux[s->uxo[t] + i] += cache[i];
    }
    if (!s->constrained) return;

    // gradients!(constraint_data) — src/gradients.jl:83-90 → src/constraints.jl:75-87
    for (int t = 0; t < T; ++t) {
        const OrcConstraint* con = s->cons[t];
        int nc = con->num_constraint;
        if (nc == 0) continue;
        const int n0 = s->nx[t], m0 = t < N ? s->nu[t] : 0;
        const double* w = s->parameters.data() + t * nw;
        const double* ut = t < N ? u + s->uo[t] : nullptr;
        for (int i = 0; i < nc * n0; ++i) cache[i] = 0.0;
        con->jacobian_state(cache, x + s->xo[t], ut, w, con->ctx);
        std::memcpy(&s->cx[s->cxoff[t]], cache, sizeof(double) * nc * n0);
        if (t == N) continue;
        for (int i = 0; i < nc * m0; ++i) cache[i] = 0.0;
        con->jacobian_action(cache, x + s->xo[t], ut, w, con->ctx);
        std::memcpy(&s->cu[s->cuoff[t]], cache, sizeof(double) * nc * m0);
    }
    // AL Gauss-Newton terms — src/gradients.jl:54-80
    for (int t = 0; t < T; ++t) {
        int nc = nc_at(s, t), off = s->coff[t];
        if (nc == 0) continue;
        const int n = s->nx[t], m = t < N ? s->nu[t] : 0;
        const double* c = &s->violations[off];      // the violations BUFFER (Q2)
        const double* cxt = &s->cx[s->cxoff[t]];    // nc×n column-major
        double* ctmp = &s->c_tmp[off];
        double* cxtmp = s->cx_tmp.data();
        double* gx = &s->gx[s->xo[t]]; double* gxx = &s->gxx[s->xxo[t]];
        // Iρ = diag(ρ∘a); c_tmp = λ + Iρ c     (:56-62)
        for (int i = 0; i < nc; ++i) {
            double irho = s->rho[off + i] * (double)s->active[off + i];
            ctmp[i] = s->lambda[off + i] + irho * c[i];
        }
        // gx += cxᵀ c_tmp    (:63)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int i = 0; i < nc; ++i) acc += cxt[j * nc + i] * ctmp[i];
            gx[j] += acc;
        }
        // cx_tmp = Iρ cx ; gxx += cxᵀ cx_tmp   (:66-67)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < nc; ++i)
                cxtmp[j * nc + i] = (s->rho[off + i] * (double)s->active[off + i]) * cxt[j * nc + i];
        for (int j = 0; j < n; ++j)
            for (int i2 = 0; i2 < n; ++i2) {
                double acc = 0.0;
                for (int i = 0; i < nc; ++i) acc += cxt[i2 * nc + i] * cxtmp[j * nc + i];
                gxx[j * n + i2] += acc;
            }
        if (t == N) continue;                        // (:69)
        const double* cut = &s->cu[s->cuoff[t]];    // nc×m
        double* cutmp = s->cu_tmp.data();
        double* gu = &s->gu[s->uo[t]]; double* guu = &s->guu[s->uuo[t]]; double* gux = &s->gux[s->uxo[t]];
        // gu += cuᵀ c_tmp   (:72)
        for (int j = 0; j < m; ++j) {
            double acc = 0.0;
            for (int i = 0; i < nc; ++i) acc += cut[j * nc + i] * ctmp[i];
            gu[j] += acc;
        }
        // cu_tmp = Iρ cu ; guu += cuᵀ cu_tmp  (:75-76)
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < nc; ++i)
                cutmp[j * nc + i] = (s->rho[off + i] * (double)s->active[off + i]) * cut[j * nc + i];
        for (int j = 0; j < m; ++j)
            for (int i2 = 0; i2 < m; ++i2) {
                double acc = 0.0;
                for (int i = 0; i < nc; ++i) acc += cut[i2 * nc + i] * cutmp[j * nc + i];
                guu[j * m + i2] += acc;
            }
        // gux += cuᵀ cx_tmp   (:79)   gux is m×n
        for (int j = 0; j < n; ++j)
            for (int i2 = 0; i2 < m; ++i2) {
                double acc = 0.0;
                for (int i = 0; i < nc; ++i) acc += cut[i2 * nc + i] * cxtmp[j * nc + i];
                gux[j * m + i2] += acc;
            }
    }
}

// -------------------------------------------------------- small dense helpers
// C(r×c) = op(A)·B [+ C]; all column-major. A is (ra×ca) stored, op = transpose if tA.
static void gemm(double* C, int r, int c, const double* A, int ra, bool tA,
                 const double* B, int rb, int inner, bool accumulate) {
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) {
            double acc = 0.0;
            for (int l = 0; l < inner; ++l) {
                double a = tA ? A[i * ra + l] : A[l * ra + i];
                acc += a * B[j * rb + l];
            }
            if (accumulate) C[j * r + i] += acc; else C[j * r + i] = acc;
        }
}
// C(r×c) = A·op(B): B stored (rb×cb); op(B)=Bᵀ has element (l,j) = B[j + l*rb]
static void gemm_nt(double* C, int r, int c, const double* A, int ra,
                    const double* B, int rb, int inner, bool accumulate) {
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) {
            double acc = 0.0;
            for (int l = 0; l < inner; ++l) acc += A[l * ra + i] * B[l * rb + j];
            if (accumulate) C[j * r + i] += acc; else C[j * r + i] = acc;
        }
}

// LAPACK dpotrf('U') on an m×m column-major matrix, unblocked (dpotf2) order.
// Returns info (0 ok, j>0: leading minor j not positive definite; the
// reference IGNORES it, src/backward_pass.jl:69, Appendix A Q3).
static int potrf_U(double* A, int m) {
    for (int j = 0; j < m; ++j) {
        double ajj = A[j * m + j];
        for (int l = 0; l < j; ++l) ajj -= A[j * m + l] * A[j * m + l];
        if (!(ajj > 0.0)) { A[j * m + j] = ajj; return j + 1; }
        ajj = std::sqrt(ajj);
        A[j * m + j] = ajj;
        const double r = 1.0 / ajj;             // dpotf2: CALL DSCAL(N-J, ONE / AJJ, A(J,J+1), LDA)
        for (int c = j + 1; c < m; ++c) {
            double v = A[c * m + j];
            for (int l = 0; l < j; ++l) v -= A[j * m + l] * A[c * m + l];
            A[c * m + j] = v * r;
        }
    }
    return 0;
}
// LAPACK dpotrs('U'): solve UᵀU X = B in place, B is m×nrhs. dpotrs = two dtrsm calls; OpenBLAS's trsm kernels
// invert the diagonal once and MULTIPLY by it (tests/test_lapack_boundary.py holds this against scipy's real
// dpotrs: bitwise for m = 1, within a few ulp of max|X| for larger m, where OpenBLAS's FMA kernels round differently).
static void potrs_U(const double* U, int m, double* B, int nrhs) {
    double rd[ORC_MAX_NC];
    for (int i = 0; i < m; ++i) rd[i] = 1.0 / U[i * m + i];
    for (int c = 0; c < nrhs; ++c) {
        double* b = B + c * m;
        for (int i = 0; i < m; ++i) {           // Uᵀ y = b (forward)
            double v = b[i];
            for (int l = 0; l < i; ++l) v -= U[i * m + l] * b[l];
            b[i] = v * rd[i];
        }
        for (int i = m - 1; i >= 0; --i) {      // U x = y (backward)
            double v = b[i];
            for (int l = i + 1; l < m; ++l) v -= U[l * m + i] * b[l];
            b[i] = v * rd[i];
        }
    }
}

// the two LAPACK stand-ins, exported so that tests can hold them against scipy's real dpotrf / dpotrs
extern "C" int orc_potrf_U(double* A, int m) { return potrf_U(A, m); }
extern "C" void orc_potrs_U(const double* U, int m, double* B, int nrhs) { potrs_U(U, m, B, nrhs); }

// --------------------------------------------------------- backward_pass.jl
extern "C" void orc_backward_pass(OrcSolver* s) {
    const int T = s->T, N = T - 1;
    // P[H] .= gxx[H]; p[H] .= gx[H]     (:39-40)
    std::memcpy(&s->P[s->xxo[N]], &s->gxx[s->xxo[N]], sizeof(double) * s->nx[N] * s->nx[N]);
    std::memcpy(&s->p[s->xo[N]], &s->gx[s->xo[N]], sizeof(double) * s->nx[N]);
    for (int t = N - 1; t >= 0; --t) {        // (:42)
        // n = num_state, m = num_action, n1 = num_next_state of step t (src/data/policy.jl:44-72: K[t] is m x n, P[t] n x n,
        // xx̂_tmp[t] n x n1, ux̂_tmp[t] m x n1)
        const int n = s->nx[t], m = s->nu[t], n1 = s->nx[t + 1];
        const double* fx = &s->fx[s->fxo[t]];  // n1×n
        const double* fu = &s->fu[s->fuo[t]];  // n1×m
        const double* Pn = &s->P[s->xxo[t + 1]];
        const double* pn = &s->p[s->xo[t + 1]];
        const double* gx = &s->gx[s->xo[t]]; const double* gu = &s->gu[s->uo[t]];
        const double* gxx = &s->gxx[s->xxo[t]]; const double* guu = &s->guu[s->uuo[t]]; const double* gux = &s->gux[s->uxo[t]];
        double* Qx = &s->Qx[s->xo[t]]; double* Qu = &s->Qu[s->uo[t]];
        double* Qxx = &s->Qxx[s->xxo[t]]; double* Quu = &s->Quu[s->uuo[t]]; double* Qux = &s->Qux[s->uxo[t]];
        double* K = &s->K[s->uxo[t]]; double* k = &s->k[s->uo[t]];
        double* P = &s->P[s->xxo[t]]; double* p = &s->p[s->xo[t]];
        // Qx = fxᵀp' + gx   (:44-45)
        gemm(Qx, n, 1, fx, n1, true, pn, n1, n1, false);
        for (int i = 0; i < n; ++i) Qx[i] += gx[i];
        // Qu = fuᵀp' + gu   (:48-49)
        gemm(Qu, m, 1, fu, n1, true, pn, n1, n1, false);
        for (int i = 0; i < m; ++i) Qu[i] += gu[i];
        // Qxx = (fxᵀP')fx + gxx   (:52-54)
        gemm(s->xx_tmp.data(), n, n1, fx, n1, true, Pn, n1, n1, false);
        gemm(Qxx, n, n, s->xx_tmp.data(), n, false, fx, n1, n1, false);
        for (int i = 0; i < n * n; ++i) Qxx[i] += gxx[i];
        // Quu = (fuᵀP')fu + guu   (:57-59)
        gemm(s->ux_hat_tmp.data(), m, n1, fu, n1, true, Pn, n1, n1, false);
        gemm(Quu, m, m, s->ux_hat_tmp.data(), m, false, fu, n1, n1, false);
        for (int i = 0; i < m * m; ++i) Quu[i] += guu[i];
        // Qux = (fuᵀP')fx + gux   (:62-64)
        gemm(s->ux_hat_tmp.data(), m, n1, fu, n1, true, Pn, n1, n1, false);
        gemm(Qux, m, n, s->ux_hat_tmp.data(), m, false, fx, n1, n1, false);
        for (int i = 0; i < m * n; ++i) Qux[i] += gux[i];
        // potrf/potrs   (:68-75)
        std::memcpy(s->uu_tmp.data(), Quu, sizeof(double) * m * m);
        int info = potrf_U(s->uu_tmp.data(), m);
        if (info != 0 && s->potrf_info == 0) s->potrf_info = info;
        std::memcpy(K, Qux, sizeof(double) * m * n);
        std::memcpy(k, Qu, sizeof(double) * m);
        potrs_U(s->uu_tmp.data(), m, K, n);
        potrs_U(s->uu_tmp.data(), m, k, 1);
        for (int i = 0; i < m * n; ++i) K[i] *= -1.0;
        for (int i = 0; i < m; ++i) k[i] *= -1.0;
        // ux_tmp = Quu K     (:79)
        gemm(s->ux_tmp.data(), m, n, Quu, m, false, K, m, m, false);
        // P = Kᵀ ux_tmp + Kᵀ Qux + Quxᵀ K + Qxx     (:81-84)
        gemm(P, n, n, K, m, true, s->ux_tmp.data(), m, m, false);
        gemm(P, n, n, K, m, true, Qux, m, m, true);
        gemm(P, n, n, Qux, m, true, K, m, m, true);
        for (int i = 0; i < n * n; ++i) P[i] += Qxx[i];
        // p = ux_tmpᵀ k + Kᵀ Qu + Quxᵀ k + Qx        (:86-89)
        gemm(p, n, 1, s->ux_tmp.data(), m, true, k, m, m, false);
        gemm(p, n, 1, K, m, true, Qu, m, m, true);
        gemm(p, n, 1, Qux, m, true, k, m, m, true);
        for (int i = 0; i < n; ++i) p[i] += Qx[i];
    }
    (void)gemm_nt;
}

// src/solve.jl:67-83
extern "C" void orc_lagrangian_gradient(OrcSolver* s) {
    const int T = s->T, N = T - 1, X = s->xo[T];       // data.indices_state / indices_action, src/data/solver.jl:23-35
    for (int t = 0; t < N; ++t) {
        for (int i = 0; i < s->nx[t]; ++i) s->gradient[s->xo[t] + i] = s->Qx[s->xo[t] + i] - s->p[s->xo[t] + i];
        for (int i = 0; i < s->nu[t]; ++i) s->gradient[X + s->uo[t] + i] = s->Qu[s->uo[t] + i];
    }
    // gradient wrt x_T is left untouched
}

// src/data/methods.jl:42-54
static void trajectory_sensitivities(OrcSolver* s) {
    const int T = s->T, N = T - 1, X = s->xo[T];
    std::fill(s->trajectory.begin(), s->trajectory.end(), 0.0);
    for (int t = 0; t < N; ++t) {
        const int n = s->nx[t], m = s->nu[t], n1 = s->nx[t + 1];
        double* zx = &s->trajectory[s->xo[t]];
        double* zu = &s->trajectory[X + s->uo[t]];
        double* zy = &s->trajectory[s->xo[t + 1]];
        for (int i = 0; i < m; ++i) zu[i] = s->k[s->uo[t] + i];
        gemm(zu, m, 1, &s->K[s->uxo[t]], m, false, zx, n, n, true);
        gemm(zy, n1, 1, &s->fu[s->fuo[t]], n1, false, zu, m, m, false);
        gemm(zy, n1, 1, &s->fx[s->fxo[t]], n1, false, zx, n, n, true);
    }
}

// src/rollout.jl:1-31
extern "C" void orc_rollout_bang(OrcSolver* s, double step_size) {
    const int nw = s->nw, T = s->T, N = T - 1;
    double* x = s->states.data(); double* u = s->actions.data();
    const double* xb = s->nominal_states.data(); const double* ub = s->nominal_actions.data();
    for (int i = 0; i < s->nx[0]; ++i) x[i] = xb[i];     // (:19)
    for (int t = 0; t < N; ++t) {
        const int n = s->nx[t], m = s->nu[t];
        const double* K = &s->K[s->uxo[t]];
        const double* xt = x + s->xo[t]; const double* xbt = xb + s->xo[t];
        double* ut = u + s->uo[t];
        for (int i = 0; i < m; ++i) ut[i] = s->k[s->uo[t] + i];        // (:24)
        for (int i = 0; i < m; ++i) ut[i] *= step_size;                // (:25)
        for (int i = 0; i < m; ++i) ut[i] += ub[s->uo[t] + i];         // (:26)
        for (int i = 0; i < m; ++i) {                                  // (:27) u += K x
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += K[j * m + i] * xt[j];
            ut[i] += acc;
        }
        for (int i = 0; i < m; ++i) {                                  // (:28) u -= K x̄
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += K[j * m + i] * xbt[j];
            ut[i] += -1.0 * acc;
        }
        const OrcDynamics* d = s->dynamics[t];
        d->evaluate(x + s->xo[t + 1], xt, ut, s->parameters.data() + t * nw, d->ctx);   // (:29)
    }
    s->rollouts++;
}

// src/data/methods.jl:32-39
static void update_nominal_trajectory(OrcSolver* s) {
    s->nominal_states = s->states;
    s->nominal_actions = s->actions;
}

// src/forward_pass.jl:1-56
extern "C" void orc_forward_pass(OrcSolver* s) {
    const double c1 = 1.0e-4;
    const int max_iterations = 25;
    s->status = false;                                  // (:10)
    double J_prev = s->objective;                       // (:13)
    orc_lagrangian_gradient(s);                         // (:16)
    double delta_grad_product = 0.0;
    if (s->opt.line_search == 1) {                      // (:18-23)
        trajectory_sensitivities(s);
        for (size_t i = 0; i < s->gradient.size(); ++i) delta_grad_product += s->gradient[i] * s->trajectory[i];
    }
    s->last_delta = delta_grad_product;                 // bookkeeping for the parity tests (not in the reference)
    s->step_size = 1.0;                                 // (:26)
    int iteration = 1;
    while (s->step_size >= s->opt.min_step_size) {      // (:28)
        if (iteration > max_iterations) break;          // (:29)
        orc_rollout_bang(s, s->step_size);              // (:34)
        double J = orc_cost_bang(s, 1);                 // (:36)  writes data.objective
        if (J <= J_prev + c1 * s->step_size * delta_grad_product) {   // (:44)  NaN ⇒ reject
            update_nominal_trajectory(s);
            s->objective = J;
            s->status = true;
            break;
        } else {
            s->step_size *= 0.5;                        // (:51)
            iteration += 1;
        }
    }
}

extern "C" void orc_reset_model_objective(OrcSolver* s) {
    // reset!(problem.model) — src/data/model.jl:19-26 ; reset!(problem.objective) — src/data/objective.jl:23-33
    std::fill(s->fx.begin(), s->fx.end(), 0.0); std::fill(s->fu.begin(), s->fu.end(), 0.0);
    std::fill(s->gx.begin(), s->gx.end(), 0.0); std::fill(s->gu.begin(), s->gu.end(), 0.0);
    std::fill(s->gxx.begin(), s->gxx.end(), 0.0); std::fill(s->guu.begin(), s->guu.end(), 0.0);
    std::fill(s->gux.begin(), s->gux.end(), 0.0);
}

static void reset_solver_data(OrcSolver* s) {
    // src/data/solver.jl:49-59
    s->objective = 0.0;
    std::fill(s->gradient.begin(), s->gradient.end(), 0.0);
    s->max_violation = 0.0;
    s->status = false;
    s->iterations = 0;
}

static double norm_inf(const std::vector<double>& v) {
    double r = 0.0;
    for (double a : v) { double f = std::fabs(a); if (f > r || f != f) r = f; }
    return r;
}

// src/solve.jl:1-54
extern "C" void orc_ilqr_solve(OrcSolver* s) {
    orc_reset_model_objective(s);                       // (:9-10)  the ONLY place Hessians are zeroed
    if (s->opt.reset_cache) reset_solver_data(s);       // (:12)
    orc_cost_bang(s, 0);                                // (:14)
    orc_gradients(s);                                   // (:16)
    orc_backward_pass(s);                               // (:18)
    double obj_prev = s->objective;                     // (:21)
    for (int i = 1; i <= s->opt.max_iterations; ++i) {  // (:22)
        orc_forward_pass(s);                            // (:23)
        if (s->opt.line_search != 0) {                  // (:27-33)
            orc_gradients(s);
            orc_backward_pass(s);
            orc_lagrangian_gradient(s);
        }
        double gradient_norm = norm_inf(s->gradient);   // (:36)
        s->gradient_norm = gradient_norm;
        s->iterations += 1;                             // (:39)
        if (s->trace && s->trace_len < s->trace_cap) {
            OrcTrace& r = s->trace[s->trace_len++];
            r.outer = s->cur_outer; r.inner = i; r.objective = s->objective; r.gradient_norm = gradient_norm;
            r.max_violation = s->max_violation; r.step_size = s->step_size; r.status = s->status ? 1 : 0;
        }
        if (s->opt.verbose)
            std::printf("iter: %d cost: %.12g gradient_norm: %.6g max_violation: %.6g step_size: %.6g\n",
                        i, s->objective, gradient_norm, s->max_violation, s->step_size);
        if (gradient_norm < s->opt.lagrangian_gradient_tolerance) break;          // (:48)
        if (std::fabs(s->objective - obj_prev) < s->opt.objective_tolerance) break; // (:49)
        else obj_prev = s->objective;
        if (!s->status) break;                                                    // (:50)
    }
}

// src/augmented_lagrangian.jl:87-110
extern "C" void orc_augmented_lagrangian_update(OrcSolver* s) {
    for (int t = 0; t < s->T; ++t) {
        const OrcConstraint* con = s->cons[t];
        int off = s->coff[t];
        for (int i = 0; i < con->num_constraint; ++i) {
            s->lambda[off + i] += s->rho[off + i] * s->violations[off + i];
            if (is_ineq(con, i)) s->lambda[off + i] = jl_max(0.0, s->lambda[off + i]);
            s->rho[off + i] = jl_min(s->opt.scaling_penalty * s->rho[off + i], s->opt.max_penalty);
        }
    }
}

// src/solve.jl:88-129
static void constrained_ilqr_solve(OrcSolver* s) {
    reset_solver_data(s);                                            // (:93)
    int C = s->coff[s->T];
    for (int i = 0; i < C; ++i) s->lambda[i] = 0.0;                  // (:96-98)
    for (int i = 0; i < C; ++i) s->rho[i] = s->opt.initial_constraint_penalty;   // (:101-103)
    s->outer_iterations = 0;
    for (int i = 1; i <= s->opt.max_dual_updates; ++i) {             // (:105)
        s->cur_outer = i;
        s->outer_iterations = i;
        orc_ilqr_solve(s);                                           // (:109)
        orc_cost_bang(s, 0);                                         // (:113)
        if (s->max_violation <= s->opt.constraint_tolerance) break;  // (:117)
        orc_augmented_lagrangian_update(s);                          // (:120-122)
    }
}

// src/solve.jl:137-143
extern "C" void orc_solve(OrcSolver* s) {
    s->potrf_info = 0; s->rollouts = 0; s->trace_len = 0;
    if (s->constrained) constrained_ilqr_solve(s);
    else { s->cur_outer = 0; s->outer_iterations = 0; orc_ilqr_solve(s); }
}

extern "C" void orc_get_stats(const OrcSolver* s, OrcStats* st) {
    st->objective = s->objective; st->gradient_norm = s->gradient_norm;
    st->max_violation = s->max_violation; st->step_size = s->step_size;
    st->iterations = s->iterations; st->outer_iterations = s->outer_iterations;
    st->status = s->status ? 1 : 0; st->potrf_info = s->potrf_info; st->rollouts = s->rollouts;
}

extern "C" void orc_set_trace(OrcSolver* s, OrcTrace* buf, int capacity) {
    s->trace = buf; s->trace_cap = capacity; s->trace_len = 0;
}
extern "C" int orc_trace_len(const OrcSolver* s) { return s->trace_len; }
extern "C" double orc_last_delta(const OrcSolver* s) { return s->last_delta; }
extern "C" void orc_set_active_set(OrcSolver* s, const double* a) {
    for (int i = 0; i < s->coff[s->T]; ++i) s->active[i] = a[i] != 0.0 ? 1 : 0;
}
// SolverData scalars written directly (parity tests that start a stage from a given state)
extern "C" void orc_set_scalars(OrcSolver* s, double objective, double max_violation, double step_size, int status) {
    s->objective = objective; s->max_violation = max_violation; s->step_size = step_size; s->status = status != 0;
}

extern "C" double* orc_buffer(OrcSolver* s, const char* name, int* len) {
#define BUF(nm, vec) if (!std::strcmp(name, nm)) { if (len) *len = (int)(vec).size(); return (vec).data(); }
    BUF("nominal_states", s->nominal_states) BUF("nominal_actions", s->nominal_actions)
    BUF("states", s->states) BUF("actions", s->actions)
    BUF("jacobian_state", s->fx) BUF("jacobian_action", s->fu)
    BUF("gradient_state", s->gx) BUF("gradient_action", s->gu)
    BUF("hessian_state_state", s->gxx) BUF("hessian_action_action", s->guu) BUF("hessian_action_state", s->gux)
    BUF("K", s->K) BUF("k", s->k) BUF("P", s->P) BUF("p", s->p)
    BUF("Qx", s->Qx) BUF("Qu", s->Qu) BUF("Qxx", s->Qxx) BUF("Quu", s->Quu) BUF("Qux", s->Qux)
    BUF("gradient", s->gradient) BUF("trajectory", s->trajectory)
#undef BUF
    int C = s->coff[s->T];
    if (!std::strcmp(name, "violations")) { if (len) *len = C; return s->violations.data(); }
    if (!std::strcmp(name, "constraint_dual")) { if (len) *len = C; return s->lambda.data(); }
    if (!std::strcmp(name, "constraint_penalty")) { if (len) *len = C; return s->rho.data(); }
    if (!std::strcmp(name, "active_set")) {
        for (int i = 0; i < C; ++i) s->active_d[i] = (double)s->active[i];
        if (len) *len = C;
        return s->active_d.data();
    }
    if (len) *len = 0;
    return nullptr;
}

// per-timestep dimensions of a problem: nx_t[T] (num_state of every step, num_next_state of the last), nu_t[T-1]
extern "C" void orc_problem_dims(const OrcProblem* p, int* nx_t, int* nu_t) {
    const int N = p->T - 1;
    for (int t = 0; t < N; ++t) { nx_t[t] = p->dynamics[t]->num_state; nu_t[t] = p->dynamics[t]->num_action; }
    nx_t[N] = p->dynamics[N - 1]->num_next_state;
}

// ------------------------------------------------------------- batch driver
extern "C" int orc_solve_batch_w(const char* model, int T, int B, const double* x1,
                                 const double* ubar, const double* w, const OrcOptions* opts, int nthreads,
                                 double* x_out, double* u_out, double* K_out, double* k_out,
                                 OrcStats* stats_out);

extern "C" int orc_solve_batch(const char* model, int T, int B, const double* x1,
                               const double* ubar, const OrcOptions* opts, int nthreads,
                               double* x_out, double* u_out, double* K_out, double* k_out,
                               OrcStats* stats_out) {
    return orc_solve_batch_w(model, T, B, x1, ubar, nullptr, opts, nthreads, x_out, u_out, K_out, k_out, stats_out);
}

// same with per-instance parameters w: [B][T][nw] (Solver(...; parameters=θ), src/solver.jl:12,29)
extern "C" int orc_solve_batch_w(const char* model, int T, int B, const double* x1,
                                 const double* ubar, const double* w, const OrcOptions* opts, int nthreads,
                                 double* x_out, double* u_out, double* K_out, double* k_out,
                                 OrcStats* stats_out) {
    OrcProblem prob;
    if (orc_problem_builtin(model, T, &prob) != 0) return -1;
    // Host arrays are PADDED to the largest dimensions of the horizon (n = prob.nx, m = prob.nu: what the device path's
    // zero-padded template uses): x[b][t][n], u[b][t][m], K[b][t][n][m] column-major m x n — for a problem with uniform
    // dimensions that is the plain layout. The solver itself works on the ragged per-timestep blocks.
    const int n = prob.nx, m = prob.nu, N = T - 1;
    std::vector<int> nxt(T), nut(N);
    for (int t = 0; t < N; ++t) { nxt[t] = prob.dynamics[t]->num_state; nut[t] = prob.dynamics[t]->num_action; }
    nxt[N] = prob.dynamics[N - 1]->num_next_state;
    int X = 0, U = 0;
    for (int t = 0; t < T; ++t) X += nxt[t];
    for (int t = 0; t < N; ++t) U += nut[t];
    int fail = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int b = 0; b < B; ++b) {
        const double* wb = (w && prob.nw > 0) ? w + (size_t)b * T * prob.nw : nullptr;
        OrcSolver* s = orc_solver_create(T, prob.dynamics, prob.costs, prob.constraints, wb, opts);
        if (!s) {
#pragma omp atomic write
            fail = 1;
            continue;
        }
        std::vector<double> xbar(X), ur(U + 1);
        for (int t = 0; t < N; ++t)
            for (int i = 0; i < nut[t]; ++i) ur[s->uo[t] + i] = ubar[((size_t)b * N + t) * m + i];
        orc_rollout(T, prob.dynamics, x1 + (size_t)b * n, ur.data(), wb, xbar.data());
        orc_initialize_controls(s, ur.data());
        orc_initialize_states(s, xbar.data());
        orc_solve(s);
        for (int t = 0; t < T; ++t) {
            if (x_out) {
                double* o = x_out + ((size_t)b * T + t) * n;
                for (int i = 0; i < n; ++i) o[i] = i < nxt[t] ? s->nominal_states[s->xo[t] + i] : 0.0;
            }
            if (t == N) continue;
            if (u_out) {
                double* o = u_out + ((size_t)b * N + t) * m;
                for (int i = 0; i < m; ++i) o[i] = i < nut[t] ? s->nominal_actions[s->uo[t] + i] : 0.0;
            }
            if (k_out) {
                double* o = k_out + ((size_t)b * N + t) * m;
                for (int i = 0; i < m; ++i) o[i] = i < nut[t] ? s->k[s->uo[t] + i] : 0.0;
            }
            if (K_out) {
                double* o = K_out + ((size_t)b * N + t) * m * n;
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i < m; ++i)
                        o[j * m + i] = (i < nut[t] && j < nxt[t]) ? s->K[s->uxo[t] + j * nut[t] + i] : 0.0;
            }
        }
        if (stats_out) orc_get_stats(s, &stats_out[b]);
        orc_solver_destroy(s);
    }
    orc_problem_free(&prob);
    return fail ? -2 : 0;
}
