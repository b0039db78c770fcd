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
    int T, n, m, nw;
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
    s->n = dynamics[0]->num_state;
    s->m = dynamics[0]->num_action;
    s->nw = dynamics[0]->num_parameter;
    const int n = s->n, m = s->m, N = T - 1;
    for (int t = 0; t < N; ++t) {
        if (dynamics[t]->num_state != n || dynamics[t]->num_next_state != n || dynamics[t]->num_action != m) {
            delete s; return nullptr;   // uniform dimensions only (see header)
        }
        s->dynamics.push_back(dynamics[t]);
    }
    for (int t = 0; t < T; ++t) s->costs.push_back(costs[t]);
    s->constrained = constraints != nullptr;
    if (s->constrained) for (int t = 0; t < T; ++t) s->cons.push_back(constraints[t]);
    if (opts) s->opt = *opts; else orc_default_options(&s->opt);

    // src/data/problem.jl:32-43 — everything zero-initialised
    s->states.assign(T * n, 0.0); s->actions.assign(N * m, 0.0);
    s->nominal_states.assign(T * n, 0.0); s->nominal_actions.assign(N * m, 0.0);
    s->parameters.assign(T * (s->nw > 0 ? s->nw : 0) + 1, 0.0);
    if (w && s->nw > 0) std::memcpy(s->parameters.data(), w, sizeof(double) * T * s->nw);
    s->trajectory.assign(T * n + N * m, 0.0);
    s->fx.assign(N * n * n, 0.0); s->fu.assign(N * n * m, 0.0);
    s->gx.assign(T * n, 0.0); s->gu.assign(N * m, 0.0);
    s->gxx.assign(T * n * n, 0.0); s->guu.assign(N * m * m, 0.0); s->gux.assign(N * m * n, 0.0);
    // src/data/policy.jl:44-78
    s->K.assign(N * m * n, 0.0); s->k.assign(N * m, 0.0);
    s->P.assign(T * n * n, 0.0); s->p.assign(T * n, 0.0);
    s->Qx.assign(N * n, 0.0); s->Qu.assign(N * m, 0.0);
    s->Qxx.assign(N * n * n, 0.0); s->Quu.assign(N * m * m, 0.0); s->Qux.assign(N * m * n, 0.0);
    s->xx_tmp.assign(n * n, 0.0); s->ux_hat_tmp.assign(m * n, 0.0);
    s->uu_tmp.assign(m * m, 0.0); s->ux_tmp.assign(m * n, 0.0);
    // src/data/solver.jl:37-46
    s->objective = std::numeric_limits<double>::infinity();
    s->max_violation = 0.0; s->step_size = 1.0;
    s->gradient.assign(T * n + N * m, 0.0);
    s->status = false; s->iterations = 0;
    // src/augmented_lagrangian.jl:13-37 — ρ=1, λ=0, a=1
    s->coff.assign(T + 1, 0); s->cxoff.assign(T + 1, 0); s->cuoff.assign(T + 1, 0);
    int maxnc = 0;
    for (int t = 0; t < T; ++t) {
        int nc = nc_at(s, t);
        if (nc > ORC_MAX_NC) { delete s; return nullptr; }
        if (nc > maxnc) maxnc = nc;
        s->coff[t + 1] = s->coff[t] + nc;
        s->cxoff[t + 1] = s->cxoff[t] + nc * n;
        s->cuoff[t + 1] = s->cuoff[t] + (t < N ? nc * m : 0);
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
    std::memcpy(s->nominal_actions.data(), u, sizeof(double) * (s->T - 1) * s->m);
}
extern "C" void orc_initialize_states(OrcSolver* s, const double* x) {
    // src/solver.jl:62-66
    std::memcpy(s->nominal_states.data(), x, sizeof(double) * s->T * s->n);
}

extern "C" void orc_rollout(int T, const OrcDynamics* const* dynamics, const double* x1,
                            const double* u, const double* w, double* x_out) {
    // src/rollout.jl:33-42 — open-loop rollout
    const int n = dynamics[0]->num_state, m = dynamics[0]->num_action, nw = dynamics[0]->num_parameter;
    std::memcpy(x_out, x1, sizeof(double) * n);
    for (int t = 0; t < T - 1; ++t) {
        const OrcDynamics* d = dynamics[t];
        d->evaluate(x_out + (t + 1) * n, x_out + t * n, u + t * m, w ? w + t * nw : nullptr, d->ctx);
    }
}

// ---------------------------------------------------------------- L1 drivers
// src/costs.jl:48-55
static double cost_objective(OrcSolver* s, const double* x, const double* u) {
    double J = 0.0;
    const int n = s->n, m = s->m, nw = s->nw, T = s->T;
    for (int t = 0; t < T; ++t) {
        double out = 0.0;
        const OrcCost* c = s->costs[t];
        c->evaluate(&out, x + t * n, t < T - 1 ? u + t * m : nullptr, s->parameters.data() + t * nw, c->ctx);
        J += out;
    }
    return J;
}

// src/constraints.jl:66-73
static void constraint_bang(OrcSolver* s, const double* x, const double* u) {
    const int n = s->n, m = s->m, nw = s->nw, T = s->T;
    for (int t = 0; t < T; ++t) {
        const OrcConstraint* con = s->cons[t];
        if (con->num_constraint == 0) continue;
        double* cache = s->cache.data();
        for (int i = 0; i < con->num_constraint; ++i) cache[i] = 0.0;
        con->evaluate(cache, x + t * n, t < T - 1 ? u + t * m : nullptr, s->parameters.data() + t * nw, con->ctx);
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
    const int n = s->n, m = s->m, nw = s->nw, T = s->T, N = T - 1;
    const double* x = s->nominal_states.data();
    const double* u = s->nominal_actions.data();
    double* cache = s->cache.data();
    // gradients!(dynamics) — src/gradients.jl:1-8 → src/dynamics.jl:41-50 (`.=`)
    for (int t = 0; t < N; ++t) {
        const OrcDynamics* d = s->dynamics[t];
        const double* w = s->parameters.data() + t * nw;
        for (int i = 0; i < n * n; ++i) cache[i] = 0.0;
        d->jacobian_state(cache, x + t * n, u + t * m, w, d->ctx);
        std::memcpy(&s->fx[t * n * n], cache, sizeof(double) * n * n);
        for (int i = 0; i < n * m; ++i) cache[i] = 0.0;
        d->jacobian_action(cache, x + t * n, u + t * m, w, d->ctx);
        std::memcpy(&s->fu[t * n * m], cache, sizeof(double) * n * m);
    }
    // gradients!(objective) — src/gradients.jl:10-21
    // cost_gradient! — src/costs.jl:57-68 (`.=`)
    for (int t = 0; t < T; ++t) {
        const OrcCost* c = s->costs[t];
        const double* w = s->parameters.data() + t * nw;
        const double* ut = t < N ? u + t * m : nullptr;
        for (int i = 0; i < n; ++i) cache[i] = 0.0;
        c->gradient_state(cache, x + t * n, ut, w, c->ctx);
        for (int i = 0; i < n; ++i) s->gx[t * n + i] = cache[i];
        if (t == N) continue;
        for (int i = 0; i < m; ++i) cache[i] = 0.0;
        c->gradient_action(cache, x + t * n, ut, w, c->ctx);
        for (int i = 0; i < m; ++i) s->gu[t * m + i] = cache[i];
    }
    // cost_hessian! — src/costs.jl:70-84 (`.+=` : ACCUMULATES, Appendix A Q1)
    for (int t = 0; t < T; ++t) {
        const OrcCost* c = s->costs[t];
        const double* w = s->parameters.data() + t * nw;
        const double* ut = t < N ? u + t * m : nullptr;
        for (int i = 0; i < n * n; ++i) cache[i] = 0.0;
        c->hessian_state_state(cache, x + t * n, ut, w, c->ctx);
        for (int i = 0; i < n * n; ++i) s->gxx[t * n * n + i] += cache[i];
        if (t == N) continue;
        for (int i = 0; i < m * m; ++i) cache[i] = 0.0;
        c->hessian_action_action(cache, x + t * n, ut, w, c->ctx);
        for (int i = 0; i < m * m; ++i) s->guu[t * m * m + i] += cache[i];
        for (int i = 0; i < m * n; ++i) cache[i] = 0.0;
        c->hessian_action_state(cache, x + t * n, ut, w, c->ctx);
        for (int i = 0; i < m * n; ++i) s->gux[t * m * n + i] += cache[i];
    }
    if (!s->constrained) return;

    // gradients!(constraint_data) — src/gradients.jl:83-90 → src/constraints.jl:75-87
    for (int t = 0; t < T; ++t) {
        const OrcConstraint* con = s->cons[t];
        int nc = con->num_constraint;
        if (nc == 0) continue;
        const double* w = s->parameters.data() + t * nw;
        const double* ut = t < N ? u + t * m : nullptr;
        for (int i = 0; i < nc * n; ++i) cache[i] = 0.0;
        con->jacobian_state(cache, x + t * n, ut, w, con->ctx);
        std::memcpy(&s->cx[s->cxoff[t]], cache, sizeof(double) * nc * n);
        if (t == N) continue;
        for (int i = 0; i < nc * m; ++i) cache[i] = 0.0;
        con->jacobian_action(cache, x + t * n, ut, w, con->ctx);
        std::memcpy(&s->cu[s->cuoff[t]], cache, sizeof(double) * nc * m);
    }
    // AL Gauss-Newton terms — src/gradients.jl:54-80
    for (int t = 0; t < T; ++t) {
        int nc = nc_at(s, t), off = s->coff[t];
        if (nc == 0) continue;
        const double* c = &s->violations[off];      // the violations BUFFER (Q2)
        const double* cxt = &s->cx[s->cxoff[t]];    // nc×n column-major
        double* ctmp = &s->c_tmp[off];
        double* cxtmp = s->cx_tmp.data();
        // Iρ = diag(ρ∘a); c_tmp = λ + Iρ c     (:56-62)
        for (int i = 0; i < nc; ++i) {
            double irho = s->rho[off + i] * (double)s->active[off + i];
            ctmp[i] = s->lambda[off + i] + irho * c[i];
        }
        // gx += cxᵀ c_tmp    (:63)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int i = 0; i < nc; ++i) acc += cxt[j * nc + i] * ctmp[i];
            s->gx[t * n + j] += acc;
        }
        // cx_tmp = Iρ cx ; gxx += cxᵀ cx_tmp   (:66-67)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < nc; ++i)
                cxtmp[j * nc + i] = (s->rho[off + i] * (double)s->active[off + i]) * cxt[j * nc + i];
        for (int j = 0; j < n; ++j)
            for (int i2 = 0; i2 < n; ++i2) {
                double acc = 0.0;
                for (int i = 0; i < nc; ++i) acc += cxt[i2 * nc + i] * cxtmp[j * nc + i];
                s->gxx[t * n * n + j * n + i2] += acc;
            }
        if (t == N) continue;                        // (:69)
        const double* cut = &s->cu[s->cuoff[t]];    // nc×m
        double* cutmp = s->cu_tmp.data();
        // gu += cuᵀ c_tmp   (:72)
        for (int j = 0; j < m; ++j) {
            double acc = 0.0;
            for (int i = 0; i < nc; ++i) acc += cut[j * nc + i] * ctmp[i];
            s->gu[t * m + j] += acc;
        }
        // cu_tmp = Iρ cu ; guu += cuᵀ cu_tmp  (:75-76)
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < nc; ++i)
                cutmp[j * nc + i] = (s->rho[off + i] * (double)s->active[off + i]) * cut[j * nc + i];
        for (int j = 0; j < m; ++j)
            for (int i2 = 0; i2 < m; ++i2) {
                double acc = 0.0;
                for (int i = 0; i < nc; ++i) acc += cut[i2 * nc + i] * cutmp[j * nc + i];
                s->guu[t * m * m + j * m + i2] += acc;
            }
        // gux += cuᵀ cx_tmp   (:79)   gux is m×n
        for (int j = 0; j < n; ++j)
            for (int i2 = 0; i2 < m; ++i2) {
                double acc = 0.0;
                for (int i = 0; i < nc; ++i) acc += cut[i2 * nc + i] * cxtmp[j * nc + i];
                s->gux[t * m * n + j * m + i2] += acc;
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
    const int n = s->n, m = s->m, T = s->T, N = T - 1;
    // P[H] .= gxx[H]; p[H] .= gx[H]     (:39-40)
    std::memcpy(&s->P[N * n * n], &s->gxx[N * n * n], sizeof(double) * n * n);
    std::memcpy(&s->p[N * n], &s->gx[N * n], sizeof(double) * n);
    for (int t = N - 1; t >= 0; --t) {        // (:42)
        const double* fx = &s->fx[t * n * n];  // n×n
        const double* fu = &s->fu[t * n * m];  // n×m
        const double* Pn = &s->P[(t + 1) * n * n];
        const double* pn = &s->p[(t + 1) * n];
        double* Qx = &s->Qx[t * n]; double* Qu = &s->Qu[t * m];
        double* Qxx = &s->Qxx[t * n * n]; double* Quu = &s->Quu[t * m * m]; double* Qux = &s->Qux[t * m * n];
        double* K = &s->K[t * m * n]; double* k = &s->k[t * m];
        double* P = &s->P[t * n * n]; double* p = &s->p[t * n];
        // Qx = fxᵀp' + gx   (:44-45)
        gemm(Qx, n, 1, fx, n, true, pn, n, n, false);
        for (int i = 0; i < n; ++i) Qx[i] += s->gx[t * n + i];
        // Qu = fuᵀp' + gu   (:48-49)
        gemm(Qu, m, 1, fu, n, true, pn, n, n, false);
        for (int i = 0; i < m; ++i) Qu[i] += s->gu[t * m + i];
        // Qxx = (fxᵀP')fx + gxx   (:52-54)
        gemm(s->xx_tmp.data(), n, n, fx, n, true, Pn, n, n, false);
        gemm(Qxx, n, n, s->xx_tmp.data(), n, false, fx, n, n, false);
        for (int i = 0; i < n * n; ++i) Qxx[i] += s->gxx[t * n * n + i];
        // Quu = (fuᵀP')fu + guu   (:57-59)
        gemm(s->ux_hat_tmp.data(), m, n, fu, n, true, Pn, n, n, false);
        gemm(Quu, m, m, s->ux_hat_tmp.data(), m, false, fu, n, n, false);
        for (int i = 0; i < m * m; ++i) Quu[i] += s->guu[t * m * m + i];
        // Qux = (fuᵀP')fx + gux   (:62-64)
        gemm(s->ux_hat_tmp.data(), m, n, fu, n, true, Pn, n, n, false);
        gemm(Qux, m, n, s->ux_hat_tmp.data(), m, false, fx, n, n, false);
        for (int i = 0; i < m * n; ++i) Qux[i] += s->gux[t * m * n + i];
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
    const int n = s->n, m = s->m, T = s->T, N = T - 1;
    for (int t = 0; t < N; ++t) {
        for (int i = 0; i < n; ++i) s->gradient[t * n + i] = s->Qx[t * n + i] - s->p[t * n + i];
        for (int i = 0; i < m; ++i) s->gradient[T * n + t * m + i] = s->Qu[t * m + i];
    }
    // gradient wrt x_T is left untouched
}

// src/data/methods.jl:42-54
static void trajectory_sensitivities(OrcSolver* s) {
    const int n = s->n, m = s->m, T = s->T, N = T - 1;
    std::fill(s->trajectory.begin(), s->trajectory.end(), 0.0);
    for (int t = 0; t < N; ++t) {
        double* zx = &s->trajectory[t * n];
        double* zu = &s->trajectory[T * n + t * m];
        double* zy = &s->trajectory[(t + 1) * n];
        for (int i = 0; i < m; ++i) zu[i] = s->k[t * m + i];
        gemm(zu, m, 1, &s->K[t * m * n], m, false, zx, n, n, true);
        gemm(zy, n, 1, &s->fu[t * n * m], n, false, zu, m, m, false);
        gemm(zy, n, 1, &s->fx[t * n * n], n, false, zx, n, n, true);
    }
}

// src/rollout.jl:1-31
extern "C" void orc_rollout_bang(OrcSolver* s, double step_size) {
    const int n = s->n, m = s->m, nw = s->nw, T = s->T, N = T - 1;
    double* x = s->states.data(); double* u = s->actions.data();
    const double* xb = s->nominal_states.data(); const double* ub = s->nominal_actions.data();
    for (int i = 0; i < n; ++i) x[i] = xb[i];            // (:19)
    for (int t = 0; t < N; ++t) {
        const double* K = &s->K[t * m * n];
        double* ut = u + t * m;
        for (int i = 0; i < m; ++i) ut[i] = s->k[t * m + i];           // (:24)
        for (int i = 0; i < m; ++i) ut[i] *= step_size;                // (:25)
        for (int i = 0; i < m; ++i) ut[i] += ub[t * m + i];            // (:26)
        for (int i = 0; i < m; ++i) {                                  // (:27) u += K x
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += K[j * m + i] * x[t * n + j];
            ut[i] += acc;
        }
        for (int i = 0; i < m; ++i) {                                  // (:28) u -= K x̄
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += K[j * m + i] * xb[t * n + j];
            ut[i] += -1.0 * acc;
        }
        const OrcDynamics* d = s->dynamics[t];
        d->evaluate(x + (t + 1) * n, x + t * n, ut, s->parameters.data() + t * nw, d->ctx);   // (:29)
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
    const int n = prob.nx, m = prob.nu, N = T - 1;
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
        std::vector<double> xbar(T * n);
        orc_rollout(T, prob.dynamics, x1 + (size_t)b * n, ubar + (size_t)b * N * m, wb, xbar.data());
        orc_initialize_controls(s, ubar + (size_t)b * N * m);
        orc_initialize_states(s, xbar.data());
        orc_solve(s);
        if (x_out) std::memcpy(x_out + (size_t)b * T * n, s->nominal_states.data(), sizeof(double) * T * n);
        if (u_out) std::memcpy(u_out + (size_t)b * N * m, s->nominal_actions.data(), sizeof(double) * N * m);
        if (K_out) std::memcpy(K_out + (size_t)b * N * m * n, s->K.data(), sizeof(double) * N * m * n);
        if (k_out) std::memcpy(k_out + (size_t)b * N * m, s->k.data(), sizeof(double) * N * m);
        if (stats_out) orc_get_stats(s, &stats_out[b]);
        orc_solver_destroy(s);
    }
    orc_problem_free(&prob);
    return fail ? -2 : 0;
}
