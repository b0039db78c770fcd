/*
 * ilqr_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A literal, single-trajectory, fp64 restatement of the hot path of
 * thowell/IterativeLQR.jl v0.2.3 (pure Julia; cannot be run in this image,
 * so this file is the checker that the HIP path is compared against).
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 *
 * PARITY PINNING: the reference ships no golden vectors and no Julia runtime
 * exists here, so whole-solve outputs (K, k, trajectories) are
 * "parity unpinned" at the BLAS/LAPACK boundary; what IS pinned are the
 * reference's own known-answer tests (test/objective.jl, test/dynamics.jl,
 * test/constraints.jl) and end-to-end property tests (test/car.jl,
 * test/acrobot.jl), all re-expressed in tests/test_oracle_*.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.
 *
 * As in the reference, every timestep has its own Dynamics / Cost / Constraint
 * object and its own dimensions (num_next_state may differ from num_state,
 * src/dynamics.jl:5-7): all buffers are ragged concatenations of per-timestep
 * blocks sized as src/data/{model,objective,policy,problem}.jl size them. For
 * uniform dimensions that is the plain [t][...] layout.
 */
#ifndef ILQR_ORACLE_H
#define ILQR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_NC 64

/* in-place callable contract of the reference: (out, x, u, w) -> nothing
 * (src/dynamics.jl:55-60, src/constraints.jl:54-64, src/costs.jl:1-15).
 * `ctx` carries what a Julia closure would capture (goal state, weights...). */
typedef void (*orc_fn)(double* out, const double* x, const double* u,
                       const double* w, const void* ctx);

/* src/dynamics.jl:1-12 */
typedef struct {
    orc_fn evaluate, jacobian_state, jacobian_action;
    int num_next_state, num_state, num_action, num_parameter;
    const void* ctx;
} OrcDynamics;

/* src/costs.jl:1-15 */
typedef struct {
    orc_fn evaluate, gradient_state, gradient_action;
    orc_fn hessian_state_state, hessian_action_action, hessian_action_state;
    int num_state, num_action; /* num_action == 0 for the terminal cost */
    const void* ctx;
} OrcCost;

/* src/constraints.jl:1-13 (indices_inequality is 0-based here) */
typedef struct {
    orc_fn evaluate, jacobian_state, jacobian_action;
    int num_constraint, num_state, num_action;
    int num_inequality;
    int indices_inequality[ORC_MAX_NC];
    const void* ctx;
} OrcConstraint;

/* src/options.jl:1-15 (line_search: 1 = :armijo, 0 = :none) */
typedef struct {
    int line_search;
    int max_iterations;
    int max_dual_updates;
    double min_step_size;
    double objective_tolerance;
    double lagrangian_gradient_tolerance;
    double constraint_tolerance;
    double constraint_norm; /* never read (src/options.jl:9) */
    double initial_constraint_penalty;
    double scaling_penalty;
    double max_penalty;
    int reset_cache;
    int verbose;
} OrcOptions;

void orc_default_options(OrcOptions* o);

typedef struct OrcSolver OrcSolver;

/* per-iteration trace record (what `verbose` prints, src/solve.jl:40-45) */
typedef struct {
    int outer, inner;
    double objective, gradient_norm, max_violation, step_size;
    int status;
} OrcTrace;

/* Solver(dynamics, costs[, constraints]) — src/solver.jl:11-46.
 * dynamics: T-1 objects, costs: T objects, constraints: T objects or NULL
 * (NULL = plain Objective path, src/solve.jl:137-139).
 * w: parameters, T*num_parameter doubles or NULL. */
OrcSolver* orc_solver_create(int T, const OrcDynamics* const* dynamics,
                             const OrcCost* const* costs,
                             const OrcConstraint* const* constraints,
                             const double* w, const OrcOptions* opts);
void orc_solver_destroy(OrcSolver* s);

/* src/solver.jl:56-66 */
void orc_initialize_controls(OrcSolver* s, const double* u /* (T-1)*m */);
void orc_initialize_states(OrcSolver* s, const double* x /* T*n */);
/* src/rollout.jl:33-42 */
void orc_rollout(int T, const OrcDynamics* const* dynamics, const double* x1,
                 const double* u, const double* w, double* x_out /* T*n */);

/* src/solve.jl:137-143 */
void orc_solve(OrcSolver* s);

/* stage-level entry points (all mode=:nominal unless stated) */
double orc_cost_bang(OrcSolver* s, int mode_current);  /* src/data/methods.jl:13-30 */
void orc_gradients(OrcSolver* s);                      /* src/gradients.jl:92-98 */
void orc_backward_pass(OrcSolver* s);                  /* src/backward_pass.jl:1-91 */
void orc_forward_pass(OrcSolver* s);                   /* src/forward_pass.jl:1-56 */
void orc_lagrangian_gradient(OrcSolver* s);            /* src/solve.jl:67-83 */
void orc_rollout_bang(OrcSolver* s, double step_size); /* src/rollout.jl:1-31 */
void orc_reset_model_objective(OrcSolver* s);          /* src/solve.jl:9-10 */
void orc_ilqr_solve(OrcSolver* s);                     /* src/solve.jl:1-54 */
void orc_augmented_lagrangian_update(OrcSolver* s);    /* src/augmented_lagrangian.jl:87-110 */

/* raw buffer access, name = reference field name:
 * "nominal_states","nominal_actions","states","actions","jacobian_state"(fx),
 * "jacobian_action"(fu),"gradient_state"(gx),"gradient_action"(gu),
 * "hessian_state_state","hessian_action_action","hessian_action_state",
 * "K","k","P","p","Qx","Qu","Qxx","Quu","Qux","gradient","trajectory",
 * "violations","constraint_dual","constraint_penalty","active_set"(as double).
 * Returns pointer into the solver (valid until destroy) and its length. */
double* orc_buffer(OrcSolver* s, const char* name, int* len);

typedef struct {
    double objective, gradient_norm, max_violation, step_size;
    int iterations, outer_iterations, status, potrf_info;
    int rollouts;
} OrcStats;
void orc_get_stats(const OrcSolver* s, OrcStats* st);

/* trace: set capacity before solve; records one row per inner iteration */
void orc_set_trace(OrcSolver* s, OrcTrace* buf, int capacity);
int orc_trace_len(const OrcSolver* s);
/* parity-test helpers: delta_grad_product of the last forward_pass! (src/forward_pass.jl:20); direct write of
 * the SolverData scalars; the LAPACK stand-ins used by the backward pass (dpotrf('U') / dpotrs('U'), column-major) */
double orc_last_delta(const OrcSolver* s);
void orc_set_active_set(OrcSolver* s, const double* a /* C doubles, 0 / 1 */);
void orc_set_scalars(OrcSolver* s, double objective, double max_violation, double step_size, int status);
int orc_potrf_U(double* A, int m);
void orc_potrs_U(const double* U, int m, double* B, int nrhs);

/* ---- built-in model zoo (oracle/models.cpp): hand-written functions with
 * forward-mode dual-number Jacobians, following the reference's examples. */
typedef struct {
    int T, nx, nu, nw;
    const OrcDynamics* const* dynamics;
    const OrcCost* const* costs;
    const OrcConstraint* const* constraints; /* NULL when unconstrained */
    void* owner;                             /* internal */
} OrcProblem;

/* name: "particle","acrobot","car","car_goal","car_obs"(nw=2),"car_tv" (distinct per-step objects),
 * "ragged" (per-step DIMENSIONS: nx = 3,3,4,4,2,2,3,3 | ..., nu = 2,1,2,1,1,2,2,1 | ...; OrcProblem.nx / .nu are the largest),
 * "synth32","synth12","pendulum_euler","pendulum_pole","kat_objective","kat_constraints","acrobot_unconstrained".
 * Returns 0 on success. The arrays live until orc_problem_free. */
int orc_problem_builtin(const char* name, int T, OrcProblem* out);
void orc_problem_free(OrcProblem* p);
/* per-timestep dimensions: nx_t[T] (num_state of step t; num_next_state of the last dynamics for t = T-1), nu_t[T-1] */
void orc_problem_dims(const OrcProblem* p, int* nx_t, int* nu_t);

/* batch driver = CPU baseline: one fresh Solver per instance, instances
 * spread over `nthreads` OpenMP threads. x1: B*n, ubar: B*(T-1)*m (states by
 * open-loop rollout). Outputs may be NULL. Returns 0 on success.
 * Problems whose dimensions vary along the horizon: the host arrays are padded to
 * n = OrcProblem.nx, m = OrcProblem.nu (the largest), padding read as / written with zeros. */
int orc_solve_batch(const char* model, int T, int B, const double* x1,
                    const double* ubar, const OrcOptions* opts, int nthreads,
                    double* x_out, double* u_out, double* K_out, double* k_out,
                    OrcStats* stats_out);

/* same with per-instance parameters w: [B][T][nw] (Solver(...; parameters = θ), src/solver.jl:12,29) */
int orc_solve_batch_w(const char* model, int T, int B, const double* x1,
                      const double* ubar, const double* w, const OrcOptions* opts, int nthreads,
                      double* x_out, double* u_out, double* K_out, double* k_out,
                      OrcStats* stats_out);

#ifdef __cplusplus
}
#endif
#endif
