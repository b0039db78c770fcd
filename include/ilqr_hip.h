/*
 * ilqr_hip.h — C-ABI of the MI355X-native batched iLQR / augmented-Lagrangian solver.
 *
 * The reference (thowell/IterativeLQR.jl v0.2.3, pure Julia) has no FFI layer;
 * this header is the boundary a Julia `ccall` wrapper (or the Python ctypes
 * mirror in iterativelqr.jl_amd/) binds to. Each entry point names the reference
 * interface it replaces (paths relative to /root/reference). One handle owns a
 * BATCH of independent problem instances of one model, on one GPU (ilqr_create)
 * or split over several (ilqr_create_sharded); every instance behaves exactly
 * like one reference `Solver`.
 *
 * Conventions: every function returns 0 on success, <0 on error (never
 * throws); ilqr_last_error() gives the message. The caller owns all host
 * pointers; the handle owns device memory. Host arrays are instance-major:
 * x[b][t][i], u[b][t][j]; matrices are column-major like Julia
 * (K[b][t][col][row]). All arithmetic is fp64 (the reference is Float64 only).
 * A handle is not thread-safe. Calls that return data synchronise the stream.
 */
#ifndef ILQR_HIP_H
#define ILQR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILQR_OK 0
#define ILQR_ERR_INVALID (-1)
#define ILQR_ERR_HIP (-2)
#define ILQR_ERR_MODEL (-3)
#define ILQR_ERR_NO_DEVICE (-4)
#define ILQR_ERR_LDS (-5)

typedef struct ilqr_handle ilqr_handle;

/* Options{T} — src/options.jl:1-15, field for field (constraint_norm is
 * carried but, as in the reference, never read). line_search: 1 = :armijo,
 * 0 = :none. verbose is ignored by the device; the host mirror prints the reference's per-iteration report from the
 * trace (ilqr_enable_trace) when it is set. */
typedef struct {
    int32_t line_search;
    int32_t max_iterations;
    int32_t max_dual_updates;
    double min_step_size;
    double objective_tolerance;
    double lagrangian_gradient_tolerance;
    double constraint_tolerance;
    double constraint_norm;
    double initial_constraint_penalty;
    double scaling_penalty;
    double max_penalty;
    int32_t reset_cache;
    int32_t verbose;
} ilqr_options;

/* What `Solver(dynamics, costs[, constraints])` needs — src/solver.jl:11-46. */
typedef struct {
    const char* model;          /* built-in model name ("acrobot", "car", ...) or the
                                   name registered by a generated model library */
    const char* model_library;  /* optional path of a generated model module (.so)
                                   to dlopen first (iterativelqr.jl_amd/codegen.py) */
    int32_t horizon;            /* T (number of states) */
    int32_t batch;              /* B independent instances */
    int32_t device;             /* HIP device ordinal */
    int32_t constrained;        /* 1: Solver(dynamics, costs, constraints) → AL solve
                                   0: Solver(dynamics, costs) → plain iLQR */
} ilqr_problem_desc;

/* SolverData scalars — src/data/solver.jl:4-18 — plus bookkeeping. */
typedef struct {
    double objective;
    double gradient_norm;
    double max_violation;
    double step_size;
    int32_t iterations;        /* data.iterations (inner iterations, all outer loops) */
    int32_t outer_iterations;  /* AL iterations executed */
    int32_t status;            /* data.status of the last forward pass */
    int32_t potrf_info;        /* first non-zero LAPACK-style info of the Quu Cholesky
                                  (the reference ignores it, src/backward_pass.jl:69) */
    int32_t rollouts;          /* closed-loop rollouts executed */
    int32_t reserved;
} ilqr_stats;

const char* ilqr_last_error(void);
int ilqr_device_count(void);

/* Options() defaults — src/options.jl:1-15 */
int ilqr_default_options(ilqr_options* opt);

/* Solver(...) — src/solver.jl:11-46. Fails loudly (ILQR_ERR_NO_DEVICE) when no
 * HIP device is present: there is no CPU fallback. */
int ilqr_create(const ilqr_problem_desc* desc, ilqr_handle** out);
/* The same Solver with its batch spread over several GPUs of the node — the reference's caller holds ONE Solver and calls
 * solve! (src/solver.jl:28-46, src/solve.jl:137-143); a Julia host has no process launcher. The instances are split into
 * contiguous ranges of ceil(batch / n_devices), range i on devices[i] (desc->device is ignored; an ordinal may repeat). Inside
 * the library every range is a handle of its own with its own workspace and stream on its device; launches are asynchronous
 * on all of them, blocking copies run on one host thread per device. Every entry point below takes the returned handle
 * unchanged: host arrays stay instance-major over the WHOLE batch and are scattered / gathered over the ranges. No data
 * crosses devices (the instances are independent). Not available on it: ilqr_initialize_rollout_device and ilqr_get_stream
 * (one device's pointers / stream); ilqr_timing_get reports the slowest device. */
int ilqr_create_sharded(const ilqr_problem_desc* desc, const int32_t* devices, int32_t n_devices, ilqr_handle** out);
int ilqr_destroy(ilqr_handle* h);
int ilqr_set_options(ilqr_handle* h, const ilqr_options* opt);   /* solver.options */
int ilqr_get_dims(const ilqr_handle* h, int32_t* nx, int32_t* nu, int32_t* nw,
                  int32_t* nc_stage, int32_t* nc_term, int32_t* horizon, int32_t* batch);

/* Solver(...; parameters = θ) — src/solver.jl:12,29, src/data/problem.jl:25-30: per-instance,
 * per-timestep parameter vectors w: [B][T][nw] (the terminal entry is used by the terminal cost /
 * constraint). Only for models with num_parameter > 0; zero until set; ilqr_reset keeps them. */
int ilqr_set_parameters(ilqr_handle* h, const double* w);

/* Fresh-solver state (all buffers zero, objective = Inf) — what constructing a
 * new reference Solver gives (src/data/problem.jl:32-38, src/data/solver.jl:37). */
int ilqr_reset(ilqr_handle* h);

/* initialize_controls!(solver, ū) / initialize_states!(solver, x̄) — src/solver.jl:56-66.
 * Host pointers. */
int ilqr_initialize_controls(ilqr_handle* h, const double* u);
int ilqr_initialize_states(ilqr_handle* h, const double* x);
/* x̄ = rollout(dynamics, x1, ū) (src/rollout.jl:33-42) followed by both
 * initialisers, executed on the device. x1: [B][nx], u: [B][T-1][nu].
 * The *_device variant takes DEVICE pointers (inputs already resident in HBM)
 * and is asynchronous on the handle's stream. */
int ilqr_initialize_rollout(ilqr_handle* h, const double* x1, const double* u);
int ilqr_initialize_rollout_device(ilqr_handle* h, const double* d_x1, const double* d_u);
/* The same from the inputs the handle keeps in HBM since the last ilqr_initialize_rollout (every device of a sharded handle
 * keeps its own range): a re-solve of the same (x1, ū) — solve!(solver) again after initialize_controls!/initialize_states!,
 * src/solver.jl:56-66 — without a host-to-device copy. Asynchronous. */
int ilqr_initialize_rollout_resident(ilqr_handle* h);

/* solve!(solver) — src/solve.jl:137-143. Asynchronous: enqueues the whole
 * AL/iLQR solve of every instance on the handle's stream. */
int ilqr_solve(ilqr_handle* h);
int ilqr_synchronize(ilqr_handle* h);

/* Stage-level entry points for parity tests (one kernel each, mode = :nominal):
 * the reference functions they run are listed per id. */
enum {
    ILQR_STAGE_COST_NOMINAL = 0,   /* cost!(data, problem, mode=:nominal) — src/data/methods.jl:13-30 */
    ILQR_STAGE_GRADIENTS = 1,      /* gradients!(problem)                 — src/gradients.jl:92-98   */
    ILQR_STAGE_BACKWARD_PASS = 2,  /* backward_pass! + lagrangian_gradient! — src/backward_pass.jl, src/solve.jl:67-83 */
    ILQR_STAGE_FORWARD_PASS = 3,   /* forward_pass!                       — src/forward_pass.jl:1-56 */
    ILQR_STAGE_RESET_MODEL_OBJECTIVE = 4, /* reset!(model); reset!(objective) — src/solve.jl:9-10 */
    ILQR_STAGE_ILQR_SOLVE = 5,     /* ilqr_solve!                         — src/solve.jl:1-54       */
    ILQR_STAGE_AL_UPDATE = 6,      /* augmented_lagrangian_update!        — src/augmented_lagrangian.jl:87-110 */
    /* host-stepped outer loop, for solve!(solver; augmented_lagrangian_callback!) — src/solve.jl:88,125:
     * AL_BEGIN once (:93-103), then AL_OUTER per outer iteration (:105-122, skipping instances that already
     * met the constraint tolerance); the caller runs its callback between AL_OUTER launches. */
    ILQR_STAGE_AL_BEGIN = 7,
    ILQR_STAGE_AL_OUTER = 8,
    /* SHARED STEP SIZE over a whole (multi-GPU) batch — not in the reference (every Solver there line-searches alone); the
     * optional mode of the north star: all instances take the SAME Armijo step, decided on the summed merit. The line search
     * of forward_pass! (src/forward_pass.jl:26-52) is stepped from the host, everything else stays per instance:
     *   SS_INNER_BEGIN            src/solve.jl:9-21 for every instance whose outer loop is still running
     *   SS_TRIAL(alpha, first)    rollout!(alpha) + cost!(mode = :current) (src/forward_pass.jl:34-36); with first != 0 also
     *                             J_prev and delta_grad_product (:13-20). The caller sums objective, "j_prev" and
     *                             "delta_grad_product" over the instances with "inner_done" == 0 (and over ranks: ONE all-reduce
     *                             of three doubles) and tests  sum J <= sum J_prev + 1e-4 alpha sum delta  (:44)
     *   SS_FINISH(alpha, accept)  update_nominal_trajectory! or line-search failure, then src/solve.jl:27-51 per instance
     *   SS_OUTER                  src/solve.jl:113-122 per instance
     * (ilqr_run_stage_param; the whole loop: ilqr_solve_shared_step below). With a batch of one it reproduces solve! exactly. */
    ILQR_STAGE_SS_INNER_BEGIN = 9,
    ILQR_STAGE_SS_TRIAL = 10,
    ILQR_STAGE_SS_FINISH = 11,
    ILQR_STAGE_SS_OUTER = 12
};
int ilqr_run_stage(ilqr_handle* h, int32_t stage);
int ilqr_run_stage_param(ilqr_handle* h, int32_t stage, double param, int32_t flag);

/* solve! with ONE step size per inner iteration for the whole batch — over every rank of a multi-GPU job when `reduce` sums across
 * them. Not a reference behaviour (every reference Solver line-searches alone; a shared step changes every iterate): the optional
 * mode of the north star, "at most an all-reduce of the line-search merit for a shared step size". The Armijo loop of
 * forward_pass! (src/forward_pass.jl:26-52) runs here, on the host, over the SUMMED merit of the instances still in their inner
 * loop: per trial one SS_TRIAL launch, one reduction of three doubles {sum J(alpha), sum J_prev, sum grad_L' dz}, then SS_FINISH;
 * linearisation, Riccati pass, convergence tests and dual updates stay per instance on the device (stages SS_* above).
 *   reduce(values, n, ctx): replace values[0..n) by their sum over all ranks, return 0 (an RCCL / MPI all-reduce in a host with
 *     several processes; NULL = one process: the identity). It is called the same number of times on every rank.
 *   steps / steps_cap / n_steps (optional): the accepted step size of every inner iteration (0 = line search failed), in order.
 *     *n_steps is the number of inner iterations the solve took; min(*n_steps, steps_cap) entries of steps are written (a caller
 *     that sized steps for fewer must not read beyond steps_cap).
 * With a batch of ONE instance the mode reproduces ilqr_solve exactly. Constrained solvers only. Synchronous. */
typedef int (*ilqr_allreduce_sum_fn)(double* values, int32_t n, void* ctx);
int ilqr_solve_shared_step(ilqr_handle* h, ilqr_allreduce_sum_fn reduce, void* ctx, double* steps, int32_t steps_cap, int32_t* n_steps);

/* get_trajectory(solver) — src/solver.jl:48-50: nominal states [B][T][nx] and
 * actions [B][T-1][nu]. */
int ilqr_get_trajectory(ilqr_handle* h, double* x, double* u);
/* solver.policy.K / .k — src/data/policy.jl:25-26. K: [B][T-1][nx][nu] (column-major nu×nx). */
int ilqr_get_policy(ilqr_handle* h, double* K, double* k);
/* solver.data.* — one record per instance. */
int ilqr_get_stats(ilqr_handle* h, ilqr_stats* stats);

/* Raw workspace access by reference field name (parity tests):
 * "nominal_states","nominal_actions","states","actions","jacobian_state",
 * "jacobian_action","gradient_state","gradient_action","hessian_state_state",
 * "hessian_action_action","hessian_action_state","K","k","P","p",
 * "gradient_state_lagrangian"(Qx−p),"gradient_action_lagrangian"(Qu),
 * "violations","constraint_dual","constraint_penalty","active_set","parameters".
 * Layout: [B][per-instance length]; ilqr_buffer_len gives the per-instance length.
 * Large models (nx > 4 or nu > 4): the kernels work on compact rows (state-dependent Jacobian entries, structurally non-zero
 * Hessian entries); the full jacobian_* / hessian_* arrays are a mirror. A value written with ilqr_set_buffer at a constant
 * Jacobian position or outside the Hessian pattern cannot be represented there and is dropped: ilqr_get_buffer afterwards
 * returns what the kernels use (the constant, a zero), not what was written. */
int ilqr_buffer_len(const ilqr_handle* h, const char* name, size_t* len);
int ilqr_get_buffer(ilqr_handle* h, const char* name, double* out);
int ilqr_set_buffer(ilqr_handle* h, const char* name, const double* in);
/* solver.policy.action_value.* — src/data/policy.jl:58-64: "Qx","Qu","Qxx","Quu","Qux" (read-only through
 * ilqr_get_buffer). The fused solve keeps them in registers; after this call the backward-pass STAGE
 * (ILQR_STAGE_BACKWARD_PASS, ILQR_STAGE_ILQR_SOLVE) also stores them to HBM. */
int ilqr_enable_action_value_buffers(ilqr_handle* h);
/* Index of a named SolverData scalar inside the "_scalars" buffer: "objective","max_violation","step_size","status",
 * "iterations","gradient_norm","outer_iterations","potrf_info","rollouts","done" (host-stepped AL loop: instance met
 * the constraint tolerance), "delta_grad_product" (∇Lᵀ·Δz of the last forward_pass!, src/forward_pass.jl:20),
 * "trace_len" (rows the last solve wrote to the trace), "count" (length of "_scalars"); shared-step mode: "obj_prev",
 * "inner_done", "j_prev", "inner_it"; "resume" (hand-over bookkeeping, 0 after a solve);
 * "literal_backward_passes" (two-wave latency kernel, models with one action: backward passes of the last solve that met a
 * non-positive pivot and were repeated in LAPACK's literal arithmetic — 0 on healthy instances); "t_start", "t_end" (when the
 * instance's workgroup started / finished its solve in the latency or large-model kernel: ticks of the device's 100 MHz
 * real-time counter) and "hw_id_wave0" … "hw_id_wave3" (where its waves ran: HW_REG_HW_ID + 2^32 * XCC_ID) — what
 * tools/finish_times.py reads. -1 if unknown. */
int ilqr_scalar_slot(const char* name);

/* Kernel variant of ilqr_solve. Large models (nx > 4 or nu > 4): 0 = auto, 1 = four waves per instance (two instances per CU),
 * 4 = ONE wave per instance (six per CU: 25 KB of LDS each) for models whose matrices are single 16x16 tiles (nx, nu <= 16) — the same phase
 * functions with the four wave roles of a phase run in turn, bitwise the four-wave results; auto takes it once the batch exceeds
 * 8 x CUs (an instance alone is faster on four waves; residency wins beyond that). Small models (nx, nu <= 4): 0 = auto — the latency kernel
 * (two waves per instance, all iteration state in LDS) while the batch fits the chip at one instance per SIMD (batch <= 4 x CUs),
 * the packed kernel beyond that and for horizons whose LDS-resident set exceeds the 160 KiB of a CU; 1 = latency; 2 = throughput
 * (one wave per instance, Jacobians in HBM / L2; superseded by the packed kernel, kept for A/B runs); 3 = packed — FOUR
 * instances per wave on the four blocks of v_mfma_f64_4x4x4, workspace streamed from HBM / L2 through a 13 KB LDS chunk buffer
 * per wave, no horizon limit — with a SECOND wave per pack as linearisation server (chunk ch - 1 linearised into a second LDS buffer
 * while the Riccati steps of chunk ch run; bitwise the one-wave results) wherever the buffers fit the CU's LDS at the batch's
 * residency (up to 4 packs per CU, i.e. batch <= 16 x CUs); 5 = packed, one wave per pack always; 6 = packed, two waves where they
 * fit (= 3; kept distinct for A/B runs). All run the same arithmetic up to the association of a few sums. */
int ilqr_set_kernel_variant(ilqr_handle* h, int32_t variant);
/* The kernel ilqr_solve launches for this handle as it stands (never 0): 1 latency / four waves per instance, 2 throughput,
 * 4 one wave per instance (large models), 5 packed with one wave per pack, 6 packed with two. */
int ilqr_resolved_kernel_variant(ilqr_handle* h, int32_t* variant);
/* Straggler hand-over of the packed kernel (no counterpart in the reference, which is one trajectory per Solver): a batched
 * launch lasts as long as its slowest instance, and in the packed kernel a straggler keeps a whole wave at 120-240 us per
 * cycle (one line-search trial per cycle). Instances that leave the packed kernel do so with their state complete in the
 * workspace block — at the start of an outer iteration of constrained_ilqr_solve! (src/solve.jl:105) or at the head of an
 * inner iteration of ilqr_solve! (src/solve.jl:22), where the block holds the nominal trajectory, the linearisation, the
 * accumulated Hessians, K, k and the loop's scalars — and a second launch on the same stream finishes exactly those with the
 * latency kernel (two waves per instance, state in LDS: 50-75 us per iteration, a rejected trial one rollout instead of a
 * cycle). The two kernels do the same arithmetic (trajectories, policies and duals bitwise: tests/test_gpu_parity.py), so
 * which instances change kernels, and when, never shows in a result.
 *   ilqr_set_handover(outer): -1 (default) = by head count, below; 0 = no hand-over; k >= 2 = an instance still unconverged
 *     when it ENTERS outer iteration k leaves at that boundary.
 *   ilqr_set_handover_live(live), used when outer = -1: once no more than `live` instances of the batch are still running,
 *     every survivor leaves at its next resumable point. -1 auto = min(1024, batch / 4), what the latency kernel holds at full
 *     speed; 0 off; n >= 1 as given.
 * One wave per pack (the packed kernel's form above four packs per CU, two packs per workgroup): the workgroups finish the
 * handed-over instances THEMSELVES — once both its packs are through, a workgroup takes instances from a device-wide queue
 * and runs the latency kernel's code on each, so the second launch finds nothing left (it stays as the net) — and a straggler
 * does not wait for the head count: under that rule
 *   ilqr_set_handover_mark(rejected): an instance whose rejected line-search trials (src/forward_pass.jl:51; those of forward
 *     passes that ended in an acceptance) exceed the mean over the batch so far by `rejected`, while more than half the batch is
 *     still running, is marked at the head of its next
 *     inner iteration; its workgroup's two packs leave at their next resumable points and the workgroup finishes the marked
 *     instance at once (BASELINE config 4: instance 2300 of shard 2 — 682 iterations, 1354 rollouts, the others 347 — is
 *     marked in its second iteration). -1 auto = 6; 0 never; n >= 1 as given. Marks are taken only while the launch is ONE round of
 *     workgroups (all resident: up to 8192 instances per GPU on an MI355X); beyond that the head-count rule alone applies.
 * The two-wave solvers (latency kernel, resume launch, the workgroups above) take the trials of a line search
 * (src/forward_pass.jl:28-52) in rounds of up to four, rolled out at once by the four 16-lane rows of one wave; trial by trial the
 * search is the reference's, and so are its results. */
int ilqr_set_handover(ilqr_handle* h, int32_t outer);
int ilqr_set_handover_live(ilqr_handle* h, int32_t live);
int ilqr_set_handover_mark(ilqr_handle* h, int32_t rejected);
/* What the last ilqr_solve did there: instances that went through the queue, instances marked as stragglers. */
int ilqr_get_handover_stats(ilqr_handle* h, int32_t* queued, int32_t* marked);

/* Per-iteration record of what the reference prints when `verbose` (src/solve.jl:40-45): for every
 * instance up to `capacity` rows of 8 doubles {outer, inner, objective, gradient_norm, max_violation,
 * step_size, status, rollouts-so-far} written by ilqr_solve. capacity 0 disables (default). */
int ilqr_enable_trace(ilqr_handle* h, int32_t capacity);
int ilqr_get_trace(ilqr_handle* h, double* out /* [B][capacity][8] */);

/* Device-side handles for callers that time or chain work themselves. */
int ilqr_get_stream(ilqr_handle* h, void** hip_stream);
/* Average device time (ms) of the solve kernel over the ilqr_solve calls since
 * the last ilqr_timing_reset, measured with HIP events on the handle's stream. */
int ilqr_timing_reset(ilqr_handle* h);
int ilqr_timing_get(ilqr_handle* h, double* solve_kernel_ms_avg, int32_t* launches);

/* Dynamics(f, fx, fu, ...) / Cost(...) / Constraint(f, fx, fu, ...) with USER-SUPPLIED callables — src/dynamics.jl:55-60,
 * src/costs.jl:1-15, src/constraints.jl:54-64 — for hosts without Python (the Julia wrapper hands over what
 * Symbolics.build_function(..., target = CTarget()) emits; a C program writes it by hand). `source` is C code defining,
 * each as `ILQR_MODEL_FN void NAME(double* out, const double* x, const double* u, const double* w)` with the reference's
 * in-place contract (out column-major, pre-zeroed by the caller; u is a dummy for the terminal functions; math.h names
 * and ilqr::sincos_fast are available):
 *     dynamics, dynamics_jacobian_state, dynamics_jacobian_action,
 *     cost_stage, cost_stage_gradient_state, cost_stage_gradient_action, cost_stage_hessian_state_state,
 *     cost_stage_hessian_action_action, cost_stage_hessian_action_state,
 *     cost_terminal, cost_terminal_gradient_state, cost_terminal_hessian_state_state,
 *     constraint_stage, constraint_stage_jacobian_state, constraint_stage_jacobian_action          (only if nc_stage > 0)
 *     constraint_terminal, constraint_terminal_jacobian_state                                       (only if nc_term > 0)
 * ineq_stage / ineq_term: bit i set = constraint row i is an inequality (indices_inequality, 0-based).
 * The library wraps the source (csrc/ilqr_model_adapter.hpp), compiles it for gfx950 with hipcc as a child process (cached
 * by a hash of the source; ILQR_HIPCC / ILQR_CSRC_DIR override the tool and the kernel headers), loads the module and
 * returns the name to put into ilqr_problem_desc.model and the module path for ilqr_problem_desc.model_library.
 * nx <= 64, nu <= 16, at most 64 constraint rows per stage here (ilqr_compile_model_rows: 256). Models with nx > 4 or nu > 4 run on the large path, which streams only the state-dependent Jacobian
 * entries and the structurally non-zero Hessian entries per timestep: for opaque callables these are found by PROBING — the
 * source is compiled a second time with the host C++ compiler (ILQR_HOSTCXX, else g++ / c++ / clang++) and the Jacobian,
 * Hessian and constraint-Jacobian callables are evaluated at 48 points spread over magnitudes 1e-3 ... 1e3 and both signs
 * (x, u and w alike; selector parameters of a lowered model cycle through every kind); an entry bitwise equal and finite at
 * all of them is a constant (the role Symbolics' sparse expressions play in the reference, src/dynamics.jl:16-34); an entry
 * that is NaN / Inf anywhere counts as state-dependent / non-zero.
 * ASSUMPTION, and its limit: a derivative entry that is constant on every probed point is taken to be constant EVERYWHERE.
 * That holds for smooth expressions; a PIECEWISE callable — a clamp or saturation (derivative 1 inside a box, 0 outside), fabs,
 * max(0, .) penalties, contact switches, terms active only in a region — can look constant on all 48 points and is then baked
 * in wrongly for instances that leave the probed regime. Models with such callables must set flags = ILQR_MODEL_DENSE_TABLES
 * (per model; ILQR_NO_STRUCTURE_PROBE in the environment does the same for the whole process). Source that does not compile
 * for the host also gets the dense tables (correct at every size, but slow: all nx (nx + nu) Jacobian entries are evaluated,
 * stored and patched per timestep). ilqr_model_compact_sizes reports what a module was built with.
 * The dynamics callable itself stays opaque: every lane of the rollout evaluates the whole vector function. */
#define ILQR_MODEL_DENSE_TABLES 1   /* flags: no structure probe for THIS model (every Jacobian entry state-dependent, every Hessian entry non-zero) */
typedef struct {
    const char* name;        /* C identifier */
    int32_t nx, nu, nw;      /* num_state, num_action, num_parameter */
    int32_t nc_stage, nc_term;
    uint64_t ineq_stage, ineq_term;
    const char* source;
    int32_t flags;           /* 0 or ILQR_MODEL_DENSE_TABLES */
} ilqr_model_source;
int ilqr_compile_model(const ilqr_model_source* src, char* registered_name, size_t name_len, char* library_path, size_t path_len);
/* The same for constraints with more than 64 rows per stage (the reference has no limit: Constraint(f, fx, fu, nc, ...;
 * indices_inequality), src/constraints.jl:54-64): the inequality rows as arrays of 64-bit words, row i = bit i % 64 of word i / 64,
 * ceil(nc_stage / 64) and ceil(nc_term / 64) words (NULL = the struct's mask, for a kind with at most 64 rows). Up to
 * ILQR_MAX_CONSTRAINT_ROWS rows per stage; beyond 64 the per-timestep constraint values no longer fit a thread's registers
 * (correct, slower). */
#define ILQR_MAX_CONSTRAINT_ROWS 256
int ilqr_compile_model_rows(const ilqr_model_source* src, const uint64_t* ineq_stage_words, const uint64_t* ineq_term_words,
                            char* registered_name, size_t name_len, char* library_path, size_t path_len);

/* ---- Vectors of DISTINCT per-step objects and time-varying DIMENSIONS — README.md:26 of the reference ("costs, constraints and
 * dynamics can differ at every timestep"): Solver(dynamics::Vector{Dynamics}, costs::Vector{Cost}, constraints) keeps one object
 * per timestep (src/solver.jl:28-46) and sizes every buffer per timestep (num_next_state may differ from num_state,
 * src/dynamics.jl:5-7, src/data/model.jl:11-16, src/data/policy.jl:44-78). The device kernels are compiled for ONE stage template
 * (one Dynamics / stage Cost / stage Constraint of fixed dimensions, one terminal Cost / Constraint), so the host side LOWERS the
 * per-step objects onto it, exactly:
 *   - the distinct objects of a category ("kinds", in order of first appearance) become one combined callable that takes one
 *     extra per-timestep parameter per kind, a one-hot SELECTOR s_k(t) in {0, 1} riding in theta_t behind the user's own
 *     parameters; the combined callable runs the kind whose selector is set (a branch, not a product: a kind that is switched
 *     off may be outside its domain at that step); a category with a single kind gets no selectors;
 *   - constraint kinds are stacked: kind k owns rows [constraint_row0[k], + nc_k) of the combined stage constraint, the rows of
 *     the kinds that are off read c = 0 with zero Jacobian — their multipliers stay 0 and they add nothing to the AL cost, its
 *     gradient, the Gauss-Newton Hessian or max_violation (src/augmented_lagrangian.jl:39-110, src/gradients.jl:23-81,
 *     src/data/constraints.jl:23-46);
 *   - dimensions are zero-padded to the largest num_state / num_action of the horizon: padded next-state rows are 0, padded
 *     actions get the stage cost u^2 / 2 (Quu stays positive definite and block-diagonal with the real block, so its Cholesky
 *     and solves leave the real block untouched and return K = 0, k = 0 for the padding), padded states enter no function.
 *     Host arrays (x1, u, trajectories, gains) are the padded ones; state_dims[t] / action_dims[t] entries of step t are real.
 * ilqr_plan_stages computes the plan from the kinds alone (no device, no compiler): template dimensions, selector columns, the
 * selector table, row offsets and inequality masks. ilqr_compile_model_stages composes the combined callables from C source
 * of the kinds' callables and compiles them like ilqr_compile_model. ilqr_set_stage_selectors hands the table to a handle.
 * (The Python host traces symbolic objects: its lowering.py asks ilqr_plan_stages for the plan and gates the expressions
 * accordingly; a Julia or C host goes through ilqr_compile_model_stages.) */
#define ILQR_MAX_STAGE_KINDS 16
typedef struct {
    int32_t horizon;                    /* T: T-1 dynamics, T costs, T constraints (src/data/problem.jl:28-30) */
    int32_t num_parameter;              /* the user's parameters per timestep (largest num_parameter of all objects) */
    /* Dynamics kinds — src/dynamics.jl:1-12: kind k maps (dynamics_nx[k], dynamics_nu[k]) -> dynamics_nx_next[k] */
    int32_t n_dynamics;
    const int32_t* dynamics_nx; const int32_t* dynamics_nu; const int32_t* dynamics_nx_next;
    const int32_t* dynamics_of_step;    /* [T-1]: kind acting at step t */
    /* stage Cost kinds — src/costs.jl:1-15 */
    int32_t n_costs;
    const int32_t* cost_nx; const int32_t* cost_nu;
    const int32_t* cost_of_step;        /* [T-1] */
    /* stage Constraint kinds — src/constraints.jl:1-13; a step without rows names a kind with constraint_nc = 0
     * (Constraint(), src/constraints.jl:45-52). n_constraints = 0: no stage constraints at all (constraint_of_step unused). */
    int32_t n_constraints;
    const int32_t* constraint_nc; const int32_t* constraint_nx; const int32_t* constraint_nu;
    const uint64_t* constraint_ineq;    /* [n_constraints][4]: four 64-bit words per kind, row i of the kind an inequality = bit i % 64 of its word i / 64 */
    const int32_t* constraint_of_step;  /* [T-1] */
    /* terminal Cost / Constraint: num_state = nx_term (must be the last dynamics' num_next_state) */
    int32_t nx_term, nc_term;
    uint64_t ineq_term[4];              /* inequality rows of the terminal constraint, as words */
} ilqr_stage_kinds;
typedef struct {
    int32_t nx, nu;                     /* template dimensions: largest num_state / num_action of the horizon */
    int32_t nw;                         /* template parameters per timestep = num_parameter + n_selectors */
    int32_t nc_stage, nc_term;          /* rows of the combined stage constraint (all kinds stacked), terminal rows */
    int32_t n_selectors;
    int32_t sel_dynamics, sel_cost, sel_constraint;   /* first parameter column of the category's one-hot block; -1: one kind, no selectors */
    int32_t constraint_row0[ILQR_MAX_STAGE_KINDS];    /* first row of kind k in the combined stage constraint */
    uint64_t ineq_stage_words[4];       /* inequality rows of the combined stage constraint (row i = bit i % 64 of word i / 64) */
} ilqr_stage_plan;
/* selectors: [T][plan->n_selectors] written densely (row T-1, the terminal step, is all zero); selectors_len = doubles available
 * (T * (n_dynamics + n_costs + n_constraints) always suffices). state_dims: [T], action_dims: [T-1]. Any output pointer but
 * `plan` may be NULL. Fails with ILQR_ERR_INVALID when the chain of dimensions is inconsistent (dynamics[t].num_next_state !=
 * dynamics[t+1].num_state, a cost or constraint whose dimensions are not its step's), as the reference would throw at
 * x[t+1] .= dynamics!(...) (src/rollout.jl:29). */
int ilqr_plan_stages(const ilqr_stage_kinds* kinds, ilqr_stage_plan* plan, double* selectors, size_t selectors_len,
                     int32_t* state_dims, int32_t* action_dims);
/* `source`: C code defining, with the contract of ilqr_compile_model (ILQR_MODEL_FN void NAME(double* out, const double* x,
 * const double* u, const double* w), out column-major IN THE KIND'S OWN DIMENSIONS and pre-zeroed; w = the user's parameters),
 *     for every Dynamics kind k:    dynamics_<k>, dynamics_<k>_jacobian_state, dynamics_<k>_jacobian_action
 *     for every stage Cost kind k:  cost_stage_<k>, cost_stage_<k>_gradient_state, cost_stage_<k>_gradient_action,
 *                                   cost_stage_<k>_hessian_state_state, cost_stage_<k>_hessian_action_action, cost_stage_<k>_hessian_action_state
 *     for every Constraint kind k with rows: constraint_stage_<k>, constraint_stage_<k>_jacobian_state, constraint_stage_<k>_jacobian_action
 *     cost_terminal, cost_terminal_gradient_state, cost_terminal_hessian_state_state,
 *     constraint_terminal, constraint_terminal_jacobian_state                                         (only if nc_term > 0)
 * The library writes the combined, padded callables of the template around them (selector branches, re-striding of the
 * matrices into the template's leading dimensions, u^2 / 2 on padded actions) and compiles the result as ilqr_compile_model
 * does. Outputs as ilqr_plan_stages plus the registered name and module path for ilqr_problem_desc. After ilqr_create:
 * ilqr_set_stage_selectors(h, selectors, plan.n_selectors). */
int ilqr_compile_model_stages(const char* name, const ilqr_stage_kinds* kinds, const char* source, ilqr_stage_plan* plan,
                              double* selectors, size_t selectors_len, int32_t* state_dims, int32_t* action_dims,
                              char* registered_name, size_t name_len, char* library_path, size_t path_len);
/* The selector table of a lowered problem, [T][n_selectors] (n_selectors <= num_parameter of the model): the handle keeps it and
 * writes it into the last n_selectors parameter columns of every instance, now and on every ilqr_set_parameters — which from
 * here on takes the USER's parameters only, w: [B][T][nw - n_selectors], as ilqr_get_dims reports them. n_selectors = 0 detaches.
 * The user's columns (the first nw - n_selectors of every timestep) keep what an earlier ilqr_set_parameters put there. */
int ilqr_set_stage_selectors(ilqr_handle* h, const double* selectors, int32_t n_selectors);

/* Synthetic inputs of the benchmark workloads, generated on the host by ONE function that every host language
 * can call, so that a Julia or C program solves the same instances as the Python bench: value (b, t, j) is a pure function of
 * (seed, b, t, j) — key = seed ^ (b * 2^20 + t * 2^4 + j), splitmix64(key) -> U(0,1), a second splitmix64 round for the Box-Muller
 * partner, z = sqrt(-2 ln u1) cos(2 pi u2) — so shards [first, first + B) of a larger batch are slices of it.
 *   model: "particle" (u = 0.1 z, examples/particle.jl:30), "acrobot" (u = z, test/acrobot.jl:86-88), "car" / "car_goal" / "car_obs"
 *   (instance 0: test/car.jl:24-29 exactly; b >= 1: u scaled by 1 + 0.5 U(-1,1)_b, (x, y) of x1 jittered by 0.05 z), "synth32"
 *   (x1 = 0.5 z, u = 0), "synth12" (x1 = 0.5 z, u = 0.1 z). x1: [B][nx], ubar: [B][T-1][nu] of that model. No device needed.
 * (bench.py measures on this function's output since round 6 — `--generator pcg64` brings back the numpy PCG64 streams the figures of
 * rounds 1-5 were measured on; workloads.make_inputs(generator = "splitmix64") calls it; bench/julia_ref.jl reads its output
 * through tools/dump_inputs.py or `ccall`s it.) */
int ilqr_synthetic_inputs(const char* model, int32_t horizon, uint64_t seed, int64_t first_instance, int32_t batch, double* x1, double* ubar);

/* Test hook: evaluates one of the device-side scalar routines of csrc/ilqr_math.hpp on cuda device 0 — "recip_fast",
 * "rsqrt_fast", "sqrt_fast" (the d of sqrt_rsqrt_fast), "sin_fast", "cos_fast" — elementwise, y[i] = f(x[i]). These replace
 * IEEE division / sqrt and libm on the serial chains (src/rollout.jl:27-29, src/backward_pass.jl:68-75); the tests pin their
 * values at 0, subnormal, huge and non-finite arguments against the IEEE results. Host pointers. */
int ilqr_device_math(const char* fn, const double* x, double* y, int32_t n);

/* Model registry (generated model modules call this from a static initialiser). */
struct ilqr_model_vtable;
int ilqr_register_model(const struct ilqr_model_vtable* vt);
int ilqr_model_count(void);
const char* ilqr_model_name(int32_t i);
/* Large models (nx > 4 or nu > 4): how many Jacobian entries per timestep the kernels treat as state-dependent and how many
 * Hessian entries as structurally non-zero (the rest are constants / zeros: generated tables for symbolic models, found by
 * probing the callables on the host for ilqr_compile_model); 0, 0 for small models. */
int ilqr_model_compact_sizes(const char* model, int32_t* jac_nvar, int32_t* hess_nnz);

#ifdef __cplusplus
}
#endif
#endif
