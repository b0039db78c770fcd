// Workspace layouts shared by host (C-ABI) and device (kernels).
//
// HBM workspace: instance-major — one contiguous block of `stride` doubles per
// problem instance, so the wavefront that owns an instance streams contiguous,
// fully-coalesced lines and instances never share a cache line. Within the
// block every reference buffer (src/data/*.jl) is stored time-major with
// column-major per-timestep matrices, i.e. exactly Julia's memory order.
#pragma once

namespace ilqr {

// scalar slots (stored as doubles) at Layout::scal
enum {
    S_OBJECTIVE = 0, S_MAX_VIOLATION, S_STEP_SIZE, S_STATUS, S_ITERATIONS, S_GRADIENT_NORM,
    S_OUTER_ITERATIONS, S_POTRF_INFO, S_ROLLOUTS, S_STATES_EQ_NOMINAL,
    S_PROF = 10,        // ..15: per-phase cycle counters of an ILQR_PROFILE build
    S_JAC_VALID = 16,   // large path: the Jacobians have been evaluated (or set by the host) since the last reset — the full
                        // jacobian_state / jacobian_action arrays are then the generated constant tables plus the compact entries
    S_DONE = 17,        // host-stepped AL loop only: instance finished
    S_DELTA = 18,       // delta_grad_product = ∇Lᵀ·Δz of the last forward_pass! (src/forward_pass.jl:20)
    S_TRACE_LEN = 19,   // rows of the per-iteration trace written by the last solve
    // host-stepped shared-step mode only (ILQR_STAGE_SS_*): loop state of ilqr_solve! that otherwise lives in registers
    S_OBJ_PREV = 20, S_INNER_DONE = 21, S_J_PREV = 22, S_INNER_IT = 23,
    // straggler hand-over: the packed kernel left this instance at the START of outer iteration S_RESUME (>= 2) for the latency
    // kernel to finish (0 = nothing to resume)
    S_RESUME = 24,
    // ... and, for a hand-over at the head of an inner iteration, the Armijo product ∇Lᵀ·Δz the backward pass produced for the
    // NEXT forward pass (adjoint form): the resumed launch uses this very number instead of summing it again in another order
    S_DELTA_NEXT = 25,
    // backward passes of the last solve that the short nu = 1 form (backward_pass_m1) had to hand to the literal code because a
    // pivot was not positive (0 on healthy instances; the two-wave latency kernel counts, the other kernels leave it alone)
    S_LITERAL_PASSES = 26,
    // when the instance's workgroup started / finished its solve in the latency kernel: s_memrealtime ticks (100 MHz, one counter for
    // the whole device), so that (S_T_END - min S_T_START) of a batch is its finishing-time profile (tools/finish_times.py)
    S_T_START = 27, S_T_END = 28,
    // ... and where its two waves ran: HW_REG_HW_ID (wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]) + 2^32 x XCC_ID
    S_HW0 = 29, S_HW1 = 30, S_HW2 = 31, S_HW3 = 32,   // (large models: the four waves of the workgroup, by hardware wave index)
    S_COUNT = 34
};

// Riccati hand-over between the two waves of a small-model instance: chunks of RING_STEPS timesteps, double-buffered
enum { RING_STEPS = 4, RING_DOUBLES = 2 * RING_STEPS * 3 * 16 };
// doubles of zeros behind Layout::gzero + 2 in every instance block: padding lanes of operands addressed with immediate offsets
// (backward_pass_m1) aim there
enum { GZERO_REGION = 32, GTRASH_REGION = 16 };     // ... and doubles behind those that result lanes with nothing to store may write

struct Layout {
    int T, nx, nu, nw, ncs, nct;
    int C;   // total number of constraints over the horizon
    // offsets (in doubles) inside one instance block
    int xb, ub, x, u, fx, fu, gx, gu, K, k, Lx, Lu, c, lam, rho, act, w;   // LDS-resident set (w: parameters θ_t)
    int zslot;                                                          // [0] always 0.0, [1] write-only trash, [2..11] wave-to-wave scalars
    int lds_doubles;                                                    // size of that set
    int lds_doubles_slim;                                               // ... without fx, fu (throughput variant)
    int ring;                                                           // LDS: two half-rings of RING_STEPS x {Quu, Qux, ux_tmp} (4x4 each)
    int gxx, guu, gux, P, p, scal, gzero;                               // HBM-only set (gzero: a 0.0)
    // large path only (0 otherwise): what the solve kernels actually stream. fv: the JV state-dependent Jacobian entries per
    // timestep (the constant ones come from generated tables); hc: the HS structurally non-zero entries of the accumulated
    // Hessians [gxx | guu | gux] per timestep; ab: a_t = alpha k_t + u_t and b_t = K_t x_t of the running line-search trial.
    // The full fx, fu, gxx, guu, gux arrays above are a host-visible mirror written on demand (materialise_large_kernel).
    int JV, HS, fv, hc, ab;
    int stride;
};

constexpr __host__ __device__ int pad2(int v) { return (v + 1) & ~1; }   // keep 16-B alignment

// Optional action-value buffers Qx, Qu, Qxx, Quu, Qux (src/data/policy.jl:58-64), written by the backward-pass STAGE
// kernel only (parity tests; the fused solve keeps them in registers). One block of `stride` doubles per instance.
struct QLayout { int Qx, Qu, Qxx, Quu, Qux, stride; };
inline __host__ __device__ QLayout make_qlayout(int nx, int nu, int T) {
    QLayout q;
    const int N = T - 1;
    int o = 0;
    q.Qx = o; o += pad2(N * nx);
    q.Qu = o; o += pad2(N * nu);
    q.Qxx = o; o += pad2(N * nx * nx);
    q.Quu = o; o += pad2(N * nu * nu);
    q.Qux = o; o += pad2(N * nu * nx);
    q.stride = (o + 15) & ~15;
    return q;
}

// models with nx > 4 or nu > 4 take the HBM-resident large path (ilqr_device_large.hpp)
inline __host__ __device__ bool is_large_model(int nx, int nu) { return nx > 4 || nu > 4; }
// LDS staging of the large path (must match LargeDims<M>::total); the tail holds a copy of the Layout so that the
// phase functions (real calls) take one pointer instead of twenty on the stack
enum { LAYOUT_LDS_DOUBLES = 24, LARGE_WAVES = 4, LARGE_CHOL = 16 * 17 };
// hess_nnz: structurally non-zero Hessian entries per timestep (the compact row). The row of the NEXT Riccati step and its
// cost gradients are staged in LDS when that still fits a CU's 160 KiB (large_stage_doubles > 0).
constexpr __host__ __device__ int large_lds_base_doubles(int n, int m) {
    const int NP = (n + 15) & ~15, MP = (m + 15) & ~15, ld = NP + 1, ldm = MP + 1;
    // 17 <= nx <= 32 (two tile rows): the hand-made schedule computes one tile of T while Qxx is being stored, so Qxx gets a
    // buffer of its own there (elsewhere it is written over P', which nobody reads any more by then)
    const int qxx = NP == 32 ? NP * ld : 0;
    return 3 * NP * ld + MP * ld + (4 * NP + 2) * ldm + 2 * MP * ldm + LARGE_CHOL + 2 * NP + 8 + LAYOUT_LDS_DOUBLES + qxx;
}
constexpr __host__ __device__ int large_stage_doubles(int n, int m, int hess_nnz) {
    const int want = pad2(pad2(hess_nnz > 0 ? hess_nnz : 1) + n + m);
    return large_lds_base_doubles(n, m) + want <= 160 * 1024 / 8 ? want : 0;
}
constexpr __host__ __device__ int large_lds_doubles(int n, int m, int hess_nnz) {
    return large_lds_base_doubles(n, m) + large_stage_doubles(n, m, hess_nnz);
}

#define ILQR_LAYOUT_FIELDS(X) X(T) X(nx) X(nu) X(nw) X(ncs) X(nct) X(C) X(xb) X(ub) X(x) X(u) X(fx) X(fu) X(gx) X(gu) X(K) X(k) X(Lx) X(Lu) \
    X(c) X(lam) X(rho) X(act) X(w) X(zslot) X(lds_doubles) X(lds_doubles_slim) X(ring) X(gxx) X(guu) X(gux) X(P) X(p) X(scal) X(gzero) X(JV) X(HS) X(fv) X(hc) X(ab) X(stride)
enum { LAYOUT_INTS = 41 };
static_assert(LAYOUT_INTS * 4 <= LAYOUT_LDS_DOUBLES * 8, "layout copy fits its LDS slot");

inline __host__ __device__ Layout make_layout(int nx, int nu, int nw, int ncs, int nct, int T, int jac_nvar = 0, int hess_nnz = 0) {
    Layout L;
    const int N = T - 1;
    L.T = T; L.nx = nx; L.nu = nu; L.nw = nw; L.ncs = ncs; L.nct = nct;
    L.C = N * ncs + nct;
    int o = 0;
    L.xb = o; o += pad2(T * nx);
    L.ub = o; o += pad2(N * nu);
    L.x = o; o += pad2(T * nx);
    L.u = o; o += pad2(N * nu);
    L.gx = o; o += pad2(T * nx);
    L.gu = o; o += pad2(N * nu);
    L.K = o; o += pad2(N * nu * nx);
    L.k = o; o += pad2(N * nu);
    L.Lx = o; o += pad2(N * nx);
    L.Lu = o; o += pad2(N * nu);
    L.c = o; o += pad2(L.C);
    L.lam = o; o += pad2(L.C);
    L.rho = o; o += pad2(L.C);
    L.act = o; o += pad2(L.C);
    L.w = o; o += pad2(T * nw);
    L.zslot = o; o += 16;              // [2..7]: scalars handed from one wave of the instance to the other; [8..11]: cost pass results (two alternating pairs)
    L.lds_doubles_slim = o;            // the throughput variant keeps the Jacobians in HBM/L2
    L.fx = o; o += pad2(N * nx * nx);
    L.fu = o; o += pad2(N * nx * nu);
    L.ring = o; o += RING_DOUBLES;      // Riccati hand-over ring between the two waves of an instance (small path)
    L.lds_doubles = o;
    L.gxx = o; o += pad2(T * nx * nx);
    L.guu = o; o += pad2(N * nu * nu);
    L.gux = o; o += pad2(N * nu * nx);
    L.P = o; o += pad2(T * nx * nx);
    L.p = o; o += pad2(T * nx);
    L.scal = o; o += S_COUNT;
    L.gzero = o; o += 2 + GZERO_REGION + GTRASH_REGION;       // [0] a 0.0, [1] write-only trash (packed kernel), GZERO_REGION zeros, a trash region
    L.JV = 0; L.HS = 0; L.fv = L.hc = L.ab = o;
    if (is_large_model(nx, nu)) {
        L.JV = pad2(jac_nvar > 0 ? jac_nvar : 1); L.HS = pad2(hess_nnz > 0 ? hess_nnz : 1);
        L.fv = o; o += N * L.JV;
        L.hc = o; o += T * L.HS;
        L.ab = o; o += pad2(2 * N * nu);
    }
    L.stride = (o + 15) & ~15;   // 128-B aligned instance blocks
    return L;
}

}  // namespace ilqr
