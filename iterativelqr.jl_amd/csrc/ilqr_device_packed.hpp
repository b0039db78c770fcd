// Packed kernel for small models (nx, nu <= 4): FOUR problem instances per wavefront, no LDS-resident state.
//
// Why: at one instance per wave every issued instruction does one instance's worth of work (the serial phases are
// wave-uniform arithmetic: <= 16 useful lanes in the MFMA Riccati step, 1 in the rollout chain), and the SIMDs saturate
// at ~2.5 such waves each — so beyond ~2 k resident instances the latency / throughput kernels serialise in rounds.
// The LDS-resident working set (20-39 KB per instance) is what caps the residency. Here
//   * the whole per-instance workspace stays in the HBM block of ilqr_layout.hpp (L2 / Infinity-Cache resident for the
//     BASELINE batches) and is STREAMED: every serial loop walks time in one direction and fetches its operands one
//     step ahead into registers, so no array needs to live on chip — and the horizon is no longer bounded by the
//     160 KiB LDS of a CU;
//   * v_mfma_f64_4x4x4 works on its four blocks at once: block beta = (lane >> 2) & 3 carries the Riccati recursion of
//     instance beta of the wave (element (r, c) on lane c + 4*beta + 16*r);
//   * the time-parallel phases (cost, linearisation) and the rollout use the ROW mapping: instance q = lane >> 4 owns
//     the 16 lanes of row q; the cooperative dynamics (one sincos / one division sequence for all arguments of a
//     dependency level) exchange values inside a row with 64-bit DPP row broadcasts (ilqr::RowBC);
//   * the reference's per-instance control flow (AL outer loop, iLQR inner loop, Armijo trials) becomes a per-instance
//     STATE MACHINE: the wave cycles  [outer-loop transitions] -> [line-search trial] -> [linearise + Riccati]  and an
//     instance takes part in a phase when its state asks for it (predicated stores). An instance that rejects a trial
//     simply sits out the following linearisation; nobody waits for anybody else inside a wave beyond that.
// Same arithmetic as the other kernels (one canonical summation order for the objective, explicit FMAs in generated model code):
// an instance may change kernels in the middle of a solve and no output shows it.
//
// Workgroups: one pack + a helper wave that linearises ahead and costs trials behind the rollout (TWO; up to four packs per CU), or
// two packs, one per wave and each on its own, whose workgroup turns into a two-wave latency solver for handed-over instances once
// both packs are through — a device-wide queue, a straggler marked by its rejected line-search trials leaving at once and getting
// its workgroup, then its CU, to itself (packed_solve_body's head of the cycle, pool_mark_step, pool_worker, solve_kernel_packed).
//
// Reference functions reproduced: see ilqr_device.hpp (same citations, paths relative to /root/reference).
#pragma once

namespace ilqr {

template <class M> struct packed_ok { static constexpr bool value = (M::NX <= 4 && M::NU <= 4); };

#ifndef ILQR_PIN_PACKED_CONSTANTS
#define ILQR_PIN_PACKED_CONSTANTS true      // the model's wave-uniform constants as opaque scalar pairs in the rollout (see ilqr_device.hpp)
#endif
#ifndef ILQR_PK_TRIALS
#define ILQR_PK_TRIALS 4      // line-search trials per cycle of the packed kernel's state machine (1 = the first version: one)
#endif
#ifndef ILQR_PK_REJECTS
#define ILQR_PK_REJECTS 8     // ... the extra ones only for an instance with that many rejected trials so far
#endif

namespace pk {

enum { ST_INIT = 0, ST_FORWARD = 1, ST_OUTER = 2, ST_DONE = 3 };

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }
// sum / NaN-propagating max over the 16 lanes of a row (rows = instances in the row mapping)
__device__ __forceinline__ double row_sum(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double row_max(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = (w > v || w != w) ? w : v; }
    return v;
}
// bit q of the result = predicate of instance q (taken from the first lane of its row)
__device__ __forceinline__ unsigned row_mask(bool p) {
    const unsigned long long b = __ballot(p);
    return (unsigned)((b & 1ull) | ((b >> 15) & 2ull) | ((b >> 30) & 4ull) | ((b >> 45) & 8ull));
}
// v_permlane32_swap(v, v) = {[lo lo], [hi hi]}: rows 2, 3 <- rows 0, 1 (first result) or rows 0, 1 <- rows 2, 3 (second)
__device__ __forceinline__ double from_lane_minus32(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
__device__ __forceinline__ double from_lane_plus32(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}

// Per-lane view of the wave's four instances.
template <class M>
struct PInst {
    // row mapping (instance q = lane >> 4, j = lane & 15): workspace block of this lane's instance
    double* g;
    // block mapping (instance beta = (lane >> 2) & 3, element (r, c) = (lane >> 4, lane & 3))
    double* gb;
    const char* wb;    // wave-uniform: block of the wave's first instance (per-lane 32-bit byte offsets are taken from it)
    Layout L;
    int lane, q, j, beta, r, c;
    bool valid_row, valid_blk;
    double* trace;
    int trace_cap;
    // SolverData of the row's instance, replicated over its 16 lanes
    double objective, max_violation, step_size, gradient_norm, obj_prev, J_prev, delta, delta_next;
    int status, iterations, outer, it, trial, rollouts, potrf_info, states_eq_nominal, trace_len, state, needB, leaving;
    int nbar;          // workgroup barriers this wave has passed (two-wave form: an order to the helper wave names the barrier it is valid from)
    // one-wave form: a workgroup holds TWO packs, one per wave, each with its own control flow and its own LDS region (lds0, in
    // doubles from the dynamic LDS base); such a wave never meets the other at a barrier inside the state machine
    int lds0;
    bool paired;
    int rej_acc;       // rejected line-search trials of the forward passes that ended in an acceptance (the straggler mark's measure); bit 30: marked
};
// every workgroup barrier of the solver wave goes through here (one-wave form: what __syncthreads() is to a lone wave — its own
// stores and loads in order: fences of WAVEFRONT scope, which cost no s_waitcnt (in a 64-thread workgroup the compiler reduces
// __syncthreads() to exactly that; workgroup-scope fences in the 128-thread workgroup drained every store at every sync point, +5 %
// on the shard) — and no s_barrier, which would tie the workgroup's two independent packs together)
template <class M> __device__ __forceinline__ void pk_sync(PInst<M>& I) {
    if (I.paired) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    else __syncthreads();
    I.nbar += 1;
}
enum { PK_LINEARISE = 1, PK_COST = 2 };
// Two-wave form: the solver wave's order to the helper wave, valid from the solver wave's NEXT barrier on. The helper looks into
// the mailbox after every barrier and may well see an order early (written after the barrier both just passed): it takes it only
// once its own barrier count has reached the one the order names.
template <class M>
__device__ __forceinline__ void pk_post(PInst<M>& I, int mbox, int kind, unsigned mask, bool constrained) {
    extern __shared__ __attribute__((aligned(16))) double pk_lds[];
    if (I.lane == 0) {
        volatile int* mb = (volatile int*)(pk_lds + mbox);
        mb[1] = (int)mask; mb[2] = constrained ? 1 : 0; mb[3] = kind; mb[4] = I.nbar + 1;
        mb[0] = mb[0] + 1;
    }
}

// ------------------------------------------------------------------ cost! (row mapping, one timestep per lane of a row)
// X / U: offsets of the trajectory inside the instance block. upd_J / upd_viol are per-lane (per-instance) predicates.
// The work of ONE timestep is split into its operand loads and its evaluation (cost_load / cost_eval), so that cost_pass can
// request a lane's next timestep before evaluating the current one — and so that the two-wave form's helper can evaluate the
// timesteps of a trial trajectory segment by segment behind the rollout (cost_follow), with the very same code.
template <class M>
struct CostIn {
    static constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT, NCM = ncs > nct ? ncs : nct, NC = NCM > 0 ? NCM : 1;
    double x[n], u[m > 0 ? m : 1], w[cdim<M::NW>::v], lam[NC], rho[NC];
};
template <class M>
__device__ __forceinline__ void cost_load(const double* g, const Layout& L, int Xo, int Uo, bool duals, int t, CostIn<M>& o) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT, NCM = CostIn<M>::NCM;
    const int N = L.T - 1;
    load_w<M::NW>(g + L.w, t, o.w);
#pragma unroll
    for (int i = 0; i < n; ++i) o.x[i] = g[Xo + t * n + i];
    if (t < N) {
#pragma unroll
        for (int i = 0; i < m; ++i) o.u[i] = g[Uo + t * m + i];
    }
    if (duals) {
        const int off = t < N ? t * ncs : N * ncs, cnt = t < N ? ncs : nct;
#pragma unroll
        for (int i = 0; i < NCM; ++i)
            if (i < cnt) { o.lam[i] = g[L.lam + off + i]; o.rho[i] = g[L.rho + off + i]; }
    }
}
// (Inlined into its two callers — cost_pass and the helper wave's cost_follow. Round 4: hipcc contracted the model's sums of products
// into FMAs differently at the two sites and the two forms of the kernel reported objectives one ulp apart on a few trials. Since
// round 5 the generated M::cost_s / cost_t / con_s / con_t carry `#pragma clang fp contract(off)`, the AL terms are the explicit fma
// chains of objective_term (ilqr_device.hpp) and the sums follow the canonical order stated there: J is the same bits in every
// kernel family, wherever an instance changes kernels.)
template <class M>
__device__ __forceinline__ void cost_eval(double* g, const Layout& L, const CostIn<M>& cur, int t, bool upd_J, bool upd_viol, bool constrained,
                                          ObjAcc& Jp, double& vp) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    const int N = L.T - 1;
    double* cbuf = g + L.c; double* act = g + L.act;
    const double (&xt)[n] = cur.x;
    const double (&w)[cdim<M::NW>::v] = cur.w;
    if (t < N) {
        double ut[m];
#pragma unroll
        for (int i = 0; i < m; ++i) ut[i] = cur.u[i];
        double l_ = 0.0;
        if (upd_J) { l_ = M::cost_s(xt, ut, w); ILQR_OPAQUE(l_); }
        if constexpr (ncs == 0) { if (upd_J) Jp.add(t, l_); }
        if constexpr (ncs > 0) {
            if (!constrained) { if (upd_J) Jp.add(t, l_); }
            if (constrained) {
                double cv[ncs];
                M::con_s(xt, ut, w, cv);
                const int off = t * ncs;
                if (upd_J) { double v_ = objective_term<M, true, ncs>(l_, cv, cur.lam, cur.rho, act + off, true); ILQR_OPAQUE(v_); Jp.add(t, v_); }
                if (upd_viol) {
#pragma unroll
                    for (int i = 0; i < ncs; ++i) {
                        cbuf[off + i] = cv[i];
                        const bool ineq = IneqMask<M>::s(i);
                        vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                    }
                }
            }
        }
    } else {
        double l_ = 0.0;
        if (upd_J) { l_ = M::cost_t(xt, w); ILQR_OPAQUE(l_); }
        if constexpr (nct == 0) { if (upd_J) Jp.add(t, l_); }
        if constexpr (nct > 0) {
            if (!constrained) { if (upd_J) Jp.add(t, l_); }
            if (constrained) {
                double cv[nct];
                M::con_t(xt, w, cv);
                const int off = N * ncs;
                if (upd_J) { double v_ = objective_term<M, false, nct>(l_, cv, cur.lam, cur.rho, act + off, true); ILQR_OPAQUE(v_); Jp.add(t, v_); }
                if (upd_viol) {
#pragma unroll
                    for (int i = 0; i < nct; ++i) {
                        cbuf[off + i] = cv[i];
                        const bool ineq = IneqMask<M>::t(i);
                        vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                    }
                }
            }
        }
    }
}
template <class M>
__device__ void cost_pass(PInst<M>& I, int Xo, int Uo, bool upd_J, bool upd_viol, bool constrained, double& J_out, double& viol_out) {
    const Layout& L = I.L;
    const int T = L.T;
    ObjAcc Jp;
    double vp = 0.0;
    // A lane walks its timesteps t = j, j + 16, ...; every pass needs x_t, u_t, λ_t, ρ_t from HBM / L2 and nothing hides that
    // round trip at one wave per SIMD (the pass used to cost one round trip per 16 timesteps: 9 of the 80 µs of a car cycle).
    // The operands of the NEXT pass are requested before the current one is evaluated.
    const bool any = upd_J || upd_viol, duals = constrained && upd_J;
    CostIn<M> cur, nxt;
    int t = any ? I.j : T;
    if (t < T) cost_load<M>(I.g, L, Xo, Uo, duals, t, cur);
    for (; t < T; t += 16) {
        if (t + 16 < T) cost_load<M>(I.g, L, Xo, Uo, duals, t + 16, nxt);
        cost_eval<M>(I.g, L, cur, t, upd_J, upd_viol, constrained, Jp, vp);
        cur = nxt;
    }
    J_out = row_sum(Jp.S());
    viol_out = row_max(vp);
    pk_sync<M>(I);
}

// cost!(data, problem, mode) for the instances with `act` set — src/data/methods.jl:13-30 (Q2: the violations buffer
// and max_violation are always taken at problem.states; one pass when states == nominal bitwise).
template <class M>
__device__ void cost_bang(PInst<M>& I, bool act, bool mode_current, bool constrained) {
    const Layout& L = I.L;
    const bool one_pass = mode_current || I.states_eq_nominal || !constrained;
    {
        double J, v;
        cost_pass<M>(I, mode_current ? L.x : L.xb, mode_current ? L.u : L.ub, act, act && one_pass, constrained, J, v);
        if (act) {
            I.objective = J;
            if (one_pass && constrained) I.max_violation = v;
        }
    }
    const bool second = act && !one_pass;
    if (__any(second)) {
        double J, v;
        cost_pass<M>(I, L.x, L.u, false, second, constrained, J, v);
        if (second) I.max_violation = v;
    }
}

// ------------------------------------------------- gradients! FUSED into backward_pass! (the B phase of the state machine)
// The Riccati recursion walks time backwards in chunks of 16 steps. Each chunk is first linearised time-parallel in the ROW
// mapping (lane j of row q: timestep 16*chunk + j of instance q) — dynamics Jacobians and cost gradients go to a 13 KB LDS
// buffer of the wave and never touch HBM, the accumulated Hessians are updated in place in the HBM block (Q1) — and then
// consumed by the four MFMA blocks. HBM sees x̄, ū (read), the Hessians (read-modify-write, then an L2-hit read) and
// K, k, ∇L (write): 46 KB per instance and iteration for the acrobot instead of 106 KB for separate sweeps.
// The Armijo slope  Δ = ∇Lᵀ·Δz  (src/forward_pass.jl:19-20, src/data/methods.jl:42-54) rides along as the ADJOINT of the
// sensitivity recursion — with  w_t = ∇L_u,t + fu_tᵀ ν_{t+1},  ν_t = ∇L_x,t + fx_tᵀ ν_{t+1} + K_tᵀ w_t,  ν_H = 0:
// Δ = Σ_t w_tᵀ k_t  — the same bilinear form summed in the opposite order (four MFMAs per step on operands that are in
// registers anyway), which saves the forward sweep over fx, fu, K, k, ∇L altogether.
// TWO: the two-wave form of the kernel (below) — a second chunk buffer, so that a helper wave linearises chunk ch - 1 while the
// Riccati steps of chunk ch run, and a mailbox through which the solver wave tells it what to linearise.
template <class M, bool TWO = false>
struct PkLds {
    static constexpr int n = M::NX, m = M::NU;
    static constexpr int SFX = (n * n) | 1, SFU = (n * m) | 1, SGX = n | 1, SGU = m | 1;      // odd strides: conflict-free lane-per-step writes
    static constexpr int FX = 0, FU = FX + 16 * SFX, GX = FU + 16 * SFU, GU = GX + 16 * SGX, IB = GU + 16 * SGU;
    static constexpr int BUF = 4 * IB, TERM = (TWO ? 2 : 1) * BUF, ZERO = TERM + 4 * n, MBOX = ZERO + 2, RES = MBOX + 4, total = TWO ? RES + 8 : MBOX;
    // mailbox (ints at MBOX): [0] sequence number (-1: leave), [1] instance mask, [2] constrained, [3] kind (PK_LINEARISE / PK_COST),
    // [4] the solver wave's barrier count at which the order becomes valid; RES: the helper's answers (J and max violation per instance)
};

// One stage timestep of gradients!: a real function, so that the (large) symbolic Jacobian code gets its own register
// allocation instead of pushing the Riccati state of the caller into scratch.
struct LinArgs { double* gr; int xb, ub, w, gxx, guu, gux, rho, act, lam, c, t, lds_row, j, constrained; };
template <class M>
__attribute__((noinline)) __device__ void linearise_stage(LinArgs a) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS;
    typedef PkLds<M> LD;
    extern __shared__ __attribute__((aligned(16))) double pk_lds[];
    double* gr = a.gr;
    double* lrow = pk_lds + a.lds_row;
    const int t = a.t;
    double w[cdim<M::NW>::v];
    load_w<M::NW>(gr + a.w, t, w);
    double xt[n], ut[m];
#pragma unroll
    for (int i = 0; i < n; ++i) xt[i] = gr[a.xb + t * n + i];
#pragma unroll
    for (int i = 0; i < m; ++i) ut[i] = gr[a.ub + t * m + i];
    {   // touch the Hessian lines of this timestep: the Riccati steps of the chunk then hit L2 even where the accumulators
        // below visit no entry of a line (structural zeros)
        double t0_ = gr[a.gxx + t * n * n], t1_ = gr[a.gxx + t * n * n + n * n - 1], t2_ = gr[a.guu + t * m * m], t3_ = gr[a.gux + t * m * n];
        asm volatile("" :: "v"(t0_), "v"(t1_), "v"(t2_), "v"(t3_));
    }
    M::dyn_jac_mem(xt, ut, w, lrow + LD::FX + a.j * LD::SFX, lrow + LD::FU + a.j * LD::SFU);   // `.=` (src/dynamics.jl:45-46)
    double gx[n], gu[m];
    M::cost_s_grad(xt, ut, w, gx, gu);                                                          // `.=` (src/costs.jl:61,65)
    double* gxx = gr + a.gxx + t * n * n;
    double* guu = gr + a.guu + t * m * m;
    double* gux = gr + a.gux + t * m * n;
    M::cost_s_hess_acc(xt, ut, w, gxx, guu, gux);                                               // `.+=` (src/costs.jl:74-80)
    if constexpr (ncs > 0) {
        if (a.constrained) {                                                                    // src/gradients.jl:54-80
            double ct[ncs], ir[ncs];
            const int off = t * ncs;
#pragma unroll
            for (int i = 0; i < ncs; ++i) al_multipliers(gr[a.rho + off + i], gr[a.act + off + i], gr[a.lam + off + i], gr[a.c + off + i], ir[i], ct[i]);
            M::al_s(xt, ut, w, ct, ir, gx, gu, gxx, guu, gux);
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) lrow[LD::GX + a.j * LD::SGX + i] = gx[i];
#pragma unroll
    for (int i = 0; i < m; ++i) lrow[LD::GU + a.j * LD::SGU + i] = gu[i];
}

template <class M, bool TWO>
__device__ void linearise_riccati(PInst<M>& I, bool act_row, unsigned mask, bool constrained,
                                  double& gnorm_row, int& info_row, double& delta_row) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    typedef PkLds<M, TWO> LD;
    static_assert(n <= 4 && m <= 4, "packed Riccati step: nx, nu <= 4");
    constexpr bool SHARE = 2 * m <= 4;      // k's right-hand side fits into the spare rows m..2m-1 of the block and shares K's solve
    extern __shared__ __attribute__((aligned(16))) double pk_lds_[];
    double* const pk_lds = pk_lds_ + I.lds0;                                        // this pack's region
    const Layout& L = I.L;
    const int lane = I.lane, r = I.r, c = I.c, N = L.T - 1;
    const bool on = I.valid_blk && ((mask >> I.beta) & 1u);
    double* gr = I.g;                       // row mapping block
    double* g = I.gb;                       // block mapping block
    const double* lblk = pk_lds + I.beta * LD::IB;
    if (lane == 0) { pk_lds[LD::ZERO] = 0.0; pk_lds[LD::ZERO + 1] = 0.0; }
    // ---- terminal linearisation (t = N): gx[N] `.=`, gxx[N] `.+=` (src/costs.jl:57-84, src/gradients.jl:54-67)
    if (act_row && I.j == 0) {
        double w[cdim<M::NW>::v];
        load_w<M::NW>(gr + L.w, N, w);
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = gr[L.xb + N * n + i];
        double gx[n];
        M::cost_t_grad(xt, w, gx);
        double* gxx = gr + L.gxx + N * n * n;
        M::cost_t_hess_acc(xt, w, gxx);
        if constexpr (nct > 0) {
            if (constrained) {
                double ct[nct], ir[nct];
                const int off = N * ncs;
#pragma unroll
                for (int i = 0; i < nct; ++i) al_multipliers(gr[L.rho + off + i], gr[L.act + off + i], gr[L.lam + off + i], gr[L.c + off + i], ir[i], ct[i]);
                M::al_t(xt, w, ct, ir, gx, gxx);
            }
        }
#pragma unroll
        for (int i = 0; i < n; ++i) { pk_lds[LD::TERM + I.q * n + i] = gx[i]; gr[L.gx + N * n + i] = gx[i]; }
    }
    if constexpr (TWO) pk_post<M>(I, LD::MBOX, PK_LINEARISE, mask, constrained);   // the helper wave's order: which instances, constrained or not (pk_helper)
    pk_sync<M>(I);
    const bool vnn = on && r < n && c < n, vnm = on && r < n && c < m, vmn = on && r < m && c < n, vmm = on && r < m && c < m;
    const bool vn1 = on && c == 0 && r < n, vm1 = on && c == 0 && r < m;
    // Addresses for INSTRUCTION COUNT (a wave issues one instruction per 5-6 clk whatever it is, tools/probes/probe_issue.hip):
    // HBM operands and results through 32-bit per-lane byte offsets from the wave's (scalar) base that walk backwards in time — one
    // v_sub per access instead of a multiply and a 64-bit add; padding lanes aim at the block's zero / trash slot with stride 0.
    // LDS chunk operands through per-lane byte addresses re-aimed at the top of every chunk.
    typedef __attribute__((address_space(3))) double ldsd;
    const char* wb = I.wb;
    const unsigned ob = (unsigned)((const char*)g - wb);
    auto GL = [&](unsigned off) -> double { return *(const double*)(wb + off); };
    auto GS = [&](unsigned off, double v) { *(double*)(const_cast<char*>(wb) + off) = v; };
    auto ldsa = [](const double* q) -> unsigned { return (unsigned)(size_t)(const ldsd*)q; };
    auto LDr = [](unsigned a_) -> double { return *(const ldsd*)(size_t)a_; };
    const unsigned ozero = ob + 8u * L.gzero, otrash = ob + 8u * (L.gzero + 1);
    unsigned oxx = vnn ? ob + 8u * (L.gxx + (N - 1) * n * n + c * n + r) : ozero;   const unsigned sxx = vnn ? 8u * n * n : 0u;
    unsigned ouu = vmm ? ob + 8u * (L.guu + (N - 1) * m * m + c * m + r) : ozero;   const unsigned suu = vmm ? 8u * m * m : 0u;
    unsigned oux = vmn ? ob + 8u * (L.gux + (N - 1) * m * n + c * m + r) : ozero;   const unsigned sux = vmn ? 8u * m * n : 0u;
    unsigned oK = vmn ? ob + 8u * (L.K + (N - 1) * m * n + c * m + r) : otrash;     const unsigned sK = vmn ? 8u * m * n : 0u;
    unsigned ok_ = vm1 ? ob + 8u * (L.k + (N - 1) * m + r) : otrash;                const unsigned sk = vm1 ? 8u * m : 0u;
    unsigned oLu = vm1 ? ob + 8u * (L.Lu + (N - 1) * m + r) : otrash;
    unsigned oLx = vn1 ? ob + 8u * (L.Lx + (N - 1) * n + r) : otrash;               const unsigned sLx = vn1 ? 8u * n : 0u;
    const unsigned lzero = ldsa(pk_lds + LD::ZERO);
    const unsigned bfx = vnn ? ldsa(lblk + LD::FX + c * n + r) : lzero;   const unsigned sfx = vnn ? 8u * LD::SFX : 0u;
    const unsigned bfu = vnm ? ldsa(lblk + LD::FU + c * n + r) : lzero;   const unsigned sfu = vnm ? 8u * LD::SFU : 0u;
    const unsigned bgx = vn1 ? ldsa(lblk + LD::GX + r) : lzero;           const unsigned sgx = vn1 ? 8u * LD::SGX : 0u;
    const unsigned bgu = vm1 ? ldsa(lblk + LD::GU + r) : lzero;           const unsigned sgu = vm1 ? 8u * LD::SGU : 0u;
    unsigned afx = bfx, afu = bfu, agx = bgx, agu = bgu;

    double P = vnn ? g[L.gxx + N * n * n + c * n + r] : 0.0;            // P[H] .= gxx[H]  (:39)
    double p = vn1 ? pk_lds[LD::TERM + I.beta * n + r] : 0.0;           // p[H] .= gx[H]   (:40)
    double nu = 0.0, dacc = 0.0;                                        // adjoint of the sensitivity recursion, Δ accumulator
    double gmax = 0.0;
    bool gnan = false;
    int pinfo = 0;
    const int rr = (m == 2) ? (r & 1) : r;                              // row inside a right-hand-side set
    // potrs('U') of right-hand sides held as Y(r, c), rows on lanes 16 apart (m == 2: two sets, rows {0,1} and {2,3})   (:70-75)
    auto solve = [&](double Y, const double (&Uc)[m * m], const double (&Ur)[m], int info) {
        if constexpr (m == 2) return solve2x2_rows(Y, (r & 1) != 0, Uc[2], Ur[0], Ur[1]);      // (ilqr_device.hpp)
        if (m == 1 && info == 0) {
            Y = Y * recip_fast(Uc[0]);                                  // see ilqr_device.hpp: 1x1 shortcut
        } else {
#pragma unroll
            for (int i = 0; i < m; ++i) {                               // U^T y = b
#pragma unroll
                for (int l = 0; l < i; ++l) {
                    const double yl = (m == 2) ? from_lane_minus16_odd_rows(Y) : __shfl(Y, lane - 16 * (i - l));
                    const double v = Y - Uc[i * m + l] * yl;
                    Y = (rr == i) ? v : Y;
                }
                const double qv = Y * Ur[i];
                Y = (rr == i) ? qv : Y;
            }
#pragma unroll
            for (int i = m - 1; i >= 0; --i) {                          // U x = y
#pragma unroll
                for (int l = i + 1; l < m; ++l) {
                    const double xl = (m == 2) ? from_lane_plus16_even_rows(Y) : __shfl(Y, lane + 16 * (l - i));
                    const double v = Y - Uc[l * m + i] * xl;
                    Y = (rr == i) ? v : Y;
                }
                const double qv = Y * Ur[i];
                Y = (rr == i) ? qv : Y;
            }
        }
        return Y * -1.0;
    };
    const int blk0 = (lane & 12);
    struct Opnd { double gxx, guu, gux, fx, fu, gx, gu; };
    auto fetch = [&](Opnd& o) {          // operands at the walking addresses (timesteps are fetched in descending order), then one step back
        o.gxx = GL(oxx); o.guu = GL(ouu); o.gux = GL(oux);
        oxx -= sxx; ouu -= suu; oux -= sux;
        o.fx = LDr(afx); o.fu = LDr(afu); o.gx = LDr(agx); o.gu = LDr(agu);
        afx -= sfx; afu -= sfu; agx -= sgx; agu -= sgu;
    };
    auto riccati_step = [&](const Opnd& o, int t) {
        ILQR_ISA_MARK("riccati_step", 4);
        const double fx = o.fx, fu = o.fu;
        const double W = mfma444(P, fx, 0.0);                           // (:52-64)
        const double Wu = mfma444(P, fu, 0.0);
        const double Qxx = mfma444(W, fx, o.gxx);
        const double Qux = mfma444(Wu, fx, o.gux);
        const double Quu = mfma444(Wu, fu, o.guu);
        const double Qx = mfma444(fx, p, o.gx);                         // (:44-49)
        const double Qu = mfma444(fu, p, o.gu);
        double K, k;
        if constexpr (m == 1) {
            // 1x1: Quu(0,0) of the block sits on lane 4*beta; a DPP quad broadcast hands it to the block's row 0, where K (a row) and
            // k (its first entry) live. Rows 1..3 are padding: Qux and Qu are exactly zero there, a pivot of 1 keeps them zero.
            // K = -(Qux (1/q)), k = -(Qu (1/q)) with the reciprocal of ilqr_device.hpp's 1x1 shortcut on EVERY lane; an instance
            // whose potrf fails (q <= 0 or NaN; info ignored by the reference, :69) redoes its lanes in LAPACK's arithmetic behind
            // one wave-uniform branch — no exec-masked region on the common path.
            const double q0 = quad_bcast<0>(Quu);
            const double qs = (r == 0) ? q0 : 1.0;                      // (:68-69)
            const bool bad = !(qs > 0.0);
            const double rinv = recip_fast(qs);
            K = (Qux * rinv) * -1.0;                                    // (:70-75)
            k = (Qu * rinv) * -1.0;
            if (__builtin_expect(__any(bad), 0)) {
                if (bad) {
                    double Uc[1] = {qs}, Ur[1];
                    const int info = potrf_U<1>(Uc, Ur);
                    if (info != 0 && pinfo == 0) pinfo = info;
                    K = ((Qux * Ur[0]) * Ur[0]) * -1.0;
                    k = ((Qu * Ur[0]) * Ur[0]) * -1.0;
                }
            }
        } else {
            double Uc[m * m];
#pragma unroll
            for (int jj = 0; jj < m; ++jj)
#pragma unroll
                for (int i = 0; i < m; ++i) Uc[jj * m + i] = (i <= jj) ? shfl_d(Quu, blk0 + jj + 16 * i) : 0.0;   // (:68-69)
            double Ur[m];
            int info = 0;
            if (__builtin_expect(__any(potrf_U_nofail<m>(Uc, Ur)), 0)) {    // (ilqr_device.hpp) a pivot failed in one of the wave's instances
                ILQR_ISA_COLD_BEGIN();
#pragma unroll
                for (int jj = 0; jj < m; ++jj)
#pragma unroll
                    for (int i = 0; i < m; ++i) Uc[jj * m + i] = (i <= jj) ? shfl_d(Quu, blk0 + jj + 16 * i) : 0.0;
                info = potrf_U<m>(Uc, Ur);
                ILQR_ISA_COLD_END();
            }
            if (info != 0 && pinfo == 0) pinfo = info;
            if constexpr (SHARE) {
                // right-hand sides: Qux in rows 0..m-1, Qu (column 0) moved down into rows m..2m-1: ONE solve for K and k
                const double Qu_dn = from_lane_minus32(Qu);
                const bool krow = (r >= m && r < 2 * m && c == 0);
                const double Y = solve(krow ? Qu_dn : Qux, Uc, Ur, info);   // (:70-75)
                K = (r < m) ? Y : 0.0;
                const double kY = krow ? Y : 0.0;
                const double k_up = from_lane_plus32(kY);
                k = (r < m && c == 0) ? k_up : 0.0;
            } else {                                                        // nu = 3, 4: no spare rows, two solves
                const double YK = solve(Qux, Uc, Ur, info);
                const double Yk = solve(Qu, Uc, Ur, info);
                K = (r < m) ? YK : 0.0;
                k = (r < m && c == 0) ? Yk : 0.0;
            }
        }
        const double uxt = mfma444(Quu, K, 0.0);                        // (:79)
        // (:81-84), (:86-89): summed as ((Qxx + K^T Qux) + Qux^T K) + K^T ux_tmp, the association of backward_pass_mfma
        // (ilqr_device.hpp) — results must not depend on which kernel variant a batch size selects
        double Pn = mfma444(K, Qux, Qxx);
        Pn = mfma444(Qux, K, Pn);
        Pn = mfma444(K, uxt, Pn);
        double pn = mfma444(K, Qu, Qx);
        pn = mfma444(Qux, k, pn);
        pn = mfma444(uxt, k, pn);
        const double Lx = Qx - pn;                                      // src/solve.jl:73-81
        asm("v_max_f64 %0, %1, |%2|" : "=v"(gmax) : "v"(gmax), "v"(Lx));    // v_max_f64 drops NaNs: they are tracked beside it
        asm("v_max_f64 %0, %1, |%2|" : "=v"(gmax) : "v"(gmax), "v"(Qu));
        gnan |= (Lx != Lx) | (Qu != Qu);
        // adjoint sensitivity step: w = ∇L_u + fuᵀν', Δ += wᵀk, ν = ∇L_x + fxᵀν' + Kᵀw
        const double wv = mfma444(fu, nu, Qu);
        dacc = mfma444(wv, k, dacc);
        double nun = mfma444(fx, nu, Lx);
        nun = mfma444(K, wv, nun);
        nu = nun;
        GS(oK, K); GS(ok_, k); GS(oLu, Qu); GS(oLx, Lx);
        oK -= sK; ok_ -= sk; oLu -= sk; oLx -= sLx;
        P = Pn; p = pn;
    };
    for (int ch = (N + 15) / 16 - 1; ch >= 0; --ch) {
        const int t0 = 16 * ch, cnt = (N - t0) < 16 ? (N - t0) : 16;
        // ---- linearise timesteps t0 .. t0 + cnt - 1, one per lane of a row (two-wave form: the helper wave has done it, into
        // buffer ch & 1, while the previous chunk's steps ran here)
        if constexpr (!TWO) {
            if (act_row && I.j < cnt) {
                LinArgs la{gr, L.xb, L.ub, L.w, L.gxx, L.guu, L.gux, L.rho, L.act, L.lam, L.c, t0 + I.j, I.lds0 + I.q * LD::IB, I.j, constrained ? 1 : 0};
                linearise_stage<M>(la);
            }
        }
        pk_sync<M>(I);
        // ---- Riccati steps of the chunk, last timestep first   (:42)
        Opnd A, B;
        int sl = cnt - 1;
        const unsigned buf = TWO ? (unsigned)(ch & 1) * 8u * LD::BUF : 0u;                         // (lzero lies behind both buffers: no offset for padding lanes)
        afx = bfx + sl * sfx + (vnn ? buf : 0u); afu = bfu + sl * sfu + (vnm ? buf : 0u);         // LDS operands of the chunk's last step
        agx = bgx + sl * sgx + (vn1 ? buf : 0u); agu = bgu + sl * sgu + (vm1 ? buf : 0u);
        fetch(A);
        // pairs of steps while a whole pair follows (both sets refilled unconditionally: a fetch behind a branch costs a copy of the
        // seven operands at the loop edge), then the chunk's last one to three steps; nothing is fetched across the chunk boundary
        // (the next chunk's Hessians are accumulated by its linearisation first)
        for (; sl >= 3; sl -= 2) {
            fetch(B);
            riccati_step(A, t0 + sl);
            fetch(A);
            riccati_step(B, t0 + sl - 1);
        }
        if (sl == 2) {
            fetch(B); mfma_block_boundary_guard(); riccati_step(A, t0 + 2);
            fetch(A); riccati_step(B, t0 + 1);
            riccati_step(A, t0);
        } else if (sl == 1) {
            fetch(B); mfma_block_boundary_guard(); riccati_step(A, t0 + 1);
            riccati_step(B, t0);
        } else {
            mfma_block_boundary_guard(); riccati_step(A, t0);
        }
        if constexpr (!TWO) pk_sync<M>(I);  // the chunk buffer is free again (two buffers: the next chunk's barrier says so)
    }
    double gm = (on && c == 0) ? (gnan ? __builtin_nan("") : gmax) : 0.0;
    { double w_; w_ = __shfl_xor(gm, 16); gm = nanmax(gm, w_); w_ = __shfl_xor(gm, 32); gm = nanmax(gm, w_); }
    gnorm_row = shfl_d(gm, 4 * I.q);
    info_row = __shfl(pinfo, 4 * I.q);
    delta_row = shfl_d(dacc, 4 * I.q);
    pk_sync<M>(I);
}

// ------------------------------------------------------------- rollout! (row mapping, cooperative inside a row)
// alpha is per instance. Lane j == 0 of an active row stores the trial trajectory. Every operand and result goes through 32-bit
// byte offsets from the wave's scalar base that advance once per three steps (immediate offsets in between), instead of a
// multiply and a 64-bit add per access. (Forming a_t = k_t α + ū_t and b_t = K_t x̄_t beforehand, as the LDS kernels do, does
// not pay here: the pass that forms them waits on HBM seven times per rollout with nothing to hide behind — measured
// 113 -> 119 ms on acrobot:8192.)
template <class M, bool FOLLOW = false>     // FOLLOW (two-wave form): a workgroup barrier per 15 steps, behind which the helper wave evaluates the trial's cost
__device__ void rollout(PInst<M>& I, bool act, double alpha) {
    constexpr int n = M::NX, m = M::NU;
    const Layout& L = I.L;
    const int N = L.T - 1;
    double* g = I.g;
    const char* wb = I.wb;
    const unsigned og = (unsigned)((const char*)g - wb);
    // (offset zero-extended FIRST, constant index added in 64 bits: the constant then folds into the instruction's immediate
    // offset and neighbouring doubles merge into one wide load; a 32-bit add could wrap, so hipcc would keep it)
    auto GL = [&](unsigned off, int idx) -> double { return ((const double*)(wb + off))[idx]; };
    auto GS = [&](unsigned off, int idx, double v) { ((double*)(const_cast<char*>(wb) + off))[idx] = v; };
    // lanes that store nothing aim at the block's trash REGION with stride 0 and use the same constant index as the writers (a
    // lane-dependent index cannot fold into the instruction's immediate: a 64-bit address add in front of every store)
    const unsigned otrash = og + 8u * (L.gzero + 2 + GZERO_REGION);
    static_assert(n <= GTRASH_REGION && m <= GTRASH_REGION, "trash region holds a row of x or u");
    const bool wr = act && I.j == 0;
    double xt[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xt[i] = g[L.xb + i];                    // (:19)
    if (wr) {
#pragma unroll
        for (int i = 0; i < n; ++i) g[L.x + i] = xt[i];
    }
    const typename M::WaveCtx wcx = M::template wave_ctx<ILQR_PIN_PACKED_CONSTANTS>(I.j);   // per-lane constants of the cooperative dynamics (I.j: lane of the row)
    // loads on every lane of the row (all lanes evaluate the policy), stores on the row's lane 0 only (trash slot, stride 0 elsewhere)
    unsigned oK = og + 8u * L.K, ok_ = og + 8u * L.k, oub = og + 8u * L.ub, oxb = og + 8u * L.xb;
    unsigned wU = wr ? og + 8u * L.u : otrash, wX = wr ? og + 8u * L.x : otrash;
    const unsigned sU = wr ? 8u * m : 0u, sX = wr ? 8u * n : 0u;
    struct Ops { double K[m * n], k[m], ub[m], xb[n]; };
    // d: step offset (0..4) from the step the walking offsets stand at
    auto fetch = [&](Ops& o, int d) {
#pragma unroll
        for (int i = 0; i < m * n; ++i) o.K[i] = GL(oK, d * m * n + i);
#pragma unroll
        for (int i = 0; i < m; ++i) { o.k[i] = GL(ok_, d * m + i); o.ub[i] = GL(oub, d * m + i); }
#pragma unroll
        for (int i = 0; i < n; ++i) o.xb[i] = GL(oxb, d * n + i);
    };
    auto step = [&](const Ops& o, int t, int d, const double (&xin)[n], double (&xout)[n]) {
        ILQR_ISA_MARK("rollout_step", 3);
        double ut[m];
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int jj = 0; jj < n; ++jj) {
                a1 = fma(o.K[jj * m + i], xin[jj], a1);
                a2 = fma(o.K[jj * m + i], o.xb[jj], a2);
            }
            ut[i] = (fma(o.k[i], alpha, o.ub[i]) + a1) - a2;            // (:24-28)
        }
        double w[cdim<M::NW>::v];
        load_w<M::NW>(g + L.w, t, w);
        M::template dyn_wave<Row16BC>(wcx, I.j, xin, ut, w, xout);      // (:29)
#pragma unroll
        for (int i = 0; i < m; ++i) GS(wU + d * sU, i, ut[i]);
#pragma unroll
        for (int i = 0; i < n; ++i) GS(wX + (d + 1) * sX, i, xout[i]);
    };
    // operands are fetched TWO steps ahead (three rotating register sets): they come from HBM / L2 here, and at one wave per
    // SIMD nothing else hides their latency
    Ops A, B, C;
    double xo[n];
    if (N > 0) fetch(A, 0);
    if (N > 1) fetch(B, 1);
    int t = 0;
    for (; t + 2 < N; t += 3) {
        fetch(C, 2);
        step(A, t, 0, xt, xo);
        if (t + 3 < N) fetch(A, 3);
        step(B, t + 1, 1, xo, xt);
        if (t + 4 < N) fetch(B, 4);
        step(C, t + 2, 2, xt, xo);
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = xo[i];
        oK += 24u * m * n; ok_ += 24u * m; oub += 24u * m; oxb += 24u * n; wU += 3u * sU; wX += 3u * sX;
        if constexpr (FOLLOW) {
            if ((t + 3) % 15 == 0) pk_sync<M>(I);       // x, u up to here are in the workspace (pk_rollout_segments counts these)
        }
    }
    if (t < N) { step(A, t, 0, xt, xo); ++t; if (t < N) step(B, t, 1, xo, xt); }
    pk_sync<M>(I);
}

// The fused B phase keeps fx, fu, gx, gu on chip. When an instance LEAVES its inner loop they are written out once, so that
// jacobian_* / gradient_* in the workspace hold the last linearisation exactly as after the other kernels (and as
// solver.problem.model / .objective do in the reference). Same functions, same inputs, no Hessian accumulation.
template <class M>
__attribute__((noinline)) __device__ void materialise_stage(LinArgs a, int fx_off, int fu_off, int gx_off, int gu_off) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS;
    double* gr = a.gr;
    const int t = a.t;
    double w[cdim<M::NW>::v];
    load_w<M::NW>(gr + a.w, t, w);
    double xt[n], ut[m];
#pragma unroll
    for (int i = 0; i < n; ++i) xt[i] = gr[a.xb + t * n + i];
#pragma unroll
    for (int i = 0; i < m; ++i) ut[i] = gr[a.ub + t * m + i];
    M::dyn_jac_mem(xt, ut, w, gr + fx_off + t * n * n, gr + fu_off + t * n * m);
    double gx[n], gu[m];
    M::cost_s_grad(xt, ut, w, gx, gu);
    if constexpr (ncs > 0) {
        if (a.constrained) {
            double ct[ncs], ir[ncs], dxx[n * n], duu[m * m], dux[m * n];
            const int off = t * ncs;
#pragma unroll
            for (int i = 0; i < ncs; ++i) al_multipliers(gr[a.rho + off + i], gr[a.act + off + i], gr[a.lam + off + i], gr[a.c + off + i], ir[i], ct[i]);
#pragma unroll
            for (int i = 0; i < n * n; ++i) dxx[i] = 0.0;
#pragma unroll
            for (int i = 0; i < m * m; ++i) duu[i] = 0.0;
#pragma unroll
            for (int i = 0; i < m * n; ++i) dux[i] = 0.0;
            M::al_s(xt, ut, w, ct, ir, gx, gu, dxx, duu, dux);       // only the gradient part is kept
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) gr[gx_off + t * n + i] = gx[i];
#pragma unroll
    for (int i = 0; i < m; ++i) gr[gu_off + t * m + i] = gu[i];
}

// 16 lanes of a row fill / copy a range of the row's instance block
template <class M>
__device__ __forceinline__ void row_fill(PInst<M>& I, bool act, int off, int len, double v) {
    if (act) for (int i = I.j; i < len; i += 16) I.g[off + i] = v;
}
template <class M>
__device__ __forceinline__ void row_copy(PInst<M>& I, bool act, int dst, int src, int len) {
    if (act) for (int i = I.j; i < len; i += 16) I.g[dst + i] = I.g[src + i];
}

}  // namespace pk

// Two-wave form (solve_kernel_packed<M, true>, 128 threads): wave 1 is a LINEARISATION SERVER. The chunk linearisations are 40 % of
// the instructions of a linearise + Riccati pass and depend on nothing the Riccati steps produce: while wave 0 takes the steps of
// chunk ch out of one LDS buffer, wave 1 linearises chunk ch - 1 into the other (one workgroup barrier per chunk). Everything else
// of the state machine stays on wave 0; wave 1 sits in a barrier loop and looks into its mailbox after every barrier of the
// workgroup (wave 0's own barriers, which a lone wave passes at once, are met by that loop). Same functions on the same inputs in
// the same order per instance: results bitwise those of the one-wave form. Used where the second buffer fits the CU's LDS at the
// batch's residency (ilqr_api.hip).
namespace pk {
// in-loop barriers of the packed rollout in the two-wave form: one per 15 steps of its three-step loop (a trial's cost is evaluated
// segment by segment behind the rollout, cost_follow below)
__device__ __forceinline__ int pk_rollout_segments(int N) { return N >= 3 ? (3 * ((N - 3) / 3 + 1)) / 15 : 0; }
template <class M>
__device__ void pk_helper(const KArgs& a) {
    constexpr int n = M::NX, m = M::NU;
    (void)n; (void)m;
    typedef PkLds<M, true> LD;
    extern __shared__ __attribute__((aligned(16))) double pk_lds[];
    const Layout& L = a.L;
    const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15, N = L.T - 1;
    const int b_row = blockIdx.x * 4 + q;
    const bool valid_row = b_row < a.B;
    double* gr = a.ws + (size_t)(valid_row ? b_row : a.B - 1) * (size_t)L.stride;
    volatile int* mb = (volatile int*)(pk_lds + LD::MBOX);
    int seen = 0, nb = 0;
    for (;;) {
        __syncthreads(); ++nb;
        const int seq = __builtin_amdgcn_readfirstlane(mb[0]);
        if (seq == seen) continue;
        if (seq < 0) return;
        if (__builtin_amdgcn_readfirstlane(mb[4]) != nb) continue;          // posted after this barrier: it is the next one's
        seen = seq;
        const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane(mb[1]);
        const int constrained = __builtin_amdgcn_readfirstlane(mb[2]), kind = __builtin_amdgcn_readfirstlane(mb[3]);
        const bool act_row = valid_row && ((mask >> q) & 1u);
        if (kind == PK_LINEARISE) {
            for (int ch = (N + 15) / 16 - 1; ch >= 0; --ch) {
                const int t0 = 16 * ch, cnt = (N - t0) < 16 ? (N - t0) : 16;
                if (act_row && j < cnt) {
                    LinArgs la{gr, L.xb, L.ub, L.w, L.gxx, L.guu, L.gux, L.rho, L.act, L.lam, L.c, t0 + j, q * LD::IB + (ch & 1) * LD::BUF, j, constrained};
                    linearise_stage<M>(la);
                }
                __syncthreads(); ++nb;
            }
        } else {
            // cost_follow: cost!(mode = current) of a line-search trial (cost_pass(I, L.x, L.u, act, act, constrained)) behind the rollout
            // that produces the trajectory — segment k of 15 timesteps once the solver wave has passed its k-th in-loop barrier, the
            // rest and the terminal timestep behind the rollout's closing barrier. Lane j takes the timesteps t with t % 16 == j in
            // ascending order, as cost_pass does: the same per-lane sums, the same reduction.
            ObjAcc Jp;
            double vp = 0.0;
            CostIn<M> in;
            const int nseg = pk_rollout_segments(N);
            for (int k = 0; k <= nseg; ++k) {                               // k == nseg: behind the rollout's closing barrier, up to the terminal timestep
                __syncthreads(); ++nb;
                const int lo = 15 * k, hi = k < nseg ? 15 * k + 15 : N + 1;
                if (act_row) {
                    for (int t = lo + ((j - lo) & 15); t < hi; t += 16) {   // this lane's timesteps of [lo, hi): one, two at most behind the last barrier
                        cost_load<M>(gr, L, L.x, L.u, constrained != 0, t, in);
                        cost_eval<M>(gr, L, in, t, true, true, constrained != 0, Jp, vp);
                    }
                }
            }
            const double J = row_sum(Jp.S()), v = row_max(vp);
            if (j == 0) { pk_lds[LD::RES + q] = J; pk_lds[LD::RES + 4 + q] = v; }
            __syncthreads(); ++nb;                                          // answers in place
        }
    }
}
}  // namespace pk

// words of KArgs::pool (ints in HBM, zeroed by the host before every launch): the device-wide queue of handed-over instances
enum { POOL_TAIL = 0, POOL_HEAD = 1, POOL_PACKS_DONE = 2, POOL_REJECTED = 3, POOL_MARKED = 4, POOL_BUSY = 5, POOL_FINISHED = 6, POOL_Q = 8 };
// control words of a one-wave-form workgroup (ints in LDS behind everything else, KArgs::pool_ctl doubles from the base)
enum { CTL_EVACUATE = 0, CTL_NEXT = 1, CTL_QUIET = 2, CTL_OWN = 8, CTL_WORDS = 16 };
#ifndef ILQR_SPEC_WORKER
#define ILQR_SPEC_WORKER 2
#endif
enum { POOL_CUS = 2048 };      // per-CU words behind the queue (KArgs::pool + POOL_Q + B): a marked straggler's CU is vacated
// (xcc, se, sh, cu) of the executing wave in 11 bits (HW_ID: cu 11:8, sh 12, se 15:13; tools/probes/probe_evict.hip: 256 distinct
// ids on an MI355X, eight one-wave-form waves under each)
__device__ __forceinline__ int cu_id() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (int)(((xcc & 15u) << 7) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u));
}

// The straggler mark's step at the head of every fourth cycle (packed_solve_body): a FUNCTION, out of line — inlined, its few
// registers and its 64-bit arithmetic changed the allocation of the rollout loop a screen further down (+3 % on an ordinary shard).
// Returns bit 0: this pack leaves; bit 1: this row's instance is marked.
__attribute__((noinline)) __device__ int pool_mark_step(int* pool, volatile int* ctl, int B, int mark, int cu_mode, int done, int rej_acc,
                                                        bool at_head, int j, int lane) {
    const int rej = rej_acc & 0x3fffffff;                   // (posted to the batch-wide sum where it grows: at the acceptance)
    bool leave = ctl[CTL_EVACUATE] != 0, mine = false;
    int* const cu_word = pool + POOL_Q + B + cu_id();
    if (cu_mode && !leave && __builtin_amdgcn_readfirstlane(__hip_atomic_load(cu_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) {
        // a straggler was marked on this CU: every pack of the CU leaves, and these workgroups keep quiet until no marked instance
        // is on its way (it has the CU — its SIMDs, LDS, instruction cache — to itself)
        if (lane == 0) { ctl[CTL_EVACUATE] = 1; ctl[CTL_QUIET] = 1; }
        leave = true;
    }
    if (mark > 0 && 2 * done < B) {
        const int sum = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pool + POOL_REJECTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const bool mk = at_head && (long long)rej * B - (long long)sum >= (long long)mark * B;
        if (mk && !(rej_acc >> 30) && j == 0) atomicAdd(pool + POOL_BUSY, 1);           // a marked instance on its way (until it is finished)
        mine = mk;
        if (__any(mk)) {
            if (lane == 0) { ctl[CTL_EVACUATE] = 1; if (cu_mode) __hip_atomic_store(cu_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            leave = true;
        }
    }
    return (leave ? 1 : 0) | (mine ? 2 : 0);
}

// solve!(solver) for the four instances of pack `pack` — src/solve.jl:1-54, 88-143 as a per-instance state machine. TWO: the
// workgroup's second wave serves linearisations and trial costs (pk_helper); PAIR: the workgroup's other wave runs a pack of its
// own. Out: per row, where the instance was handed over (0: it was not) and whether it was marked as a straggler.
template <class M, bool TWO, bool PAIR>
__device__ __forceinline__ void packed_solve_body(const KArgs& a, const int pack, const int lds0, int& resume_out, bool& marked_out) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS;
    using namespace pk;
    PInst<M> I;
    const Layout& L = a.L;
    I.L = L;
    I.nbar = 0; I.lds0 = lds0; I.paired = PAIR; I.rej_acc = 0;
    I.lane = threadIdx.x & 63; I.q = I.lane >> 4; I.j = I.lane & 15; I.beta = (I.lane >> 2) & 3; I.r = I.lane >> 4; I.c = I.lane & 3;
    const int b_row = pack * 4 + I.q, b_blk = pack * 4 + I.beta;
    I.valid_row = b_row < a.B; I.valid_blk = b_blk < a.B;
    I.g = a.ws + (size_t)(I.valid_row ? b_row : a.B - 1) * (size_t)L.stride;
    I.gb = a.ws + (size_t)(I.valid_blk ? b_blk : a.B - 1) * (size_t)L.stride;
    I.wb = (const char*)(a.ws + (size_t)(pack * 4) * (size_t)L.stride);
    I.trace = a.trace ? a.trace + (size_t)(I.valid_row ? b_row : 0) * (size_t)a.trace_cap * TRACE_W : nullptr;
    I.trace_cap = a.trace_cap; I.trace_len = 0;
    const ilqr_options& opt = a.opt;
    const bool constrained = a.constrained != 0, al_outer = constrained;
    const int T = L.T, N = T - 1, C = L.C;
    double* scal = I.g + L.scal;
    I.objective = scal[S_OBJECTIVE]; I.max_violation = scal[S_MAX_VIOLATION]; I.step_size = scal[S_STEP_SIZE];
    I.gradient_norm = scal[S_GRADIENT_NORM]; I.status = (int)scal[S_STATUS]; I.iterations = (int)scal[S_ITERATIONS];
    I.states_eq_nominal = (int)scal[S_STATES_EQ_NOMINAL];
    I.potrf_info = 0; I.rollouts = 0; I.outer = 0; I.delta = 0.0; I.delta_next = 0.0; I.obj_prev = 0.0; I.J_prev = 0.0; I.it = 0; I.trial = 1; I.needB = 0; I.leaving = 0;
    const bool live = I.valid_row;
    if (al_outer) {
        // reset!(solver.data) (src/solve.jl:93); λ ← 0, ρ ← ρ0 (:96-103)
        I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0; I.gradient_norm = 0.0;
        row_fill<M>(I, live, L.Lx, N * n, 0.0);
        row_fill<M>(I, live, L.Lu, N * m, 0.0);
        row_fill<M>(I, live, L.lam, C, 0.0);
        row_fill<M>(I, live, L.rho, C, opt.initial_constraint_penalty);
        I.outer = 1;
    }
    I.state = live ? ST_INIT : ST_DONE;
    int resume = 0, resume_it = 0;
    double resume_obj_prev = 0.0;
    // hand-over by head count (KArgs::handover_live): this wave's finished instances are added to the batch-wide counter at the
    // head of every cycle; once the survivors of the whole batch are few enough, every one of them leaves at its next resumable point
    int n_prev = __popcll(__ballot(live && I.j == 0));
    bool ho_now = false;
    int mark_cycle = 0;
    auto write_scalars = [&]() {
        scal[S_OBJECTIVE] = I.objective; scal[S_MAX_VIOLATION] = I.max_violation;
        scal[S_STEP_SIZE] = I.step_size; scal[S_GRADIENT_NORM] = I.gradient_norm;
        scal[S_STATUS] = (double)I.status; scal[S_ITERATIONS] = (double)I.iterations;
        scal[S_OUTER_ITERATIONS] = (double)(al_outer ? I.outer : 0); scal[S_POTRF_INFO] = (double)I.potrf_info;
        scal[S_ROLLOUTS] = (double)I.rollouts; scal[S_STATES_EQ_NOMINAL] = (double)I.states_eq_nominal;
        scal[S_DELTA] = I.delta;
        scal[S_TRACE_LEN] = (double)(I.trace_len < I.trace_cap ? I.trace_len : I.trace_cap);
        scal[S_RESUME] = (double)resume; scal[S_INNER_IT] = (double)resume_it; scal[S_OBJ_PREV] = resume_obj_prev;
        scal[S_DELTA_NEXT] = I.delta_next;
    };
    pk_sync<M>(I);
    const int outer_max = al_outer ? opt.max_dual_updates : 1;
    if (outer_max < 1) I.state = ST_DONE;

    // Phase-timing hook (tools/packed_phases.py; ILQR_PK_DEBUG in the environment of the library): bit 0 = run exactly
    // max_iterations cycles in which every instance takes every phase and no trial is accepted (fixed, data-independent
    // work); bits 1..5 = leave out the delta sweep / linearisation / Riccati pass / rollout / cost pass. Never set in product use.
    const int dbg = a.stage;
    int dbg_cycles = 0;
    for (;;) {
        if (dbg & 1) {
            if (dbg_cycles++ >= opt.max_iterations) break;
            if (live && dbg_cycles > 1) { I.state = ST_FORWARD; I.trial = 1; I.needB = 0; I.it = 1; }
        }
        if (a.handover_live > 0 && !ho_now && al_outer) {
            const int n_now = __popcll(__ballot(I.state != ST_DONE && I.j == 0));
            if (n_now < n_prev && (threadIdx.x & 63) == 0) atomicAdd(a.done_counter, n_prev - n_now);
            n_prev = n_now;
            const int done = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.done_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            ho_now = a.B - done <= a.handover_live;
            if constexpr (PAIR) {
                if (a.pool != nullptr && !ho_now && (++mark_cycle & 3) == 0) {          // (every fourth cycle: two more loads from L2)
                    // STRAGGLERS LEAVE AT ONCE. An instance that will need twice the iterations of the others shows early: its
                    // line searches reject before they accept (instance 2300 of config 4's shard 2: 21 rejected trials in its first
                    // six iterations; the ordinary acrobot instance rejects none before its fourth outer iteration). The measure is
                    // the number of rejected trials in forward passes that ENDED IN AN ACCEPTANCE — a search that fails outright
                    // ends its inner solve, and the instances that do nothing else (four per shard of config 4: ten outer iterations
                    // of one failed search each) are no stragglers. A row whose count exceeds the batch's mean so far by
                    // KArgs::pool_mark, at the head of an inner iteration and while most of the batch is still running, is marked: BOTH packs of its workgroup leave at their next resumable points
                    // (as under the head-count rule) and the workgroup turns into a latency solver — two waves, LDS-resident state —
                    // for the marked instance, then for whatever the device-wide queue holds (solve_kernel_packed below). The other
                    // seven instances wait in that queue for the first workgroup whose packs are done. Which instances change
                    // kernels never shows in a result (same arithmetic), so the rule may depend on timing.
                    extern __shared__ __attribute__((aligned(16))) double pk_lds_[];
                    const int f = pool_mark_step(a.pool, (volatile int*)(pk_lds_ + a.pool_ctl), a.B, a.pool_mark, a.pool_cu, done, I.rej_acc,
                                                 I.valid_row && I.state == ST_FORWARD && I.trial == 1, I.j, I.lane);
                    if (f & 2) I.rej_acc |= 1 << 30;
                    const bool leave = (f & 1) != 0;
                    if (leave) {                // the pack's instances count as gone for the head-count rule
                        if (n_now > 0 && I.lane == 0) atomicAdd(a.done_counter, n_now);
                        n_prev = 0; ho_now = true;
                    }
                }
            }
        }
        // ------------------------------------------------ A: outer-loop transitions (src/solve.jl:105-126) and ilqr_solve! entry (:9-18)
        {
            const bool at_outer = I.state == ST_OUTER;
            if (__any(at_outer)) {
                cost_bang<M>(I, at_outer, false, true);                                   // cost!(mode = :nominal)   (:113)
                const bool conv = at_outer && (I.max_violation <= opt.constraint_tolerance);   // (:117)
                const bool upd = at_outer && !conv;
                if (upd) {                                                                // augmented_lagrangian_update! (:120-122)
                    double* g = I.g;
                    for (int i = I.j; i < C; i += 16) {
                        const int ns = N * ncs;
                        bool ineq;
                        if (i < ns) ineq = ncs > 0 ? IneqMask<M>::s(i % (ncs > 0 ? ncs : 1)) : false;
                        else ineq = IneqMask<M>::t(i - ns);
                        double lam = g[L.lam + i], rho = g[L.rho + i];
                        dual_update_row(lam, rho, g[L.c + i], ineq, opt);
                        g[L.lam + i] = lam; g[L.rho + i] = rho;
                    }
                }
                if (conv || (upd && I.outer >= outer_max)) I.state = ST_DONE;
                else if (upd) {
                    I.outer += 1;
                    // straggler hand-over: this instance's workspace block is complete at the boundary (duals and penalties
                    // updated, nominal trajectory in place); it leaves for the latency kernel's resume launch
                    if ((a.handover_outer > 1 && I.outer >= a.handover_outer) || ho_now) { I.state = ST_DONE; resume = I.outer; }
                    else I.state = ST_INIT;
                }
                pk_sync<M>(I);
            }
            const bool at_init = I.state == ST_INIT;
            if (__any(at_init)) {
                // reset!(problem.model); reset!(problem.objective)   (:9-10)
                row_fill<M>(I, at_init, L.fx, N * n * n, 0.0);
                row_fill<M>(I, at_init, L.fu, N * n * m, 0.0);
                row_fill<M>(I, at_init, L.gx, T * n, 0.0);
                row_fill<M>(I, at_init, L.gu, N * m, 0.0);
                row_fill<M>(I, at_init, L.gxx, T * n * n, 0.0);
                row_fill<M>(I, at_init, L.guu, N * m * m, 0.0);
                row_fill<M>(I, at_init, L.gux, N * m * n, 0.0);
                if (at_init && opt.reset_cache) { I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0; }
                pk_sync<M>(I);
                cost_bang<M>(I, at_init, false, constrained);                             // (:14)
                if (at_init) { I.it = 0; I.needB = 1; }
            }
        }
        if (!__any(I.state != ST_DONE)) break;
        // ------------------------------------------------ C: line-search trials of forward_pass! (src/forward_pass.jl)
        // One trial for every instance in its forward pass — and up to three more, at once, for an instance whose trial was rejected
        // and that has ILQR_PK_REJECTS rejected trials behind it already: for such an instance (the stragglers: instance 2300 of
        // config 4's shard 2 spends 682 trials on 725 iterations) a rejected trial then costs one rollout + cost pass instead of a
        // whole cycle, at the price of one such pass for the wave's other instances (which is why the ordinary instance, a dozen
        // rejections in 350 iterations, does not get them: car:4096 was 1.5 % slower with extra trials for everybody). Every
        // instance takes the same steps in the same order either way.
        {
            const bool fw = I.state == ST_FORWARD;
            if (__any(fw)) {
                const bool first = fw && I.trial == 1;
                if (first) { I.status = 0; I.J_prev = I.objective; I.delta = 0.0; I.step_size = 1.0; }   // (:10, :13, :26)
                const bool want_delta = first && opt.line_search == 1;
                if (want_delta) I.delta = I.delta_next;                                   // (:16-20) came out of the backward pass (adjoint form)
#pragma clang loop unroll(disable)
                for (int rep = 0; rep < ILQR_PK_TRIALS; ++rep) {
                    // while step_size >= min_step_size && iteration <= 25   (:28-29)
                    const bool go = fw && !I.needB && (I.step_size >= opt.min_step_size) && (I.trial <= 25) &&
                                    (rep == 0 || I.rollouts - I.iterations >= ILQR_PK_REJECTS);
                    if (!__any(go)) break;
                    bool followed = false;
                    if constexpr (TWO) {
                        if (!(dbg & 48)) {
                            // the helper wave evaluates cost!(mode = current) (:36) segment by segment behind the rollout (:34)
                            followed = true;
                            pk_post<M>(I, PkLds<M, true>::MBOX, PK_COST, row_mask(go), constrained);
                            pk_sync<M>(I);                                              // the order is taken here
                            rollout<M, true>(I, go, I.step_size);
                            pk_sync<M>(I);                                              // the answers are in place
                            extern __shared__ __attribute__((aligned(16))) double pk_lds[];
                            const double Jh = pk_lds[PkLds<M, true>::RES + I.q], vh = pk_lds[PkLds<M, true>::RES + 4 + I.q];
                            if (go) { I.objective = Jh; if (constrained) I.max_violation = vh; }
                        }
                    }
                    if (!followed && !(dbg & 16)) {                                         // (:34)
                        rollout<M>(I, go, I.step_size);
                    }
                    if (go) { I.rollouts += 1; I.states_eq_nominal = 0; }
                    if (!followed && !(dbg & 32)) cost_bang<M>(I, go, true, constrained);   // (:36)
                    const bool acc = go && !(dbg & 1) && (I.objective <= I.J_prev + 1.0e-4 * I.step_size * I.delta);   // (:44) NaN ⇒ reject
                    if (__any(acc)) {                                                     // update_nominal_trajectory!
                        row_copy<M>(I, acc, L.xb, L.x, T * n);
                        row_copy<M>(I, acc, L.ub, L.u, N * m);
                        pk_sync<M>(I);
                    }
                    if (acc) { I.states_eq_nominal = 1; I.status = 1; I.needB = 1; }
                    if constexpr (PAIR) {
                        if (acc && I.trial > 1) {
                            I.rej_acc += I.trial - 1;
                            if (a.pool != nullptr && I.j == 0) atomicAdd(a.pool + POOL_REJECTED, I.trial - 1);
                        }
                    }
                    if (go && !acc) { I.step_size *= 0.5; I.trial += 1; }                 // (:51)
                }
                // the loop ends without acceptance: the forward pass is over with status = false
                if (fw && !I.needB && !((I.step_size >= opt.min_step_size) && (I.trial <= 25))) I.needB = 1;
            }
        }
        // ------------------------------------------------ B: gradients! + backward_pass! + lagrangian_gradient!, then src/solve.jl:36-51
        {
            const bool nb = I.needB != 0 || ((dbg & 1) && live);
            const bool lin = nb && (I.it == 0 || opt.line_search != 0);                   // (:16-18), (:27-33)
            const unsigned bmask = row_mask(lin);
            if (bmask) {
                double gn = 0.0, dn = 0.0; int info = 0;
                if (!(dbg & 8)) linearise_riccati<M, TWO>(I, lin, bmask, constrained, gn, info, dn);
                if (lin) I.delta_next = dn;
                if (lin) {
                    I.gradient_norm = gn;
                    if (info != 0 && I.potrf_info == 0) I.potrf_info = info;
                }
            }
            if (nb && !(dbg & 1)) {
                I.needB = 0;
                bool end_inner = false;
                if (I.it == 0) {
                    I.obj_prev = I.objective;                                             // (:21)
                    if (opt.max_iterations < 1) end_inner = true;
                } else {
                    I.iterations += 1;                                                    // (:39)
                    if (I.trace != nullptr && I.trace_len < I.trace_cap && I.j == 0) {    // verbose record (:40-45)
                        double* rw = I.trace + (size_t)I.trace_len * TRACE_W;
                        rw[0] = (double)(al_outer ? I.outer : 0); rw[1] = (double)I.it; rw[2] = I.objective; rw[3] = I.gradient_norm;
                        rw[4] = I.max_violation; rw[5] = I.step_size; rw[6] = (double)I.status; rw[7] = (double)I.rollouts;
                    }
                    I.trace_len += 1;
                    if (I.gradient_norm < opt.lagrangian_gradient_tolerance) end_inner = true;               // (:48)
                    else if (fabs(I.objective - I.obj_prev) < opt.objective_tolerance) end_inner = true;     // (:49)
                    else {
                        I.obj_prev = I.objective;
                        if (!I.status) end_inner = true;                                  // (:50)
                        else if (I.it >= opt.max_iterations) end_inner = true;            // (:22)
                    }
                }
                if (end_inner) I.state = al_outer ? ST_OUTER : ST_DONE;
                else { I.it += 1; I.trial = 1; I.state = ST_FORWARD; }
                I.leaving = end_inner ? 1 : 0;
                if (!end_inner && al_outer && ho_now) {
                    // hand-over inside the inner solve, at the head of iteration I.it: the block holds the nominal trajectory, K, k,
                    // the accumulated Hessians and (written out below like at the end of an inner solve) this linearisation
                    resume = I.outer; resume_it = I.it; resume_obj_prev = I.obj_prev;
                    I.state = ST_DONE; I.leaving = 1;
                }
            }
            if (__any(I.leaving != 0)) {       // write the last linearisation of the inner solve out (terminal gx[N] is already there)
                if (I.leaving) {
                    for (int t = I.j; t < N; t += 16) {
                        LinArgs la{I.g, L.xb, L.ub, L.w, L.gxx, L.guu, L.gux, L.rho, L.act, L.lam, L.c, t, 0, I.j, constrained ? 1 : 0};
                        materialise_stage<M>(la, L.fx, L.fu, L.gx, L.gu);
                    }
                }
                I.leaving = 0;
                pk_sync<M>(I);
            }
        }
    }
    if (a.handover_live > 0 && !ho_now && al_outer && n_prev > 0 && (threadIdx.x & 63) == 0) atomicAdd(a.done_counter, n_prev);
    if constexpr (TWO) {                    // the helper wave's leave
        extern __shared__ __attribute__((aligned(16))) double pk_lds[];
        if (I.lane == 0) ((volatile int*)(pk_lds + PkLds<M, true>::MBOX))[0] = -1;
    }
    pk_sync<M>(I);
    if (live && I.j == 0) write_scalars();
    resume_out = live ? resume : 0; marked_out = (I.rej_acc >> 30) != 0;
}

// The workgroup as a worker of the pool, once its two packs are through. A FUNCTION, and one that reads the launch's arguments
// from the kernel-argument segment itself: inlined behind the state machine, the latency solver's needs (all of KArgs, to the end
// of the kernel) changed that one's scalar register allocation — 28 kernel-argument reloads inside its cycle, three in the rollout
// loop, +5 % on a shard; as a function with a KArgs reference it would read them through vector loads.
template <class M>
__device__ __forceinline__ void pool_worker(const unsigned long long kernarg, const int pack, const int resume, const int marked) {
    // the kernel's one parameter lies at offset 0 of its argument segment; the address comes in a vector register pair (a function
    // argument) and goes back to scalar ones, as a pointer to constant memory: scalar loads
    typedef const __attribute__((address_space(4))) KArgs* kargs_p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)kernarg), hi = __builtin_amdgcn_readfirstlane((unsigned)(kernarg >> 32));
    const KArgs a = *(const KArgs*)(kargs_p)(((unsigned long long)hi << 32) | lo);     // (a copy: loaded once, like a kernel's own arguments)
    extern __shared__ __attribute__((aligned(16))) double pk_lds_[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, packs = (a.B + 3) / 4;
    volatile int* ctl = (volatile int*)(pk_lds_ + a.pool_ctl);
    // this wave's leavers: the block is complete in HBM (release), then the marked ones stay with the workgroup and the others
    // go to the queue (entry = instance + 1; 0 = not written yet)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... while a marked instance is still on its way: the launch will last as long as that one does, and the others must not wait
    // behind it. Otherwise they are left to the resume launch — the latency solver in a kernel of its own is some 10 % faster than
    // here behind the state machine (register allocation), and with no straggler about the second launch starts when the packs end.
    if ((lane & 15) == 0 && marked) atomicAdd(a.pool + POOL_MARKED, 1);
    if ((lane & 15) == 0 && marked && resume == 0) atomicSub(a.pool + POOL_BUSY, 1);      // marked, and finished where it was
    if ((lane & 15) == 0 && resume > 0) {
        const int b = pack * 4 + (lane >> 4);
        if (marked) ctl[CTL_OWN + wave * 4 + (lane >> 4)] = b;
        else if (__hip_atomic_load(a.pool + POOL_BUSY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0 ||
                 __hip_atomic_load(a.pool + POOL_TAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >
                 __hip_atomic_load(a.pool + POOL_FINISHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {      // (or queued work is still being done)
            const int i = atomicAdd(a.pool + POOL_TAIL, 1);
            __hip_atomic_store(a.pool + POOL_Q + i, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) { const int mine = packs - blockIdx.x * 2; atomicAdd(a.pool + POOL_PACKS_DONE, mine < 2 ? mine : 2); }
    // a worker's instructions go first where it shares a SIMD with a pack's wave (a marked straggler runs beside six of them, and
    // the launch lasts as long as it does)
    __builtin_amdgcn_s_setprio(3);
    bool own = false, queued = false, quiet = false;
    if (threadIdx.x == 0) {        // vacated for another workgroup's straggler: nothing to do here until the packs of the launch are through
        quiet = ctl[CTL_QUIET] != 0;
        for (int i = 0; i < 8; ++i) if (ctl[CTL_OWN + i] >= 0) quiet = false;
    }
    for (;;) {
        if (threadIdx.x == 0) {
            int b = -1;
            if (own) {                                                                     // the marked instance of the last round is finished
                atomicSub(a.pool + POOL_BUSY, 1); own = false;
                if (a.pool_cu) __hip_atomic_store(a.pool + POOL_Q + a.B + cu_id(), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the CU is everybody's again
            }
            else if (queued) atomicAdd(a.pool + POOL_FINISHED, 1);
            queued = false;
            for (int i = 0; i < 8; ++i) if (ctl[CTL_OWN + i] >= 0) { b = ctl[CTL_OWN + i]; ctl[CTL_OWN + i] = -1; own = true; break; }
            while (b == -1) {
                const int done = __hip_atomic_load(a.pool + POOL_PACKS_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // BEFORE the tail
                if (quiet && done < packs && __hip_atomic_load(a.pool + POOL_BUSY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) { __builtin_amdgcn_s_sleep(127); continue; }
                const int head = __hip_atomic_load(a.pool + POOL_HEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int tail = __hip_atomic_load(a.pool + POOL_TAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (head < tail) {
                    if (atomicCAS(a.pool + POOL_HEAD, head, head + 1) == head) {
                        int e;
                        while ((e = __hip_atomic_load(a.pool + POOL_Q + head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(8);
                        b = e - 1; queued = true;
                    }
                } else if (done >= packs || !a.pool_cu || __hip_atomic_load(a.pool + POOL_BUSY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) b = -2;
                else __builtin_amdgcn_s_sleep(127);        // (no marked instance on its way: nothing more will be queued, see above)
            }
            ctl[CTL_NEXT] = b;
        }
        __syncthreads();
        const int b = ctl[CTL_NEXT];
        __syncthreads();
        if (b < 0) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        solve_instance<M, true, ILQR_SPEC_WORKER>(a, pk_lds_, b);   // (line search in rounds of four trials: forward_pass<M, SPEC>)
        __syncthreads();
    }
}

// Four instances per wave. TWO: one pack per workgroup, the second wave its helper. Otherwise (the one-wave form) TWO PACKS per
// workgroup, one per wave, each on its own — and once both are through (every instance finished or handed over), the workgroup is
// a WORKER of the device-wide pool: it takes handed-over instances from the queue in HBM and finishes each with the latency
// kernel's code (solve_instance: two waves, LDS-resident state), until every pack of the launch is through and the queue is empty.
// So a batch that fills the chip to the last register (8192 instances: 2048 waves of 256 VGPRs) still has somewhere for a
// straggler to go — its own workgroup, the moment it is marked (packed_solve above) — and the survivors of the head-count rule
// are finished without a second launch (a pending launch on another stream is not given the slots that free up: measured,
// tools/probes/probe_evict.hip). The resume launch that follows on the stream finds nothing left to do; it stays as the net.
template <class M, bool TWO = false>
__global__ __launch_bounds__(128, 2) void solve_kernel_packed(KArgs a) {
    using namespace pk;
    extern __shared__ __attribute__((aligned(16))) double pk_lds_[];
    int resume = 0;
    bool marked = false;
    if constexpr (TWO) {
        if (threadIdx.x < 8) ((volatile int*)(pk_lds_ + PkLds<M, true>::MBOX))[threadIdx.x] = 0;
        // which of the two waves solves and which serves: one solver wave per SIMD (KArgs::cu_slots, pick_roles — the same geometry
        // as the latency kernel: four two-wave workgroups per CU, and the SIMD's arbiter serves its oldest wave first)
        __shared__ int role_sm[4];
        int swap_roles = 0;
        if (a.cu_slots != nullptr) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            if ((threadIdx.x & 63) == 0) role_sm[threadIdx.x >> 6] = (int)((hwid >> 4) & 3u);
            __syncthreads();
            if (threadIdx.x == 0) {
                const int cu = (int)(((xcc & 7u) << 8) | (((hwid >> 13) & 7u) << 5) | (((hwid >> 12) & 1u) << 4) | ((hwid >> 8) & 15u));
                role_sm[2] = pick_roles(a.cu_slots + CU_SLOT_INTS * cu, role_sm[0], role_sm[1], a.cu_expect);
            }
            __syncthreads();
            swap_roles = role_sm[2];
        } else __syncthreads();
#ifndef ILQR_PK_HELPER_PRIO
#define ILQR_PK_HELPER_PRIO 0
#endif
#ifndef ILQR_PK_SOLVER_PRIO
#define ILQR_PK_SOLVER_PRIO 0
#endif
        if ((int)(threadIdx.x >> 6) != swap_roles) { __builtin_amdgcn_s_setprio(ILQR_PK_HELPER_PRIO); pk_helper<M>(a); return; }
        __builtin_amdgcn_s_setprio(ILQR_PK_SOLVER_PRIO);
        packed_solve_body<M, true, false>(a, blockIdx.x, 0, resume, marked);
    } else {
        constexpr int PACK = (PkLds<M, false>::total + 1) & ~1;
        // (wave index through readfirstlane: the pack's base addresses stay in scalar registers)
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), packs = (a.B + 3) / 4, pack = blockIdx.x * 2 + wave;
        volatile int* ctl = (volatile int*)(pk_lds_ + a.pool_ctl);
        if (threadIdx.x < CTL_WORDS) ctl[threadIdx.x] = threadIdx.x >= CTL_OWN ? -1 : 0;
        __syncthreads();
        if (pack < packs) packed_solve_body<M, false, true>(a, pack, wave * PACK, resume, marked);
        if (a.pool == nullptr) return;
        pool_worker<M>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr(), pack, resume, marked ? 1 : 0);
    }
}

}  // namespace ilqr
