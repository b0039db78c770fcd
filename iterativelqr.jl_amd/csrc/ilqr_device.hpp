// Batched iLQR / augmented-Lagrangian solve for gfx950 (MI355X), hand-written HIP.
//
// Mapping: ONE WORKGROUP PER PROBLEM INSTANCE — two wavefronts in the latency kernel (see waves_of<M>: the
// second wave takes the vector chain of the Riccati recursion, the sensitivity sweep and half of the
// linearisation), one wavefront in the throughput kernel.
//   * the per-instance working set (trajectories, Jacobians, gradients, gains)
//     lives in LDS for the whole solve; HBM sees one load and one store of it;
//   * time-parallel stages (cost, linearisation) put one timestep on each lane;
//   * time-sequential stages: the Riccati recursion runs on the fp64 matrix cores
//     (v_mfma_f64_4x4x4, every 4x4 operand in the MFMA lane layout), the closed-loop
//     rollout wave-cooperatively (one fused sincos per step for all trig arguments);
//   * the accumulated cost Hessians (reference quirk Q1) stay in HBM/L2 and are
//     prefetched one step ahead by the backward pass.
// All control flow of the reference's solve loops is replicated per instance on
// the device, so a launch never synchronises with the host.
//
// Reference functions reproduced here (paths relative to /root/reference):
//   cost!/cost (AL)           src/data/methods.jl:13-30, src/augmented_lagrangian.jl:39-85,
//                             src/data/constraints.jl:23-46, src/costs.jl:48-55
//   gradients!                src/gradients.jl:1-98, src/dynamics.jl:41-50, src/costs.jl:57-84,
//                             src/constraints.jl:75-87
//   backward_pass!            src/backward_pass.jl:1-91
//   lagrangian_gradient!      src/solve.jl:67-83
//   forward_pass!, rollout!   src/forward_pass.jl:1-56, src/rollout.jl:1-31,
//                             src/data/methods.jl:32-54
//   ilqr_solve!, constrained_ilqr_solve!, augmented_lagrangian_update!
//                             src/solve.jl:1-54,88-129, src/augmented_lagrangian.jl:87-110
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/ilqr_hip.h"
#include "ilqr_layout.hpp"
#include "ilqr_math.hpp"
#include "ilqr_ric_schedule.hpp"

namespace ilqr {

struct KArgs {
    double* ws;          // HBM workspace, B instance blocks
    Layout L;
    int B;
    int constrained;
    int stage;           // stage kernel only
    ilqr_options opt;
    const double* x1;    // init kernel only (device pointers)
    const double* u_in;
    double* trace;       // optional per-iteration trace [B][trace_cap][TRACE_W] (null = off)
    int trace_cap;
    double* qv;          // optional action-value buffers [B][QL.stride] (backward-pass stage kernel only; null = off)
    QLayout QL;
    double stage_param;  // stage kernel only: step size of ILQR_STAGE_SS_TRIAL / SS_FINISH
    int stage_flag;      //                    first trial / accepted
    // Straggler hand-over (small models, batches on the packed kernel): an instance that enters outer iteration
    // handover_outer (> 1; 0 = never) leaves the packed kernel at that boundary; a second launch of the latency kernel with
    // resume = 1 picks up exactly the instances so marked (S_RESUME) and finishes them with two waves and LDS-resident state each.
    int handover_outer;
    int resume;
    // ... and by head count: once no more than handover_live instances of the batch are still running (done_counter, zeroed by
    // the host before the launch, counts the finished ones), every survivor leaves at its next resumable point — the start of
    // an outer iteration, or the start of inner iteration S_INNER_IT of one (S_OBJ_PREV carries the loop's previous objective;
    // the linearisation, the accumulated Hessians, K and k are in the workspace block). 0 = off.
    int handover_live;
    int* done_counter;
    // One-wave form of the packed kernel: its workgroups finish handed-over instances themselves (ilqr_device_packed.hpp,
    // solve_kernel_packed). pool: queue words in HBM (null = hand-overs wait for the resume launch); pool_mark: rejected line-search
    // trials above the batch's mean at which an instance is marked a straggler and leaves at once (0 = never); pool_lds: bytes of
    // LDS solve_instance needs (the launcher drops the pool where that would cost the packed kernel residency); pool_ctl: where the
    // workgroup's control words lie in LDS (set by the launcher); pool_cu: every pack on a marked straggler's CU leaves with it.
    int* pool;
    int pool_mark, pool_lds, pool_ctl, pool_cu;
    // Latency kernel, which wave of an instance is its CRITICAL one (role 0: rollout, matrix chain): the hardware places the two waves
    // of a workgroup on two SIMDs of its own choosing, and on 3 % of the SIMDs of a full chip two critical waves end up together
    // (tools/finish_times.py). cu_slots: CU_SLOT_INTS words per CU (zeroed by the host before the launch; null = roles as launched)
    // — the workgroups of a CU write down which two SIMDs they sit on, wait (bounded) until cu_expect of them have, and all take
    // the same assignment of critical waves to SIMDs: as many distinct SIMDs as the placement allows (pick_roles).
    int* cu_slots;
    int cu_expect;         // workgroups the launch puts on a CU (<= 4)
};
enum { CU_SLOT_CUS = 8 * 8 * 2 * 16, CU_SLOT_INTS = 8 };      // xcc x se x sh x cu of HW_REG_XCC_ID / HW_REG_HW_ID; [0] arrivals, [1..4] their SIMD pairs
enum { TRACE_W = 8 };   // outer, inner, objective, gradient_norm, max_violation, step_size, status, rollouts

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __shfl_xor(v, o);
        v = (w > v || w != w) ? w : v;   // NaN-propagating like Julia's max/norm
    }
    return v;
}
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }

// out[i] = value held by lane i (i < N), made wave-uniform
template <int N>
__device__ __forceinline__ void bcast_array(double v, double (&out)[N]) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
#pragma unroll
    for (int i = 0; i < N; ++i)
        out[i] = __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
}

#ifndef ILQR_PIN_ROLLOUT_CONSTANTS
#define ILQR_PIN_ROLLOUT_CONSTANTS true     // A/B switch (false: the compiler's own handling of the constants)
#endif
template <int N_> struct cdim { static constexpr int v = N_ > 0 ? N_ : 1; };
// Which constraint rows are inequalities (indices_inequality, src/constraints.jl:54-64): a 64-bit mask per stage kind for up to 64
// rows (M::INEQ_S, M::INEQ_T), an array of such words beyond (M::INEQ_WORDS, M::INEQ_S_W[], M::INEQ_T_W[]: emitted by the generator
// and by ilqr_compile_model_rows only when a model needs them, so that every other model's header stays what it was)
template <class M, class = void>
struct IneqMask {
    static_assert(M::NCS <= 64 && M::NCT <= 64, "more than 64 constraint rows per stage need the word arrays (M::INEQ_WORDS)");
    __host__ __device__ static constexpr bool s(int i) { return (M::INEQ_S >> i) & 1ull; }
    __host__ __device__ static constexpr bool t(int i) { return (M::INEQ_T >> i) & 1ull; }
};
template <class...> struct ilqr_void { typedef void type; };
template <class M>
struct IneqMask<M, typename ilqr_void<decltype(M::INEQ_WORDS)>::type> {
    __host__ __device__ static constexpr bool s(int i) { return (M::INEQ_S_W[i >> 6] >> (i & 63)) & 1ull; }
    __host__ __device__ static constexpr bool t(int i) { return (M::INEQ_T_W[i >> 6] >> (i & 63)) & 1ull; }
};

template <class M> struct is_large { static constexpr bool value = (M::NX > 4 || M::NU > 4); };


// Throughput ("slim") variant of a small model: the Jacobians fx, fu stay in HBM/L2 instead of LDS, so the
// LDS set shrinks (acrobot T=101: 36 KB -> 20 KB, 8 instances per CU) and the kernel is compiled for two
// waves per SIMD (<= 256 registers). Two fp64-bound waves interleave almost perfectly on one SIMD
// (tools/probes/probe_clock.hip), so batches larger than the number of SIMDs run ~1.8x faster.
template <class M> struct Slim : M { static constexpr bool SLIM = true; };
template <class M, class = void> struct slim_of { static constexpr bool value = false; };
template <class M> struct slim_of<M, decltype((void)M::SLIM)> { static constexpr bool value = M::SLIM; };
// Waves per instance (= per workgroup): TWO, except in the throughput variant (there the second wave of a SIMD
// belongs to another instance). Both waves execute the whole solve with identical, wave-uniform control flow, so
// every workgroup barrier is reached by both by construction. Cheap phases with idempotent effects simply run on
// both; the linearisation and the dual update (read-modify-write) split their index range; wave 1 runs the
// sensitivity sweep while wave 0 runs the first rollout; the small-model Riccati recursion stays on wave 0 (a
// 1 k-cycle step cannot pay for a barrier), the large-model one splits its MFMA tiles. Scalars produced by one
// wave are handed to the other through LDS.
// One-wave ("mid") variant of a LARGE model whose matrices are single 16x16 tiles (nx <= 16): the same phase functions as the
// four-wave large path, with the four wave roles of a phase executed one after the other by ONE wave per instance — four times
// the instances per CU (two waves per SIMD all the same) for batches beyond what the four-wave kernel holds resident.
template <class M> struct Mid : M { static constexpr bool MID = true; };
template <class M, class = void> struct mid_of { static constexpr bool value = false; };
template <class M> struct mid_of<M, decltype((void)M::MID)> { static constexpr bool value = M::MID; };
template <class M> struct mid_ok { static constexpr bool value = is_large<M>::value && M::NX <= 16 && M::NU <= 16; };
template <class M> struct waves_of { static constexpr int value = is_large<M>::value ? (mid_of<M>::value ? 1 : LARGE_WAVES) : (slim_of<M>::value ? 1 : 2); };
// Two-wave latency kernel, models with one action: the Lagrangian gradient ∇L (src/solve.jl:67-83) is written straight to its place
// in the instance's HBM block by the short Riccati form (backward_pass_m1) and never kept current in LDS — nothing reads it back
// inside a launch (‖∇L‖∞ and ∇Lᵀ·Δz are carried in registers); the LDS copy a launch starts with (loaded with the rest of the set)
// serves the stage kernels' forward sensitivity sweep, and the write-back leaves the HBM values alone.
template <class M> struct lagrangian_home_is_hbm { static constexpr bool value = !is_large<M>::value && !slim_of<M>::value && M::NU == 1; };

// parameters of timestep t (empty when NW == 0)
template <int NW>
__device__ __forceinline__ void load_w(const double* W, int t, double (&w)[cdim<NW>::v]) {
    w[0] = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = W[t * NW + i];
}

// Optional per-phase cycle accounting (build with -DILQR_PROFILE): shader-clock
// ticks (s_memtime) spent in each phase are accumulated per instance and written
// to scalar slots S_PROF.. of the workspace (tools/phase_cycles.py prints them).
// -DILQR_PROFILE -DILQR_PROFILE_SUB instead splits ONE phase into up to six sub-phases (ILQR_SUB_MARK).
#if defined(ILQR_PROFILE) && defined(ILQR_PROFILE_SUB)
#define ILQR_PROF_BEGIN() do {} while (0)
#define ILQR_PROF_END(I, slot) do {} while (0)
#define ILQR_SUB_BEGIN() long long sub_t0_ = clock64()
#define ILQR_SUB_MARK(I, slot) do { const long long t1_ = clock64(); (I).prof[slot] += (double)(t1_ - sub_t0_); sub_t0_ = t1_; } while (0)
#if defined(ILQR_SUB_SET) && ILQR_SUB_SET == 2      // the alternative set of mark sites (work / wait of the first two windows)
#define ILQR_SUB_MARK1(I, slot) do {} while (0)
#define ILQR_SUB_MARK2(I, slot) ILQR_SUB_MARK(I, slot)
#else
#define ILQR_SUB_MARK1(I, slot) ILQR_SUB_MARK(I, slot)
#define ILQR_SUB_MARK2(I, slot) do {} while (0)
#endif
#elif defined(ILQR_PROFILE)
#define ILQR_PROF_BEGIN() const long long prof_t0_ = clock64()
#define ILQR_PROF_END(I, slot) (I).prof[slot] += (double)(clock64() - prof_t0_)
#define ILQR_SUB_BEGIN() do {} while (0)
#define ILQR_SUB_MARK(I, slot) do {} while (0)
#define ILQR_SUB_MARK1(I, slot) do {} while (0)
#define ILQR_SUB_MARK2(I, slot) do {} while (0)
#else
#define ILQR_SUB_BEGIN() do {} while (0)
#define ILQR_SUB_MARK(I, slot) do {} while (0)
#define ILQR_SUB_MARK1(I, slot) do {} while (0)
#define ILQR_SUB_MARK2(I, slot) do {} while (0)
#define ILQR_PROF_BEGIN() do {} while (0)
#define ILQR_PROF_END(I, slot) do {} while (0)
#endif
// -DILQR_PROFILE -DILQR_PROFILE_BAR: time the two waves of a small-model instance spend in the chunk barriers of the Riccati
// recursion (wave 0 -> the `delta` slot, wave 1 -> LDS slot zs[7], reported in the `cost` slot INSTEAD of the cost pass)
#if defined(ILQR_PROFILE) && defined(ILQR_PROFILE_BAR)
#define ILQR_BAR_BEGIN() const long long bar_t0_ = clock64()
#define ILQR_BAR_END(I, slot) (I).prof[slot] += (double)(clock64() - bar_t0_)
#else
#define ILQR_BAR_BEGIN() do {} while (0)
#define ILQR_BAR_END(I, slot) do {} while (0)
#endif
enum { PROF_COST = 0, PROF_GRAD, PROF_BACKWARD, PROF_DELTA, PROF_ROLLOUT, PROF_OTHER, PROF_N };
// Comment markers at the head of every timestep body of the serial loops: tools/issue_model.py finds the loops in the assembly
// the build keeps (-save-temps, csrc/Makefile) and reads their per-step instruction counts off it. An empty asm statement: it
// emits no instruction (same-box A/B of the library with and without the markers: profiles/r03_ab_markers.txt).
#define ILQR_ISA_MARK(name, role) asm volatile("; ILQR_MARK " name " %0" ::"i"(role))
// a rarely taken straight-line region inside a serial loop (tools/issue_model.py leaves what lies between the two out of the step's list)
#define ILQR_ISA_COLD_BEGIN() asm volatile("; ILQR_COLD_BEGIN")
#define ILQR_ISA_COLD_END() asm volatile("; ILQR_COLD_END")

// Per-instance context. LDS pointers first, then HBM pointers, then the
// wave-uniform SolverData scalars (src/data/solver.jl:4-18) kept in registers.
template <class M>
struct Inst {
    static constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    double *xb, *ub, *x, *u, *fx, *fu, *gx, *gu, *K, *k, *Lx, *Lu, *c, *lam, *rho, *act;
    const double* w;       // parameters θ_t (problem.parameters, src/data/problem.jl:25-30), T x NW
    double *gxx, *guu, *gux, *P, *p, *scal;
    double *gbase;         // HBM: this instance's workspace block
    int fv_off, hc_off;    // large path: offsets of the compact Jacobian / Hessian rows inside the block
    int hLx, hLu;          // offsets of the Lagrangian gradient inside the block (see lagrangian_home_is_hbm)
    double *zs;            // LDS: zs[0] == 0.0 always, zs[1] is a write-only trash slot
    double *ring;          // LDS: Riccati hand-over ring between the two waves (small path)
    double *lds;           // large path: LDS staging area (the workspace itself stays in HBM)
    double *trace;         // per-iteration trace rows of this instance (what `verbose` prints, src/solve.jl:40-45)
    int trace_cap, trace_len;
    double *Q;             // HBM: this instance's block of the optional Qx, Qu, Qxx, Quu, Qux buffers (null = not stored)
    QLayout QL;
    double delta;          // delta_grad_product of the last forward_pass! (src/forward_pass.jl:20)
    double delta_next;     // the same product for the NEXT forward pass, when the backward pass produced it (adjoint form, two-wave kernel)
    int delta_next_ok;
    const double* gzero;   // HBM: a 0.0
    int T, N, C, lane, wave;
    double objective, max_violation, step_size, gradient_norm;
    int status, iterations, outer_iterations, potrf_info, rollouts, states_eq_nominal;
    int cost_par;          // two-wave kernel: which pair of LDS slots carries the next cost pass's results from wave 0 to wave 1
#ifdef ILQR_PROFILE
    double prof[PROF_N];
#endif
};

// large-model path (ilqr_device_large.hpp)
template <class M> __device__ void gradients_large(Inst<M>& I, bool constrained);
template <class M> __device__ void cost_pass_large(Inst<M>& I, bool at_states, bool upd_J, bool upd_viol, bool constrained, double& J_out, double& viol_out);
template <class M> __device__ void reset_model_objective_large(Inst<M>& I, bool literal);
template <class M, bool STORE_VALUE> __device__ void backward_pass_large(Inst<M>& I);
template <class M> __device__ void rollout_large(Inst<M>& I, double alpha, bool want_delta, double& delta_out);

// ------------------------------------------------------------------ the objective, in ONE arithmetic for every small-model kernel
// J (src/augmented_lagrangian.jl:39-66, src/costs.jl:48-55) feeds the Armijo test (src/forward_pass.jl:44) and the objective
// tolerance (src/solve.jl:49) and is reported (solver.data.objective, the trace), and an instance may change kernels in the middle
// of a solve (straggler hand-over, §3.2 of DESIGN.md): every kernel family therefore forms it in the same order, bit for bit —
//   v_t  = (l_t + λ_tᵀ c_t) + ½ Σ_{a_i = 1} ρ_i c_i²     one timestep's term (objective_term: explicit fma chains, no contraction;
//                                                         the generated M::cost_*, M::con_* carry `fp contract(off)` themselves)
//   A_l  = v_l + v_{l+64} + v_{l+128} + …                 ascending, l = 0 … 63 (what lane l of a 64-lane cost pass sums)
//   S_j  = (A_j + A_{j+32}) + (A_{j+16} + A_{j+48})       j = 0 … 15 (the first two butterfly steps of wave_sum)
//   J    = butterfly over the 16 S_j (xor 8, 4, 2, 1: wave_sum's last four steps = the packed kernel's row_sum)
// The 64-lane kernels get this from wave_sum as it stands; the packed kernel (16 lanes per instance, lane j walking t = j, j + 16,
// …) keeps the four A of its lane apart (ObjAcc) and combines them at the end.
template <class M, bool STAGE, int NC>
__device__ __forceinline__ double objective_term(double l, const double (&cv)[NC], const double* lam, const double* rho, double* act, bool with_al) {
#pragma clang fp contract(off)
    if (!with_al) return l;
    double dot = 0.0, pen = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const double lm = lam[i];
        const bool ineq = STAGE ? IneqMask<M>::s(i) : IneqMask<M>::t(i);
        const bool inactive = ineq && cv[i] < 0.0 && lm == 0.0;                 // active_set!, src/augmented_lagrangian.jl:77-82
        act[i] = inactive ? 0.0 : 1.0;
        dot = __builtin_fma(lm, cv[i], dot);
        const double c2 = cv[i] * cv[i], hr = 0.5 * rho[i];
        if (!inactive) pen = __builtin_fma(hr, c2, pen);
    }
    return (l + dot) + pen;
}
// Iρ = ρ∘a and c̃ = λ + Iρ c of the Gauss-Newton AL terms (src/gradients.jl:56-62), one row: c̃ as ONE fma in every kernel
__device__ __forceinline__ void al_multipliers(double rho, double act, double lam, double c, double& ir, double& ct) {
#pragma clang fp contract(off)
    ir = rho * act;
    ct = __builtin_fma(ir, c, lam);
}
// the four residue sums A_j, A_{j+16}, A_{j+32}, A_{j+48} of a lane that walks t = j (mod 16): timestep t adds to A[(t >> 4) & 3].
// Adding +0.0 to the others is exact (a sum that starts at +0.0 never becomes -0.0), so the select is on the addend.
struct ObjAcc {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    __device__ __forceinline__ void add(int t, double v) {
        const int q = (t >> 4) & 3;
        a0 += q == 0 ? v : 0.0; a1 += q == 1 ? v : 0.0; a2 += q == 2 ? v : 0.0; a3 += q == 3 ? v : 0.0;
    }
    __device__ __forceinline__ double S() const { return (a0 + a2) + (a1 + a3); }
};

// ------------------------------------------------------------------ cost!
// One timestep per lane. upd_J: evaluate J and the active set at (X,U)
// (src/augmented_lagrangian.jl:39-85); upd_viol: overwrite the violations
// buffer and compute max_violation at (X,U) (src/data/constraints.jl:23-46).
template <class M>
__device__ void cost_pass(Inst<M>& I, const double* X, const double* U, bool upd_J, bool upd_viol,
                          bool constrained, double& J_out, double& viol_out) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    ILQR_PROF_BEGIN();
    double Jp = 0.0, vp = 0.0;
    // Two waves per instance: the pass is wave 0's; wave 1 waits at the barrier and takes the two results from LDS (it used to run
    // the same pass — idempotent, but its issue slots are another instance's critical wave's on the SIMD they share)
    constexpr bool PARK = waves_of<M>::value == 2;
    const int t_first = (PARK && I.wave != 0) ? I.T : I.lane;
    for (int t = t_first; t < I.T; t += 64) {
        double w[cdim<M::NW>::v];
        load_w<M::NW>(I.w, t, w);
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = X[t * n + i];
        if (t < I.N) {
            double ut[m];
#pragma unroll
            for (int i = 0; i < m; ++i) ut[i] = U[t * m + i];
            double l_ = 0.0;
            if (upd_J) { l_ = M::cost_s(xt, ut, w); ILQR_OPAQUE(l_); }
            if constexpr (ncs == 0) { if (upd_J) Jp += l_; }
            if constexpr (ncs > 0) {
                if (!constrained) { if (upd_J) Jp += l_; }
                if (constrained) {
                    double cv[ncs];
                    M::con_s(xt, ut, w, cv);
                    const int off = t * ncs;
                    if (upd_J) { double v_ = objective_term<M, true, ncs>(l_, cv, I.lam + off, I.rho + off, I.act + off, true); ILQR_OPAQUE(v_); Jp += v_; }
                    if (upd_viol) {
#pragma unroll
                        for (int i = 0; i < ncs; ++i) {
                            I.c[off + i] = cv[i];
                            const bool ineq = IneqMask<M>::s(i);
                            vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                        }
                    }
                }
            }
        } else {
            double l_ = 0.0;
            if (upd_J) { l_ = M::cost_t(xt, w); ILQR_OPAQUE(l_); }
            if constexpr (nct == 0) { if (upd_J) Jp += l_; }
            if constexpr (nct > 0) {
                if (!constrained) { if (upd_J) Jp += l_; }
                if (constrained) {
                    double cv[nct];
                    M::con_t(xt, w, cv);
                    const int off = I.N * ncs;
                    if (upd_J) { double v_ = objective_term<M, false, nct>(l_, cv, I.lam + off, I.rho + off, I.act + off, true); ILQR_OPAQUE(v_); Jp += v_; }
                    if (upd_viol) {
#pragma unroll
                        for (int i = 0; i < nct; ++i) {
                            I.c[off + i] = cv[i];
                            const bool ineq = IneqMask<M>::t(i);
                            vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                        }
                    }
                }
            }
        }
    }
    if constexpr (PARK) {
        double* hand = I.zs + 8 + 2 * I.cost_par;      // alternating pairs: wave 1 reads pair p while wave 0 may already be in the next pass
        I.cost_par ^= 1;
        if (I.wave == 0) {
            J_out = wave_sum(Jp);
            viol_out = wave_max(vp);
            if (I.lane == 0) { hand[0] = J_out; hand[1] = viol_out; }
        }
        __syncthreads();
        if (I.wave != 0) { J_out = hand[0]; viol_out = hand[1]; }
    } else {
        J_out = wave_sum(Jp);
        viol_out = wave_max(vp);
        __syncthreads();
    }
#ifndef ILQR_PROFILE_BAR
    ILQR_PROF_END(I, PROF_COST);
#endif
}

// cost!(data, problem, mode) — src/data/methods.jl:13-30.
// The violations buffer / max_violation are ALWAYS taken at problem.states
// (SURVEY Appendix A, Q2); when states == nominal bitwise one pass suffices.
template <class M>
__device__ void cost_bang(Inst<M>& I, bool mode_current, bool constrained) {
    // one or two passes through a SINGLE instantiation of cost_pass (code size)
    const bool one_pass = mode_current || I.states_eq_nominal || !constrained;
    const int npass = one_pass ? 1 : 2;
    for (int pass = 0; pass < npass; ++pass) {
        const bool at_states = mode_current || pass == 1;
        const bool upd_J = pass == 0;
        const bool upd_viol = one_pass || pass == 1;
        double J, v;
        if constexpr (is_large<M>::value) cost_pass_large<M>(I, at_states, upd_J, upd_viol, constrained, J, v);
        else cost_pass<M>(I, at_states ? I.x : I.xb, at_states ? I.u : I.ub, upd_J, upd_viol, constrained, J, v);
        if (upd_J) I.objective = J;
        if (upd_viol && constrained) I.max_violation = v;
    }
}

// -------------------------------------------------------------- gradients!
// One timestep per lane: dynamics Jacobians (`.=`), cost gradients (`.=`),
// cost Hessians (`.+=` — accumulate, Appendix A Q1) and the Gauss-Newton AL
// terms of src/gradients.jl:54-80 using the violations BUFFER (Q2).
template <class M>
__device__ void gradients_small(Inst<M>& I, bool constrained) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT, W = waves_of<M>::value;
    ILQR_PROF_BEGIN();
    for (int t = I.lane + 64 * I.wave; t < I.T; t += 64 * W) {          // Hessians accumulate: each timestep exactly once
        double w[cdim<M::NW>::v];
        load_w<M::NW>(I.w, t, w);
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = I.xb[t * n + i];
        if (t < I.N) {
            double ut[m];
#pragma unroll
            for (int i = 0; i < m; ++i) ut[i] = I.ub[t * m + i];
            {
                double fx[n * n], fu[n * m];
                M::dyn_jac(xt, ut, w, fx, fu);
#pragma unroll
                for (int i = 0; i < n * n; ++i) I.fx[t * n * n + i] = fx[i];
#pragma unroll
                for (int i = 0; i < n * m; ++i) I.fu[t * n * m + i] = fu[i];
            }
            double gx[n], gu[m];
            M::cost_s_grad(xt, ut, w, gx, gu);
            if constexpr (ncs == 0) {
                // no stage constraints: the accumulated Hessians (HBM) change only where the stage cost's Hessian is structurally
                // non-zero — read-modify-write THOSE entries (acrobot: 3 of 21 doubles per timestep) instead of the dense blocks
                M::cost_s_hess_acc(xt, ut, w, I.gxx + t * n * n, I.guu + t * m * m, I.gux + t * m * n);   // `.+=` (src/costs.jl:74-80)
#pragma unroll
                for (int i = 0; i < n; ++i) I.gx[t * n + i] = gx[i];
#pragma unroll
                for (int i = 0; i < m; ++i) I.gu[t * m + i] = gu[i];
                continue;
            }
            // `.+=` of the cost Hessian (src/costs.jl:74-80), then the Gauss-Newton AL terms (src/gradients.jl:54-80), both straight into
            // the accumulated arrays through the model's structural accumulators — the SAME two calls, in the same order, as the packed
            // kernel's linearise_stage: whichever kernel linearises an instance, the accumulated Hessians are the same bits (round 4
            // formed them here with dense loops over cx, cu in registers: equal algebra, other roundings — car_obs showed it)
            M::cost_s_hess_acc(xt, ut, w, I.gxx + t * n * n, I.guu + t * m * m, I.gux + t * m * n);
            if constexpr (ncs > 0) {
                if (constrained) {
                    double ct[ncs], ir[ncs];
                    const int off = t * ncs;
#pragma unroll
                    for (int i = 0; i < ncs; ++i) al_multipliers(I.rho[off + i], I.act[off + i], I.lam[off + i], I.c[off + i], ir[i], ct[i]);
                    M::al_s(xt, ut, w, ct, ir, gx, gu, I.gxx + t * n * n, I.guu + t * m * m, I.gux + t * m * n);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) I.gx[t * n + i] = gx[i];
#pragma unroll
            for (int i = 0; i < m; ++i) I.gu[t * m + i] = gu[i];
        } else {
            double gx[n];
            M::cost_t_grad(xt, w, gx);
            M::cost_t_hess_acc(xt, w, I.gxx + t * n * n);
            if constexpr (nct > 0) {
                if (constrained) {
                    double ct[nct], ir[nct];
                    const int off = I.N * ncs;
#pragma unroll
                    for (int i = 0; i < nct; ++i) al_multipliers(I.rho[off + i], I.act[off + i], I.lam[off + i], I.c[off + i], ir[i], ct[i]);
                    M::al_t(xt, w, ct, ir, gx, I.gxx + t * n * n);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) I.gx[t * n + i] = gx[i];
        }
    }
    __syncthreads();
    ILQR_PROF_END(I, PROF_GRAD);
}

// ------------------------------------------------ LAPACK potrf('U') / potrs('U')
// Unblocked right-looking order of dpotf2; returns LAPACK info (ignored by the
// reference, src/backward_pass.jl:69 — we only record it).
// Unblocked dpotf2 order: row j is scaled by ONE / AJJ (DSCAL), as LAPACK and OpenBLAS's potf2 do. R[j] = 1 / U(j,j)
// is handed out for solves that multiply by the inverted diagonal (what OpenBLAS's TRSM kernels do); after a
// failed factorisation (info > 0, ignored by the reference) R holds the reciprocals of whatever is on the diagonal.
template <int m>
__device__ __forceinline__ int potrf_U(double (&A)[m * m], double (&R)[m]) {
    // selects only on the common path, so that the scheduler can interleave it with independent MFMA work
    int info = 0;
#pragma unroll
    for (int j = 0; j < m; ++j) {
        double ajj = A[j * m + j];
#pragma unroll
        for (int l = 0; l < j; ++l) ajj -= A[j * m + l] * A[j * m + l];
        const bool ok = (info == 0) && (ajj > 0.0);
        const bool fail_now = (info == 0) && !(ajj > 0.0);
        info = fail_now ? j + 1 : info;
        double dq, rq;
        sqrt_rsqrt_fast(ok ? ajj : 1.0, dq, rq);                       // pivot and its reciprocal in one sequence (ilqr_math.hpp)
        const double dbad = fail_now ? ajj : A[j * m + j];             // after a failed factorisation: whatever is on the diagonal
        const double d = ok ? dq : dbad;
        A[j * m + j] = d;
        double r = rq;
        if (__builtin_expect(!ok, 0)) r = 1.0 / dbad;                  // rare (diverged instances): a real branch keeps the division off the common path
        R[j] = r;
#pragma unroll
        for (int c = j + 1; c < m; ++c) {
            double v = A[c * m + j];
#pragma unroll
            for (int l = 0; l < j; ++l) v -= A[j * m + l] * A[c * m + l];
            A[c * m + j] = ok ? v * r : A[c * m + j];
        }
    }
    return info;
}
template <int m>
__device__ __forceinline__ int potrf_U(double (&A)[m * m]) {
    double R[m];
    return potrf_U<m>(A, R);
}
// The same factorisation WITHOUT the failure handling: no select, no exec-masked division — for a matrix whose pivots all come out
// positive, operation for operation what potrf_U computes (the selects there only choose between these values and the failure
// path's). Returns whether a pivot was not positive (or NaN): the caller then repeats the step's factorisation with potrf_U behind
// one wave-uniform branch (a select on a double is two instructions; potrf_U<2> carried 14 of them and two masked regions).
template <int m>
__device__ __forceinline__ bool potrf_U_nofail(double (&A)[m * m], double (&R)[m]) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < m; ++j) {
        double ajj = A[j * m + j];
#pragma unroll
        for (int l = 0; l < j; ++l) ajj -= A[j * m + l] * A[j * m + l];
        bad |= !(ajj > 0.0);
        double dq, rq;
        sqrt_rsqrt_fast(ajj, dq, rq);
        A[j * m + j] = dq;
        R[j] = rq;
#pragma unroll
        for (int c = j + 1; c < m; ++c) {
            double v = A[c * m + j];
#pragma unroll
            for (int l = 0; l < j; ++l) v -= A[j * m + l] * A[c * m + l];
            A[c * m + j] = v * rq;
        }
    }
    return bad;
}
// potrs('U') with the inverted diagonal R
template <int m, int nrhs>
__device__ __forceinline__ void potrs_U_rdiag(const double (&U)[m * m], const double (&R)[m], double (&B)[m * nrhs]) {
#pragma unroll
    for (int c = 0; c < nrhs; ++c) {
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double v = B[c * m + i];
#pragma unroll
            for (int l = 0; l < i; ++l) v -= U[i * m + l] * B[c * m + l];
            B[c * m + i] = v * R[i];
        }
#pragma unroll
        for (int i = m - 1; i >= 0; --i) {
            double v = B[c * m + i];
#pragma unroll
            for (int l = i + 1; l < m; ++l) v -= U[l * m + i] * B[c * m + l];
            B[c * m + i] = v * R[i];
        }
    }
}
template <int m, int nrhs>
__device__ __forceinline__ void potrs_U(const double (&U)[m * m], double (&B)[m * nrhs]) {
#pragma unroll
    for (int c = 0; c < nrhs; ++c) {
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double v = B[c * m + i];
#pragma unroll
            for (int l = 0; l < i; ++l) v -= U[i * m + l] * B[c * m + l];
            B[c * m + i] = v / U[i * m + i];
        }
#pragma unroll
        for (int i = m - 1; i >= 0; --i) {
            double v = B[c * m + i];
#pragma unroll
            for (int l = i + 1; l < m; ++l) v -= U[l * m + i] * B[c * m + l];
            B[c * m + i] = v / U[i * m + i];
        }
    }
}

// ------------------------------------------------- backward_pass! on the matrix cores
// For nx <= 4, nu <= 4 the whole Riccati step runs on v_mfma_f64_4x4x4 (4 blocks).
// Measured on gfx950 (tools/probes/probe_issue.hip): one wave issues an fp64 VALU
// instruction every 5-6 clk and a 4x4x4 f64 MFMA every ~17 clk, dependent or not —
// one MFMA replaces ~16 DP FMAs plus the cross-lane traffic they would need.
// Lane layout of the instruction (decoded by the probe), block beta = (lane>>2)&3:
//     A[i][k] at lane i + 4*beta + 16*k,  B[k][j] at lane j + 4*beta + 16*k,
//     C/D[i][j] at lane j + 4*beta + 16*i.
// Hence with every (zero-padded) 4x4 matrix X held as X(r,c) on lane c + 4*beta + 16*r,
//     mfma(A<-X, B<-Y, C<-Z) = X^T * Y + Z   in the same layout,
// and D feeds the next instruction's operands without any data movement.
// Block 0 carries the recursion; block 1 is borrowed for the k column during the
// triangular solves so that K and k share the (expensive) fp64 divisions.
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {   // uniform value from one lane
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}
// Cross-block moves inside a 16-lane row. Written as volatile asm: hipcc sinks the
// update_dpp builtin into the divergent arm of a following select, where the source
// lanes are masked off and DPP then reads 0. hipcc inserts no hazard wait states around
// inline asm: the leading s_nops cover MFMA(f64 4x4x4)->VALU-read (the operand often comes
// straight out of an MFMA in VGPR form; with too few wait states the low dword arrives stale
// and k loses ~2^-24 relative accuracy), VALU->DPP and EXEC->DPP; the trailing one covers
// DPP result -> MFMA operand.
__device__ __forceinline__ double row_from_next_quad(double v) {   // lane l <- lane l+4 (row_shl:4)
    int lo = __double2loint(v), hi = __double2hiint(v), olo, ohi;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b32_dpp %0, %2 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %1, %3 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1"
                 : "=&v"(olo), "=&v"(ohi) : "v"(lo), "v"(hi));
    return __hiloint2double(ohi, olo);
}
__device__ __forceinline__ double row_from_prev_quad(double v) {   // lane l <- lane l-4 (row_shr:4)
    int lo = __double2loint(v), hi = __double2hiint(v), olo, ohi;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b32_dpp %0, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %1, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1"
                 : "=&v"(olo), "=&v"(ohi) : "v"(lo), "v"(hi));
    return __hiloint2double(ohi, olo);
}
// gfx950 v_permlane16_swap: rows of 16 lanes exchanged pairwise in one VALU instruction (no LDS crossbar trip like
// ds_bpermute). swap(v, v) = {[r0 r0 r2 r2], [r1 r1 r3 r3]}: the first result hands an odd row its lower neighbour
// (lane l <- l - 16), the second hands an even row its upper neighbour (lane l <- l + 16).
__device__ __forceinline__ double from_lane_minus16_odd_rows(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
__device__ __forceinline__ double from_lane_plus16_even_rows(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
// value of the partner row of a row PAIR (0,1), (2,3): one v_permlane16_swap per half serves odd and even rows at once
__device__ __forceinline__ double from_partner_row(double v, bool odd_row) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(odd_row ? (int)b[0] : (int)b[1], odd_row ? (int)a[0] : (int)a[1]);
}
// potrs('U') for a 2x2 factor on right-hand sides held as Y(r, c), the two rows of a set on lanes 16 apart: every lane fetches
// its partner row ONCE and runs the whole 2x2 forward / backward substitution locally — the operations of the general
// row-by-row form below, each rounded the same (y0 = b0 r0; y1 = (b1 - u01 y0) r1; x1 = y1 r1; x0 = (y0 - u01 x1) r0),
// in 6 fp64 instructions, 6 selects and one row exchange instead of four dependent exchange-and-select rounds.
__device__ __forceinline__ double solve2x2_rows(double Y, bool odd_row, double u01, double r0, double r1) {
    const double oth = from_partner_row(Y, odd_row);
    const double b0 = odd_row ? oth : Y, b1 = odd_row ? Y : oth;
    const double y0 = b0 * r0;
    const double y1 = fma(-u01, y0, b1) * r1;
    const double x1 = y1 * r1;
    const double x0 = fma(-u01, x1, y0) * r0;
    return (odd_row ? x1 : x0) * -1.0;                                  // K .*= -1, k .*= -1
}
__device__ __forceinline__ double mfma444(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// A step of the recursion ends with the MFMA that produces P (or p) and the next one starts with MFMAs that read it as an A / B
// operand. Inside one basic block hipcc inserts the wait states a dependent f64 MFMA needs; on a path that ENTERS the next step
// through a taken branch (the skipped operand prefetch of the last pair, the odd tail) it did not (ROCm 7.2, observed: the first
// MFMA of step t = 0 read a stale P whenever nothing but the branch lay between). Those rare paths carry their own wait states.
__device__ __forceinline__ void mfma_block_boundary_guard() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7");
    __builtin_amdgcn_sched_barrier(0);
}

// ROLE 0: the whole recursion on one wave (throughput variant). With two waves per instance the recursion is
// split along its data flow — the value-function MATRIX chain (W, Wu, Qxx, Qux, Quu, potrf, K, ux_tmp, P) never
// reads the VECTOR chain (Qx, Qu, k, p, ∇L), which only consumes Quu, Qux, ux_tmp and K of the same timestep:
//   ROLE 1 (wave 0) runs the matrix chain and hands {Quu, Qux, ux_tmp} over through a small LDS ring (K goes to
//          its LDS array anyway), RING_STEPS timesteps per chunk, one workgroup barrier per chunk;
//   ROLE 2 (wave 1) runs the vector chain one chunk behind (double-buffered ring), redoing the tiny Cholesky of
//          Quu so that k comes out of bit-identical arithmetic.
// Wave 0's step loses 5 of 14 MFMAs, the k/∇L stores and both DPP row moves with their hazard nops.
template <class M, bool STORE_VALUE, int ROLE>
__device__ void backward_pass_mfma(Inst<M>& I) {
    constexpr int n = M::NX, m = M::NU;
    constexpr bool MAT = ROLE != 2, VEC = ROLE != 1;     // which chain(s) this wave runs
    static_assert(n <= 4 && m <= 4, "MFMA Riccati step handles nx, nu <= 4");
    const int lane = I.lane, r = lane >> 4, c = lane & 3, blk = (lane >> 2) & 3;
    const bool vnn = r < n && c < n, vnm = r < n && c < m, vmn = r < m && c < n, vmm = r < m && c < m;
    const bool vn1 = c == 0 && r < n, vm1 = c == 0 && r < m;
    const int N = I.N;
    // Per-lane operand pointers that walk backwards in time. Lanes outside the real nx/nu
    // extent of the zero-padded 4x4 operands aim at a slot holding 0.0 with stride 0, so the
    // padding costs no selects and no divergent branches; the same trick with a trash slot
    // makes the result stores unconditional inside block 0.
    constexpr bool SL = slim_of<M>::value;   // slim: fx, fu live in HBM -> base + non-negative offset like the Hessians
    const double* pfx = vnn ? I.fx + (SL ? 0 : (N - 1) * n * n) + c * n + r : (SL ? I.gzero : I.zs);   const int sfx = vnn ? n * n : 0;
    const double* pfu = vnm ? I.fu + (SL ? 0 : (N - 1) * n * m) + c * n + r : (SL ? I.gzero : I.zs);   const int sfu = vnm ? n * m : 0;
    const double* pgx = vn1 ? I.gx + (N - 1) * n + r : I.zs;               const int sgx = vn1 ? n : 0;
    const double* pgu = vm1 ? I.gu + (N - 1) * m + r : I.zs;               const int sgu = vm1 ? m : 0;
    // HBM-resident accumulated Hessians: base + t * stride with a NON-NEGATIVE offset (walking a global
    // pointer backwards made hipcc 7.2 emit saddr-form loads whose 32-bit offset went negative ->
    // HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION)
    const double* bxx = vnn ? I.gxx + c * n + r : I.gzero;   const int sxx = vnn ? n * n : 0;
    const double* buu = vmm ? I.guu + c * m + r : I.gzero;   const int suu = vmm ? m * m : 0;
    const double* bux = vmn ? I.gux + c * m + r : I.gzero;   const int sux = vmn ? m * n : 0;
    const bool b0 = blk == 0;
    double* qK = (b0 && vmn) ? I.K + (N - 1) * m * n + c * m + r : I.zs + 1;   const int sK = (b0 && vmn) ? m * n : 0;
    double* qk = (b0 && vm1) ? I.k + (N - 1) * m + r : I.zs + 1;               const int sk = (b0 && vm1) ? m : 0;
    double* qLu = (b0 && vm1) ? I.Lu + (N - 1) * m + r : I.zs + 1;
    double* qLx = (b0 && vn1) ? I.Lx + (N - 1) * n + r : I.zs + 1;             const int sLx = (b0 && vn1) ? n : 0;
    // two-wave hand-over: element (r, c) of {Quu, Qux, ux_tmp} of ring slot s at ring[s*48 + q*16 + r*4 + c]
    double* wring = (ROLE == 1 && b0) ? I.ring + r * 4 + c : I.zs + 1;   const int swr = (ROLE == 1 && b0) ? 1 : 0;   // writer (block 0)
    const double* rring = I.ring + r * 4 + c;                                                                      // reader (every block)
    const double* pKr = vmn ? I.K + (N - 1) * m * n + c * m + r : I.zs;   const int sKr = vmn ? m * n : 0;      // ROLE 2 reads K back

    double P = (MAT && vnn) ? I.gxx[N * n * n + c * n + r] : 0.0;      // P[H] .= gxx[H]  (:39)
    double p = (VEC && vn1) ? I.gx[N * n + r] : 0.0;                   // p[H] .= gx[H]   (:40)
    if (STORE_VALUE && b0) {
        if (MAT && vnn) I.P[N * n * n + c * n + r] = P;
        if (VEC && vn1) I.p[N * n + r] = p;
    }
    double gmax = 0.0;
    bool gnan = false;      // ‖·‖∞ must propagate NaN like Julia's norm: v_max_f64 drops NaNs, so they are tracked beside it (3 instead of 6 slots per value)
    // operands of step t are fetched one step ahead (accumulated Hessians from HBM/L2, the rest from
    // LDS); the time loop is unrolled by two with ping-pong operand sets so no register copies are needed
    struct Opnd { double gxx, guu, gux, fx, fu, gx, gu; };
    auto fetch_first = [&](Opnd& o, int tp) {
        if constexpr (MAT) { o.gxx = bxx[tp * sxx]; o.guu = buu[tp * suu]; o.gux = bux[tp * sux]; }
        else { o.gxx = 0.0; o.guu = 0.0; o.gux = 0.0; }
        if constexpr (SL) { o.fx = pfx[tp * sfx]; o.fu = pfu[tp * sfu]; }
        else { o.fx = *pfx; o.fu = *pfu; }
        if constexpr (VEC) { o.gx = *pgx; o.gu = *pgu; }
        else { o.gx = 0.0; o.gu = 0.0; }
    };
    auto fetch_prev = [&](Opnd& o, int tp) {     // operands of step tp = (previously fetched step) - 1
        if constexpr (!SL) { pfx -= sfx; pfu -= sfu; }
        pgx -= sgx; pgu -= sgu;
        fetch_first(o, tp);
    };
    // potrs('U') of one right-hand-side set held as Y(r, c) (rows on lanes 16 apart), given the factor   (:70-75)
    auto solve = [&](double Y, const double (&Uc)[m * m], const double (&Ur)[m], int info) {
        if (m == 1 && info == 0) {
            // 1x1: LAPACK (OpenBLAS trsm) computes (b * (1/sqrt(q))) * (1/sqrt(q)); here b * (1/q) with the reciprocal from
            // v_rcp_f64 + two Newton steps: within 1.5 ulp of either, without the sqrt and the 11-instruction IEEE division
            // sequences (~220 clk) on the serial chain. The literal path below still runs when potrf fails.
            Y = Y * recip_fast(Uc[0]);
        } else {
#pragma unroll
            for (int i = 0; i < m; ++i) {                               // U^T y = b
#pragma unroll
                for (int l = 0; l < i; ++l) {
                    const double yl = (m == 2) ? from_lane_minus16_odd_rows(Y) : __shfl(Y, lane - 16 * (i - l));
                    const double v = Y - Uc[i * m + l] * yl;
                    Y = (r == i) ? v : Y;
                }
                const double q = Y * Ur[i];
                Y = (r == i) ? q : Y;
            }
#pragma unroll
            for (int i = m - 1; i >= 0; --i) {                          // U x = y
#pragma unroll
                for (int l = i + 1; l < m; ++l) {
                    const double xl = (m == 2) ? from_lane_plus16_even_rows(Y) : __shfl(Y, lane + 16 * (l - i));
                    const double v = Y - Uc[l * m + i] * xl;
                    Y = (r == i) ? v : Y;
                }
                const double q = Y * Ur[i];
                Y = (r == i) ? q : Y;
            }
        }
        return Y * -1.0;                                                // K .*= -1, k .*= -1
    };
    auto riccati_step = [&](const Opnd& o, int t, int slot) {
        ILQR_ISA_MARK("riccati_step", ROLE);
        const double gxx = o.gxx, guu = o.guu, gux = o.gux, fx = o.fx, fu = o.fu, gx = o.gx, gu = o.gu;
        double Qxx = 0.0, Qux, Quu, Qx = 0.0, Qu = 0.0, K, k = 0.0, uxt;
        if constexpr (MAT) {
            // W = P'^T fx (= (fx^T P')^T), Wu = P'^T fu; Qxx = (fx^T P') fx + gxx, Qux = (fu^T P') fx + gux,
            // Quu = (fu^T P') fu + guu   (:52-64)
            const double W = mfma444(P, fx, 0.0);
            const double Wu = mfma444(P, fu, 0.0);
            Qxx = mfma444(W, fx, gxx);
            Qux = mfma444(Wu, fx, gux);
            Quu = mfma444(Wu, fu, guu);
        } else {
            Quu = rring[slot * 48]; Qux = rring[slot * 48 + 16]; uxt = rring[slot * 48 + 32];
        }
        if constexpr (VEC) {
            Qx = mfma444(fx, p, gx);                                    // Qx = fx^T p' + gx, Qu = fu^T p' + gu   (:44-49)
            Qu = mfma444(fu, p, gu);
        }
        // potrf('U') of Quu on wave-uniform scalars (upper triangle only, info ignored)   (:68-69)
        double Uc[m * m];
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
            for (int i = 0; i < m; ++i) Uc[j * m + i] = (i <= j) ? lane_bcast(Quu, j + 16 * i) : 0.0;
        int info = 0;
        double Ur[m];                                                   // inverted diagonal of the factor
        if (m == 1) info = (Uc[0] > 0.0) ? 0 : 1;
        else if (__builtin_expect(__any(potrf_U_nofail<m>(Uc, Ur)), 0)) {     // a pivot failed somewhere in the wave: LAPACK's sequence with its failure handling
            ILQR_ISA_COLD_BEGIN();
#pragma unroll
            for (int j = 0; j < m; ++j)
#pragma unroll
                for (int i = 0; i < m; ++i) Uc[j * m + i] = (i <= j) ? lane_bcast(Quu, j + 16 * i) : 0.0;
            info = potrf_U<m>(Uc, Ur);
            ILQR_ISA_COLD_END();
        }
        if (m == 1 && info != 0) potrf_U<m>(Uc, Ur);
        if (MAT && info != 0 && I.potrf_info == 0) I.potrf_info = info;
        if constexpr (ROLE == 0) {
            // K (block 0: m x n) and k (block 1, column 0) through the same solve
            const double Qu_b1 = row_from_prev_quad(Qu);                // block 0 -> block 1
            const double Y = solve((blk == 1) ? Qu_b1 : Qux, Uc, Ur, info);
            K = Y;                                                      // valid in block 0
            k = row_from_next_quad(Y);                                  // block 1 -> block 0, column 0
        } else if constexpr (ROLE == 1) {
            K = solve(Qux, Uc, Ur, info);
        } else {
            k = solve(Qu, Uc, Ur, info);
            K = *pKr; pKr -= sKr;
        }
        if constexpr (MAT) {
            uxt = mfma444(Quu, K, 0.0);                                 // ux_tmp = Quu K   (:79)   (Quu^T K; Quu is symmetric up to rounding)
            // P = K^T ux_tmp + K^T Qux + Qux^T K + Qxx   (:81-84). The four terms are summed as ((Qxx + K^T Qux) + Qux^T K) + K^T ux_tmp:
            // the first two products need only K, so they issue while ux_tmp = Quu K is still in the matrix pipe, and the chain
            // behind ux_tmp is ONE dependent MFMA instead of three plus an addition (fewer hazard wait states on wave 0's stream).
            // Same terms, other association than the reference's mul!(P, ..., 1.0, 1.0) sequence: rounding-level.
            double Pn = mfma444(K, Qux, Qxx);
            Pn = mfma444(Qux, K, Pn);
            Pn = mfma444(K, uxt, Pn);
            if (b0) {
                *qK = K; qK -= sK;
                if (STORE_VALUE && vnn) I.P[t * n * n + c * n + r] = Pn;
                if (STORE_VALUE && I.Q != nullptr) {                    // policy.action_value.* (src/data/policy.jl:58-64)
                    if (vnn) I.Q[I.QL.Qxx + t * n * n + c * n + r] = Qxx;
                    if (vmn) I.Q[I.QL.Qux + t * m * n + c * m + r] = Qux;
                    if (vmm) I.Q[I.QL.Quu + t * m * m + c * m + r] = Quu;
                }
            }
            if constexpr (ROLE == 1) {                                  // hand Quu, Qux, ux_tmp to the vector chain
                wring[slot * 48 * swr] = Quu; wring[(slot * 48 + 16) * swr] = Qux; wring[(slot * 48 + 32) * swr] = uxt;
            }
            P = Pn;
        }
        if constexpr (VEC) {
            // p = ux_tmp^T k + K^T Qu + Qux^T k + Qx   (:86-89)
            // (same association as P: Qx seeds the accumulator, the ux_tmp term goes last)
            double pn = mfma444(K, Qu, Qx);
            pn = mfma444(Qux, k, pn);
            pn = mfma444(uxt, k, pn);
            // lagrangian_gradient!: Lx = Qx - p[t], Lu = Qu   (src/solve.jl:73-81). The padding of Lx and
            // Qu is exactly zero, so the running ∞-norm needs no lane predicate (block 0 is picked at the end).
            const double Lx = Qx - pn;
            gmax = fmax(gmax, fabs(Lx));
            gmax = fmax(gmax, fabs(Qu));
            gnan |= (Lx != Lx) | (Qu != Qu);
            if (b0) {
                *qk = k; *qLu = Qu; *qLx = Lx;
                qk -= sk; qLu -= sk; qLx -= sLx;
                if (STORE_VALUE && vn1) I.p[t * n + r] = pn;
                if (STORE_VALUE && I.Q != nullptr) {
                    if (vn1) I.Q[I.QL.Qx + t * n + r] = Qx;
                    if (vm1) I.Q[I.QL.Qu + t * m + r] = Qu;
                }
            }
            p = pn;
        }
    };
    Opnd A, B;
    if (N > 0) fetch_first(A, N - 1);
    int t = N - 1;
    if constexpr (ROLE == 0) {
        for (; t >= 1; t -= 2) {                                        // (:42)
            fetch_prev(B, t - 1);
            riccati_step(A, t, 0);
            if (t >= 2) fetch_prev(A, t - 2);
            else mfma_block_boundary_guard();
            riccati_step(B, t - 1, 0);
        }
        if (t == 0) { mfma_block_boundary_guard(); riccati_step(A, 0, 0); }
    } else {
        // chunks of RING_STEPS timesteps; the vector chain works one chunk behind the matrix chain
        for (int chunk = 0; t >= 0; ++chunk) {                          // (:42)
            if constexpr (ROLE == 2) __syncthreads();                   // chunk `chunk` of the ring is complete
            const int base = (chunk & 1) * RING_STEPS, cnt = t + 1 < RING_STEPS ? t + 1 : RING_STEPS;
            int i = 0;
            for (; i + 1 < cnt; i += 2, t -= 2) {
                fetch_prev(B, t - 1);
                riccati_step(A, t, base + i);
                if (t >= 2) fetch_prev(A, t - 2);
                else mfma_block_boundary_guard();
                riccati_step(B, t - 1, base + i + 1);
            }
            if (i < cnt) { mfma_block_boundary_guard(); riccati_step(A, t, base + i); t -= 1; }      // odd tail: only ever the very last step
            if constexpr (ROLE == 1) __syncthreads();                   // hand the chunk over (the other half-ring is free again)
        }
    }
    if constexpr (VEC) I.gradient_norm = wave_max((b0 && c == 0) ? (gnan ? __builtin_nan("") : gmax) : 0.0);
}

// ---- the two-wave form of the recursion (latency kernel), written for INSTRUCTION COUNT: a wave issues one instruction per
// ~5-6 clk whatever its class (tools/probes/probe_issue.hip), so the step is as long as its instruction list.
//   ROLE 1 (wave 0): matrix chain W, Wu, Qxx, Qux, Quu, potrf, K, ux_tmp, P.   ROLE 2 (wave 1): vector chain Qx, Qu, k, p, ∇L,
//   one ring chunk behind. All four MFMA blocks of a wave hold the SAME operands (the lane's address depends on (r, c) only),
//   so every result store is unconditional: the lanes of the four blocks write the same value to the same address, padding
//   lanes write to a trash slot with stride 0 — no exec mask is set up or restored anywhere in the step.
//   * LDS operands and results go through per-lane BYTE ADDRESSES held in VGPRs that walk backwards in time (one v_add each);
//     the accumulated Hessians come from the instance's HBM block through 32-bit per-lane offsets from its (scalar) base.
//   * Hand-over ring slot (48 doubles): tiles Qux and ux_tmp, then the SOLVE SCALARS wave 0 has already paid for — for nu = 1
//     the pair (a, b) with k = -((Qu a) b): (1 / Quu, 1) normally, (1 / d, 1 / d) in LAPACK's arithmetic after a failed potrf;
//     for nu >= 2 the factor's off-diagonal entries and inverted diagonal. Wave 1 reads them back as wave-uniform LDS loads
//     instead of re-factorising Quu (which it did before, from a Quu tile): same operands, same arithmetic for k.
template <class M, bool STORE_VALUE, int ROLE>
__device__ void backward_pass_split(Inst<M>& I) {
    constexpr int n = M::NX, m = M::NU;
    constexpr bool MAT = ROLE == 1, VEC = ROLE == 2;
    static_assert(ROLE == 1 || ROLE == 2, "two-wave recursion");
    static_assert(n <= 4 && m <= 4 && !slim_of<M>::value, "LDS-resident small models");
    static_assert(m * (m + 1) / 2 + 2 <= 16, "solve scalars fit their ring tile");
    typedef __attribute__((address_space(3))) double ldsd;
    const int lane = I.lane, r = lane >> 4, c = lane & 3;
    const bool vnn = r < n && c < n, vnm = r < n && c < m, vmn = r < m && c < n, vmm = r < m && c < m;
    const bool vn1 = c == 0 && r < n, vm1 = c == 0 && r < m, b0 = ((lane >> 2) & 3) == 0;
    const int N = I.N;
    auto lds = [](const double* q) -> unsigned { return (unsigned)(size_t)(const ldsd*)q; };
    auto LD = [](unsigned a, int off) -> double { return *(const ldsd*)(size_t)(a + 8u * off); };
    auto ST = [](unsigned a, int off, double v) { *(ldsd*)(size_t)(a + 8u * off) = v; };
    const unsigned zero = lds(I.zs), trash = lds(I.zs + 1);
    // operand addresses of step N-1 (stride 0 at the zero slot for padding lanes)
    unsigned afx = vnn ? lds(I.fx + (N - 1) * n * n + c * n + r) : zero;   const unsigned sfx = vnn ? 8u * n * n : 0u;
    unsigned afu = vnm ? lds(I.fu + (N - 1) * n * m + c * n + r) : zero;   const unsigned sfu = vnm ? 8u * n * m : 0u;
    unsigned agx = vn1 ? lds(I.gx + (N - 1) * n + r) : zero;               const unsigned sgx = vn1 ? 8u * n : 0u;
    unsigned agu = vm1 ? lds(I.gu + (N - 1) * m + r) : zero;               const unsigned sgu = vm1 ? 8u * m : 0u;
    unsigned aK = vmn ? lds(I.K + (N - 1) * m * n + c * m + r) : (MAT ? trash : zero);   const unsigned sK = vmn ? 8u * m * n : 0u;
    unsigned ak = vm1 ? lds(I.k + (N - 1) * m + r) : trash;                const unsigned sk = vm1 ? 8u * m : 0u;
    unsigned aLu = vm1 ? lds(I.Lu + (N - 1) * m + r) : trash;
    unsigned aLx = vn1 ? lds(I.Lx + (N - 1) * n + r) : trash;              const unsigned sLx = vn1 ? 8u * n : 0u;
    // accumulated Hessians: byte offsets into the instance's HBM block
    const char* gb = (const char*)I.gbase;
    auto goff = [&](const double* q) -> unsigned { return (unsigned)((const char*)q - gb); };
    unsigned oxx = vnn ? goff(I.gxx + (N - 1) * n * n + c * n + r) : goff(I.gzero);   const unsigned sxx = vnn ? 8u * n * n : 0u;
    unsigned ouu = vmm ? goff(I.guu + (N - 1) * m * m + c * m + r) : goff(I.gzero);   const unsigned suu = vmm ? 8u * m * m : 0u;
    unsigned oux = vmn ? goff(I.gux + (N - 1) * m * n + c * m + r) : goff(I.gzero);   const unsigned sux = vmn ? 8u * m * n : 0u;
    const unsigned aring0 = lds(I.ring + r * 4 + c);        // tile element (r, c) of slot 0
    const unsigned asc0 = lds(I.ring + 32);                 // solve scalars of slot 0 (wave-uniform address)

    double P = (MAT && vnn) ? I.gxx[N * n * n + c * n + r] : 0.0;      // P[H] .= gxx[H]  (:39)
    double p = (VEC && vn1) ? I.gx[N * n + r] : 0.0;                   // p[H] .= gx[H]   (:40)
    if (STORE_VALUE && b0) {
        if (MAT && vnn) I.P[N * n * n + c * n + r] = P;
        if (VEC && vn1) I.p[N * n + r] = p;
    }
    double gmax = 0.0, one = 1.0;
    ILQR_OPAQUE(one);                   // stays in a register pair instead of being rebuilt where the rare path joins
    // Δ = ∇Lᵀ·Δz (src/forward_pass.jl:16-20, src/data/methods.jl:42-54) as the ADJOINT of the sensitivity recursion, carried by the
    // vector chain: w = ∇L_u + fuᵀν′, Δ += wᵀk, ν = ∇L_x + fxᵀν′ + Kᵀw — four MFMAs on operands this step holds anyway, instead of
    // a forward sweep of 84 instructions per timestep on wave 1 beside the first rollout (which shared the SIMD's fp64 pipe with
    // another instance's critical wave). Same number up to rounding; it only enters the Armijo test (c1 = 1e-4).
    double nu = 0.0, dacc = 0.0;
    unsigned long long nanmask = 0;     // ‖·‖∞ must propagate NaN like Julia's norm; v_max_f64 drops NaNs, so they are tracked beside it
    struct Opnd { double gxx, guu, gux, fx, fu, gx, gu; };
    auto fetch = [&](Opnd& o) {          // operands at the walking addresses, then one step back in time
        if constexpr (MAT) {
            o.gxx = *(const double*)(gb + oxx); o.guu = *(const double*)(gb + ouu); o.gux = *(const double*)(gb + oux);
            oxx -= sxx; ouu -= suu; oux -= sux;
        } else { o.gxx = 0.0; o.guu = 0.0; o.gux = 0.0; }
        o.fx = LD(afx, 0); o.fu = LD(afu, 0);
        afx -= sfx; afu -= sfu;
        if constexpr (VEC) { o.gx = LD(agx, 0); o.gu = LD(agu, 0); agx -= sgx; agu -= sgu; }
        else { o.gx = 0.0; o.gu = 0.0; }
    };
    // potrs('U') of a right-hand-side set held as Y(r, c), given the factor's off-diagonal entries and inverted diagonal   (:70-75)
    auto solve = [&](double Y, const double (&Uc)[m * m], const double (&Ur)[m]) {
        if constexpr (m == 2) return solve2x2_rows(Y, (r & 1) != 0, Uc[2], Ur[0], Ur[1]);
#pragma unroll
        for (int i = 0; i < m; ++i) {                                   // U^T y = b
#pragma unroll
            for (int l = 0; l < i; ++l) {
                const double yl = (m == 2) ? from_lane_minus16_odd_rows(Y) : __shfl(Y, lane - 16 * (i - l));
                const double v = Y - Uc[i * m + l] * yl;
                Y = (r == i) ? v : Y;
            }
            const double q = Y * Ur[i];
            Y = (r == i) ? q : Y;
        }
#pragma unroll
        for (int i = m - 1; i >= 0; --i) {                              // U x = y
#pragma unroll
            for (int l = i + 1; l < m; ++l) {
                const double xl = (m == 2) ? from_lane_plus16_even_rows(Y) : __shfl(Y, lane + 16 * (l - i));
                const double v = Y - Uc[l * m + i] * xl;
                Y = (r == i) ? v : Y;
            }
            const double q = Y * Ur[i];
            Y = (r == i) ? q : Y;
        }
        return Y * -1.0;                                                // K .*= -1, k .*= -1
    };
    auto riccati_step = [&](const Opnd& o, int t, unsigned aslot, unsigned asc) {
        ILQR_ISA_MARK("riccati_step", ROLE);
        if constexpr (MAT) {
            // W = P'^T fx, Wu = P'^T fu; Qxx = W^T fx + gxx, Qux = Wu^T fx + gux, Quu = Wu^T fu + guu   (:52-64)
            const double W = mfma444(P, o.fx, 0.0);
            const double Wu = mfma444(P, o.fu, 0.0);
            const double Qxx = mfma444(W, o.fx, o.gxx);
            const double Qux = mfma444(Wu, o.fx, o.gux);
            const double Quu = mfma444(Wu, o.fu, o.guu);
            double K;
            if constexpr (m == 1) {
                // 1x1: LAPACK (OpenBLAS trsm) computes (b (1/sqrt q)) (1/sqrt q); here b (1/q) with the reciprocal from v_rcp_f64 + two
                // Newton steps: within 1.5 ulp of either. A failed potrf (q <= 0 or NaN; info ignored by the reference, :69) leaves q
                // on the diagonal and potrs divides by it twice: the literal arithmetic, on a branch of its own.
                const double q = lane_bcast(Quu, 0);
                double sa, sb = one;
                if (__builtin_expect(q > 0.0, 1)) {
                    sa = recip_fast(q);
                    K = (Qux * sa) * -1.0;                              // (x 1.0 is exact: the vector chain's ((Qu a) b) agrees)
                } else {
                    double Uc[1] = {q}, Ur[1];
                    const int info = potrf_U<1>(Uc, Ur);
                    if (info != 0 && I.potrf_info == 0) I.potrf_info = info;
                    sa = Ur[0]; sb = Ur[0];
                    K = ((Qux * sa) * sb) * -1.0;
                }
                ST(asc, 0, sa); ST(asc, 1, sb);
            } else {
                double Uc[m * m], Ur[m];
#pragma unroll
                for (int j = 0; j < m; ++j)
#pragma unroll
                    for (int i = 0; i < m; ++i) Uc[j * m + i] = (i <= j) ? lane_bcast(Quu, j + 16 * i) : 0.0;
                int info = 0;                                           // (:68-69) upper triangle only, info ignored
                if (__builtin_expect(__any(potrf_U_nofail<m>(Uc, Ur)), 0)) {         // a failed pivot: LAPACK's sequence with its failure handling
                    ILQR_ISA_COLD_BEGIN();
#pragma unroll
                    for (int j = 0; j < m; ++j)
#pragma unroll
                        for (int i = 0; i < m; ++i) Uc[j * m + i] = (i <= j) ? lane_bcast(Quu, j + 16 * i) : 0.0;
                    info = potrf_U<m>(Uc, Ur);
                    ILQR_ISA_COLD_END();
                }
                if (info != 0 && I.potrf_info == 0) I.potrf_info = info;
                K = solve(Qux, Uc, Ur);
                int q_ = 0;
#pragma unroll
                for (int j = 0; j < m; ++j) {
#pragma unroll
                    for (int i = 0; i < j; ++i) ST(asc, q_++, Uc[j * m + i]);
                }
#pragma unroll
                for (int i = 0; i < m; ++i) ST(asc, q_++, Ur[i]);
            }
            const double uxt = mfma444(Quu, K, 0.0);                    // ux_tmp = Quu K   (:79)
            // P = K^T ux_tmp + K^T Qux + Qux^T K + Qxx   (:81-84), summed as ((Qxx + K^T Qux) + Qux^T K) + K^T ux_tmp (see backward_pass_mfma)
            double Pn = mfma444(K, Qux, Qxx);
            Pn = mfma444(Qux, K, Pn);
            Pn = mfma444(K, uxt, Pn);
            ST(aK, 0, K); aK -= sK;
            ST(aslot, 0, Qux); ST(aslot, 16, uxt);
            if (STORE_VALUE && b0) {
                if (vnn) I.P[t * n * n + c * n + r] = Pn;
                if (I.Q != nullptr) {                                   // policy.action_value.* (src/data/policy.jl:58-64)
                    if (vnn) I.Q[I.QL.Qxx + t * n * n + c * n + r] = Qxx;
                    if (vmn) I.Q[I.QL.Qux + t * m * n + c * m + r] = Qux;
                    if (vmm) I.Q[I.QL.Quu + t * m * m + c * m + r] = Quu;
                }
            }
            P = Pn;
        } else {
            const double Qx = mfma444(o.fx, p, o.gx);                   // Qx = fx^T p' + gx, Qu = fu^T p' + gu   (:44-49)
            const double Qu = mfma444(o.fu, p, o.gu);
            const double Qux = LD(aslot, 0), uxt = LD(aslot, 16);
            const double K = LD(aK, 0); aK -= sK;
            double k;
            if constexpr (m == 1) {
                const double sa = LD(asc, 0), sb = LD(asc, 1);
                k = ((Qu * sa) * sb) * -1.0;
            } else {
                double Uc[m * m], Ur[m];
                int q_ = 0;
#pragma unroll
                for (int j = 0; j < m; ++j) {
#pragma unroll
                    for (int i = 0; i < m; ++i) Uc[j * m + i] = 0.0;
                }
#pragma unroll
                for (int j = 0; j < m; ++j) {
#pragma unroll
                    for (int i = 0; i < j; ++i) Uc[j * m + i] = LD(asc, q_++);
                }
#pragma unroll
                for (int i = 0; i < m; ++i) Ur[i] = LD(asc, q_++);
                k = solve(Qu, Uc, Ur);
            }
            // p = ux_tmp^T k + K^T Qu + Qux^T k + Qx   (:86-89), same association as P
            double pn = mfma444(K, Qu, Qx);
            pn = mfma444(Qux, k, pn);
            pn = mfma444(uxt, k, pn);
            // lagrangian_gradient!: Lx = Qx - p[t], Lu = Qu   (src/solve.jl:73-81); the padding of Lx and Qu is exactly zero
            const double Lx = Qx - pn;
            asm("v_max_f64 %0, %1, |%2|" : "=v"(gmax) : "v"(gmax), "v"(Lx));
            asm("v_max_f64 %0, %1, |%2|" : "=v"(gmax) : "v"(gmax), "v"(Qu));
            nanmask |= __builtin_amdgcn_ballot_w64((Lx != Lx) | (Qu != Qu));
            ST(ak, 0, k); ST(aLu, 0, Qu); ST(aLx, 0, Lx);
            ak -= sk; aLu -= sk; aLx -= sLx;
            const double wv = mfma444(o.fu, nu, Qu);
            dacc = mfma444(wv, k, dacc);
            const double nun = mfma444(o.fx, nu, Lx);
            nu = mfma444(K, wv, nun);
            if (STORE_VALUE && b0) {
                if (vn1) I.p[t * n + r] = pn;
                if (I.Q != nullptr) {
                    if (vn1) I.Q[I.QL.Qx + t * n + r] = Qx;
                    if (vm1) I.Q[I.QL.Qu + t * m + r] = Qu;
                }
            }
            p = pn;
        }
    };
    // operands of step t are fetched one step ahead; chunks of RING_STEPS timesteps, the vector chain one chunk behind
    Opnd A, B;
    if (N > 0) fetch(A);
    int t = N - 1;
    for (int chunk = 0; t >= 0; ++chunk) {                              // (:42)
        if constexpr (ROLE == 2) __syncthreads();                       // chunk `chunk` of the ring is complete
        const int base = (chunk & 1) * RING_STEPS, cnt = t + 1 < RING_STEPS ? t + 1 : RING_STEPS;
        unsigned aslot = aring0 + 8u * 48u * base, asc = asc0 + 8u * 48u * base;
        int i = 0;
        for (; i + 1 < cnt; i += 2, t -= 2) {
            fetch(B);
            riccati_step(A, t, aslot, asc);
            // operands of step t - 2, unconditionally: for the last pair they are reads below the arrays (inside this instance's
            // LDS set and HBM block: every array fetched here sits behind x̄, ū) whose values are never used — no branch, and no
            // step entered through one (mfma_block_boundary_guard)
            fetch(A);
            riccati_step(B, t - 1, aslot + 8u * 48u, asc + 8u * 48u);
            aslot += 16u * 48u; asc += 16u * 48u;
        }
        if (i < cnt) { mfma_block_boundary_guard(); riccati_step(A, t, aslot, asc); t -= 1; }        // odd tail: only ever the very last step
        if constexpr (ROLE == 1) __syncthreads();                       // hand the chunk over (the other half-ring is free again)
    }
    if constexpr (VEC) {
        I.gradient_norm = wave_max(nanmask != 0 ? __builtin_nan("") : gmax);
        I.delta_next = lane_bcast(dacc, 0);                             // element (0, 0)
    }
}

// ---- nu = 1 (acrobot, pendulum, particle): the two-wave recursion with the action dimension folded OUT of the matrix pipe.
// v_mfma_f64_4x4x4 accumulates the four products of an element as ONE k-ascending fma chain (tools/probes/probe_mfma_order.hip,
// profiles/r04_probe_mfma_order.txt: 128000 of 128000 elements bitwise), so rank-1 terms that the general form issues as
// dependent MFMAs with one non-zero k slice each can share an instruction — or leave the matrix pipe — without changing a bit:
//   matrix chain (wave 0), SIX MFMAs instead of nine and no lane read-out:
//     * fu enters REPLICATED over the columns of its tile, gux over the rows, guu everywhere: WuR = P'ᵀ fuR has Wu in every column,
//       QuxR = WuRᵀ fx + guxR has the row vector Qux in EVERY row, q = WuRᵀ fuR + guu has Quu on EVERY lane (the products and the
//       order of element (0,0) of the general form) — the reciprocal is formed per lane, no v_readlane, no scalar hazard states;
//     * the three rank-1 updates P = ((Qxx + Kᵀ Qux) + Quxᵀ K) + Kᵀ ux_tmp (src/backward_pass.jl:79-84) are ONE MFMA on tiles
//       stacked along k: A3 = rows [K; Qux; K; 0], B3 = rows [Qux; K; ux_tmp; 0], both QuxR scaled row-wise by per-lane
//       multipliers (K = Qux (-1/q), ux_tmp = K q: the roundings of the general form);
//     * ONE LDS store hands over everything: row 0 of B3 (Qux) and row 2 (ux_tmp) go to the ring, row 1 (K) to its array;
//     * a non-positive or NaN pivot is not looked for inside the loop (the reference ignores potrf's info, src/backward_pass.jl:69,
//       but what LAPACK leaves behind then is another arithmetic): the loop keeps min 1 / q, and a pass that ends with a negative one or
//       a NaN in P is REPEATED by the literal code (backward_pass_split) — rare (diverged instances), exact;
//   vector chain (wave 1), FOUR MFMAs instead of nine: with p, ν, gx replicated over columns and fu, gu everywhere, Qu and
//     w = ∇L_u + fuᵀν′ come out of their MFMAs on every lane, k = Qu (-1/q) is one multiply, and p = ((Qx + Kᵀ Qu) + Quxᵀ k) +
//     ux_tmpᵀ k, Δ += w k, ν = (fxᵀν′ + ∇L_x) + Kᵀ w are plain fp64 FMAs in the order of the k slices they replace.
// Addresses: every operand and result stands at the LOWEST timestep of its four-step chunk (chunks are aligned to t & ~3, the
// first one may be short) and step t uses the immediate offset (t & 3) * stride; bases move once per chunk. Padding lanes
// (nx < 4) aim at a zero REGION long enough for those offsets (LDS: in the ring area, zeroed by each wave before its pass; HBM:
// behind Layout::gzero), result lanes that store nothing at a trash region.
enum { M1_RQ = 0, M1_RU = 32, M1_RS = 64, M1_TRASH = 96, M1_ZERO = 128, M1_ZERO_N = 80 };    // doubles from Inst::ring (every row 8 slots of nx)
static_assert(M1_ZERO + M1_ZERO_N <= RING_DOUBLES, "nu = 1 ring map fits the ring");
template <class M, bool STORE_VALUE, int ROLE>
__device__ bool backward_pass_m1(Inst<M>& I) {
    constexpr int n = M::NX;
    constexpr bool MAT = ROLE == 1, VEC = ROLE == 2;
    static_assert(ROLE == 1 || ROLE == 2, "two-wave recursion");
    static_assert(M::NU == 1 && n <= 4 && !slim_of<M>::value, "LDS-resident small models with one action");
    static_assert(RING_STEPS == 4, "chunks of four steps");
    static_assert(n == 4 || (3 * n * n < GZERO_REGION && M1_RS + 3 * n < M1_ZERO_N && 3 * n < GTRASH_REGION), "zero regions cover the largest immediate offset (nx = 4 has no padding lanes)");
    typedef __attribute__((address_space(3))) double ldsd;
    const int lane = I.lane, r = lane >> 4, c = lane & 3;
    const bool vnn = r < n && c < n, vr = r < n, vc = c < n, b0 = ((lane >> 2) & 3) == 0;
    const int N = I.N;
    if (N <= 0) {
        if constexpr (VEC) { I.gradient_norm = 0.0; I.delta_next = 0.0; }
        return false;
    }
    auto lds = [](const double* q) -> unsigned { return (unsigned)(size_t)(const ldsd*)q; };
    auto LD = [](unsigned a, int off) -> double { return *(const ldsd*)(size_t)(a + 8u * off); };
    auto ST = [](unsigned a, int off, double v) { *(ldsd*)(size_t)(a + 8u * off) = v; };
    if (n < 4) {
        for (int i = lane; i < M1_ZERO_N; i += 64) I.ring[M1_ZERO + i] = 0.0;   // this wave's own reads follow its own writes (LDS serves a wave in order)
    }
    const unsigned zero = lds(I.ring + M1_ZERO), trash = lds(I.ring + M1_TRASH);
    const char* gb = (const char*)I.gbase;
    auto goff = [&](const double* q) -> unsigned { return (unsigned)((const char*)q - gb); };
    auto GL = [&](unsigned o, int off) -> double { return *(const double*)(gb + (size_t)o + (ptrdiff_t)(8 * off)); };
    const unsigned gz = goff(I.gzero + 2);                                // HBM zero region
    const int t0 = N - 1, tl = t0 & ~3;
    // Jacobians, shared by both chains: fx(r, c); fu(r) in every column
    unsigned afx = vnn ? lds(I.fx + tl * n * n + c * n + r) : zero;   const unsigned dfx = vnn ? 32u * n * n : 0u;     // per chunk
    unsigned afu = vr ? lds(I.fu + tl * n + r) : zero;                const unsigned dfu = vr ? 32u * n : 0u;
    struct Opnd { double fx, fu, c0, c1, c2; };       // c0.. : gxx, gux(c) in every row, guu (matrix chain) / gx(r) in every column, gu (vector chain)
    Opnd A, B;
    int t = t0, par = 0;

    if constexpr (MAT) {
        unsigned oxx = vnn ? goff(I.gxx + tl * n * n + c * n + r) : gz;   const unsigned dxx = vnn ? 32u * n * n : 0u;
        unsigned oux = vc ? goff(I.gux + tl * n + c) : gz;                const unsigned dux = vc ? 32u * n : 0u;
        unsigned ouu = goff(I.guu + tl);
        // the ONE result store: row 0 -> ring Qux, row 1 -> K[t], row 2 -> ring ux_tmp, row 3 -> ring 1 / q
        const bool toK = r == 1 && vc, toR = (r != 1) && vc;
        unsigned ast = toK ? lds(I.K + tl * n + c) : (toR ? lds(I.ring + (r == 0 ? M1_RQ : (r == 2 ? M1_RU : M1_RS)) + c) : trash);
        const unsigned dA = toK ? (unsigned)(-32 * n) : (toR ? (unsigned)(32 * n) : 0u);      // to the next chunk, ring parity 0 -> 1
        const unsigned dB = toK ? (unsigned)(-32 * n) : (toR ? (unsigned)(-32 * n) : 0u);     //                    ring parity 1 -> 0
        double P = vnn ? I.gxx[N * n * n + c * n + r] : 0.0;              // P[H] .= gxx[H]  (:39)
        if (STORE_VALUE && b0 && vnn) I.P[N * n * n + c * n + r] = P;
        // per-lane multipliers of the stacked tiles (rows 0..3)
        double c02 = (r == 0 || r == 2) ? 1.0 : 0.0, c1 = r == 1 ? 1.0 : 0.0, c12 = (r == 1 || r == 2) ? 1.0 : 0.0,
               c0 = r == 0 ? 1.0 : 0.0, c2 = r == 2 ? 1.0 : 0.0, c013 = r != 2 ? 1.0 : 0.0, c3 = r == 3 ? 1.0 : 0.0, qmin = 1.0;
        ILQR_OPAQUE(c02); ILQR_OPAQUE(c1); ILQR_OPAQUE(c12); ILQR_OPAQUE(c0); ILQR_OPAQUE(c2); ILQR_OPAQUE(c013); ILQR_OPAQUE(c3);
        // LDS operands one step ahead on two sets (A: odd t & 3, B: even); the accumulated Hessians come from HBM / L2, whose round
        // trip is longer than a step of this form: TWO steps ahead, four sets (G0..G3 by t & 3), the chunk below at negative immediates
        struct Hess { double xx, ux, uu; };
        Hess G0, G1, G2, G3;
        auto fetch = [&](Opnd& o, int i) {                              // i = t & 3 of the step fetched
            o.fx = LD(afx, i * n * n); o.fu = LD(afu, i * n);
        };
        auto fetch_g = [&](int i) -> Hess {                             // i = t & 3 of the step fetched, -4..-1: in the chunk below
            Hess g;
            g.xx = GL(oxx, i * n * n); g.ux = GL(oux, i * n); g.uu = GL(ouu, i);
            return g;
        };
        auto step = [&](const Opnd& o, const Hess& g, int i) {
            ILQR_ISA_MARK("riccati_step", ROLE);
            const double W = mfma444(P, o.fx, 0.0);                     // W = P'ᵀ fx, WuR = P'ᵀ fuR   (:52-64)
            const double WuR = mfma444(P, o.fu, 0.0);
            const double Qxx = mfma444(W, o.fx, g.xx);
            const double QuxR = mfma444(WuR, o.fx, g.ux);
            const double q = mfma444(WuR, o.fu, g.uu);
            const double sa = recip_fast(q);                            // potrf + potrs of the 1x1 system as b (1 / q), see backward_pass_split
            // min over the pass of 1 / q: negative iff a pivot was (q = 0, NaN or Inf end as a NaN in P). Taken on the VALU result:
            // hipcc puts no hazard wait states around inline asm, so asm must not read an MFMA result; __builtin_fmin would add
            // two canonicalising v_max per step
            asm("v_min_f64 %0, %1, %2" : "=v"(qmin) : "v"(qmin), "v"(sa));
            const double mA = fma(c02, -sa, c1);                        // rows [-1/q, 1, -1/q, 0]
            const double mB1 = fma(c12, -sa, c0);                       // rows [1, -1/q, -1/q, 0]
            const double mB2 = fma(c2, q, c013);                        // rows [1, 1, q, 1]
            const double A3 = QuxR * mA;                                // rows [K; Qux; K; 0]
            // rows [Qux; K; ux_tmp = K q; 1 / q]   (:70-75, :79) — the fourth row rides along for the vector chain: it meets the zero
            // row of A3 in the product
            const double B3 = fma(QuxR * mB1, mB2, c3 * sa);
            const double Pn = mfma444(A3, B3, Qxx);                     // (:81-84)
            ST(ast, i * n, B3);
            if constexpr (STORE_VALUE) {
                if (b0) {
                    if (vnn) I.P[t * n * n + c * n + r] = Pn;
                    if (I.Q != nullptr) {                               // policy.action_value.* (src/data/policy.jl:58-64)
                        if (vnn) I.Q[I.QL.Qxx + t * n * n + c * n + r] = Qxx;
                        if (r == 0 && vc) I.Q[I.QL.Qux + t * n + c] = QuxR;
                        if (r == 0 && c == 0) I.Q[I.QL.Quu + t] = q;
                    }
                }
            }
            P = Pn;
            --t;
        };
        if (t0 & 1) fetch(A, t0 & 3); else fetch(B, t0 & 3);           // steps with odd t & 3 work on set A
        switch (t0 & 3) {                                               // Hessians of the first two steps
            case 3: G3 = fetch_g(3); G2 = fetch_g(2); break;
            case 2: G2 = fetch_g(2); G1 = fetch_g(1); break;
            case 1: G1 = fetch_g(1); G0 = fetch_g(0); break;
            default: G0 = fetch_g(0); G3 = fetch_g(-1);
        }
        for (int e = t0 & 3;; e = 3) {                                  // e: t & 3 of the chunk's first step (< 3 only in the first chunk)
            // requests below t = 0 are reads inside this instance's own LDS set and HBM block (every array fetched here sits behind
            // x̄, ū) whose values are never used
            switch (e) {
                case 3: fetch(B, 2); G1 = fetch_g(1); step(A, G3, 3); [[fallthrough]];
                case 2: fetch(A, 1); G0 = fetch_g(0); step(B, G2, 2); [[fallthrough]];
                case 1: fetch(B, 0); G3 = fetch_g(-1); step(A, G1, 1); [[fallthrough]];
                default:
                    afx -= dfx; afu -= dfu;
                    fetch(A, 3); G2 = fetch_g(-2); step(B, G0, 0);
            }
            oxx -= dxx; oux -= dux; ouu -= 32u;
            ast += par ? dB : dA;
            par ^= 1;
            asm volatile("" : "+v"(afx), "+v"(afu), "+v"(ast));          // (see the vector chain: no re-basing onto negative DS offsets)
            ILQR_BAR_BEGIN();
            __syncthreads();                                            // hand the chunk over (the other half of the ring is free again)
            ILQR_BAR_END(I, PROF_DELTA);
            if (t < 0) break;
        }
        return __builtin_amdgcn_ballot_w64(!(qmin > 0.0) || P != P) != 0;
    } else {
        unsigned agx = vr ? lds(I.gx + tl * n + r) : zero;             const unsigned dgx = vr ? 32u * n : 0u;
        unsigned agu = lds(I.gu + tl);
        unsigned aK = vr ? lds(I.K + tl * n + r) : zero;               // K(r), Qux(r), ux_tmp(r) in every column
        unsigned arq = vr ? lds(I.ring + M1_RQ + r) : zero;            const unsigned dq = vr ? 32u * n : 0u;
        unsigned asa = lds(I.ring + M1_RS);
        unsigned ak = lds(I.k + tl);
        // (the three wave-uniform LDS addresses in VGPRs: left to itself the compiler keeps them in scalar registers and moves them
        // to a vector register in front of every access, 10 of the chunk's 148 issue slots)
        asm volatile("" : "+v"(agu), "+v"(asa), "+v"(ak));
        // the Lagrangian gradient goes straight to its place in the instance's HBM block (plain stores, nobody in the workgroup
        // reads it back during a solve: ‖∇L‖∞ and ∇Lᵀ·Δz are carried in registers) — an LDS store costs the wave 25 clk with the
        // CUs busy (profiles/r02_probe_lds.txt), a global one its issue slot
        unsigned oLu = goff(I.gbase + I.hLu + tl);
        unsigned oLx = vr ? goff(I.gbase + I.hLx + tl * n + r) : goff(I.gzero + 2 + GZERO_REGION);   const unsigned dLx = vr ? 32u * n : 0u;
        auto GS = [&](unsigned o, int off, double v) { *(double*)(const_cast<char*>(gb) + (size_t)o + (ptrdiff_t)(8 * off)) = v; };
        double p = vr ? I.gx[N * n + r] : 0.0;                          // p[H] .= gx[H]   (:40), in every column
        if (STORE_VALUE && b0 && vr && c == 0) I.p[N * n + r] = p;
        double gmax = 0.0, nu = 0.0, dacc = 0.0;
        unsigned long long nanmask = 0;     // ‖·‖∞ must propagate NaN like Julia's norm; v_max_f64 drops NaNs, so they are tracked beside it
        auto fetch = [&](Opnd& o, int i) {
            o.fx = LD(afx, i * n * n); o.fu = LD(afu, i * n);
            o.c0 = LD(agx, i * n); o.c1 = LD(agu, i);
        };
        auto step = [&](const Opnd& o, int i) {
            ILQR_ISA_MARK("riccati_step", ROLE);
            const double Qx = mfma444(o.fx, p, o.c0);                   // Qx = fxᵀ p' + gx (every column), Qu = fuᵀ p' + gu (every lane)   (:44-49)
            const double Qu = mfma444(o.fu, p, o.c1);
            const double sa = LD(asa, i * n);
            const double Kc = LD(aK, i * n), Quxc = LD(arq, i * n), uxtc = LD(arq, M1_RU - M1_RQ + i * n);
            const double k = (Qu * sa) * -1.0;                          // (:72-75)
            // p = ux_tmpᵀ k + Kᵀ Qu + Quxᵀ k + Qx   (:86-89), summed as ((Qx + Kᵀ Qu) + Quxᵀ k) + ux_tmpᵀ k like P
            const double pn = fma(uxtc, k, fma(Quxc, k, fma(Kc, Qu, Qx)));
            // lagrangian_gradient!: Lx = Qx - p[t], Lu = Qu   (src/solve.jl:73-81); the padding of Lx is exactly zero
            const double Lx = Qx - pn;
            // (Lx as a third operand: Qu is an MFMA result, and hipcc puts no hazard wait states around inline asm — behind Lx, which
            // a chain of compiler-scheduled VALU instructions derives from Qu, the read is safe)
            asm("v_max_f64 %0, %1, |%2|" : "=v"(gmax) : "v"(gmax), "v"(Lx));
            asm("v_max_f64 %0, %1, |%2| ; %3" : "=v"(gmax) : "v"(gmax), "v"(Qu), "v"(Lx));
            nanmask |= __builtin_amdgcn_ballot_w64(__builtin_isunordered(Lx, Qu));   // either one NaN
            ST(ak, i, k); GS(oLu, i, Qu); GS(oLx, i * n, Lx);
            // Δ = ∇Lᵀ·Δz as the adjoint of the sensitivity recursion (see backward_pass_split): w = ∇L_u + fuᵀν′, Δ += w k,
            // ν = (∇L_x + fxᵀν′) + Kᵀ w
            const double wv = mfma444(o.fu, nu, Qu);
            dacc = fma(wv, k, dacc);
            const double nun = mfma444(o.fx, nu, Lx);
            nu = fma(Kc, wv, nun);
            if constexpr (STORE_VALUE) {
                if (b0 && c == 0) {
                    if (vr) I.p[t * n + r] = pn;
                    if (I.Q != nullptr) {
                        if (vr) I.Q[I.QL.Qx + t * n + r] = Qx;
                        if (r == 0) I.Q[I.QL.Qu + t] = Qu;
                    }
                }
            }
            p = pn;
            --t;
        };
        if (t0 & 1) fetch(A, t0 & 3); else fetch(B, t0 & 3);
        for (int e = t0 & 3;; e = 3) {
            ILQR_BAR_BEGIN();
            __syncthreads();                                            // this chunk of the ring is complete
            ILQR_BAR_END(I, PROF_COST);
            switch (e) {
                case 3: fetch(B, 2); step(A, 3); [[fallthrough]];
                case 2: fetch(A, 1); step(B, 2); [[fallthrough]];
                case 1: fetch(B, 0); step(A, 1); [[fallthrough]];
                default:
                    afx -= dfx; afu -= dfu; agx -= dgx; agu -= 32u;
                    fetch(A, 3); step(B, 0);
            }
            aK -= dq; ak -= 32u; oLu -= 32u; oLx -= dLx;
            arq += par ? (0u - dq) : dq;
            asa += par ? (unsigned)(-32 * n) : 32u * n;
            par ^= 1;
            // (the walking LDS addresses are opaque from here on: re-based on ONE induction variable by the compiler, the accesses of a
            // chunk end up at negative offsets, which a DS instruction cannot encode — an add in front of every access, 20 per chunk)
            asm volatile("" : "+v"(afx), "+v"(afu), "+v"(agx), "+v"(agu), "+v"(aK), "+v"(ak), "+v"(arq), "+v"(asa));
            if (t < 0) break;
        }
        I.gradient_norm = wave_max(nanmask != 0 ? __builtin_nan("") : gmax);
        I.delta_next = lane_bcast(dacc, 0);
#ifdef ILQR_PROFILE_BAR
        if (lane == 0) I.zs[7] += I.prof[PROF_COST];                    // wave 1's barrier waits, picked up by wave 0 at write-back
        I.prof[PROF_COST] = 0.0;
#endif
        return false;
    }
}

template <class M, bool STORE_VALUE>
__device__ __forceinline__ void backward_pass(Inst<M>& I) {
    ILQR_PROF_BEGIN();
    if constexpr (is_large<M>::value) backward_pass_large<M, STORE_VALUE>(I);
    else {
        if constexpr (waves_of<M>::value == 1) {
            backward_pass_mfma<M, STORE_VALUE, 0>(I);
            __syncthreads();
        } else {
            // matrix chain on wave 0, vector chain on wave 1 (one ring chunk behind); scalars through LDS
            bool literal = true;
            if constexpr (M::NU == 1) {
                // one action: the short form, which assumes positive pivots; a pass that met another kind is repeated literally
                if (I.wave == 0) {
                    const bool bad = backward_pass_m1<M, STORE_VALUE, 1>(I);
                    if (I.lane == 0) { I.zs[6] = bad ? 1.0 : 0.0; I.zs[3] = (double)I.potrf_info; }
                } else {
                    backward_pass_m1<M, STORE_VALUE, 2>(I);
                    if (I.lane == 0) { I.zs[2] = I.gradient_norm; I.zs[5] = I.delta_next; }
                }
                __syncthreads();
                literal = I.zs[6] != 0.0;
#ifdef ILQR_PROFILE
                if (literal) I.prof[PROF_DELTA] += 1.0;                 // (slot unused by the two-wave kernel: counts repeated passes)
#endif
                if (literal) {
                    if (I.wave == 0 && I.lane == 0) I.scal[S_LITERAL_PASSES] += 1.0;
                    __syncthreads();                                    // both waves have read the flag before the ring is reused
                }
            }
            if (literal) {
                if (I.wave == 0) {
                    backward_pass_split<M, STORE_VALUE, 1>(I);
                    if (I.lane == 0) I.zs[3] = (double)I.potrf_info;
                } else {
                    backward_pass_split<M, STORE_VALUE, 2>(I);
                    if (I.lane == 0) { I.zs[2] = I.gradient_norm; I.zs[5] = I.delta_next; }
                }
                __syncthreads();
                if constexpr (lagrangian_home_is_hbm<M>::value) {       // the literal code leaves ∇L in LDS: take it home
                    for (int i = threadIdx.x; i < I.N * M::NX; i += 128) I.gbase[I.hLx + i] = I.Lx[i];
                    for (int i = threadIdx.x; i < I.N * M::NU; i += 128) I.gbase[I.hLu + i] = I.Lu[i];
                }
            }
            I.gradient_norm = I.zs[2]; I.potrf_info = (int)I.zs[3];
            I.delta_next = I.zs[5]; I.delta_next_ok = 1;
        }
    }
    ILQR_PROF_END(I, PROF_BACKWARD);
}

// ------------------------------------------------------------- rollout!
// Closed-loop rollout u = αk + ū + Kx − Kx̄ in the reference's operation order
// (src/rollout.jl:24-28), wave-uniform; lane 0 writes the trial trajectory.
// With `with_delta` (first line-search trial) the loop also carries the sensitivity
// recursion Δu = k + KΔx, Δx⁺ = fuΔu + fxΔx (src/data/methods.jl:42-54) and the product
// ∇Lᵀ·Δz (src/forward_pass.jl:20) as three f64 MFMAs per step on column vectors in
// the MFMA lane layout: the matrix pipe works asynchronously beside the VALU dynamics
// chain, so Δ costs issue slots only (it used to be a separate 54 k-cycle serial loop).
// MULTI: up to four trials of the line search at once, one per ROW of 16 lanes — the small models' cooperative dynamics work within
// rows everywhere (four instances per wave in the packed kernel, four identical copies of the one instance here): row r rolls out
// step size alpha / 2^min(r, nt - 1) into its own trial buffer (trial 0: I.x, I.u; trial j >= 1: buf1 + (j - 1) * bsz, its actions
// xs doubles further). One instruction stream, four trajectories: a round of trials costs what one trial did.
template <class M, bool MULTI = false>
__device__ void rollout_small(Inst<M>& I, double alpha, bool with_delta, double& delta_out, int nt = 1, double* buf1 = nullptr,
                              int xs = 0, int bsz = 0) {
    constexpr int n = M::NX, m = M::NU;
    constexpr bool MF = (n <= 4 && m <= 4) && waves_of<M>::value == 1;   // with two waves Δ is wave 1's job (delta_small)
    // The rollout is ONE instruction stream per instance and a wave issues one instruction per ~5-6 clk whatever it is
    // (tools/probes/probe_issue.hip): every instruction taken out of the step is time. So everything of (:24-28) that does not
    // depend on the running state is formed beforehand, one timestep per lane: a_t = k_t α + ū_t and b_t = K_t x̄_t, parked in
    // the trial buffers where step t will put u_t and x_{t+1} (a_t in u[t], b_t in the first nu entries of x[t+1]); the step then
    // reads K_t, a_t, b_t through three walking LDS addresses with immediate offsets and forms u = (a_t + K_t x) − b_t — the
    // operations of the reference's order, each rounded as before.
    constexpr bool PRE = (m <= n);
    typedef __attribute__((address_space(3))) double ldsd;
    const int lane = I.lane;
    // this row's trial: buffers and step size (MULTI; alpha * 0.5^j is what j halvings of the step size give, src/forward_pass.jl:51)
    double* xr = I.x; double* ur = I.u;
    const double alpha0 = alpha;
    (void)alpha0;
    if constexpr (MULTI) {
        const int jr = (lane >> 4) < nt - 1 ? (lane >> 4) : nt - 1;
        if (jr > 0) { xr = buf1 + (jr - 1) * bsz; ur = xr + xs; }
        for (int j = 0; j < jr; ++j) alpha *= 0.5;
    }
    if constexpr (PRE) {
        for (int t = lane; t < I.N; t += 64) {
#pragma unroll
            for (int i = 0; i < m; ++i) {
                double b = 0.0;
#pragma unroll
                for (int j = 0; j < n; ++j) b = fma(I.K[t * m * n + j * m + i], I.xb[t * n + j], b);
                if constexpr (MULTI) {
                    double al = alpha0;              // every trial's a_t is formed here, whoever's row it is
                    for (int j = 0; j < nt; ++j) {
                        double* xj = j == 0 ? I.x : buf1 + (j - 1) * bsz;
                        double* uj = j == 0 ? I.u : xj + xs;
                        uj[t * m + i] = fma(I.k[t * m + i], al, I.ub[t * m + i]);
                        xj[(t + 1) * n + i] = b;
                        al *= 0.5;
                    }
                } else {
                    const double a = fma(I.k[t * m + i], alpha, I.ub[t * m + i]);     // (:24-26)
                    I.u[t * m + i] = a;
                    I.x[(t + 1) * n + i] = b;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // same wave reads them back: LDS operations of a wave complete in order
        __builtin_amdgcn_wave_barrier();
    }
    double xt[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xt[i] = I.xb[i];                      // (:19)
#pragma unroll
    for (int i = 0; i < n; ++i) xr[i] = xt[i];                        // every lane (of a row) the same value to the same address
    // sensitivity state (MFMA layout: element (r, c) on lane c + 4*blk + 16*r, vectors in column 0)
    const int r = lane >> 4, c = lane & 3, blk = (lane >> 2) & 3;
    const bool vnn = r < n && c < n, vnm = r < n && c < m, vmn = r < m && c < n;
    const bool vn1 = c == 0 && r < n, vm1 = c == 0 && r < m;
    double zx = 0.0, dacc = 0.0;
    (void)blk;
    const typename M::WaveCtx wcx = M::template wave_ctx<ILQR_PIN_ROLLOUT_CONSTANTS>(lane);   // per-lane constants of the cooperative dynamics, built once; the model's
                                                                                            // wave-uniform constants as opaque scalar pairs (8 fewer s_mov_b32 per acrobot step)
    // LDS byte addresses of K_t, u[t], x[t] held in VGPRs (opaque to the compiler, which would otherwise rebuild every address
    // from scalar registers with shift / add / move triples); they advance by two steps per loop trip, all other offsets are
    // immediates. Operands of step t are fetched one step ahead; the loop is unrolled by two with ping-pong operand sets.
    unsigned aK = (unsigned)(size_t)(ldsd*)I.K, aU = (unsigned)(size_t)(ldsd*)ur, aX = (unsigned)(size_t)(ldsd*)xr;
    unsigned aKb = (unsigned)(size_t)(ldsd*)I.k, aUb = (unsigned)(size_t)(ldsd*)I.ub, aXb = (unsigned)(size_t)(ldsd*)I.xb;
    asm volatile("" : "+v"(aK), "+v"(aU), "+v"(aX));
    if constexpr (!PRE) asm volatile("" : "+v"(aKb), "+v"(aUb), "+v"(aXb));
    auto L = [](unsigned a, int off) -> double { return *(const ldsd*)(size_t)(a + 8u * off); };
    auto S = [](unsigned a, int off, double v) { *(ldsd*)(size_t)(a + 8u * off) = v; };
    struct Ops { double K[m * n], a[m], b[m], xb[PRE ? 1 : n]; };
    // d: step offset (0, 1, 2) from the step the walking addresses stand at
    auto fetch = [&](Ops& o, int d) {
#pragma unroll
        for (int i = 0; i < m * n; ++i) o.K[i] = L(aK, d * m * n + i);
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < m; ++i) { o.a[i] = L(aU, d * m + i); o.b[i] = L(aX, (d + 1) * n + i); }
        } else {
#pragma unroll
            for (int i = 0; i < m; ++i) { o.a[i] = L(aKb, d * m + i); o.b[i] = L(aUb, d * m + i); }    // k_t, ū_t
#pragma unroll
            for (int i = 0; i < n; ++i) o.xb[i] = L(aXb, d * n + i);
        }
    };
    // Sensitivity operands, transposed straight from LDS (mfma(A<-X^T, B<-v, C) = X v + C), through
    // per-lane pointers that walk forward in time; padding lanes aim at the zero slot with stride 0.
    const double* pKT = (MF && vnm) ? I.K + r * m + c : I.zs;     const int sKT = (MF && vnm) ? m * n : 0;   // K^T(r,c) = K(c,r)
    const double* zsrc = slim_of<M>::value ? I.gzero : I.zs;      // fx, fu: HBM (slim) or LDS
    const double* pfxT = (MF && vnn) ? I.fx + r * n + c : zsrc;   const int sfxT = (MF && vnn) ? n * n : 0;  // fx^T(r,c) = fx(c,r)
    const double* pfuT = (MF && vmn) ? I.fu + r * n + c : zsrc;   const int sfuT = (MF && vmn) ? n * m : 0;  // fu^T(r,c) = fu(c,r)
    const double* pkc = (MF && vm1) ? I.k + r : I.zs;             const int skc = (MF && vm1) ? m : 0;
    const double* pLx = (MF && vn1) ? I.Lx + r : I.zs;            const int sLx = (MF && vn1) ? n : 0;
    const double* pLu = (MF && vm1) ? I.Lu + r : I.zs;
    auto step = [&](const Ops& o, int t, int d, const double (&xin)[n], double (&xout)[n]) {
        ILQR_ISA_MARK("rollout_step", MF ? 1 : 0);
        // sensitivity recursion: operand loads first (their latency hides under the policy evaluation)
        double KT = 0.0, fxT = 0.0, fuT = 0.0, kc = 0.0, Lxc = 0.0, Luc = 0.0;
        if constexpr (MF) {
            if (with_delta) {
                KT = *pKT; fxT = *pfxT; fuT = *pfuT; kc = *pkc; Lxc = *pLx; Luc = *pLu;
                pKT += sKT; pfxT += sfxT; pfuT += sfuT; pkc += skc; pLx += sLx; pLu += skc;
            }
        }
        double ut[m];
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double a1 = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) a1 = fma(o.K[j * m + i], xin[j], a1);
            if constexpr (PRE) {
                ut[i] = (o.a[i] + a1) - o.b[i];                          // (:27), (:28)
            } else {
                double a2 = 0.0;
#pragma unroll
                for (int j = 0; j < n; ++j) a2 = fma(o.K[j * m + i], o.xb[j], a2);
                ut[i] = (fma(o.a[i], alpha, o.b[i]) + a1) - a2;          // (:24-28)
            }
        }
        // first half of the recursion goes to the matrix pipe before the long VALU dynamics chain,
        // the dependent second half is issued after it: Δ costs issue slots only
        double zu = 0.0, fz = 0.0;
        if constexpr (MF) {
            if (with_delta) {
                zu = mfma444(KT, zx, kc);                       // Δu = k + K Δx
                fz = mfma444(fxT, zx, 0.0);                     // fx Δx
                dacc = mfma444(Lxc, zx, dacc);                  // += ∇L_x · Δx
            }
        }
        double w[cdim<M::NW>::v];
        load_w<M::NW>(I.w, t, w);
        M::dyn_wave(wcx, lane, xin, ut, w, xout);                     // (:29)
        // the trial trajectory: every lane stores the same values to the same addresses (no exec mask to set up and restore)
#pragma unroll
        for (int i = 0; i < m; ++i) S(aU, d * m + i, ut[i]);
#pragma unroll
        for (int i = 0; i < n; ++i) S(aX, (d + 1) * n + i, xout[i]);
        if constexpr (MF) {
            if (with_delta) {
                dacc = mfma444(Luc, zu, dacc);                         // += ∇L_u · Δu
                zx = mfma444(fuT, zu, fz);                             // Δx⁺ = fu Δu + fx Δx
            }
        }
    };
    auto advance2 = [&]() {
        aK += 16u * m * n; aU += 16u * m; aX += 16u * n;
        if constexpr (!PRE) { aKb += 16u * m; aUb += 16u * m; aXb += 16u * n; }
    };
    Ops A, B;
    double xo[n];
    if (I.N > 0) fetch(A, 0);
    int t = 0;
    // (a counted loop and an unconditional fetch of step t + 2: behind the last pair that is a read past K, u, x — inside the
    // instance's LDS set, every one of these arrays is followed by others — whose values nobody uses; five scalar instructions
    // per pair less than the guarded form)
    for (int pairs = I.N >> 1; pairs > 0; --pairs, t += 2) {
        fetch(B, 1);
        step(A, t, 0, xt, xo);
        fetch(A, 2);
        step(B, t + 1, 1, xo, xt);
        advance2();
    }
    if (t < I.N) {
        step(A, t, 0, xt, xo);
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = xo[i];
    }
    if constexpr (MF) {
        if (with_delta) delta_out = lane_bcast(dacc, 0);               // element (0,0) of block 0
    }
}

// trajectory_sensitivities (src/data/methods.jl:42-54) and the product gradient^T dz (src/forward_pass.jl:20) as a
// sweep of its own: wave 1 runs it while wave 0 runs the first rollout (all operands are read-only LDS data by then).
// Wave-uniform arithmetic (every lane the same values), broadcast LDS reads.
template <class M>
__device__ double delta_small(Inst<M>& I) {
    constexpr int n = M::NX, m = M::NU;
    double zx[n], d = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) zx[i] = 0.0;
    for (int t = 0; t < I.N; ++t) {
        ILQR_ISA_MARK("delta_step", 0);
        double du[m];
#pragma unroll
        for (int i = 0; i < m; ++i) {                                   // Δu = k + K Δx
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) acc += I.K[t * m * n + j * m + i] * zx[j];
            du[i] = I.k[t * m + i] + acc;
        }
#pragma unroll
        for (int j = 0; j < n; ++j) d += I.Lx[t * n + j] * zx[j];
#pragma unroll
        for (int i = 0; i < m; ++i) d += I.Lu[t * m + i] * du[i];
        double zn[n];
#pragma unroll
        for (int i = 0; i < n; ++i) {                                   // Δx⁺ = fu Δu + fx Δx
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) a1 += I.fu[t * n * m + j * n + i] * du[j];
#pragma unroll
            for (int j = 0; j < n; ++j) a2 += I.fx[t * n * n + j * n + i] * zx[j];
            zn[i] = a1 + a2;
        }
#pragma unroll
        for (int i = 0; i < n; ++i) zx[i] = zn[i];
    }
    return d;
}

}  // namespace ilqr
#include "ilqr_device_large.hpp"
namespace ilqr {

// ---- dispatch between the LDS-resident small path and the HBM-resident large path
template <class M>
__device__ __forceinline__ void gradients(Inst<M>& I, bool constrained) {
    if constexpr (is_large<M>::value) gradients_large<M>(I, constrained);
    else gradients_small<M>(I, constrained);
}
// (MULTI, nt trials of the line search at once: rollout_small<M, true>; forward_pass<M, SPEC>)
template <class M, bool MULTI = false>
__device__ __forceinline__ void rollout_bang(Inst<M>& I, double alpha, bool with_delta, double& delta_out, int nt = 1,
                                             double* buf1 = nullptr, int xs = 0, int bsz = 0) {
    if constexpr (is_large<M>::value) rollout_large<M>(I, alpha, with_delta, delta_out);
    else {
        ILQR_PROF_BEGIN();
        if constexpr (waves_of<M>::value == 1) {
            rollout_small<M>(I, alpha, with_delta, delta_out);
            __syncthreads();
        } else {
            if (I.wave == 0) {
                double unused = 0.0;
                if constexpr (MULTI) rollout_small<M, true>(I, alpha, false, unused, nt, buf1, xs, bsz);
                else rollout_small<M>(I, alpha, false, unused);
            } else if (with_delta && !I.delta_next_ok) {                // stage kernels: the backward pass was another launch
                const double d = delta_small<M>(I);
                if (I.lane == 0) I.zs[4] = d;
            }
            __syncthreads();
            if (with_delta) delta_out = I.delta_next_ok ? I.delta_next : I.zs[4];
        }
        I.rollouts += 1;
        I.states_eq_nominal = 0;
        ILQR_PROF_END(I, PROF_ROLLOUT);
    }
}

// --------------------------------------------------------- forward_pass!
// SPEC (two-wave small models; 1: from the second trial on — the resume launch; 2: from the first — the packed kernel's workers, whose
// instances are the ones that reject; 0: the latency kernel): the line search takes its trials in ROUNDS of up to four, rolled out at
// once by the four rows of wave 0 (rollout_small<M, true>) into x, u and up to three more trial buffers in the LDS of the dynamics
// Jacobians fx, fu — dead between the backward pass and the next gradients!, which rewrites them; saved to their (stale) HBM home
// before a search's first round and brought back if the whole search fails, the one case in which nothing rewrites them. Trial by
// trial the search then does what it always did — cost!, the Armijo test, the counters, problem.states = the last trial evaluated
// (Q2 reads them) — only that the next trials of a round are rolled out already when one is rejected: a rejected trial is a rollout
// and a cost pass, 40 us of an iteration's 50, and the stragglers of a batch are the instances that reject (instance 7609 of
// config 4's shard 6: 1914 rollouts for 917 iterations). Same arithmetic on the same inputs in the same order per trial: results
// bitwise unchanged.
template <class M, int SPEC = 0>
__device__ void forward_pass(Inst<M>& I, const ilqr_options& opt, bool constrained) {
    constexpr int n = M::NX, m = M::NU;
    constexpr bool MF = true;       // Δ rides along the first rollout on both paths (small: MFMAs beside the VALU chain; large: wave 1)
    constexpr bool SP = SPEC != 0 && !is_large<M>::value && waves_of<M>::value == 2;
    const double c1 = 1.0e-4;
    const int max_iterations = 25;
    I.status = 0;                                                     // (:10)
    const double J_prev = I.objective;                                // (:13)
    // lagrangian_gradient! (:16) was produced by the backward pass (Lx, Lu in LDS).
    // trajectory_sensitivities (src/data/methods.jl:42-54) fused with the product
    // gradientᵀ·Δz (:20): on the MFMA path it rides along the first rollout below.
    double delta = 0.0;
    I.delta = 0.0;
    I.step_size = 1.0;                                                // (:26)
    int iteration = 1;
    // trial buffers 1..nb behind x, u: in the LDS of fx, fu
    const int xs = (I.T * n + 1) & ~1, bsz = xs + ((I.N * m + 1) & ~1);
    const int region = ((I.N * n * n + 1) & ~1) + ((I.N * n * m + 1) & ~1);
    const int nb = SP ? (region / bsz < 3 ? region / bsz : 3) : 0;
    double* const buf1 = I.fx;
    double* const home = I.gbase + (I.fx - I.lds);                    // the HBM home of fx, fu (stale during a launch)
    bool saved = false;
    int last = 0;                                                     // buffer of the last trial evaluated
    auto accept = [&](const double* X, const double* U) {
        // update_nominal_trajectory! (src/data/methods.jl:32-39)
        constexpr int CS = 64 * waves_of<M>::value;     // each element once (a barrier follows: forward_pass's last lines)
        const int c0 = is_large<M>::value ? (int)threadIdx.x : I.lane + 64 * I.wave;
        for (int i = c0; i < I.T * n; i += CS) I.xb[i] = X[i];
        for (int i = c0; i < I.N * m; i += CS) I.ub[i] = U[i];
        I.states_eq_nominal = 1;
        I.status = 1;
    };
    while (I.step_size >= opt.min_step_size) {                        // (:28)
        if (iteration > max_iterations) break;                        // (:29)
        const bool want_delta = MF && iteration == 1 && opt.line_search == 1;
        int nt = 1;                                                   // trials of this round: those the loop would get to one after the other
        if constexpr (SP) {
            if (iteration >= (SPEC == 2 ? 1 : 2) && (!want_delta || I.delta_next_ok)) {
                double s = 0.5 * I.step_size;
                while (nt < 1 + nb && s >= opt.min_step_size && iteration + nt <= max_iterations) { nt += 1; s *= 0.5; }
            }
            if (nt > 1 && !saved) {
                for (int i = (int)threadIdx.x; i < nb * bsz; i += 128) home[i] = buf1[i];
                saved = true;
                __syncthreads();                                      // read out before wave 0 writes trials there
            }
        }
        double d = 0.0;
        rollout_bang<M, SP>(I, I.step_size, want_delta, d, nt, buf1, xs, bsz);     // (:34)
        if (want_delta) { delta = d; I.delta = d; }
        bool done = false;
        for (int j = 0; j < nt; ++j) {                                // the round's trials, in the order of the search
            double* const xs0 = I.x; double* const us0 = I.u;
            if (j > 0) { I.rollouts += 1; I.x = buf1 + (j - 1) * bsz; I.u = I.x + xs; }
            cost_bang<M>(I, true, constrained);                       // (:36)
            const double* X = I.x; const double* U = I.u;
            I.x = xs0; I.u = us0;
            last = j;
            if (I.objective <= J_prev + c1 * I.step_size * delta) {   // (:44) NaN ⇒ reject
                accept(X, U);
                done = true;
                break;
            }
            I.step_size *= 0.5;                                       // (:51)
            iteration += 1;
        }
        if (done) break;
    }
    if (I.status || last > 0) __syncthreads();
    if constexpr (SP) {
        if (last > 0) {               // problem.states, problem.actions = the last trial evaluated (Q2 reads them)
            const double* X = buf1 + (last - 1) * bsz;
            for (int i = (int)threadIdx.x; i < I.T * n; i += 128) I.x[i] = X[i];
            for (int i = (int)threadIdx.x; i < I.N * m; i += 128) I.u[i] = X[xs + i];
            __syncthreads();
        }
        if (saved && (!I.status || opt.line_search == 0)) {     // no gradients! will rewrite fx, fu: bring them back
            for (int i = (int)threadIdx.x; i < nb * bsz; i += 128) buf1[i] = home[i];
            __syncthreads();
        }
    }
}

// data.gradient .= 0 of reset!(solver.data) (src/data/solver.jl:49-59), wherever this kernel keeps it
template <class M>
__device__ __forceinline__ void zero_lagrangian_gradient(Inst<M>& I) {
    for (int i = I.lane; i < I.N * M::NX; i += 64) I.Lx[i] = 0.0;
    for (int i = I.lane; i < I.N * M::NU; i += 64) I.Lu[i] = 0.0;
    if constexpr (lagrangian_home_is_hbm<M>::value) {
        for (int i = I.lane; i < I.N * M::NX; i += 64) I.gbase[I.hLx + i] = 0.0;
        for (int i = I.lane; i < I.N * M::NU; i += 64) I.gbase[I.hLu + i] = 0.0;
    }
}

// reset!(problem.model); reset!(problem.objective) — src/solve.jl:9-10
template <class M>
__device__ void reset_model_objective(Inst<M>& I, bool literal = true) {
    constexpr int n = M::NX, m = M::NU;
    if constexpr (is_large<M>::value) {       // compact representation, work split over the four waves
        reset_model_objective_large<M>(I, literal);
        return;
    }
    for (int i = I.lane; i < I.N * n * n; i += 64) I.fx[i] = 0.0;
    for (int i = I.lane; i < I.N * n * m; i += 64) I.fu[i] = 0.0;
    for (int i = I.lane; i < I.T * n; i += 64) I.gx[i] = 0.0;
    for (int i = I.lane; i < I.N * m; i += 64) I.gu[i] = 0.0;
    for (int i = I.lane; i < I.T * n * n; i += 64) I.gxx[i] = 0.0;
    for (int i = I.lane; i < I.N * m * m; i += 64) I.guu[i] = 0.0;
    for (int i = I.lane; i < I.N * m * n; i += 64) I.gux[i] = 0.0;
    __syncthreads();
}

// augmented_lagrangian_update! — src/augmented_lagrangian.jl:87-110
// augmented_lagrangian_update! for one constraint row (src/augmented_lagrangian.jl:100-108): λ ← λ + ρ c (one fma: the same bits in
// every kernel that runs it — left to the compiler, the latency and the packed kernel contracted it differently on car_obs),
// inequalities clamped at 0 (Julia's max propagates NaN), ρ ← min(scaling ρ, max_penalty) (Julia's min propagates NaN)
__device__ __forceinline__ void dual_update_row(double& lam, double& rho, double c, bool ineq, const ilqr_options& opt) {
#pragma clang fp contract(off)
    double l = __builtin_fma(rho, c, lam);
    if (ineq) l = nanmax(0.0, l);
    lam = l;
    const double r = opt.scaling_penalty * rho;
    rho = (r < opt.max_penalty || r != r) ? r : opt.max_penalty;
}
template <class M>
__device__ void al_update(Inst<M>& I, const ilqr_options& opt) {
    constexpr int ncs = M::NCS, W = waves_of<M>::value;
    for (int i = I.lane + 64 * I.wave; i < I.C; i += 64 * W) {            // read-modify-write: each index exactly once
        const int ns = I.N * ncs;
        bool ineq;
        if (i < ns) ineq = ncs > 0 ? IneqMask<M>::s(i % (ncs > 0 ? ncs : 1)) : false;
        else ineq = IneqMask<M>::t(i - ns);
        double lam = I.lam[i], rho = I.rho[i];
        dual_update_row(lam, rho, I.c[i], ineq, opt);
        I.lam[i] = lam; I.rho[i] = rho;
    }
    __syncthreads();
}

// solve!(solver): constrained_ilqr_solve! (src/solve.jl:88-129) around ilqr_solve!
// (src/solve.jl:1-54), written as ONE loop nest in which every heavy phase
// (linearise, Riccati, line search) is instantiated exactly once.
//   al_outer = true : AL outer loop (Solver with constraints)
//   al_outer = false: a single ilqr_solve! (plain Objective, or the stage test)
template <class M, bool STORE_VALUE, int SPEC = 0>
__device__ void solve_loops(Inst<M>& I, const ilqr_options& opt, bool constrained, bool al_outer, int o_start = 1, int it_start = 0,
                            double obj_prev0 = 0.0) {
    if (al_outer && o_start == 1 && it_start == 0) {
        // reset!(solver.data) (:93, src/data/solver.jl:49-59); λ ← 0, ρ ← ρ0 (:96-103)
        I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0; I.gradient_norm = 0.0;
        zero_lagrangian_gradient<M>(I);
        for (int i = I.lane; i < I.C; i += 64) {
            I.lam[i] = 0.0;
            I.rho[i] = opt.initial_constraint_penalty;
        }
        __syncthreads();
        I.outer_iterations = 0;
    }
    const int outer_max = al_outer ? opt.max_dual_updates : 1;
    for (int o = o_start; o <= outer_max; ++o) {                      // src/solve.jl:105 (o_start > 1: resumed after a hand-over)
        if (al_outer) I.outer_iterations = o;
        // ---------------- ilqr_solve! (src/solve.jl:1-54)
        // (resumed inside an inner solve, it_start >= 1: the workspace block holds the loop's state at the head of that iteration)
        const bool mid = it_start > 0 && o == o_start;
        if (!mid) {
            reset_model_objective<M>(I, false);                       // (:9-10)
            if (opt.reset_cache) {                                    // (:12) reset!(data)
                I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0;
            }
        }
        double obj_prev = mid ? obj_prev0 : 0.0;
        for (int it = mid ? it_start : 0; it <= opt.max_iterations; ++it) {     // it = 0: (:14-21); it ≥ 1: (:22-51)
            if (it == 0) cost_bang<M>(I, false, constrained);         // (:14)
            else forward_pass<M, SPEC>(I, opt, constrained);          // (:23)
            if (it == 0 || opt.line_search != 0) {                    // (:16-18), (:27-33)
                gradients<M>(I, constrained);
                backward_pass<M, STORE_VALUE>(I);
            }
            if (it == 0) { obj_prev = I.objective; continue; }        // (:21)
            I.iterations += 1;                                        // (:39)
            if (I.trace != nullptr && I.trace_len < I.trace_cap && I.lane == 0) {     // verbose record (:40-45)
                double* r = I.trace + (size_t)I.trace_len * TRACE_W;
                r[0] = (double)(al_outer ? o : I.outer_iterations); r[1] = (double)it; r[2] = I.objective; r[3] = I.gradient_norm;
                r[4] = I.max_violation; r[5] = I.step_size; r[6] = (double)I.status; r[7] = (double)I.rollouts;
            }
            I.trace_len += 1;
            if (I.gradient_norm < opt.lagrangian_gradient_tolerance) break;          // (:48)
            if (fabs(I.objective - obj_prev) < opt.objective_tolerance) break;       // (:49)
            obj_prev = I.objective;
            if (!I.status) break;                                     // (:50)
        }
        if (!al_outer) break;
        cost_bang<M>(I, false, true);                                 // src/solve.jl:113
        if (I.max_violation <= opt.constraint_tolerance) break;       // (:117)
        al_update<M>(I, opt);                                         // (:120-122)
    }
}

// ------------------------------------------------------------ kernel glue
template <class M>
__device__ __forceinline__ void inst_setup(Inst<M>& I, const KArgs& a, double* smem, int b, int swap_roles = 0) {
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    I.xb = smem + L.xb; I.ub = smem + L.ub; I.x = smem + L.x; I.u = smem + L.u;
    I.fx = smem + L.fx; I.fu = smem + L.fu; I.gx = smem + L.gx; I.gu = smem + L.gu;
    I.K = smem + L.K; I.k = smem + L.k; I.Lx = smem + L.Lx; I.Lu = smem + L.Lu;
    I.c = smem + L.c; I.lam = smem + L.lam; I.rho = smem + L.rho; I.act = smem + L.act;
    I.zs = smem + L.zslot; I.gzero = g + L.gzero; I.w = smem + L.w; I.ring = smem + L.ring;
    I.gxx = g + L.gxx; I.guu = g + L.guu; I.gux = g + L.gux; I.P = g + L.P; I.p = g + L.p; I.scal = g + L.scal;
    I.T = L.T; I.N = L.T - 1; I.C = L.C; I.lane = threadIdx.x & 63; I.wave = (int)(threadIdx.x >> 6) ^ swap_roles;
    I.cost_par = 0;
    I.lds = smem; I.gbase = g; I.fv_off = 0; I.hc_off = 0; I.hLx = L.Lx; I.hLu = L.Lu;
    I.trace = a.trace ? a.trace + (size_t)b * (size_t)a.trace_cap * TRACE_W : nullptr;
    I.trace_cap = a.trace_cap; I.trace_len = 0;      // (the stage kernel continues from the stored count, see there)
    I.Q = a.qv ? a.qv + (size_t)b * (size_t)a.QL.stride : nullptr; I.QL = a.QL;
    I.delta = I.scal[S_DELTA]; I.delta_next = 0.0; I.delta_next_ok = 0;
    if constexpr (is_large<M>::value) {
        // large path: every buffer stays in the HBM workspace, LDS is staging only
        I.xb = g + L.xb; I.ub = g + L.ub; I.x = g + L.x; I.u = g + L.u;
        I.fx = g + L.fx; I.fu = g + L.fu; I.gx = g + L.gx; I.gu = g + L.gu;
        I.K = g + L.K; I.k = g + L.k; I.Lx = g + L.Lx; I.Lu = g + L.Lu;
        I.c = g + L.c; I.lam = g + L.lam; I.rho = g + L.rho; I.act = g + L.act;
        I.zs = g + L.zslot; I.w = g + L.w;
        I.fv_off = L.fv; I.hc_off = L.hc;
        if (threadIdx.x == 0) store_layout_lds<M>(L);     // read back by the phase functions (real calls) instead of twenty stack arguments
    } else {
        // LDS-resident set: one coalesced 16-B-per-lane stream from HBM
        const double2* src = reinterpret_cast<const double2*>(g);
        double2* dst = reinterpret_cast<double2*>(smem);
        const int nd = slim_of<M>::value ? L.lds_doubles_slim : L.lds_doubles;
        for (int i = I.lane + 64 * I.wave; i < nd / 2; i += 64 * waves_of<M>::value) dst[i] = src[i];
        __syncthreads();
        if (I.lane == 0 && I.wave == 0) { I.zs[0] = 0.0; I.zs[1] = 0.0; }
#ifdef ILQR_PROFILE_BAR
        if (I.lane == 0 && I.wave == 0) I.zs[7] = 0.0;
#endif
        if constexpr (slim_of<M>::value) { I.fx = g + L.fx; I.fu = g + L.fu; }
    }
    I.objective = I.scal[S_OBJECTIVE]; I.max_violation = I.scal[S_MAX_VIOLATION];
    I.step_size = I.scal[S_STEP_SIZE]; I.gradient_norm = I.scal[S_GRADIENT_NORM];
    I.status = (int)I.scal[S_STATUS]; I.iterations = (int)I.scal[S_ITERATIONS];
    I.outer_iterations = (int)I.scal[S_OUTER_ITERATIONS]; I.potrf_info = (int)I.scal[S_POTRF_INFO];
    I.rollouts = (int)I.scal[S_ROLLOUTS]; I.states_eq_nominal = (int)I.scal[S_STATES_EQ_NOMINAL];
#ifdef ILQR_PROFILE
    for (int i = 0; i < PROF_N; ++i) I.prof[i] = 0.0;
#endif
    __syncthreads();
}

template <class M>
__device__ __forceinline__ void inst_writeback(Inst<M>& I, const KArgs& a, double* smem, int b) {
    __syncthreads();
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    if constexpr (!is_large<M>::value) {
        double2* dst = reinterpret_cast<double2*>(g);
        const double2* src = reinterpret_cast<const double2*>(smem);
        const int nd = slim_of<M>::value ? L.lds_doubles_slim : L.lds_doubles;
        if constexpr (lagrangian_home_is_hbm<M>::value) {
            // everything but the Lagrangian gradient, whose HBM values are the current ones (Lx, Lu: adjacent, even offsets and lengths)
            const int lo = L.Lx / 2, hi = (L.Lu + pad2(I.N * M::NU)) / 2;
            for (int i = I.lane + 64 * I.wave; i < nd / 2; i += 64 * waves_of<M>::value)
                if (i < lo || i >= hi) dst[i] = src[i];
        } else {
            for (int i = I.lane + 64 * I.wave; i < nd / 2; i += 64 * waves_of<M>::value) dst[i] = src[i];
        }
    }
    if (I.lane == 0 && I.wave == 0) {
        I.scal[S_OBJECTIVE] = I.objective; I.scal[S_MAX_VIOLATION] = I.max_violation;
        I.scal[S_STEP_SIZE] = I.step_size; I.scal[S_GRADIENT_NORM] = I.gradient_norm;
        I.scal[S_STATUS] = (double)I.status; I.scal[S_ITERATIONS] = (double)I.iterations;
        I.scal[S_OUTER_ITERATIONS] = (double)I.outer_iterations; I.scal[S_POTRF_INFO] = (double)I.potrf_info;
        I.scal[S_ROLLOUTS] = (double)I.rollouts; I.scal[S_STATES_EQ_NOMINAL] = (double)I.states_eq_nominal;
        I.scal[S_DELTA] = I.delta;
#ifdef ILQR_PROFILE
        for (int i = 0; i < PROF_N; ++i) I.scal[S_PROF + i] = I.prof[i];
#ifdef ILQR_PROFILE_BAR
        if constexpr (!is_large<M>::value) I.scal[S_PROF + PROF_COST] = I.zs[7];
#endif
#endif
    }
}

#ifndef ILQR_SPEC_LATENCY
#define ILQR_SPEC_LATENCY 1         // line-search trials in rounds of up to four (forward_pass<M, 1 | 2>: from the second / first trial on) in the latency kernel ...
#endif
#ifndef ILQR_SPEC_RESUME
#define ILQR_SPEC_RESUME 1          // ... and in the resume launch
#endif
// Which wave of this workgroup takes role 0 (KArgs::cu_slots): the workgroups of a CU publish the SIMDs (s0, s1) of their two waves,
// wait until `expect` of them have (bounded: a launch that puts fewer on this CU decides on what is there), and evaluate the same
// rule on the same table: of all assignments the one that puts critical waves on the most distinct SIMDs, then the fewest swaps,
// then the lowest mask. Returns 1 if this workgroup's roles are swapped. (A greedy first-come choice through counters was the
// first version: in a third of the launches the arrival order left one SIMD with two critical waves.)
__device__ inline int pick_roles(int* c, int s0, int s1, int expect) {
    const int k = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k >= 4) return 0;                                             // a later round of a long launch: as launched
    __hip_atomic_store(c + 1 + k, role_entry(s0, s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (expect > 4) expect = 4;
    int e[4], n = 0;
    for (int tries = 0; tries < 64; ++tries) {
        n = 0;
        for (int i = 0; i < 4; ++i) {
            e[i] = __hip_atomic_load(c + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (e[i] & 16) n = i + 1; 
        }
        bool all = n >= expect;
        for (int i = 0; i < n; ++i) all = all && (e[i] & 16);
        if (all) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (n <= k) n = k + 1;
    e[k] = role_entry(s0, s1);
    return (role_mask(e, n) >> k) & 1;                                // (the rule: ilqr_ric_schedule.hpp, checked on the host)
}

// one instance from its workspace block to the end of solve!: from the start, or (resumed) from where the packed kernel left it
template <class M, bool RESUMED, int SPEC = 0>
__device__ __forceinline__ void solve_instance(const KArgs& a, double* smem, int b) {
    constexpr bool resumed = RESUMED;
    int o_start = 1, it_start = 0;
    double obj_prev0 = 0.0;
    if constexpr (RESUMED) {
        const double* sc = a.ws + (size_t)b * (size_t)a.L.stride + a.L.scal;
        o_start = (int)sc[S_RESUME];
        it_start = (int)sc[S_INNER_IT];
        obj_prev0 = sc[S_OBJ_PREV];
    }
    Inst<M> I;
    int swap_roles = 0;
    unsigned hwid = 0, xcc = 0;
    constexpr bool TWO = !is_large<M>::value && waves_of<M>::value == 2;
    if constexpr (TWO) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (a.cu_slots != nullptr) {
            // one critical wave per SIMD (KArgs::cu_slots)
            int* sm = reinterpret_cast<int*>(smem);
            if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = (int)((hwid >> 4) & 3u);
            __syncthreads();
            if (threadIdx.x == 0) {
                const int cu = (int)(((xcc & 7u) << 8) | (((hwid >> 13) & 7u) << 5) | (((hwid >> 12) & 1u) << 4) | ((hwid >> 8) & 15u));
                sm[2] = pick_roles(a.cu_slots + CU_SLOT_INTS * cu, sm[0], sm[1], a.cu_expect);
            }
            __syncthreads();
            swap_roles = sm[2];
            __syncthreads();                                          // read before inst_setup fills the LDS set
        }
    }
    inst_setup<M>(I, a, smem, b, swap_roles);
    if (I.lane == 0 && I.wave == 0) I.scal[S_T_START] = (double)wall_clock64();
    if constexpr (is_large<M>::value && waves_of<M>::value == LARGE_WAVES) {
        unsigned hw_, xc_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc_));
        if (I.lane == 0) I.scal[S_HW0 + I.wave] = (double)hw_ + 4294967296.0 * (double)(xc_ & 15u);
    }
    if constexpr (TWO) {
        if (I.lane == 0) I.scal[S_HW0 + I.wave] = (double)hwid + 4294967296.0 * (double)(xcc & 15u);
        // The SIMD's arbiter serves its OLDEST wave first: of two waves that share a SIMD the one dispatched later gets what the other
        // leaves (measured: the fourth workgroup of a CU ran its iterations in 67 us, the first in 53; tools/finish_times.py).
        // Wave 1 works a third of the time and its instance's critical wave waits for it at the chunk barriers of the Riccati
        // recursion: it goes first whenever it has work, whatever its age; every critical wave then yields to ONE foreign wave 1.
#ifndef ILQR_NO_SETPRIO
        if (I.wave != 0) __builtin_amdgcn_s_setprio(3);
#endif
    }
    if (!resumed) {
        I.potrf_info = 0; I.rollouts = 0; I.outer_iterations = 0;
        if (I.lane == 0 && I.wave == 0) I.scal[S_LITERAL_PASSES] = 0.0;
    } else {
        I.trace_len = (int)I.scal[S_TRACE_LEN];
        // resumed at the head of an inner iteration: the first forward pass takes the Armijo product the packed kernel's backward
        // pass left behind (same adjoint form, same bits) instead of the forward sensitivity sweep
        if (it_start >= 1) { I.delta_next = I.scal[S_DELTA_NEXT]; I.delta_next_ok = 1; }
    }
    {
        ILQR_PROF_BEGIN();
        solve_loops<M, false, SPEC>(I, a.opt, a.constrained != 0, a.constrained != 0, o_start, it_start, obj_prev0);
        ILQR_PROF_END(I, PROF_OTHER);   // total; phases are subtracted on the host
    }
    if (I.lane == 0 && I.wave == 0) {
        I.scal[S_TRACE_LEN] = (double)(I.trace_len < I.trace_cap ? I.trace_len : I.trace_cap);
        I.scal[S_RESUME] = 0.0;
        I.scal[S_T_END] = (double)wall_clock64();
    }
    inst_writeback<M>(I, a, smem, b);
    if constexpr (TWO) __builtin_amdgcn_s_setprio(0);                 // (the packed kernel's workers go on after this)
}

// solve!(solver) for every instance — src/solve.jl:137-143
template <class M>
__global__ __launch_bounds__(64 * waves_of<M>::value, 2) void solve_kernel(KArgs a) {      // (second parameter: waves per SIMD -> 256 VGPRs)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    solve_instance<M, false, ILQR_SPEC_LATENCY>(a, smem, b);
}

// The same for the instances the packed kernel handed over, in the launch that follows it on the stream: workgroup b finishes
// instance b if it was marked (S_RESUME). A kernel of its own: the entry inside an inner solve costs registers the plain kernel
// does not have to spare.
template <class M>
__global__ __launch_bounds__(64 * waves_of<M>::value, 2) void solve_kernel_resume(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    if ((int)(a.ws + (size_t)b * (size_t)a.L.stride)[a.L.scal + S_RESUME] < 1) return;
    solve_instance<M, true, ILQR_SPEC_RESUME>(a, smem, b);
}

// throughput variant: two waves per SIMD (<= 256 registers), Jacobians in HBM (see Slim<M>)
template <class M>
__global__ __launch_bounds__(64, 2) void solve_kernel_slim(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    typedef Slim<M> MS;
    Inst<MS> I;
    inst_setup<MS>(I, a, smem, b);
    I.potrf_info = 0; I.rollouts = 0; I.outer_iterations = 0;
    solve_loops<MS, false>(I, a.opt, a.constrained != 0, a.constrained != 0);
    if (I.lane == 0) I.scal[S_TRACE_LEN] = (double)(I.trace_len < I.trace_cap ? I.trace_len : I.trace_cap);
    inst_writeback<MS>(I, a, smem, b);
}

// single stages for parity tests (STORE_VALUE: P, p and, when enabled, Qx..Qux are written to HBM); instantiated for
// the latency mapping (M) and, for small models, the throughput mapping (Slim<M>)
template <class M>
__global__ __launch_bounds__(64 * waves_of<M>::value, 2) void stage_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    Inst<M> I;
    inst_setup<M>(I, a, smem, b);
    const bool con = a.constrained != 0;
    // host-stepped AL loop (solve! with augmented_lagrangian_callback!, src/solve.jl:88,125): instances that
    // already met the constraint tolerance sit out the remaining outer iterations
    const bool done = I.scal[S_DONE] != 0.0 && a.stage > ILQR_STAGE_AL_BEGIN;      // AL_BEGIN itself re-arms a finished instance
    // per-iteration trace across the launches of a host-stepped solve (augmented_lagrangian_callback! path): AL_BEGIN and a
    // stand-alone ILQR_SOLVE start a new record, every later stage appends to it
    const bool trace_restart = a.stage == ILQR_STAGE_AL_BEGIN || a.stage == ILQR_STAGE_ILQR_SOLVE;
    I.trace_len = trace_restart ? 0 : (int)I.scal[S_TRACE_LEN];
    if (!done) switch (a.stage) {
        case ILQR_STAGE_COST_NOMINAL: cost_bang<M>(I, false, con); break;
        case ILQR_STAGE_GRADIENTS: gradients<M>(I, con); break;
        case ILQR_STAGE_BACKWARD_PASS: backward_pass<M, true>(I); break;
        case ILQR_STAGE_FORWARD_PASS: forward_pass<M>(I, a.opt, con); break;
        case ILQR_STAGE_RESET_MODEL_OBJECTIVE: reset_model_objective<M>(I); break;
        case ILQR_STAGE_ILQR_SOLVE: solve_loops<M, true>(I, a.opt, con, false); break;
        case ILQR_STAGE_AL_UPDATE: al_update<M>(I, a.opt); break;
        case ILQR_STAGE_AL_BEGIN: {      // src/solve.jl:93-103: reset!(data), λ ← 0, ρ ← ρ0
            I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0; I.gradient_norm = 0.0;
            I.outer_iterations = 0; I.potrf_info = 0; I.rollouts = 0;
            zero_lagrangian_gradient<M>(I);
            for (int i = I.lane; i < I.C; i += 64) { I.lam[i] = 0.0; I.rho[i] = a.opt.initial_constraint_penalty; }
            if (I.lane == 0 && I.wave == 0) I.scal[S_DONE] = 0.0;
            __syncthreads();
        } break;
        case ILQR_STAGE_AL_OUTER: {      // one pass of the loop body src/solve.jl:105-122 (callback runs on the host)
            I.outer_iterations += 1;
            solve_loops<M, false>(I, a.opt, true, false);             // ilqr_solve!            (:109)
            cost_bang<M>(I, false, true);                             // cost!(mode = :nominal) (:113)
            if (I.max_violation <= a.opt.constraint_tolerance) {      // (:117)
                if (I.lane == 0 && I.wave == 0) I.scal[S_DONE] = 1.0;
            } else {
                al_update<M>(I, a.opt);                               // (:120-122)
            }
        } break;
        // ---- shared step size over the whole batch: the line search is stepped from the host (include/ilqr_hip.h)
        case ILQR_STAGE_SS_INNER_BEGIN: {                             // src/solve.jl:9-21
            I.outer_iterations += 1;
            reset_model_objective<M>(I, false);
            if (a.opt.reset_cache) { I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0; }
            cost_bang<M>(I, false, con);                              // (:14)
            gradients<M>(I, con);                                     // (:16)
            backward_pass<M, false>(I);                               // (:18)
            if (I.lane == 0 && I.wave == 0) { I.scal[S_OBJ_PREV] = I.objective; I.scal[S_INNER_DONE] = 0.0; I.scal[S_INNER_IT] = 0.0; }
        } break;
        case ILQR_STAGE_SS_TRIAL: {                                   // src/forward_pass.jl:10-20 (first trial), :34-36
            if (I.scal[S_INNER_DONE] == 0.0) {
                const bool first = a.stage_flag != 0;
                if (first) {
                    I.status = 0;
                    if (I.lane == 0 && I.wave == 0) I.scal[S_J_PREV] = I.objective;
                    I.delta = 0.0;
                }
                double d = 0.0;
                const bool want_delta = first && a.opt.line_search == 1;
                rollout_bang<M>(I, a.stage_param, want_delta, d);
                if (want_delta) I.delta = d;
                cost_bang<M>(I, true, con);
            }
        } break;
        case ILQR_STAGE_SS_FINISH: {                                  // src/forward_pass.jl:44-52, then src/solve.jl:27-51
            if (I.scal[S_INNER_DONE] == 0.0) {
                constexpr int n = M::NX, m = M::NU;
                I.step_size = a.stage_param;
                if (a.stage_flag != 0) {                              // update_nominal_trajectory!
                    for (int i = I.lane; i < I.T * n; i += 64) I.xb[i] = I.x[i];
                    for (int i = I.lane; i < I.N * m; i += 64) I.ub[i] = I.u[i];
                    I.states_eq_nominal = 1; I.status = 1;
                    __syncthreads();
                } else {
                    I.status = 0;
                }
                if (a.opt.line_search != 0) { gradients<M>(I, con); backward_pass<M, false>(I); }
                I.iterations += 1;
                const int it = (int)I.scal[S_INNER_IT] + 1;
                const double obj_prev = I.scal[S_OBJ_PREV];
                bool stop = false;
                if (I.gradient_norm < a.opt.lagrangian_gradient_tolerance) stop = true;
                else if (fabs(I.objective - obj_prev) < a.opt.objective_tolerance) stop = true;
                else if (!I.status) stop = true;
                else if (it >= a.opt.max_iterations) stop = true;
                __syncthreads();
                if (I.lane == 0 && I.wave == 0) {
                    I.scal[S_INNER_IT] = (double)it;
                    if (stop) I.scal[S_INNER_DONE] = 1.0; else I.scal[S_OBJ_PREV] = I.objective;
                }
            }
        } break;
        case ILQR_STAGE_SS_OUTER: {                                   // src/solve.jl:113-122
            cost_bang<M>(I, false, true);
            if (I.max_violation <= a.opt.constraint_tolerance) {
                if (I.lane == 0 && I.wave == 0) I.scal[S_DONE] = 1.0;
            } else {
                al_update<M>(I, a.opt);
            }
        } break;
        default: break;
    }
    if (I.lane == 0 && I.wave == 0 && (trace_restart || (!done && I.trace != nullptr)))
        I.scal[S_TRACE_LEN] = (double)(I.trace_len < I.trace_cap ? I.trace_len : I.trace_cap);
    inst_writeback<M>(I, a, smem, b);
}

// x̄ = rollout(dynamics, x1, ū) (src/rollout.jl:33-42) + initialize_controls!/
// initialize_states! (src/solver.jl:56-66). One LANE per instance here (the
// open-loop rollout has no intra-instance parallelism and runs once per solve).
template <class M>
__global__ __launch_bounds__(64) void init_rollout_kernel(KArgs a) {
    constexpr int n = M::NX, m = M::NU;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    const int N = L.T - 1;
    double xt[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { xt[i] = a.x1[(size_t)b * n + i]; g[L.xb + i] = xt[i]; }
    for (int t = 0; t < N; ++t) {
        double ut[m], y[n];
#pragma unroll
        for (int i = 0; i < m; ++i) { ut[i] = a.u_in[((size_t)b * N + t) * m + i]; g[L.ub + t * m + i] = ut[i]; }
        double w[cdim<M::NW>::v];
        load_w<M::NW>(g + L.w, t, w);
        M::dyn(xt, ut, w, y);
#pragma unroll
        for (int i = 0; i < n; ++i) { xt[i] = y[i]; g[L.xb + (t + 1) * n + i] = y[i]; }
    }
    g[L.scal + S_STATES_EQ_NOMINAL] = 0.0;
}

// The same for large models: ONE WAVE per instance, one state component per lane — row i of the affine part of the dynamics
// (generated table M::DYN_AFF, coefficients in registers for the whole horizon) on lane i plus the wave-cooperative
// remainder, x and u exchanged through LDS; all HBM traffic coalesced (the one-lane-per-instance kernel above walks
// 2.4 MB-strided blocks and took 1.0 ms for 512 synth32 instances, 16 % of a BASELINE solve step).
template <class M>
__global__ __launch_bounds__(64) void init_rollout_large_kernel(KArgs a) {
    constexpr int n = M::NX, m = M::NU;
    __shared__ double sx[n], su[m];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= a.B) return;
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    const int N = L.T - 1;
    DynAff<M> aff;
    aff.init(lane);
    const int xrow = DynAff<M>::SPLIT ? (lane & 31) : lane;
    double xl = xrow < n ? a.x1[(size_t)b * n + xrow] : 0.0;
    if (lane < n) g[L.xb + lane] = xl;
    double ul = (lane < m && N > 0) ? a.u_in[(size_t)b * N * m + lane] : 0.0;
    for (int t = 0; t < N; ++t) {
        if (lane < n) sx[lane] = xl;
        if (lane < m) { su[lane] = ul; g[L.ub + t * m + lane] = ul; }
        wave_lds_fence();
        const double u_next = (lane < m && t + 1 < N) ? a.u_in[((size_t)b * N + t + 1) * m + lane] : 0.0;
        double ua[m];
#pragma unroll
        for (int j = 0; j < m; ++j) ua[j] = su[j];
        const double y = dyn_row<M>(aff, sx, ua, xl, lane, g + L.w, t);
        xl = y;
        ul = u_next;
        if (lane < n) g[L.xb + (t + 1) * n + lane] = y;
        wave_lds_fence();
    }
    if (lane == 0) g[L.scal + S_STATES_EQ_NOMINAL] = 0.0;
}

}  // namespace ilqr
#include "ilqr_device_packed.hpp"

// Model module interface: what a compiled model (built-in or generated by
// iterativelqr.jl_amd/codegen.py) registers with the library.
// the packed kernel's two-wave form fits: its workgroups' LDS at per_cu workgroups per CU (one rule for the launcher and for
// ilqr_solve / ilqr_resolved_kernel_variant on the host)
inline bool packed2_fits(int lds2_bytes, int per_cu) { return lds2_bytes > 0 && per_cu >= 1 && per_cu <= 4 && (long long)per_cu * (lds2_bytes + 512) <= 160ll * 1024; }
#define ILQR_MODEL_ABI_VERSION 10   /* bump whenever KArgs, Layout or this struct change: stale model modules are refused */
extern "C" struct ilqr_model_vtable {
    int abi_version;     // ILQR_MODEL_ABI_VERSION the module was compiled against
    int kargs_bytes;     // sizeof(ilqr::KArgs) it was compiled against
    const char* name;
    int nx, nu, nw, ncs, nct;
    unsigned long long ineq_s, ineq_t;
    int (*launch_solve)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);
    int (*launch_stage)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);
    int (*launch_init)(const ilqr::KArgs* a, void* stream);
    int (*launch_solve_slim)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);   // null for large models
    int (*launch_stage_slim)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);   // null for large models
    int (*launch_solve_packed)(const ilqr::KArgs* a, void* stream);                   // four instances per wave, no LDS; null for large models
    // large models only (null otherwise): the compact representation the kernels stream — row lengths for make_layout and
    // the kernel that writes the host-visible full arrays from it (dir 0) or reads them back (dir 1)
    int jac_nvar, hess_nnz;
    int (*launch_mirror)(const ilqr::KArgs* a, int dir, void* stream);
    // large models whose matrices are single 16x16 tiles (nx, nu <= 16), null otherwise: the one-wave-per-instance variant
    int (*launch_solve_mid)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);
    int (*launch_stage_mid)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);
    // LDS bytes of a workgroup of the packed kernel's two-wave form (0: no packed kernel): launch_solve_packed takes that form
    // when KArgs::stage_flag == 2 and per_cu * (this + 512) <= 160 KiB — the host asks for it under the same condition
    int packed2_lds_bytes;
};

namespace ilqr {
template <class M>
struct ModelModule {
    static int launch_solve(const KArgs* a, size_t lds, void* stream) {
        // dynamic LDS above the 64 KiB default needs the per-device function attribute
        // (set on every such launch: handles may live on different devices of one process)
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<M>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        if constexpr (packed_ok<M>::value) {         // hand-over exists where the packed kernel does
            if (a->resume) {
                if (lds > 64 * 1024 &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel_resume<M>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
                hipLaunchKernelGGL(solve_kernel_resume<M>, dim3(a->B), dim3(64 * waves_of<M>::value), lds, (hipStream_t)stream, *a);
                return hipGetLastError() == hipSuccess ? 0 : -1;
            }
        }
        hipLaunchKernelGGL(solve_kernel<M>, dim3(a->B), dim3(64 * waves_of<M>::value), lds, (hipStream_t)stream, *a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static int launch_stage(const KArgs* a, size_t lds, void* stream) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<M>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
        hipLaunchKernelGGL(stage_kernel<M>, dim3(a->B), dim3(64 * waves_of<M>::value), lds, (hipStream_t)stream, *a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static int launch_solve_slim(const KArgs* a, size_t lds, void* stream) {
        if constexpr (!is_large<M>::value) {
            if (lds > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel_slim<M>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            hipLaunchKernelGGL(solve_kernel_slim<M>, dim3(a->B), dim3(64), lds, (hipStream_t)stream, *a);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    static int launch_stage_slim(const KArgs* a, size_t lds, void* stream) {
        if constexpr (!is_large<M>::value) {
            if (lds > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<Slim<M>>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            hipLaunchKernelGGL(stage_kernel<Slim<M>>, dim3(a->B), dim3(64), lds, (hipStream_t)stream, *a);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    // KArgs::stage_flag == 2 asks for the two-wave form (a linearisation server beside the solver wave, ilqr_device_packed.hpp) with
    // KArgs::stage_param = workgroups per CU at this batch size: taken when its two chunk buffers fit a CU's LDS at that residency
    static int launch_solve_packed(const KArgs* a, void* stream) {
        if constexpr (packed_ok<M>::value) {
            constexpr size_t lds2 = sizeof(double) * pk::PkLds<M, true>::total;
            const double per_cu = a->stage_param >= 1.0 ? a->stage_param : 1.0;
            if (a->stage_flag == 2 && packed2_fits((int)lds2, (int)per_cu)) {
                KArgs b = *a;
                b.stage_flag = 0; b.stage_param = 0.0;
                hipLaunchKernelGGL((solve_kernel_packed<M, true>), dim3((a->B + 3) / 4), dim3(128), lds2, (hipStream_t)stream, b);
            } else {
                // one-wave form: two packs per workgroup (four workgroups of 256-register waves per CU); with the pool where the
                // latency solver's LDS fits without costing residency
                KArgs b = *a;
                b.stage_flag = 0; b.stage_param = 0.0;
                constexpr size_t pack = sizeof(double) * ((pk::PkLds<M, false>::total + 1) & ~1);
                size_t lds = 2 * pack;
                const size_t cu = 160 * 1024, ctl = sizeof(int) * CTL_WORDS;
                auto per = [&](size_t bytes) { const size_t k = cu / ((bytes + 511) & ~(size_t)511); return k < 4 ? k : 4; };
                if (b.pool != nullptr && b.pool_lds > 0) {
                    const size_t with_pool = lds > (size_t)b.pool_lds ? lds : (size_t)b.pool_lds;
                    if (per(with_pool + ctl) >= per(lds + ctl)) lds = with_pool; else b.pool = nullptr;
                } else b.pool = nullptr;
                b.pool_ctl = (int)(lds / sizeof(double));
                lds += ctl;
                // marks, vacated CUs and waiting workers belong to launches whose workgroups are ALL resident (a waiting workgroup
                // would hold the slots the next round needs): with fewer than that resident — long horizons, per() < 4 — nobody is
                // marked and nobody waits (advisor finding, round 5: the host's test assumed four workgroups per CU)
                {
                    int dev = 0, cus = 0;
                    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
                    const size_t wgs = (size_t)(((a->B + 3) / 4 + 1) / 2);
                    if (per(lds) * (size_t)cus < wgs) { b.pool_cu = 0; b.pool_mark = 0; }
                }
                if (lds > 64 * 1024 &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel_packed<M, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
                hipLaunchKernelGGL((solve_kernel_packed<M, false>), dim3(((a->B + 3) / 4 + 1) / 2), dim3(128), lds, (hipStream_t)stream, b);
            }
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    static int launch_solve_mid(const KArgs* a, size_t lds, void* stream) {
        if constexpr (mid_ok<M>::value) {
            if (lds > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<Mid<M>>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            hipLaunchKernelGGL(solve_kernel<Mid<M>>, dim3(a->B), dim3(64), lds, (hipStream_t)stream, *a);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    static int launch_stage_mid(const KArgs* a, size_t lds, void* stream) {
        if constexpr (mid_ok<M>::value) {
            if (lds > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<Mid<M>>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            hipLaunchKernelGGL(stage_kernel<Mid<M>>, dim3(a->B), dim3(64), lds, (hipStream_t)stream, *a);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    static int launch_init(const KArgs* a, void* stream) {
        if constexpr (is_large<M>::value) hipLaunchKernelGGL(init_rollout_large_kernel<M>, dim3(a->B), dim3(64), 0, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL(init_rollout_kernel<M>, dim3((a->B + 63) / 64), dim3(64), 0, (hipStream_t)stream, *a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static int launch_mirror(const KArgs* a, int dir, void* stream) {
        if constexpr (is_large<M>::value) {
            hipLaunchKernelGGL(mirror_large_kernel<M>, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a, dir);
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else {
            return -1;
        }
    }
    template <class MM = M> static constexpr int packed2_lds() { if constexpr (packed_ok<MM>::value) return (int)(sizeof(double) * pk::PkLds<MM, true>::total); else return 0; }
    template <class MM = M> static constexpr int jac_nvar() { if constexpr (is_large<MM>::value) return MM::JAC_NVAR; else return 0; }
    template <class MM = M> static constexpr int hess_nnz() { if constexpr (is_large<MM>::value) return MM::HESS_NXX + MM::HESS_NUU + MM::HESS_NUX; else return 0; }
    static const ilqr_model_vtable* vtable() {
        static const ilqr_model_vtable vt = {ILQR_MODEL_ABI_VERSION, (int)sizeof(KArgs),
                                             M::NAME, M::NX, M::NU, M::NW, M::NCS, M::NCT, M::INEQ_S, M::INEQ_T,
                                             &launch_solve, &launch_stage, &launch_init,
                                             is_large<M>::value ? nullptr : &launch_solve_slim,
                                             is_large<M>::value ? nullptr : &launch_stage_slim,
                                             packed_ok<M>::value ? &launch_solve_packed : nullptr,
                                             jac_nvar(), hess_nnz(), is_large<M>::value ? &launch_mirror : nullptr,
                                             mid_ok<M>::value ? &launch_solve_mid : nullptr, mid_ok<M>::value ? &launch_stage_mid : nullptr,
                                             packed2_lds<M>()};
        return &vt;
    }
};
}  // namespace ilqr

#define ILQR_DEFINE_MODEL(MODEL)                                                              \
    namespace {                                                                               \
    struct Registrar_##MODEL {                                                                \
        Registrar_##MODEL() { ilqr_register_model(ilqr::ModelModule<MODEL>::vtable()); }      \
    } registrar_##MODEL;                                                                      \
    }
