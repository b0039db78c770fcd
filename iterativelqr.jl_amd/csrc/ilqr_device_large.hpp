// Large-model path (nx > 4 or nu > 4, up to nx = 64 — one state component per lane — and nu = 16), e.g. BASELINE config "synth32".
//
// One problem instance per workgroup of FOUR waves (two instances per CU: two waves per SIMD, 256 VGPRs each). The per-instance
// workspace stays in HBM; what the solve streams per timestep is COMPACT:
//   * Jacobians: entries that do not depend on (x, u, θ) — 1248 of 1280 for synth32 — never touch HBM: the generated tables
//     M::JAC_CONST_* are put into LDS once per Riccati pass and stay there; only the M::JAC_NVAR state-dependent entries are
//     evaluated (M::dyn_jac_var), stored per timestep (Layout::fv) and patched into the LDS copy step by step;
//   * accumulated cost Hessians (reference quirk Q1): one row per timestep with the structurally non-zero entries only
//     (Layout::hc, order M::HESS_IDX: [gxx sorted by 16x16 tile | guu | gux]); the Riccati step adds them to Qxx, Quu, Qux
//     where they fall — bitwise what adding the dense arrays gives, zeros contribute nothing;
//   * the full jacobian_* / hessian_* arrays of the reference are a host-visible mirror, written on demand by
//     mirror_large_kernel and read back into the compact form by it after a host write.
// Riccati step (src/backward_pass.jl:42-90): 136 v_mfma_f64_16x16x4_f64 for n = 32, m = 8 around the serial Cholesky chain,
// scheduled statically (RicSchedule) over the four waves in windows that end in one workgroup barrier each, and compiled into one
// straight-line instruction stream per wave (tile coordinates are immediates). For 17 <= nx <= 32 (round 5: THREE windows):
//     B   waves 0,1: Qux = ûx fx + gux    wave 2: Quu = ûx fu + guu      wave 3: T(0,0) = fxᵀP′
//         (row nu of ûx is p′ᵀ: row nu of the Qux / Quu tiles is Qx - gx / Qu - gu — the step's two matrix-vector products for free)
//     C   wave 0: potrf and potrs fused in registers → K, k; p, ∇L       waves 1-3: T(0,1), T(1,.) (flags in LDS), Qxx = T fx + gxx
//     D   every wave one tile of P = Kᵀûxt + KᵀQux + QuxᵀK + Qxx and, straight from that tile's registers, its share of the NEXT
//         step's ûx = fuᵀP′ (ds_add_f64 into the zeroed buffer: two contributions per element); the next step's operands go
//         into LDS in its shadow
//     other sizes keep the four-window schedule: A  ûx = fuᵀP′ | T   B  Qux, Quu | T, (Qu, Qx)   C  chain | T, Qxx   D  P
// Operand fragments are read from zero-padded LDS matrices with odd leading dimensions (transposition = stride pattern, no
// bounds checks), all fragments of a tile before its first MFMA, the MFMAs of a tile back to back.
// Forward sweep: wave 0 runs the closed-loop rollout, wave 1 the sensitivity recursion Δz with ∇Lᵀ·Δz, waves 2,3 stage K_t, a_t,
// b_t into an LDS ring a chunk ahead; a_t = α k_t + ū_t and b_t = K_t x̄_t are formed beforehand time-parallel. Matrix-vector
// products use every lane: K x as eight (nu <= 8) or four partial sums per action, a dynamics / sensitivity row as two halves
// on lanes i and i + 32 (nx <= 32), joined by DPP / permlane moves; actions reach all lanes by DPP row broadcasts.
// gradients!: state-dependent Jacobian entries one (timestep, entry) pair per thread where they are elementwise
// (M::JAC_VAR_ELEMENTWISE), else on waves 0, 1 beside the cost / AL terms on waves 2, 3.
// Same reference semantics and citations as ilqr_device.hpp.
#pragma once
#include "ilqr_ric_schedule.hpp"

namespace ilqr {

constexpr int r16(int v) { return (v + 15) & ~15; }
constexpr int r4(int v) { return (v + 3) & ~3; }

template <class M>
struct LargeDims {
    static constexpr int n = M::NX, m = M::NU, W = waves_of<M>::value, NT = 64 * W;      // W = 4, or 1 for the one-wave variant (Mid<M>)
    static constexpr int NP = r16(n), MP = r16(m);          // whole 16x16 tiles; the padding is kept at zero
    static constexpr int TN = NP / 16;
    static constexpr int ld = NP + 1, ldm = MP + 1;         // odd leading dimensions: conflict-free LDS column walks
    static constexpr int n4 = r4(n), m4 = r4(m);
    static constexpr int JV = M::JAC_NVAR, JVP = pad2(JV > 0 ? JV : 1);
    static constexpr int HXX = M::HESS_NXX, HUU = M::HESS_NUU, HUX = M::HESS_NUX, HS = HXX + HUU + HUX, HSP = pad2(HS > 0 ? HS : 1);
    // LDS carve (doubles); must match large_lds_doubles() of ilqr_layout.hpp
    static constexpr int oFx = 0, oFu = oFx + NP * ld, oP = oFu + MP * ld, oT = oP + NP * ld, oUh = oT + NP * ld,
                         oQux = oUh + NP * ldm, oK = oQux + (NP + 1) * ldm, oGux = oK + (NP + 1) * ldm, oQuu = oGux + NP * ldm,
                         oGuu = oQuu + MP * ldm, oChol = oGuu + MP * ldm, oVec = oChol + LARGE_CHOL, oLay = oVec + 2 * NP + 8,      // Qux, K: one more column for Qu, k
                         oQxx = oLay + LAYOUT_LDS_DOUBLES, QXB = TN == 2 ? NP * ld : 0,                      // Qxx's own buffer (TN = 2 schedule)
                         oStg = oQxx + QXB, STG = large_stage_doubles(n, m, HS), total = oStg + STG;
    static constexpr bool STAGE = STG > 0;                  // next step's compact Hessian row and cost gradients staged in LDS
    // forward sweep: fx, fu keep their place (sensitivity recursion); behind them the sweep's vectors and the K ring
    static constexpr int oFw = oP, oRing = oFw + 2 * NP + 2 * MP, ringDoubles = oVec - oRing;
    static constexpr int RSTEP = m * n + 2 * m;             // ring entry of a timestep: K_t, a_t, b_t
    static constexpr int CH = ringDoubles / (2 * RSTEP) < 32 ? ringDoubles / (2 * RSTEP) : 32;     // timesteps per ring half
    static_assert(CH >= 1, "LDS ring of the forward sweep holds at least one timestep of K per half");
};

typedef double double4_t __attribute__((ext_vector_type(4)));

// D-layout of a tile: element r of lane (li, lk) is (row I0 + lk + 4r, column J0 + li)
template <int LDD>
__device__ __forceinline__ void tile_store(double* D, double4_t acc, int I0, int J0, int li, int lk) {
    double* p = D + (J0 + li) * LDD + I0 + lk;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[4 * r] = acc[r];
}

template <int LDD, class PTR>         // PTR: const double* or its LDS-qualified form
__device__ __forceinline__ double4_t tile_load(PTR D, int I0, int J0, int li, int lk) {
    const auto p = D + (J0 + li) * LDD + I0 + lk;
    double4_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = p[4 * r];
    return v;
}

// One 16x16 output tile on v_mfma_f64_16x16x4_f64: acc += sum_{k<KD} A(i,k) B(k,j), A(i,k) at Ab[AI*i + AK*k], B(k,j) at
// Bb[BK*k + BJ*j] (i, j < 16: the caller offsets Ab / Bb to the tile; KD and the strides are compile-time; operands are
// zero-padded, so there are no bounds checks). All fragments are read first, then the MFMAs issue back to back.
// A[i][k]: i = lane&15, k = lane>>4; B[k][j]: k = lane>>4, j = lane&15; C/D[i][j]: j = lane&15, i = (lane>>4) + 4*reg.
template <int KD, int AI, int AK, int BK, int BJ>
__device__ __forceinline__ double4_t tile_mm(const double* Ab, const double* Bb, int li, int lk, double4_t acc = double4_t{0, 0, 0, 0}) {
    constexpr int KS = KD / 4;
    double fa[KS], fb[KS];
    const double* pa = Ab + AI * li + AK * lk;
    const double* pb = Bb + BK * lk + BJ * li;
#pragma unroll
    for (int s = 0; s < KS; ++s) { fa[s] = pa[AK * 4 * s]; fb[s] = pb[BK * 4 * s]; }
    __builtin_amdgcn_sched_barrier(0);        // all fragment reads in flight before the first MFMA: one LDS round trip per tile, not one per k-step
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[s], acc, 0, 0, 0);
    return acc;
}

// Phases are real (noinline) functions: each gets its own register allocation (the dense 32-state model code inlined next to
// the Riccati loop once pushed that loop into hundreds of spills). Pointers cross the call with their address space attached
// (a plain double* would turn every access into a flat_load); LDS is re-derived from the dynamic shared symbol.
typedef __attribute__((address_space(1))) double gdbl;
template <class T> __device__ __forceinline__ gdbl* as_global(T* p) { return (gdbl*)p; }
// LDS traffic of ONE wave is in order: a wave-local exchange between lanes needs no workgroup barrier
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ gdbl* uniform_ptr(gdbl* p) {
    unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gdbl*)(((unsigned long long)hi << 32) | lo);
}
// What the phase functions need of one instance. Only the instance's block pointer crosses the call; the Layout is read back
// from the tail of the dynamic LDS, where the kernel parked it (store_layout_lds), and the pointers are rebuilt wave-uniform.
struct LargeArgs {
    gdbl *xb, *ub, *x, *u, *gx, *gu, *K, *k, *Lx, *Lu, *c, *lam, *rho, *act, *w, *P, *p, *scal, *fv, *hc, *ab;
    int T, N;
};
template <class M>
__device__ __forceinline__ void store_layout_lds(const Layout& L) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    int* dst = reinterpret_cast<int*>(lds_dyn + LargeDims<M>::oLay);
    int i = 0;
#define ILQR_X(f) dst[i++] = L.f;
    ILQR_LAYOUT_FIELDS(ILQR_X)
#undef ILQR_X
}
template <class M>
__device__ __forceinline__ LargeArgs large_args_from_lds(gdbl* base) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const int* src = reinterpret_cast<const int*>(lds_dyn + LargeDims<M>::oLay);
    Layout L;
    int i = 0;
#define ILQR_X(f) L.f = __builtin_amdgcn_readfirstlane(src[i++]);
    ILQR_LAYOUT_FIELDS(ILQR_X)
#undef ILQR_X
    gdbl* g = uniform_ptr(base);
    return LargeArgs{g + L.xb, g + L.ub, g + L.x, g + L.u, g + L.gx, g + L.gu, g + L.K, g + L.k, g + L.Lx, g + L.Lu,
                     g + L.c, g + L.lam, g + L.rho, g + L.act, g + L.w, g + L.P, g + L.p, g + L.scal, g + L.fv, g + L.hc, g + L.ab,
                     L.T, L.T - 1};
}

// ---------------------------------------------------------------- cost! (one timestep per thread of the workgroup)
// cost_pass of ilqr_device.hpp for the four-wave workgroup: every timestep is evaluated by exactly one thread; the per-lane
// partial sums of the waves are combined through LDS in wave order (the order a single wave walking t = lane, lane + 64, ...
// would add them in), then summed over the lanes.
template <class M>
__attribute__((noinline)) __device__ double cost_pass_large_fn(gdbl* base, int at_states, int upd_J, int upd_viol, int constrained,
                                                               double* viol_out) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT, NT = LargeDims<M>::NT, W = LargeDims<M>::W;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const LargeArgs A = large_args_from_lds<M>(base);
    at_states = __builtin_amdgcn_readfirstlane(at_states); upd_J = __builtin_amdgcn_readfirstlane(upd_J);
    upd_viol = __builtin_amdgcn_readfirstlane(upd_viol); constrained = __builtin_amdgcn_readfirstlane(constrained);
    const gdbl* X = at_states ? A.x : A.xb;
    const gdbl* U = at_states ? A.u : A.ub;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), T = A.T, N = A.N;
    double Jp = 0.0, vp = 0.0;
    for (int t = tid; t < T; t += NT) {
        double Jt = 0.0;                                  // this timestep's terms first: the sum then does not depend on how many waves share the walk
        double w[cdim<M::NW>::v];
        load_w<M::NW>((const double*)A.w, t, w);
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = X[t * n + i];
        if (t < N) {
            double ut[m];
#pragma unroll
            for (int i = 0; i < m; ++i) ut[i] = U[t * m + i];
            if (upd_J) Jt += M::cost_s(xt, ut, w);
            if constexpr (ncs > 0) {
                if (constrained) {
                    double cv[ncs];
                    M::con_s(xt, ut, w, cv);
                    const int off = t * ncs;
                    if (upd_J) {
                        double dot = 0.0, pen = 0.0;
#pragma unroll
                        for (int i = 0; i < ncs; ++i) {
                            const double lam = A.lam[off + i];
                            const bool ineq = IneqMask<M>::s(i);
                            const bool inactive = ineq && cv[i] < 0.0 && lam == 0.0;
                            A.act[off + i] = inactive ? 0.0 : 1.0;
                            dot += lam * cv[i];
                            if (!inactive) pen += 0.5 * A.rho[off + i] * (cv[i] * cv[i]);
                        }
                        Jt += dot;
                        Jt += pen;
                    }
                    if (upd_viol) {
#pragma unroll
                        for (int i = 0; i < ncs; ++i) {
                            A.c[off + i] = cv[i];
                            const bool ineq = IneqMask<M>::s(i);
                            vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                        }
                    }
                }
            }
        } else {
            if (upd_J) Jt += M::cost_t(xt, w);
            if constexpr (nct > 0) {
                if (constrained) {
                    double cv[nct];
                    M::con_t(xt, w, cv);
                    const int off = N * ncs;
                    if (upd_J) {
                        double dot = 0.0, pen = 0.0;
#pragma unroll
                        for (int i = 0; i < nct; ++i) {
                            const double lam = A.lam[off + i];
                            const bool ineq = IneqMask<M>::t(i);
                            const bool inactive = ineq && cv[i] < 0.0 && lam == 0.0;
                            A.act[off + i] = inactive ? 0.0 : 1.0;
                            dot += lam * cv[i];
                            if (!inactive) pen += 0.5 * A.rho[off + i] * (cv[i] * cv[i]);
                        }
                        Jt += dot;
                        Jt += pen;
                    }
                    if (upd_viol) {
#pragma unroll
                        for (int i = 0; i < nct; ++i) {
                            A.c[off + i] = cv[i];
                            const bool ineq = IneqMask<M>::t(i);
                            vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                        }
                    }
                }
            }
        }
        Jp += Jt;
    }
    // combine the waves: every wave ends with the same two numbers (identical control flow afterwards)
    double* comb = lds_dyn;                               // 2 x W x 64 doubles at the head of the staging area (dead between phases)
    comb[wave * 64 + lane] = Jp;
    comb[(W + wave) * 64 + lane] = vp;
    __syncthreads();
    double Js = 0.0, vs = 0.0;
#pragma unroll
    for (int q = 0; q < W; ++q) { Js += comb[q * 64 + lane]; vs = nanmax(vs, comb[(W + q) * 64 + lane]); }
    const double J = wave_sum(Js);
    *viol_out = wave_max(vs);
    __syncthreads();
    return J;
}
template <class M>
__device__ __forceinline__ void cost_pass_large(Inst<M>& I, bool at_states, bool upd_J, bool upd_viol, bool constrained,
                                                double& J_out, double& viol_out) {
    ILQR_PROF_BEGIN();
    double v = 0.0;
    J_out = cost_pass_large_fn<M>(as_global(I.gbase), at_states ? 1 : 0, upd_J ? 1 : 0, upd_viol ? 1 : 0, constrained ? 1 : 0, &v);
    viol_out = v;
    ILQR_PROF_END(I, PROF_COST);
}

// ---------------------------------------------------------------- gradients! (one timestep per thread)
// State-dependent Jacobian entries → fv (`.=`, src/dynamics.jl:45-46), cost gradients (`.=`, src/costs.jl:61,65), cost
// Hessians and Gauss-Newton AL terms ACCUMULATED (`.+=`, src/costs.jl:74-80, src/gradients.jl:54-80) into the compact row hc[t].
template <class M>
__attribute__((noinline)) __device__ void gradients_large_fn(gdbl* base, int constrained) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT, NT = LD::NT;
    const LargeArgs A = large_args_from_lds<M>(base);
    constrained = __builtin_amdgcn_readfirstlane(constrained);
    const int tid = threadIdx.x, T = A.T, N = A.N;
    // A timestep's linearisation is one long serial instruction stream per thread (4.4 k instructions for synth32, two thirds of
    // them the 32 cosines of the state-dependent Jacobian entries), so it is cut along its function boundaries.
    // Where every state-dependent Jacobian entry is the same function of ONE state component (M::JAC_VAR_ELEMENTWISE), those
    // entries are evaluated one (timestep, entry) pair per thread by all four waves, coalesced; the cost gradients, the accumulated
    // Hessian row and the Gauss-Newton AL terms follow, one timestep per thread. Otherwise waves 0, 1 evaluate the Jacobian entries
    // of timestep t while waves 2, 3 do the rest of the same t (wave-uniform roles: no divergence; both halves read x̄_t, ū_t).
    // Hessians accumulate: each timestep exactly once.
    constexpr bool JE = M::JAC_VAR_ELEMENTWISE;
    constexpr bool ONE = LD::W == 1;                       // one wave: no roles, every thread does both halves of its timestep
    constexpr int NH = (JE || ONE) ? NT : NT / 2;
    const int half = (JE || ONE) ? 1 : __builtin_amdgcn_readfirstlane(tid / NH);
    if constexpr (JE) {
        constexpr int JV = LD::JV;
        for (int e = tid; e < N * JV; e += NT) {
            const int t = e / JV, q = e - t * JV;
            double w[cdim<M::NW>::v];
            load_w<M::NW>((const double*)A.w, t, w);
            A.fv[(size_t)t * LD::JVP + q] = M::dyn_jac_var_own(A.xb[t * n + M::JAC_VAR_SRC[q]], w) + M::JAC_VAR_ADD[0][q];
        }
    }
    for (int t = tid % NH; t < T; t += NH) {
        double w[cdim<M::NW>::v];
        load_w<M::NW>((const double*)A.w, t, w);
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = A.xb[t * n + i];
        if constexpr (!JE) {
            if (ONE || half == 0) {
                if (t < N) {
                    double ut[m];
#pragma unroll
                    for (int i = 0; i < m; ++i) ut[i] = A.ub[t * m + i];
                    double v[cdim<LD::JV>::v];
                    v[0] = 0.0;
                    M::dyn_jac_var(xt, ut, w, v);
#pragma unroll
                    for (int q = 0; q < LD::JV; ++q) A.fv[(size_t)t * LD::JVP + q] = v[q];
                }
                if (!ONE) continue;
            }
        }
        double* hrow = (double*)(A.hc + (size_t)t * LD::HSP);
        if (t < N) {
            double ut[m];
#pragma unroll
            for (int i = 0; i < m; ++i) ut[i] = A.ub[t * m + i];
            double gx[n], gu[m];
            M::cost_s_grad(xt, ut, w, gx, gu);
            M::cost_s_hess_c(xt, ut, w, hrow);
            if constexpr (ncs > 0) {
                if (constrained) {                                                            // src/gradients.jl:54-80
                    double ct[ncs], ir[ncs];
                    const int off = t * ncs;
#pragma unroll
                    for (int i = 0; i < ncs; ++i) {
                        ir[i] = A.rho[off + i] * A.act[off + i];
                        ct[i] = A.lam[off + i] + ir[i] * A.c[off + i];
                    }
                    M::al_s_c(xt, ut, w, ct, ir, gx, gu, hrow);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) A.gx[t * n + i] = gx[i];
#pragma unroll
            for (int i = 0; i < m; ++i) A.gu[t * m + i] = gu[i];
        } else {
            double gx[n];
            M::cost_t_grad(xt, w, gx);
            M::cost_t_hess_c(xt, w, hrow);
            if constexpr (nct > 0) {
                if (constrained) {
                    double ct[nct], ir[nct];
                    const int off = N * ncs;
#pragma unroll
                    for (int i = 0; i < nct; ++i) {
                        ir[i] = A.rho[off + i] * A.act[off + i];
                        ct[i] = A.lam[off + i] + ir[i] * A.c[off + i];
                    }
                    M::al_t_c(xt, w, ct, ir, gx, hrow);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) A.gx[t * n + i] = gx[i];
        }
    }
    if (tid == 0) A.scal[S_JAC_VALID] = 1.0;
    __syncthreads();
}
template <class M>
__device__ __forceinline__ void gradients_large(Inst<M>& I, bool constrained) {
    ILQR_PROF_BEGIN();
    gradients_large_fn<M>(as_global(I.gbase), constrained ? 1 : 0);
    ILQR_PROF_END(I, PROF_GRAD);
}

// reset!(problem.model); reset!(problem.objective) (src/solve.jl:9-10) on the compact representation: the Jacobians are
// overwritten (`.=`) by the gradients! call that follows at once, so only the accumulating Hessian rows and the gradients
// are zeroed. literal = true (the stage entry point) also forgets the Jacobians.
template <class M>
__device__ __forceinline__ void reset_model_objective_large(Inst<M>& I, bool literal) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, NT = LD::NT;
    const int tid = threadIdx.x;
    double* hc = I.gbase + I.hc_off;
    for (int i = tid; i < I.T * LD::HSP; i += NT) hc[i] = 0.0;
    for (int i = tid; i < I.T * n; i += NT) I.gx[i] = 0.0;
    for (int i = tid; i < I.N * m; i += NT) I.gu[i] = 0.0;
    if (literal) {
        double* fv = I.gbase + I.fv_off;
        for (int i = tid; i < I.N * LD::JVP; i += NT) fv[i] = 0.0;
        if (tid == 0) I.scal[S_JAC_VALID] = 0.0;
    }
    __syncthreads();
}

// v + (v of the partner row of the 16-lane row pair (0,1), (2,3)) and v + (v of the other 32-lane half): gfx950 v_permlane16_swap /
// v_permlane32_swap, no LDS crossbar trip. swap(v, v) = {[r0 r0 r2 r2], [r1 r1 r3 r3]} resp. {[lo lo], [hi hi]}, so every lane
// adds the two partial sums in the same order.
__device__ __forceinline__ double sum_row_pairs(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double sum_halves(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double sum_quarters(double v) { return sum_halves(sum_row_pairs(v)); }   // (r0 + r1) + (r2 + r3)

// potrf('U') on the common path, column c = lane & 15 of the matrix on every lane of the four 16-lane rows (four identical
// copies): the entries of column j a lane needs arrive by ONE v_mov_b64_dpp row_newbcast each and stay in VGPRs; no compare
// steers the pivots — the factorisation runs through and reports whether every pivot was positive. If not (rare: diverged
// instances) the caller repeats it with potrf_U_lanes, which reproduces what dpotf2 leaves behind after a failure.
template <int m, int J = 0>
__device__ __forceinline__ void potrf_U_rows_step(double (&a)[m], double (&R)[m], int col, bool& bad) {
    if constexpr (J < m) {
        double v = a[J];
#pragma unroll
        for (int l = 0; l < J; ++l) v -= row_bcast<J>(a[l]) * a[l];             // A(j,c) - sum_l U(l,j) U(l,c); on lane j: the pivot
        const double ajj = row_bcast<J>(v);
        bad = bad || !(ajj > 0.0);
        double d, r;
        sqrt_rsqrt_fast(ajj, d, r);
        a[J] = (col == J) ? d : v * r;
        R[J] = r;
        potrf_U_rows_step<m, J + 1>(a, R, col, bad);
    }
}
template <int m>
__device__ __forceinline__ bool potrf_U_rows(double (&a)[m], double (&R)[m], int col) {
    bool bad = false;
    potrf_U_rows_step<m, 0>(a, R, col, bad);
    return bad;
}
// out[i] = v of lane i of the 16-lane row, for every i < m (four identical rows: the same on every lane)
template <int m, int I = 0>
__device__ __forceinline__ void bcast_all(double v, double (&out)[m]) {
    if constexpr (I < m) {
        out[I] = row_bcast<I>(v);
        bcast_all<m, I + 1>(v, out);
    }
}
template <int m, int NR, int I, int L>
__device__ __forceinline__ void back_terms(const double (&a)[m], const double (&b)[NR][m], double (&w)[NR]) {
    if constexpr (L < m) {
        const double u = row_bcast<L>(a[I]);                                 // U(I,L) from the lanes that hold column L
#pragma unroll
        for (int q = 0; q < NR; ++q) w[q] -= u * b[q][L];
        back_terms<m, NR, I, L + 1>(a, b, w);
    }
}
// potrf('U') and potrs('U') in one sequence, the factor never leaving the registers: column c = lane & 15 of Quu on every lane
// (a, as in potrf_U_rows) and NR right-hand sides per lane (b). The entries U(l,J), l < J, that pivot step J broadcasts for the
// factorisation are exactly the ones row J of the forward substitution multiplies with, so that row rides along with its pivot
// step (and fills the issue slots its dependent chain leaves empty); the back substitution re-broadcasts them (one DPP move
// each). Same arithmetic, operation for operation, as potrf_U_rows + potrs_U_lds; the diagonal of the factor is not formed
// (nobody reads it: the solves use R = 1 / diag). Returns whether a pivot was not positive (the caller then repeats the step
// on the careful path).
template <int m, int NR, int J = 0>
__device__ __forceinline__ void chol_solve_rows_fwd(double (&a)[m], double (&R)[m], double (&b)[NR][m], bool& bad) {
    if constexpr (J < m) {
        double v = a[J], w[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) w[q] = b[q][J];
#pragma unroll
        for (int l = 0; l < J; ++l) {
            const double u = row_bcast<J>(a[l]);                             // U(l,J), the same on every lane
            v -= u * a[l];                                                   // A(J,c) - sum_l U(l,J) U(l,c)
#pragma unroll
            for (int q = 0; q < NR; ++q) w[q] -= u * b[q][l];                // row J of U^T y = b
        }
        const double ajj = row_bcast<J>(v);
        bad = bad || !(ajj > 0.0);
        const double r = rsqrt_fast(ajj);
        a[J] = v * r;
        R[J] = r;
#pragma unroll
        for (int q = 0; q < NR; ++q) b[q][J] = w[q] * r;
        chol_solve_rows_fwd<m, NR, J + 1>(a, R, b, bad);
    }
}
template <int m, int NR, int I>
__device__ __forceinline__ void chol_solve_rows_back(const double (&a)[m], const double (&R)[m], double (&b)[NR][m]) {
    if constexpr (I >= 0) {
        double w[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) w[q] = b[q][I];
        back_terms<m, NR, I, I + 1>(a, b, w);
#pragma unroll
        for (int q = 0; q < NR; ++q) b[q][I] = w[q] * R[I];
        chol_solve_rows_back<m, NR, I - 1>(a, R, b);
    }
}
template <int m, int NR>
__device__ __forceinline__ bool chol_solve_rows(double (&a)[m], double (&R)[m], double (&b)[NR][m]) {
    bool bad = false;
    chol_solve_rows_fwd<m, NR, 0>(a, R, b, bad);
    chol_solve_rows_back<m, NR, m - 1>(a, R, b);
    return bad;
}
// potrf('U') with column c of the matrix on lane c (a[i] = A(i, c), i <= c): dpotf2 order, element for element the arithmetic of
// potrf_U (ilqr_device.hpp) — there every lane repeats the whole factorisation (m^3/3 FMAs on the serial chain), here a lane
// updates its own column and the entries of column j it needs arrive by v_readlane: 3 instructions per (pivot, row) pair
// instead of one per (pivot, row, column) triple. R[j] = 1 / U(j,j), wave-uniform. Returns LAPACK's info.
template <int m>
__device__ __forceinline__ int potrf_U_lanes(double (&a)[m], double (&R)[m], int lane) {
    int info = 0;
#pragma unroll
    for (int j = 0; j < m; ++j) {
        double v = a[j];
#pragma unroll
        for (int l = 0; l < j; ++l) v -= lane_bcast(a[l], j) * a[l];        // A(j,c) - sum_l U(l,j) U(l,c); on lane j: the pivot
        const double ajj = lane_bcast(v, j);
        const bool ok = (info == 0) && (ajj > 0.0);
        const bool fail_now = (info == 0) && !(ajj > 0.0);
        info = fail_now ? j + 1 : info;
        double d, r;
        sqrt_rsqrt_fast(ok ? ajj : 1.0, d, r);
        if (__builtin_expect(!ok, 0)) {                                       // rare (diverged instances): what potrf_U leaves behind
            d = fail_now ? ajj : lane_bcast(a[j], j);
            r = 1.0 / d;
        }
        a[j] = (lane == j) ? d : (ok ? v * r : a[j]);
        R[j] = r;
    }
    return info;
}
// potrs('U') with the factor in LDS (column c at U + c m, broadcast reads) and the inverted diagonal R; one right-hand side per lane
template <int m>
__device__ __forceinline__ void potrs_U_lds(const double* U, const double (&R)[m], double (&b)[m]) {
#pragma unroll
    for (int i = 0; i < m; ++i) {
        double v = b[i];
#pragma unroll
        for (int l = 0; l < i; ++l) v -= U[i * m + l] * b[l];
        b[i] = v * R[i];
    }
#pragma unroll
    for (int i = m - 1; i >= 0; --i) {
        double v = b[i];
#pragma unroll
        for (int l = i + 1; l < m; ++l) v -= U[l * m + i] * b[l];
        b[i] = v * R[i];
    }
}

// ---------------------------------------------------------------- backward_pass! (MFMA 16x16x4 tiles, four waves)
// Static schedule of the tiles of one Riccati step over (window, wave). Windows end in one workgroup barrier each:
//   A  ûx = fuᵀP′ (TN tiles, what the chain waits for)          B  Qux = ûx fx (TN), Quu = ûx fu (1)
//   C  wave 0: the potrf / potrs chain; others Qxx = T fx + gxx  D  P tiles (waves 1..3); wave 0: p, ∇L
// T = fxᵀP′ (NQ = TN² tiles, needed by Qxx only) fills the idle slots of A and B and, for what is left, the head of C.
// TN = 2 (17 <= nx <= 32, e.g. synth32) is scheduled by hand so that no wave has more than one tile in A or in B:
//   A: w0 ûx0, w1 ûx1, w2 T(0,0), w3 T(0,1)   B: w0 Qux0, w1 Qux1 (+ Qu), w2 Quu, w3 T(1,0)
//   C: w1 Qx, Qxx(0,0); w2 Qxx(0,1); w3 T(1,1), Qxx(1,0), Qxx(1,1)   D: w1 P0, P3; w2 P1; w3 P2
// (the ninth tile of windows A + B has no free slot and lands in C on wave 3, whose three tiles are about as long as the chain.
// Forming that tile twice — on waves 1 and 3, two tiles each in C — was measured: the chain no longer waits 0.45 k clk per
// timestep, but the eight extra MFMAs cost more when both instances of a CU are active: 3.83 -> 3.94 ms on BASELINE C5.)
// Other sizes: critical tiles alternate over waves 0, 1; T tiles over waves 2, 3 (half in A, half in B); Qxx / P over waves 1..3.
// compile-time loop and a switch on the (uniform) wave index that hands the index over as a constant: the Riccati step's static
// schedule is compiled into four straight-line instruction streams, one per wave, with every tile coordinate an immediate
template <int V> struct IntC { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IntC<I>{}); static_for<I + 1, N>(f); }
}
template <class F>
__device__ __forceinline__ void wave_switch(int wave, F&& f) {
    switch (wave) {
        case 0: f(IntC<0>{}); break;
        case 1: f(IntC<1>{}); break;
        case 2: f(IntC<2>{}); break;
        default: f(IntC<3>{}); break;
    }
}

// the four roles of a window: one per wave, or all of them in turn on the instance's only wave (one-wave variant: every tile list
// of a window is independent of the others', and windows stay separated by the barriers, which a single wave passes at once)
template <int NWAVES, class F>
__device__ __forceinline__ void role_switch(int wave, F&& f) {
    if constexpr (NWAVES == 1) static_for<0, 4>(f);
    else wave_switch(wave, f);
}

struct RiccatiOut {
    double gradient_norm; int potrf_info;
#if defined(ILQR_PROFILE) && defined(ILQR_PROFILE_SUB)
    double prof[6];
#endif
};

template <class M, bool STORE_VALUE>
__attribute__((noinline)) __device__ RiccatiOut backward_pass_large_fn(gdbl* base, gdbl* Qbase) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, NP = LD::NP, ld = LD::ld, ldm = LD::ldm, TN = LD::TN, NT = LD::NT;
    constexpr int n4 = LD::n4, m4 = LD::m4, JV = LD::JV, JVP = LD::JVP, HXX = LD::HXX, HUU = LD::HUU, HUX = LD::HUX, HSP = LD::HSP;
    constexpr int NWV = LD::W;                             // 4, or 1: the four roles of a window run one after the other on the instance's only wave
    constexpr int NS = NWV == 1 ? 64 : NT - 64;            // threads that fetch the next step's operands: waves 1..3 (wave 0 only stores)
    constexpr int FOFF = NWV == 1 ? 0 : 64;                // ... and the first of them
    constexpr int EJ = (JV + NS - 1) / NS > 0 ? (JV + NS - 1) / NS : 1;            // Jacobian patch entries per fetching thread
    constexpr int EU = (HUU + HUX + 63) / 64 > 0 ? (HUU + HUX + 63) / 64 : 1;      // guu / gux entries per lane of wave 0
    typedef RicSchedule<TN> RS;
    constexpr int SLOTS = RS::SLOTS;                       // Qxx / P tiles per wave (waves 1..3)
    constexpr bool PAIRED = NWV == 1;                      // one wave: independent tiles of a window share their fragment reads and interleave their MFMAs
    // Three windows per step (RicSchedule<2>, 17 <= nx <= 32 on four waves): window A is merged into window D — every wave forms its
    // share of the next step's ûx from the P tile it holds in registers (partial sum over the tile's rows of P′) and ADDS it into
    // the zeroed ûx buffer with ds_add_f64: two contributions per element, a + b = b + a bitwise, so the order the waves arrive in
    // does not show
    constexpr bool XA = TN == 2 && NWV == 4;
    // ... and the step's two matrix-vector products ride in the padding of ûx's tiles: p′ᵀ is row nu of ûx, so row nu of Qux = ûx fx
    // is (fxᵀp′)ᵀ = Qx - gx and row nu of Quu = ûx fu is (fuᵀp′)ᵀ = Qu - gu (src/backward_pass.jl:44-49 as part of the products of :57-64)
    constexpr bool VPAD = XA && m < LD::MP;
    constexpr int VR = m / 4, VK = m % 4;                  // row nu of a tile: register VR of the lanes with lk == VK
    static_assert(n <= 64 && m <= 16, "large path: nx <= 64 (one state component per lane), nu <= 16");
    static_assert(LD::total == large_lds_doubles(n, m, LD::HS), "LDS carve and host-side size disagree");
    static_assert(NWV == LARGE_WAVES || (NWV == 1 && TN == 1), "the Riccati step is scheduled over four waves per instance, or run by one when every matrix is a single tile");
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const LargeArgs A = large_args_from_lds<M>(base);
    struct { double prof[6]; } I;
    for (int q = 0; q < 6; ++q) I.prof[q] = 0.0;
    int potrf_info = 0;
    gdbl* const Qv = (STORE_VALUE && Qbase != nullptr) ? uniform_ptr(Qbase) : nullptr;   // optional action-value buffers (stage kernel only)
    const QLayout QL = make_qlayout(n, m, A.T);
#ifndef ILQR_LARGE_ROT
#define ILQR_LARGE_ROT 2
#endif
    // The roles of the four waves are rotated by two for every other workgroup: the MFMA load of the roles is uneven (20 / 44 / 36 / 36
    // instructions per step in the three-window schedule) and the two instances of a CU share its SIMDs. Measured on
    // synth32_tight11:512, same box, rotation 0 / 1 / 2 / 3: 45.81 / 45.85 / 45.27 / 46.30 ms.
    const int hwv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, wave = NWV == 1 ? hwv : ((hwv + (int)(blockIdx.x & 1) * ILQR_LARGE_ROT) & 3), tid = wave * 64 + lane;
    const int N = A.N, li = lane & 15, lk = lane >> 4;
    double* S = lds_dyn;
    double *sFx = S + LD::oFx, *sFu = S + LD::oFu, *sP = S + LD::oP, *sT = S + LD::oT, *sUh = S + LD::oUh, *sQux = S + LD::oQux,
           *sK = S + LD::oK, *sQuu = S + LD::oQuu;
    double *sp = S + LD::oVec, *sQx = sp + NP, *sOut = sQx + NP;
    double* sQu = sQux + NP * ldm;                                   // Qu and k ride along as column NP of Qux and K
    double* sU = S + LD::oChol;                                            // the Cholesky factor, column c at sU + c m
    constexpr int oQ = LD::QXB > 0 ? LD::oQxx : LD::oP;                    // where Qxx waits for P: its own buffer or P′'s place
    double* sQ = S + oQ;

    // ---- prologue: zero padding, constant Jacobian entries, P[H] = gxx[H], p[H] = gx[H]   (:39-40)
    for (int e = tid; e < LD::oVec; e += NT) S[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) sFx[(e / n) * ld + e % n] = M::JAC_CONST_FX[0][e];
    for (int e = tid; e < n * m; e += NT) sFu[(e / n) * ld + e % n] = M::JAC_CONST_FU[0][e];
    for (int q = tid; q < HXX; q += NT) {
        const int idx = M::HESS_IDX[q];
        sP[(idx / n) * ld + idx % n] = A.hc[(size_t)N * HSP + q];
    }
    for (int i = tid; i < n; i += NT) {
        const double v = A.gx[N * n + i];
        sp[i] = v;
        if (STORE_VALUE) A.p[N * n + i] = v;
    }
    // flags of window C, each holding the step whose datum is ready: [4 + w] = what wave w forms of T in that window
    if (tid < 3) sOut[5 + tid] = -1.0;
    // where this thread's entries go (offsets into S; -1 = none)
    int poff[EJ];
#pragma unroll
    for (int j = 0; j < EJ; ++j) {
        const int q = (tid - FOFF) + NS * j;
        poff[j] = -1;
        if (tid >= FOFF && q < JV) {
            const int idx = M::JAC_VAR_IDX[q];
            poff[j] = idx < n * n ? LD::oFx + (idx / n) * ld + idx % n : LD::oFu + ((idx - n * n) / n) * ld + (idx - n * n) % n;
        }
    }
    int uoff[EU];
#pragma unroll
    for (int j = 0; j < EU; ++j) {
        const int q = lane + 64 * j;
        uoff[j] = -1;
        if (q < HUU + HUX) {
            const int idx = M::HESS_IDX[HXX + q];
            uoff[j] = (q < HUU ? LD::oQuu : LD::oQux) + (idx / m) * ldm + idx % m;
        }
    }
    // Qxx tiles of this wave in window C: slot s holds tile RS::tab.qxx[wave][s]; its gxx entries sit at
    // [HESS_XX_TILE_START[q], HESS_XX_TILE_START[q + 1]) of the compact row, lane x of them at +lane (+64, ...)
    constexpr int EXT = 4;                                 // a 16x16 tile has at most 256 entries
    constexpr int XR = NWV == 1 ? 4 : 1;                   // one wave: it plays every role, so it keeps every role's table
    int xoff[XR][SLOTS][EXT];
#pragma unroll
    for (int rr = 0; rr < XR; ++rr) {
        const int role = NWV == 1 ? rr : wave;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int q = RS::tab.qxx[role][s] < 0 ? -1 : (RS::tab.qxx[role][s] & ~RS::RIC_WAIT_T);
            const bool have = q >= 0;
            const int xbeg = have ? M::HESS_XX_TILE_START[have ? q : 0] : 0;
            const int xcnt = have ? M::HESS_XX_TILE_START[(have ? q : 0) + 1] - xbeg : 0;
#pragma unroll
            for (int x = 0; x < EXT; ++x) {
                xoff[rr][s][x] = -1;
                const int e = lane + 64 * x;
                if (e < xcnt) {
                    const int idx = M::HESS_IDX[xbeg + e];
                    xoff[rr][s][x] = oQ + (idx / n) * ld + idx % n;
                }
            }
        }
    }
    __syncthreads();
    if (STORE_VALUE) {
        for (int e = tid; e < n * n; e += NT) A.P[(size_t)N * n * n + e] = sP[(e / n) * ld + e % n];
    }
    // the state-dependent Jacobian entries of the first step
    if (N > 0) {
#pragma unroll
        for (int j = 0; j < EJ; ++j)
            if (poff[j] >= 0) S[poff[j]] = A.fv[(size_t)(N - 1) * JVP + (tid - FOFF) + NS * j];
    }
    // Compact operands of a step: the state-dependent Jacobian entries go straight into the LDS copies of fx, fu (pval), the
    // Hessian row and the cost gradients [hc | gx | gu] into an LDS staging row. Both are requested at the top of the PREVIOUS
    // step by the threads of waves 1..3 alike (thread-indexed; wave 0 only ever stores, so its load counter never makes it wait) and written to LDS at its end, so that no load is in
    // flight across the loop edge and no role's code path can reuse a register another role's load is still aimed at (the
    // in-order load counter would stall it until those loads land).
    constexpr bool STAGE = LD::STAGE;
    constexpr int NG = n + m, ER = (HSP + NG + NS - 1) / NS;
    double *sH = S + LD::oStg, *sG = sH + HSP;
    double pval[EJ], rval[ER];
#pragma unroll
    for (int j = 0; j < EJ; ++j) pval[j] = 0.0;
    // (one load per element through a selected ADDRESS: selecting among three loaded values would need them at once)
    // element e of a thread is element (rstride t + rbase) of the instance block, counted from hc (the three arrays sit in the same
    // block): stride and base are fixed per thread, so a step's address is one multiply-add
    int rbase[ER], rstride[ER];
#pragma unroll
    for (int j = 0; j < ER; ++j) {
        const int e = (tid - FOFF) + NS * j;
        const int gx_rel = (int)(A.gx - A.hc), gu_rel = (int)(A.gu - A.hc);
        rstride[j] = e < HSP ? HSP : (e < HSP + n ? n : m);
        rbase[j] = e < HSP ? e : (e < HSP + n ? gx_rel + (e - HSP) : gu_rel + (e - HSP - n));
        if (!(tid >= FOFF && e < HSP + NG)) rstride[j] = 0;                  // (0 = no element; bases may be negative: gx, gu lie before hc)
    }
    auto stage_load = [&](int t, double (&R)[ER]) {
#pragma unroll
        for (int j = 0; j < ER; ++j) R[j] = rstride[j] > 0 ? (double)A.hc[t * rstride[j] + rbase[j]] : 0.0;
    };
    // where a staged element goes: the gxx entries and the gradients into the staging row; the guu / gux entries into dense
    // matrices Guu, Gux (zero elsewhere, for good: the same positions are rewritten every step) that the waves forming Quu / Qux
    // add to their tiles before storing them — the chain never touches them
    int soff[ER];
#pragma unroll
    for (int j = 0; j < ER; ++j) {
        const int e = (tid - FOFF) + NS * j;
        soff[j] = -1;
        if (tid >= FOFF && e < HSP + NG) {
            soff[j] = LD::oStg + e;
            if (e >= HXX && e < HXX + HUU + HUX) {
                const int idx = M::HESS_IDX[e];
                soff[j] = (e < HXX + HUU ? LD::oGuu : LD::oGux) + (idx / m) * ldm + idx % m;
            }
        }
    }
    // (Guu / Gux go through an LDS-qualified pointer, here and in the tiles' epilogues: with the generic one hipcc merged the Quu
    // and Qux roles' epilogues for nx = 5, nu = 1, left a constant flat -> LDS cast in the merged pointer and died on that cast's
    // own null check, "Illegal instruction detected: Operand has incorrect register class")
    typedef __attribute__((address_space(3))) double ldsd;
    ldsd* const S3 = (ldsd*)lds_dyn;
    auto stage_store = [&](const double (&R)[ER]) {
#pragma unroll
        for (int j = 0; j < ER; ++j)
            if (soff[j] >= 0) S3[soff[j]] = R[j];
    };
    if (STAGE && N > 0) { stage_load(N - 1, rval); stage_store(rval); }
    double gmax = 0.0;
    __syncthreads();
    int tcur = N - 1;                                                  // the step the loop is at (the tiles' epilogues read its gx, gu without a staging row)
    auto run_tiles = [&](auto Wc, auto WINc) {                        // window WIN's tasks of wave W, all compile-time
        constexpr int W = decltype(Wc)::value, WIN = decltype(WINc)::value;
        static_for<0, RS::MAXL>([&](auto Ic) {
            constexpr int task = WIN == 0 ? RS::tab.a[W][decltype(Ic)::value] : WIN == 1 ? RS::tab.b[W][decltype(Ic)::value] : RS::tab.ct[W][decltype(Ic)::value];
            if constexpr (task >= 0) {
                constexpr int kind = task & 0xe0, idx = task & 0x1f;
                if constexpr (kind == RIC_UH) {
                    const double4_t acc = tile_mm<n4, ld, 1, 1, ld>(sFu, sP + ld * 16 * idx, li, lk);
                    tile_store<ldm>(sUh, acc, 0, 16 * idx, li, lk);
                } else if constexpr (kind == RIC_T) {
                    constexpr int a = idx / TN, c = idx % TN;
                    const double4_t acc = tile_mm<n4, ld, 1, 1, ld>(sFx + ld * 16 * a, sP + ld * 16 * c, li, lk);
                    tile_store<ld>(sT, acc, 16 * a, 16 * c, li, lk);
                } else if constexpr (kind == RIC_QUX) {                  // Qux = ûx fx + gux (:63-64)
                    double4_t acc = tile_mm<n4, 1, ldm, 1, ld>(sUh, sFx + ld * 16 * idx, li, lk);
                    const int eo = (16 * idx + li) * ldm + lk;       // ONE element offset for the staged tile and the result, both through the LDS-qualified base
                                                                     // (tile_load on it beside tile_store on the generic pointer computed the offset twice: 1 % of the pass)
                    if constexpr (STAGE) { for (int r4 = 0; r4 < 4; ++r4) acc[r4] += S3[LD::oGux + eo + 4 * r4]; }
                    for (int r4 = 0; r4 < 4; ++r4) S3[LD::oQux + eo + 4 * r4] = acc[r4];
                    if constexpr (VPAD) {                                    // Qx = fxᵀp′ + gx (:44-46): row nu of the tile
                        const int j = 16 * idx + li;
                        if (lk == VK && j < n) sQx[j] = acc[VR] + (STAGE ? sG[j] : (double)A.gx[tcur * n + j]);
                    }
                } else {                                                 // Quu = ûx fu + guu (:58-59)
                    double4_t acc = tile_mm<n4, 1, ldm, 1, ld>(sUh, sFu, li, lk);
                    const int eo = li * ldm + lk;
                    if constexpr (STAGE) { for (int r4 = 0; r4 < 4; ++r4) acc[r4] += S3[LD::oGuu + eo + 4 * r4]; }
                    for (int r4 = 0; r4 < 4; ++r4) S3[LD::oQuu + eo + 4 * r4] = acc[r4];
                    if constexpr (VPAD) {                                    // Qu = fuᵀp′ + gu (:47-49): row nu of the tile
                        if (lk == VK && li < m) sQu[li] = acc[VR] + (STAGE ? sG[n + li] : (double)A.gu[tcur * m + li]);
                    }
                }
            }
        });
    };
    if constexpr (XA) {
        // prologue of the three-window schedule: ûx of the first step from P[H] in LDS (the tiles of the old window A)
        if (N > 0) wave_switch(wave, [&](auto Wc) { run_tiles(Wc, IntC<0>{}); });
        __syncthreads();
        if constexpr (VPAD) {
            for (int i = tid; i < n; i += NT) sUh[i * ldm + m] = sp[i];    // p′ = p[H] as row nu of ûx
            __syncthreads();
        }
    }
    // wave 0 carries the step's critical path in every window (ûx, Qux, the chain, a P tile) and shares its SIMD with a wave of the
    // CU's other instance: its instructions go first whenever they can issue
    if (NWV > 1 && wave == 0) __builtin_amdgcn_s_setprio(3);
    for (int t = N - 1; t >= 0; --t) {                                    // (:42)
        ILQR_SUB_BEGIN();
        tcur = t;
        const int tn = t > 0 ? t - 1 : 0;                                 // (t = 0: a harmless re-read instead of a branch)
        // operands of the NEXT step, requested now (issuing them in the shadow of the window's first tile instead was measured:
        // the window got 450 clk longer)
        bool placed = false;                                              // (one wave: the next step's operands are put into LDS once, not once per role)
        if (NWV == 1 || wave != 0) {                                      // (wave 0 requests nothing: a scalar branch past it all)
            if (STAGE) stage_load(tn, rval);
#pragma unroll
            for (int j = 0; j < EJ; ++j) pval[j] = poff[j] >= 0 ? A.fv[tn * JVP + (tid - FOFF) + NS * j] : 0.0;
        }
        // ------------------------------------------------ window A: ûx = fuᵀP′ (:57) | T = fxᵀP′ (:52)
        if constexpr (PAIRED) {
            // one wave, one tile per product: ûx = fuᵀP′ and T = fxᵀP′ share their B operand. Fragments of both tiles are read once,
            // their MFMAs alternate (two independent accumulators keep the matrix pipe busy where one waits for itself); every
            // accumulator still sees its own products in its own order: bitwise the tiles of the four-wave kernel
            constexpr int KS = n4 / 4;
            double fu_[KS], fx_[KS], fp_[KS];
            const double *pu = sFu + ld * li + lk, *px = sFx + ld * li + lk, *pp = sP + lk + ld * li;
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) { fu_[sx] = pu[4 * sx]; fx_[sx] = px[4 * sx]; fp_[sx] = pp[4 * sx]; }
            __builtin_amdgcn_sched_barrier(0);
            double4_t au = double4_t{0, 0, 0, 0}, at = double4_t{0, 0, 0, 0};
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) {
                au = __builtin_amdgcn_mfma_f64_16x16x4f64(fu_[sx], fp_[sx], au, 0, 0, 0);
                at = __builtin_amdgcn_mfma_f64_16x16x4f64(fx_[sx], fp_[sx], at, 0, 0, 0);
            }
            tile_store<ldm>(sUh, au, 0, 0, li, lk);
            tile_store<ld>(sT, at, 0, 0, li, lk);
        } else if constexpr (!XA) {
            role_switch<NWV>(wave, [&](auto Wc) { run_tiles(Wc, IntC<0>{}); });
        }
        ILQR_SUB_MARK2(I, 0);
        if constexpr (!XA) __syncthreads();                               // (B1) ûx complete (three-window schedule: it was formed in the previous step's window D)
        ILQR_SUB_MARK1(I, 0); ILQR_SUB_MARK2(I, 1);
        // ------------------------------------------------ window B: Qux = ûx fx (:63), Quu = ûx fu (:58), Qu (:47-49) | T
        if constexpr (PAIRED) {
            // Qux = ûx fx + gux and Quu = ûx fu + guu share their A operand; Qxx = T fx + gxx (window C's tile in the four-wave
            // schedule) needs nothing later than T either and goes where P′ was, which nobody reads after window A: three
            // independent accumulators, one LDS round trip
            constexpr int KS = n4 / 4;
            double fh_[KS], ft_[KS], fx_[KS], fu_[KS];
            const double *ph = sUh + li + ldm * lk, *pt = sT + li + ld * lk, *px = sFx + lk + ld * li, *pu = sFu + lk + ld * li;
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) { fh_[sx] = ph[ldm * 4 * sx]; ft_[sx] = pt[ld * 4 * sx]; fx_[sx] = px[4 * sx]; fu_[sx] = pu[4 * sx]; }
            double4_t gux_t = double4_t{0, 0, 0, 0}, guu_t = double4_t{0, 0, 0, 0};
            if constexpr (STAGE) { gux_t = tile_load<ldm>(S3 + LD::oGux, 0, 0, li, lk); guu_t = tile_load<ldm>(S3 + LD::oGuu, 0, 0, li, lk); }
            __builtin_amdgcn_sched_barrier(0);
            double4_t aq = double4_t{0, 0, 0, 0}, auu = double4_t{0, 0, 0, 0}, axx = double4_t{0, 0, 0, 0};
#pragma unroll
            for (int sx = 0; sx < KS; ++sx) {
                aq = __builtin_amdgcn_mfma_f64_16x16x4f64(fh_[sx], fx_[sx], aq, 0, 0, 0);
                auu = __builtin_amdgcn_mfma_f64_16x16x4f64(fh_[sx], fu_[sx], auu, 0, 0, 0);
                axx = __builtin_amdgcn_mfma_f64_16x16x4f64(ft_[sx], fx_[sx], axx, 0, 0, 0);
            }
            if constexpr (STAGE) { aq += gux_t; auu += guu_t; }
            tile_store<ldm>(sQux, aq, 0, 0, li, lk);
            tile_store<ldm>(sQuu, auu, 0, 0, li, lk);
            tile_store<ld>(sQ, axx, 0, 0, li, lk);
            constexpr int QW = 2, xb = M::HESS_XX_TILE_START[0], xc = M::HESS_XX_TILE_START[1] - xb;      // (RicSchedule<1>: the Qxx tile is role 2's)
            static_assert(RS::tab.qxx[QW][0] == 0, "one-wave pairing: Qxx tile of RicSchedule<1>");
            if constexpr (xc > 0) {
                wave_lds_fence();
#pragma unroll
                for (int x = 0; x < (xc + 63) / 64; ++x)
                    if (xoff[QW][0][x] >= 0) S[xoff[QW][0][x]] += STAGE ? sH[xb + lane + 64 * x] : (double)A.hc[(size_t)t * HSP + xb + lane + 64 * x];
            }
            if (STORE_VALUE && Qv != nullptr) {
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = lk + 4 * r, col = li;
                    if (row < n && col < n) Qv[QL.Qxx + (size_t)t * n * n + col * n + row] = sQ[col * ld + row];
                }
            }
        } else {
            role_switch<NWV>(wave, [&](auto Wc) { run_tiles(Wc, IntC<1>{}); });
        }
        // wave 3 (no tile in this window when nx <= 32): the two matrix-vector products of the step
        // (three-window schedule: they come out of the Qux / Quu tiles, see VPAD)
        if (!VPAD && (NWV == 1 || wave == 3)) {
            {
                // Qu = fuᵀp′ + gu (:47-49): action i on lanes i, i + 16, i + 32, i + 48, each a quarter of the sum
                constexpr int JP = (n + 3) / 4;
                const int i = li < m ? li : m - 1;
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < JP; ++q) {
                    const int l = lk * JP + q;
                    if ((n % 4 == 0) || l < n) acc += sFu[i * ld + (l < n ? l : 0)] * sp[l < n ? l : 0];
                }
                acc = sum_quarters(acc);
                if (lane < m) sQu[lane] = acc + (STAGE ? sG[n + lane] : (double)A.gu[t * m + lane]);
                if (STORE_VALUE && Qv != nullptr && lane < m) Qv[QL.Qu + t * m + lane] = sQu[lane];
            }
            {
                // Qx = fxᵀp′ + gx (:44-46): state i on lanes i and i + 32 (half of the sum each) when n <= 32
                if constexpr (n <= 32) {
                    constexpr int JP = (n + 1) / 2;
                    const int i = (lane & 31) < n ? (lane & 31) : n - 1, h = lane >> 5;
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < JP; ++q) {
                        const int l = h * JP + q;
                        if ((n % 2 == 0) || l < n) acc += sFx[i * ld + (l < n ? l : 0)] * sp[l < n ? l : 0];
                    }
                    acc = sum_halves(acc);
                    if (lane < n) sQx[lane] = acc + (STAGE ? sG[lane] : (double)A.gx[t * n + lane]);
                } else {
                    const int i = lane < n ? lane : n - 1;
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < n; ++l) acc += sFx[i * ld + l] * sp[l];
                    if (lane < n) sQx[lane] = acc + (STAGE ? sG[lane] : (double)A.gx[t * n + lane]);
                }
                if (STORE_VALUE && Qv != nullptr && lane < n) Qv[QL.Qx + t * n + lane] = sQx[lane];
            }
        }
        ILQR_SUB_MARK2(I, 2);
        __syncthreads();                                                  // (B2) Qux, Quu, T complete
        ILQR_SUB_MARK1(I, 1); ILQR_SUB_MARK2(I, 3);
        // ------------------------------------------------ window C: the serial chain and p, ∇L | Qx, Qu, Qxx
        double pn = 0.0;
        if (NWV == 1 || wave == 0) {
            double Lxv, Luv;
            // Quu += guu, Qux += gux (:59, :64): the structurally non-zero entries only
            // (with the staging row the producing waves have added them already)
            if constexpr (!STAGE) {
#pragma unroll
                for (int j = 0; j < EU; ++j)
                    if (uoff[j] >= 0) S[uoff[j]] += (double)A.hc[(size_t)t * HSP + HXX + lane + 64 * j];
                wave_lds_fence();
            }
            if (STORE_VALUE && Qv != nullptr) {                           // policy.action_value.* (src/data/policy.jl:58-64)
                for (int e = lane; e < m * n; e += 64) Qv[QL.Qux + (size_t)t * m * n + e] = sQux[(e / m) * ldm + e % m];
                for (int e = lane; e < m * m; e += 64) Qv[QL.Quu + (size_t)t * m * m + e] = sQuu[(e / m) * ldm + e % m];
                if constexpr (VPAD) {
                    if (lane < m) Qv[QL.Qu + t * m + lane] = sQu[lane];
                    if (lane < n) Qv[QL.Qx + t * n + lane] = sQx[lane];
                }
            }
            // potrf('U') (info ignored, :68-69) and potrs('U') on [Qux | Qu] (:70-75) fused: column c of Quu on lanes c, c + 16, ...,
            // column j of [Qux | Qu] per lane (Qu is column NP of the LDS matrix, k of K's; nx = 64 leaves no lane for k: lane 0 carries
            // it as a second right-hand side)
            constexpr int NR = n < 64 ? 1 : 2;
            double Ua[m], Ur[m], b[NR][m];
            const int ucol = li < m ? li : m - 1;
#pragma unroll
            for (int i = 0; i < m; ++i) Ua[i] = sQuu[ucol * ldm + i];
            const int col0 = n < 64 ? (lane < n ? lane : NP) : lane;
            // K = -Quu^-1 Qux, k = -Quu^-1 Qu (:72-75): the solves run on the NEGATED right-hand sides (exact; the sign rides on the
            // first instruction that reads them)
#pragma unroll
            for (int i = 0; i < m; ++i) b[0][i] = -sQux[col0 * ldm + i];
            if constexpr (NR == 2) {
#pragma unroll
                for (int i = 0; i < m; ++i) b[1][i] = -sQux[NP * ldm + i];
            }
            ILQR_SUB_MARK1(I, 2);
            if (__builtin_expect(chol_solve_rows<m, NR>(Ua, Ur, b), 0)) {
                // a pivot was not positive (rare: diverged instances): what dpotf2 leaves behind, then the solves against it
#pragma unroll
                for (int i = 0; i < m; ++i) Ua[i] = sQuu[ucol * ldm + i];
                const int info = potrf_U_lanes<m>(Ua, Ur, lane);
                if (info != 0 && potrf_info == 0) potrf_info = info;
                if (lane < m) {
#pragma unroll
                    for (int i = 0; i < m; ++i) sU[lane * m + i] = Ua[i];
                }
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const int col = q == 0 ? col0 : NP;
#pragma unroll
                    for (int i = 0; i < m; ++i) b[q][i] = -sQux[col * ldm + i];
                    potrs_U_lds<m>(sU, Ur, b[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const bool mine = q == 0 ? (n < 64 ? lane <= n : true) : lane == 0;
                if (mine) {
                    const int col = q == 0 ? col0 : NP;
                    gdbl* dst = col != NP ? A.K + (size_t)t * m * n + col * m : A.k + t * m;
#pragma unroll
                    for (int i = 0; i < m; ++i) {
                        sK[col * ldm + i] = b[q][i];
                        dst[i] = b[q][i];
                    }
                }
            }
            // p = ux_tmp^T k + K^T Qu + Qux^T k + Qx (:86-89) with ux_tmp^T k = (Quu K)^T k formed as K^T (Quu^T k), and the Lagrangian
            // gradient (src/solve.jl:73-81), here rather than in window D: column j of K is still in lane j's registers and k in those
            // of lane nx (lane 0's second right-hand side when nx = 64); the m-vector Quu^T k takes m FMAs with column c of Quu per lane.
            {
                double kk[m], Qk[m];
#pragma unroll
                for (int l = 0; l < m; ++l) kk[l] = lane_bcast(b[NR - 1][l], n < 64 ? n : 0);
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < m; ++r) acc += sQuu[ucol * ldm + r] * kk[r];
                bcast_all<m>(acc, Qk);                                      // Qk[l] = (Quu^T k)[l], the same on every lane
                const int j = lane < n ? lane : 0;
                double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) {
                    a1 += b[0][l] * Qk[l];
                    a2 += b[0][l] * sQu[l];
                    a3 += sQux[j * ldm + l] * kk[l];
                }
                const double qx = sQx[j];
                pn = ((a1 + a2) + a3) + qx;
                Lxv = qx - pn;
                if (lane < n) gmax = nanmax(gmax, fabs(Lxv));
                if (lane < m) {
                    Luv = sQu[lane];
                    gmax = nanmax(gmax, fabs(Luv));
                }
                // stored here, a window ahead of the loop edge: the counter of outstanding memory operations is waited to zero at the
                // top of a step (the fetches of waves 1..3 share the code), and a store issued just before it would be waited for there
                if (lane < n) {
                    A.Lx[t * n + lane] = Lxv;
                    if (STORE_VALUE) A.p[t * n + lane] = pn;
                }
                if (lane < m) A.Lu[t * m + lane] = Luv;
                if constexpr (VPAD) { if (lane < n) sp[lane] = pn; }        // p′ of the next step: window D puts it into row nu of ûx (nobody reads sp in this window)
            }
            ILQR_SUB_MARK1(I, 3);
        }
        if constexpr (XA) {
            // ûx of this step has been consumed (window B): zero the buffer for the partial sums window D adds into it
            if (wave >= 2) {
                for (int e = (wave - 2) * 64 + lane; e < NP * ldm; e += 128) sUh[e] = 0.0;
            }
        }
        if constexpr (!PAIRED) if (NWV == 1 || wave != 0) role_switch<NWV>(wave, [&](auto Wc) {
            constexpr int W = decltype(Wc)::value;
            if constexpr (W > 0 && RS::tab.ct[W][0] >= 0) {     // what is left of T; a Qxx tile on another wave may wait for it
                run_tiles(Wc, IntC<2>{});
                wave_lds_fence();
                if (lane == 0) __hip_atomic_store(&sOut[4 + W], (double)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // Qxx = T fx + gxx (:53-54), written where P′ was (generic schedule: nobody reads P′ after window B) or to its own buffer
            // (TN = 2: a tile of T is still being formed from P′ in this window); the wave that stored a tile adds its gxx entries
            if constexpr (W > 0) static_for<0, SLOTS>([&](auto Sc) {
                constexpr int s = decltype(Sc)::value, e = RS::tab.qxx[W][s];
                if constexpr (e >= 0) {
                    constexpr int q = e & ~RS::RIC_WAIT_T, a = q / TN, c = q % TN;
                    if constexpr ((e & RS::RIC_WAIT_T) != 0) static_for<1, 4>([&](auto Vc) {
                        constexpr int V = decltype(Vc)::value;
                        if constexpr (V != W && RS::tab.ct[V][0] >= 0)
                            while (__hip_atomic_load(&sOut[4 + V], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (double)t) __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");     // nothing of the tile may be read ahead of the flag (the flag is data in LDS, served in order:
                                                            // a compiler barrier is all the acquire this needs — a real fence would wait for the pending global loads too)
                    });
                    const double4_t acc = tile_mm<n4, 1, ld, 1, ld>(sT + 16 * a, sFx + ld * 16 * c, li, lk);
                    tile_store<ld>(sQ, acc, 16 * a, 16 * c, li, lk);
                    constexpr int xb = M::HESS_XX_TILE_START[q], xc = M::HESS_XX_TILE_START[q + 1] - xb;
                    if constexpr (xc > 0) {
                        wave_lds_fence();
#pragma unroll
                        for (int x = 0; x < (xc + 63) / 64; ++x)
                            if (xoff[NWV == 1 ? W : 0][s][x] >= 0) S[xoff[NWV == 1 ? W : 0][s][x]] += STAGE ? sH[xb + lane + 64 * x] : (double)A.hc[(size_t)t * HSP + xb + lane + 64 * x];
                    }
                    if (STORE_VALUE && Qv != nullptr) {
                        wave_lds_fence();
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * a + lk + 4 * r, col = 16 * c + li;
                            if (row < n && col < n) Qv[QL.Qxx + (size_t)t * n * n + col * n + row] = sQ[col * ld + row];
                        }
                    }
                }
            });
        });
        if constexpr (XA) {
            // the next step's state-dependent entries of fu: window D reads them (partial ûx), nobody reads fu in this window
            if (wave != 0) {
#pragma unroll
                for (int j = 0; j < EJ; ++j)
                    if (poff[j] >= LD::oFu) S[poff[j]] = pval[j];
            }
        }
        __syncthreads();                                                  // (B3) K, k, Qx in LDS; T, fx, fu no longer needed
        ILQR_SUB_MARK1(I, 4); ILQR_SUB_MARK2(I, 4);
        // ------------------------------------------------ window D: P (:79-84) | p, ∇L (:86-89, src/solve.jl:73-81); next step's Jacobian entries
        // P = K^T ux_tmp + K^T Qux + Qux^T K + Qxx   (:81-84), accumulated in this order, on one tile per wave when nx <= 32; ux_tmp = Quu K (:79) comes out of its MFMAs in exactly the layout the next MFMA's B operand wants
        // (k = lane>>4 + 4 reg, j = lane&15). All fragments of a tile are read first, then its eight MFMAs issue back to back.
        // The next step's operands go into LDS (they were requested in window A; nobody reads their places in this window) in the
        // shadow of the wave's first P tile, by the threads that requested them.
        // (unconditional: at t = 0 it rewrites step 0's own values, and the loads must be consumed on every path through the loop —
        // a path that skipped them would leave them pending at the loop edge and every iteration would start by waiting)
        auto place_next = [&]() {
            if (NWV == 1 && placed) return;
            placed = true;
#pragma unroll
            for (int j = 0; j < EJ; ++j)
                if (poff[j] >= 0 && (!XA || poff[j] < LD::oFu)) S[poff[j]] = pval[j];       // (three-window schedule: the fu entries went in in window C)
            if (STAGE) stage_store(rval);
        };
        role_switch<NWV>(wave, [&](auto Wc) { static_for<0, SLOTS>([&](auto Sc) {
            constexpr int q = RS::tab.p[decltype(Wc)::value][decltype(Sc)::value];
            constexpr bool first = decltype(Sc)::value == 0;
            if constexpr (q < 0 && first) place_next();
            if constexpr (q >= 0) {
                constexpr int KS = m4 / 4;
                constexpr int a = q / TN, c = q % TN;
                const double* pKa = sK + ldm * (16 * a + li) + lk;                     // A(i,k) = K[k][16a + i]
                const double* pKc = sK + ldm * (16 * c + li) + lk;                     // B(k,j) = K[k][16c + j]
                const double* pQa = sQux + ldm * (16 * a + li) + lk;                   // A(i,k) = Qux[k][16a + i]
                const double* pQc = sQux + ldm * (16 * c + li) + lk;                   // B(k,j) = Qux[k][16c + j]
                const double* pU = sQuu + li + ldm * lk;                               // A(i,k) = Quu[i][k]
                const double* pq = sQ + (16 * c + li) * ld + 16 * a + lk;              // Qxx tile (D layout)
                double fKa[KS], fKc[KS], fQa[KS], fQc[KS], fU[KS];
                double4_t qxx;
#pragma unroll
                for (int sx = 0; sx < KS; ++sx) { fU[sx] = pU[4 * ldm * sx]; fKc[sx] = pKc[4 * sx]; fKa[sx] = pKa[4 * sx]; fQc[sx] = pQc[4 * sx]; fQa[sx] = pQa[4 * sx]; }
#pragma unroll
                for (int r = 0; r < 4; ++r) qxx[r] = pq[4 * r];
                // (three-window schedule) the next step's fu, rows 16a .. 16a + 15 of it: A(i, k) = fu[16a + k][i]
                double fF[4], pnext = 0.0;
                if constexpr (XA) {
                    const double* pF = sFu + ld * li + 16 * a + lk;
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) fF[sx] = pF[4 * sx];
                    if constexpr (VPAD && a == 0) pnext = sp[16 * c + li];      // p[t] (stored at the end of the chain): row nu of the next step's ûx
                }
                __builtin_amdgcn_sched_barrier(0);
                double4_t ux = double4_t{0, 0, 0, 0}, acc = double4_t{0, 0, 0, 0};
#pragma unroll
                for (int sx = 0; sx < KS; ++sx) ux = __builtin_amdgcn_mfma_f64_16x16x4f64(fU[sx], fKc[sx], ux, 0, 0, 0);
#pragma unroll
                for (int sx = 0; sx < KS; ++sx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fKa[sx], ux[sx], acc, 0, 0, 0);
#pragma unroll
                for (int sx = 0; sx < KS; ++sx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fKa[sx], fQc[sx], acc, 0, 0, 0);
#pragma unroll
                for (int sx = 0; sx < KS; ++sx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fQa[sx], fKc[sx], acc, 0, 0, 0);
                if constexpr (first) place_next();
                const double4_t v = acc + qxx;
                tile_store<ld>(sP, v, 16 * a, 16 * c, li, lk);
                if constexpr (XA) {
                    // ûx(t - 1)(., 16c .. 16c + 15) += fu(t - 1)[16a .. 16a + 15, .]ᵀ P(a, c): the tile in registers is the B operand
                    // (row lk + 4 sx of the tile in register sx = k-slice sx); rows a = 0 and a = 1 go to separate buffers
                    double4_t au = double4_t{0, 0, 0, 0};
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) au = __builtin_amdgcn_mfma_f64_16x16x4f64(fF[sx], v[sx], au, 0, 0, 0);
                    if constexpr (VPAD && a == 0) au[VR] = lk == VK ? pnext : au[VR];     // (fu's rows beyond nu are zero padding: that row of the product is free)
                    ldsd* pu_ = S3 + LD::oUh + (16 * c + li) * ldm + lk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) __hip_atomic_fetch_add(pu_ + 4 * r, au[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (STORE_VALUE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + lk + 4 * r, col = 16 * c + li;
                        if (row < n && col < n) A.P[(size_t)t * n * n + col * n + row] = v[r];
                    }
                }
            }
        }); });
        if (!VPAD && (NWV == 1 || wave == 0) && lane < n) sp[lane] = pn;               // p' of the next step (its readers of this step are past B3)
        __syncthreads();                                                  // (B0) P′, p′, patched fx, fu visible
        ILQR_SUB_MARK1(I, 5); ILQR_SUB_MARK2(I, 5);
    }
    __builtin_amdgcn_s_setprio(0);
    // the serial chain lived on wave 0: hand its scalars to all waves (identical control flow afterwards)
    const double gn = wave_max(gmax);
    if (tid == 0) { sOut[0] = gn; sOut[1] = (double)potrf_info; }
    __syncthreads();
    RiccatiOut out;
    out.gradient_norm = sOut[0];
    out.potrf_info = (int)sOut[1];
#if defined(ILQR_PROFILE) && defined(ILQR_PROFILE_SUB)
    for (int q = 0; q < 6; ++q) out.prof[q] = I.prof[q];
#endif
    __syncthreads();
    return out;
}

template <class M, bool STORE_VALUE>
__device__ __forceinline__ void backward_pass_large(Inst<M>& I) {
    const RiccatiOut o = backward_pass_large_fn<M, STORE_VALUE>(as_global(I.gbase), as_global(I.Q));
    I.gradient_norm = o.gradient_norm;
    if (o.potrf_info != 0 && I.potrf_info == 0) I.potrf_info = o.potrf_info;
#if defined(ILQR_PROFILE) && defined(ILQR_PROFILE_SUB)
    for (int q = 0; q < 6; ++q) I.prof[q] += o.prof[q];
#endif
}

// ---------------------------------------------------------------- forward sweep: rollout! ∥ trajectory_sensitivities
// The closed-loop rollout (src/rollout.jl:1-31) and, on the first line-search trial, the sensitivity recursion with its dot
// product (src/data/methods.jl:42-54, src/forward_pass.jl:20) are independent forward sweeps over t: wave 0 runs the rollout,
// wave 1 the sensitivities, waves 2 and 3 copy K_t into an LDS ring one chunk of CH timesteps ahead (one workgroup barrier
// per chunk). Before the sweep all four waves form a_t = α k_t + ū_t and b_t = K_t x̄_t (src/rollout.jl:24-28 is
// u = ((α k + ū) + K x) − K x̄ in that order) for every t, so that the serial chain of a step is K x, the affine row and the
// remainder only. K x and K x̄ use the same summation scheme (four partial sums per action, combined pairwise), so the
// feedback term vanishes exactly where x = x̄.
// v + (v of lane li ^ 8 of the same 16-lane row)
__device__ __forceinline__ double sum_row_halves(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true);      // row_ror:8
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true);
    return v + __hiloint2double(hi, lo);
}
// Σ_j K[i][j] x[j] for action i: the summation scheme shared by the rollout's K x, the precomputed K x̄ and the sensitivity sweep's
// K Δx (the feedback term must vanish exactly where x = x̄). Lane (li, lk) of the wave sums the lk-th quarter of the states for
// action li; with nu <= 8 half of every 16-lane row would repeat its neighbour's work, so lane li + 8 takes the second half of
// the quarter instead (one DPP rotation joins them) — four terms per lane on the serial chain instead of eight. Then the
// quarters are combined pairwise. K(j, i) and x(j) come through the two callables.
template <class M, class FK, class FX>
__device__ __forceinline__ double kx_sum(int li, int lk, FK&& Kji, FX&& xj) {
    constexpr int n = M::NX, m = M::NU, JP = (n + 3) / 4;
    constexpr bool EIGHTHS = m <= 8 && JP % 2 == 0;
    constexpr int JQ = EIGHTHS ? JP / 2 : JP;
    const int ia = EIGHTHS ? (li & 7) : li, i = ia < m ? ia : m - 1;
    const int j0 = lk * JP + (EIGHTHS ? (li >> 3) * JQ : 0);
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < JQ; ++q) {
        const int j = j0 + q;
        if ((n % 4 == 0) || j < n) acc += Kji(j < n ? j : 0, i) * xj(j < n ? j : 0);
    }
    if constexpr (EIGHTHS) acc = sum_row_halves(acc);
    return sum_quarters(acc);
}
template <class M>
__device__ __forceinline__ double kx_partial(const double* Kt, const double* xv, int li, int lk) {
    return kx_sum<M>(li, lk, [&](int j, int i) { return Kt[j * M::NU + i]; }, [&](int j) { return xv[j]; });
}

// row i of x⁺ = f(x, u) (src/rollout.jl:29): affine part from the lane's coefficients, remainder either elementwise on the lane's
// own component (M::dyn_rem_own) or through the generated wave-cooperative code. nx <= 32 leaves half the wave without a row:
// lane i + 32 then takes the second half of row i's state terms (and the action terms), and the halves meet in one
// v_permlane32_swap — 24 instead of 40 FMAs and half the LDS reads per lane on the rollout's serial chain.
template <class M>
struct DynAff {
    static constexpr int n = M::NX, m = M::NU;
    static constexpr bool SPLIT = n <= 32;
    static constexpr int HX = SPLIT ? (n + 1) / 2 : n;           // state terms per lane
    double cx[HX], cu[m], c0;
    int x0;                                                      // first state term of this lane
    __device__ __forceinline__ void init(int lane) {
        const int r = SPLIT ? (lane & 31) : lane, row = r < n ? r : n - 1, h = SPLIT ? lane >> 5 : 0;
        x0 = h * HX;
#pragma unroll
        for (int j = 0; j < HX; ++j) cx[j] = x0 + j < n ? M::DYN_AFF[row][x0 + j < n ? x0 + j : 0] : 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) cu[j] = (!SPLIT || h == 1) ? M::DYN_AFF[row][n + j] : 0.0;
        c0 = h == 0 ? M::DYN_AFF[row][n + m] : 0.0;
    }
};
template <class M>
__device__ __forceinline__ double dyn_row(const DynAff<M>& aff, const double* sx, const double (&ua)[M::NU], double xl, int lane,
                                          const double* W, int t) {
    constexpr int n = M::NX, m = M::NU, HX = DynAff<M>::HX;
    // four interleaved partial sums: a dependent fp64 FMA chain advances one link per ~8 clk on a lone wave
    double y4[4] = {aff.c0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < HX; ++j) {                               // (a padding term of an odd nx re-reads x_0 against a zero coefficient)
        const int jj = (!DynAff<M>::SPLIT || n % 2 == 0 || aff.x0 + j < n) ? aff.x0 + j : 0;
        y4[j & 3] += aff.cx[j] * sx[jj];
    }
#pragma unroll
    for (int j = 0; j < m; ++j) y4[j & 3] += aff.cu[j] * ua[j];
    double y = (y4[0] + y4[1]) + (y4[2] + y4[3]);
    if constexpr (DynAff<M>::SPLIT) y = sum_halves(y);
    if constexpr (M::DYN_HAS_REM) {
        double w[cdim<M::NW>::v];
        load_w<M::NW>(W, t, w);
        if constexpr (M::DYN_REM_ELEMENTWISE) {
            y += M::dyn_rem_own(xl, ua, w);
        } else {
            double xa[n], r[n];
#pragma unroll
            for (int j = 0; j < n; ++j) xa[j] = sx[j];
            M::dyn_rem_wave(lane, xa, ua, w, r);
            double rl = r[0];
#pragma unroll
            for (int i = 1; i < n; ++i) rl = (lane == i) ? r[i] : rl;
            y += rl;
        }
    }
    return y;
}

// K, a, b of [c0, c0 + CH) into the ring half (c0 / CH) & 1 by `nthreads` threads (this one is number `me` of them): every load of
// a thread is issued before its first LDS write (a load-store loop would pay one HBM round trip per element: seventeen in a row
// made the stagers slower than the rollout)
template <class M>
__device__ __forceinline__ void fw_stage_chunk(const LargeArgs& A, double* ring, int c0, int nthreads, int me) {
    typedef LargeDims<M> LD;
    constexpr int m = M::NU, CH = LD::CH, KN = M::NU * M::NX;
    constexpr int NE = LD::W == 1 ? (CH * LD::RSTEP + 63) / 64 : (CH * LD::RSTEP + 127) / 128;     // elements per thread with the fewest stagers (one wave / two waves)
    const int N = A.N, steps = (N - c0) < CH ? (N - c0) : CH;
    double* dst = ring + ((c0 / CH) & 1) * CH * LD::RSTEP;
    const gdbl* srcK = A.K + (size_t)c0 * KN;
    const long ab_rel = (A.ab + (size_t)c0 * 2 * m) - srcK;            // same instance block: one base, selected offsets
    double v[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = me + nthreads * q, st = e / LD::RSTEP, r = e % LD::RSTEP;
        const long off = r < KN ? (long)st * KN + r : ab_rel + st * 2 * m + (r - KN);
        v[q] = e < steps * LD::RSTEP ? (double)srcK[off] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = me + nthreads * q;
        if (e < steps * LD::RSTEP) dst[e] = v[q];
    }
}

// wave 0: the closed-loop rollout, one chunk of the K ring per workgroup barrier
template <class M>
__attribute__((noinline)) __device__ void fw_rollout_wave(gdbl* base) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, CH = LD::CH, KN = m * n;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const LargeArgs A = large_args_from_lds<M>(base);
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, N = A.N;
    double *sx = lds_dyn + LD::oFw, *ring = lds_dyn + LD::oRing;
    DynAff<M> aff;
    aff.init(lane);
    // (with the split rows lanes 32.. carry a copy of x_{lane - 32}: their sine argument stays finite and their row sum is the same)
    const int xrow = DynAff<M>::SPLIT ? (lane & 31) : lane;
    double xl = xrow < n ? A.xb[xrow] : 0.0;                                  // x[1] = x̄[1]  (:19)
    if (lane < n) A.x[lane] = xl;
    const int ui = li < m ? li : 0;                                           // (every row of 16 lanes forms the same m actions)
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int c1 = (c0 + CH) < N ? (c0 + CH) : N;
        for (int t = c0; t < c1; ++t) {
            const double* Kt = ring + (t % (2 * CH)) * LD::RSTEP;
            if (lane < n) sx[lane] = xl;
            wave_lds_fence();
            const double a_t = Kt[KN + ui], b_t = Kt[KN + m + ui];            // α k + ū and K x̄, formed beforehand (no global load on this chain)
            const double acc = kx_partial<M>(Kt, sx, li, lk);
            double v = a_t;                                                   // α k + ū   (:24-26)
            v += acc;                                                         // + K x      (:27)
            v += -1.0 * b_t;                                                  // − K x̄     (:28)
            if (lane < m) A.u[t * m + lane] = v;
            // action i sits on lane i of every 16-lane row (the quarter sums leave four identical rows): the row's lanes get all of
            // u by one DPP row broadcast each — no LDS round trip between the feedback term and the dynamics
            double ua[m];
            bcast_all<m>(v, ua);
            const double y = dyn_row<M>(aff, sx, ua, xl, lane, (const double*)A.w, t);      // (:29)
            xl = y;
            if (lane < n) A.x[(t + 1) * n + lane] = y;
            wave_lds_fence();
        }
        if (LD::W == 1 && c0 + CH < N) fw_stage_chunk<M>(A, ring, c0 + CH, 64, lane);     // one wave: it stages its own next chunk
        __syncthreads();
    }
}

// wave 1 on the first line-search trial: Δu = k + KΔx on lanes 0..m-1 (quarter sums like the rollout), Δx⁺ = fuΔu + fxΔx one row
// per lane, and gradientᵀ·Δz along the way (src/data/methods.jl:42-54, src/forward_pass.jl:20)
template <class M>
__attribute__((noinline)) __device__ double fw_delta_wave(gdbl* base) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, NP = LD::NP, MP = LD::MP, ld = LD::ld, CH = LD::CH, JV = LD::JV, JVP = LD::JVP;
    constexpr int EJ = (JV + 63) / 64 > 0 ? (JV + 63) / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const LargeArgs A = large_args_from_lds<M>(base);
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, N = A.N;
    double* S = lds_dyn;
    double *sFx = S + LD::oFx, *sFu = S + LD::oFu, *zx = S + LD::oFw + NP + MP, *ring = S + LD::oRing;
    int poff[EJ];
    double pval[EJ], dpart = 0.0, kv = 0.0, Luv = 0.0, Lxv = 0.0;
    constexpr bool DSPLIT = n <= 32;
    constexpr int HXD = DSPLIT ? (n + 1) / 2 : n;
    const int xr = DSPLIT ? (lane & 31) : lane, j0 = DSPLIT ? (lane >> 5) * HXD : 0;
    const int ui = li < m ? li : m - 1, xi = xr < n ? xr : n - 1;             // (every row of 16 lanes forms the same m Δu)
#pragma unroll
    for (int j = 0; j < EJ; ++j) {
        const int q = lane + 64 * j;
        poff[j] = -1; pval[j] = 0.0;
        if (q < JV) {
            const int idx = M::JAC_VAR_IDX[q];
            poff[j] = idx < n * n ? LD::oFx + (idx / n) * ld + idx % n : LD::oFu + ((idx - n * n) / n) * ld + (idx - n * n) % n;
            if (N > 0) pval[j] = A.fv[q];
        }
    }
    if (lane < n) zx[lane] = 0.0;
#pragma unroll
    for (int j = 0; j < EJ; ++j)
        if (poff[j] >= 0) S[poff[j]] = pval[j];                               // fx_0, fu_0
    wave_lds_fence();
    if (N > 0) { kv = A.k[ui]; Luv = A.Lu[ui]; Lxv = A.Lx[xi]; }
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int c1 = (c0 + CH) < N ? (c0 + CH) : N;
#ifdef ILQR_DBG_NODELTA      // timing experiment only: the sweep's barriers without its work
        if (c1 > 0) { __syncthreads(); continue; }
#endif
        for (int t = c0; t < c1; ++t) {
            const double* Kt = ring + (t % (2 * CH)) * LD::RSTEP;
            // (fx_t's state-dependent entries are in place: written at the end of the previous step, behind its closing fence)
            const int t1 = t + 1 < N ? t + 1 : t;
#pragma unroll
            for (int j = 0; j < EJ; ++j) pval[j] = poff[j] >= 0 ? A.fv[(size_t)t1 * JVP + lane + 64 * j] : 0.0;
            const double kv_n = A.k[t1 * m + ui], Luv_n = A.Lu[t1 * m + ui], Lxv_n = A.Lx[t1 * n + xi];
            // Δx⁺ = fu Δu + fx Δx: fx Δx needs nothing of this step, so it goes first (four interleaved partial sums: a dependent
            // fp64 chain advances one link per ~8 clk) and runs while the feedback term's operands are on their way
            // (nx <= 32: lane i + 32 takes the second half of row i's terms, the halves meet in one v_permlane32_swap)
            double a2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < HXD; ++j) {
                const int jj = (!DSPLIT || n % 2 == 0 || j0 + j < n) ? j0 + j : n;     // (row n of the padded LDS copy of fx is zero)
                a2[j & 3] += sFx[jj * ld + xi] * zx[jj < n ? jj : 0];
            }
            const double zown = zx[xi];
            const double acc = kx_partial<M>(Kt, zx, li, lk);
            const double du = kv + acc;
            if (lane < m) dpart += Luv * du;
            if (lane < n) dpart += Lxv * zown;
            // Δu_i sits on lane i of every 16-lane row: one DPP row broadcast each instead of an LDS round trip
            double dua[m];
            bcast_all<m>(du, dua);
            double a1 = 0.0;
#pragma unroll
            for (int j = 0; j < m; ++j) a1 += sFu[j * ld + xi] * dua[j];
            wave_lds_fence();                                                 // every read of Δx and of fx_t, fu_t is done
            double a2s = (a2[0] + a2[1]) + (a2[2] + a2[3]);
            if constexpr (DSPLIT) a2s = sum_halves(a2s);
            if (lane < n) zx[lane] = a1 + a2s;
#pragma unroll
            for (int j = 0; j < EJ; ++j)
                if (poff[j] >= 0) S[poff[j]] = pval[j];                       // fx_{t+1}, fu_{t+1} (requested at the head of this step)
            kv = kv_n; Luv = Luv_n; Lxv = Lxv_n;
            wave_lds_fence();
        }
        if (LD::W == 1 && c0 + CH < N) fw_stage_chunk<M>(A, ring, c0 + CH, 64, lane);
        __syncthreads();
    }
    return wave_sum(dpart);
}

template <class M>
__attribute__((noinline)) __device__ double forward_sweep_large_fn(gdbl* base, double alpha, int want_delta) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, NP = LD::NP, ld = LD::ld, NT = LD::NT, CH = LD::CH, KN = m * n;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const LargeArgs A = large_args_from_lds<M>(base);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
    want_delta = __builtin_amdgcn_readfirstlane(want_delta);
    const int N = A.N;
    double* S = lds_dyn;
    double *sFx = S + LD::oFx, *sFu = S + LD::oFu, *ring = S + LD::oRing;
    double* sOut = S + LD::oVec + 2 * NP;
    // ---- a_t, b_t for every timestep (wave per timestep, same lane mapping as the rollout's K x)
    for (int t = wave; t < N; t += LD::W) {
        const double acc = kx_sum<M>(li, lk, [&](int j, int i) { return (double)A.K[(size_t)t * KN + j * m + i]; },
                                     [&](int j) { return (double)A.xb[t * n + j]; });
        if (lane < m) {
            double v = A.k[t * m + lane] * alpha;                         // (:24-25)
            v += A.ub[t * m + lane];                                      // (:26)
            A.ab[2 * t * m + lane] = v;
            A.ab[(2 * t + 1) * m + lane] = acc;
        }
    }
    // constant Jacobian entries for the sensitivity recursion
    if (want_delta) {
        for (int e = tid; e < LD::oP; e += NT) S[e] = 0.0;
        __syncthreads();
        for (int e = tid; e < n * n; e += NT) sFx[(e / n) * ld + e % n] = M::JAC_CONST_FX[0][e];
        for (int e = tid; e < n * m; e += NT) sFu[(e / n) * ld + e % n] = M::JAC_CONST_FU[0][e];
    }
    auto stage = [&](int c0, int nthreads, int me) { fw_stage_chunk<M>(A, ring, c0, nthreads, me); };
    __syncthreads();                                                      // a_t, b_t (global, written and read by this workgroup only) visible
    if (N > 0) stage(0, NT, tid);
    __syncthreads();                                                      // ring half 0 ready
    double d = 0.0;
    if constexpr (LD::W == 1) {
        // one wave per instance: the rollout over the whole horizon (staging its own chunks), then — first trial — the
        // sensitivity recursion the same way
        fw_rollout_wave<M>(base);
        if (want_delta) {
            if (N > 0) stage(0, NT, tid);
            __syncthreads();
            d = fw_delta_wave<M>(base);
        }
        return d;
    }
    if (wave == 0) fw_rollout_wave<M>(base);
    else if (wave == 1 && want_delta) d = fw_delta_wave<M>(base);
    else {
        const int first = want_delta ? 2 : 1, cnt = LD::W - first;        // the stagers: next chunk into the other ring half
        for (int c0 = 0; c0 < N; c0 += CH) {
            if (c0 + CH < N) stage(c0 + CH, 64 * cnt, tid - 64 * first);
            __syncthreads();
        }
    }
    if (want_delta) {                     // hand wave 1's scalar to all waves (identical control flow afterwards)
        if (tid == 64) sOut[2] = d;
        __syncthreads();
        d = sOut[2];
        __syncthreads();
    }
    return d;
}
template <class M>
__device__ __forceinline__ void rollout_large(Inst<M>& I, double alpha, bool want_delta, double& delta_out) {
    ILQR_PROF_BEGIN();
    const double d = forward_sweep_large_fn<M>(as_global(I.gbase), alpha, want_delta ? 1 : 0);
    if (want_delta) delta_out = d;
    I.rollouts += 1;
    I.states_eq_nominal = 0;
    ILQR_PROF_END(I, PROF_ROLLOUT);
}

// ---------------------------------------------------------------- host-visible mirror of the compact representation
// dir = 0: materialise — jacobian_state / jacobian_action = generated constants + fv (all zero while the Jacobians have not
// been evaluated since the last reset), hessian_* = zeros + hc.   dir = 1: gather — the reverse, after the host wrote one of
// the full arrays (entries outside the structural pattern cannot be represented and are dropped; the constant Jacobian
// entries are the generated ones whatever the host wrote there).
template <class M>
__global__ __launch_bounds__(256) void mirror_large_kernel(KArgs a, int dir) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= a.B) return;
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    const int T = L.T, N = T - 1;
    if (dir == 0) {
        const bool valid = g[L.scal + S_JAC_VALID] != 0.0;
        for (int e = tid; e < N * n * n; e += 256) g[L.fx + e] = valid ? M::JAC_CONST_FX[0][e % (n * n)] : 0.0;
        for (int e = tid; e < N * n * m; e += 256) g[L.fu + e] = valid ? M::JAC_CONST_FU[0][e % (n * m)] : 0.0;
        for (int e = tid; e < T * n * n; e += 256) g[L.gxx + e] = 0.0;
        for (int e = tid; e < N * m * m; e += 256) g[L.guu + e] = 0.0;
        for (int e = tid; e < N * m * n; e += 256) g[L.gux + e] = 0.0;
        __syncthreads();
        if (valid)
            for (int e = tid; e < N * LD::JV; e += 256) {
                const int t = e / (LD::JV > 0 ? LD::JV : 1), q = e % (LD::JV > 0 ? LD::JV : 1), idx = M::JAC_VAR_IDX[q];
                if (idx < n * n) g[L.fx + t * n * n + idx] = g[L.fv + t * LD::JVP + q];
                else g[L.fu + t * n * m + idx - n * n] = g[L.fv + t * LD::JVP + q];
            }
        for (int e = tid; e < T * LD::HS; e += 256) {
            const int t = e / (LD::HS > 0 ? LD::HS : 1), q = e % (LD::HS > 0 ? LD::HS : 1), idx = M::HESS_IDX[q];
            const double v = g[L.hc + t * LD::HSP + q];
            if (q < LD::HXX) g[L.gxx + t * n * n + idx] = v;
            else if (t < N && q < LD::HXX + LD::HUU) g[L.guu + t * m * m + idx] = v;
            else if (t < N) g[L.gux + t * m * n + idx] = v;
        }
    } else {
        for (int e = tid; e < N * LD::JV; e += 256) {
            const int t = e / (LD::JV > 0 ? LD::JV : 1), q = e % (LD::JV > 0 ? LD::JV : 1), idx = M::JAC_VAR_IDX[q];
            g[L.fv + t * LD::JVP + q] = idx < n * n ? g[L.fx + t * n * n + idx] : g[L.fu + t * n * m + idx - n * n];
        }
        for (int e = tid; e < T * LD::HS; e += 256) {
            const int t = e / (LD::HS > 0 ? LD::HS : 1), q = e % (LD::HS > 0 ? LD::HS : 1), idx = M::HESS_IDX[q];
            double v = 0.0;
            if (q < LD::HXX) v = g[L.gxx + t * n * n + idx];
            else if (t < N && q < LD::HXX + LD::HUU) v = g[L.guu + t * m * m + idx];
            else if (t < N) v = g[L.gux + t * m * n + idx];
            g[L.hc + t * LD::HSP + q] = v;
        }
    }
}

}  // namespace ilqr
