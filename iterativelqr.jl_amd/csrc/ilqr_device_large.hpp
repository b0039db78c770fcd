// Large-model path (nx > 4 or nu > 4, up to nx = 64 — one state component per lane — and nu = 16), e.g. BASELINE config "synth32".
//
// Differences from the LDS-resident small-model path of ilqr_device.hpp:
//   * the per-instance workspace (2.4 MB for nx=32, nu=8, T=101) stays in HBM; the wave streams the
//     per-timestep matrices through LDS staging buffers, fetching the next step's operands into registers
//     while the current step computes;
//   * every phase is a real (noinline) function with its own register allocation, taking typed global
//     pointers; inside the fused solve kernel the dense model code otherwise pushes the Riccati loop into
//     hundreds of spills whose scratch reloads wait for the HBM prefetch;
//   * linearisation: the Jacobian entries that are constants (generated tables M::JAC_CONST_*) are written by
//     coalesced wave-wide stores, only the state-dependent ones are evaluated per timestep
//     (M::dyn_jac_var_mem); Hessians accumulate only their structurally non-zero entries
//     (M::cost_*_hess_acc); the Gauss-Newton AL terms are derived symbolically (M::al_s / M::al_t);
//   * the Riccati step's contractions fx^T P' fx, fu^T P' fx, ... are real matrix products here and run as
//     register-blocked 16x16x4 fp64 MFMA tiles (v_mfma_f64_16x16x4_f64) out of zero-padded LDS operands
//     (A[i][k]: i = lane&15, k = lane>>4; B[k][j]: k = lane>>4, j = lane&15;
//      C/D[i][j]: j = lane&15, i = (lane>>4) + 4*reg  — cdna_hip_programming.md §3);
//   * Cholesky of Quu runs on wave-uniform registers (dpotf2 order), the triangular solves one column per lane
//     with the inverted diagonal;
//   * rollout: one state component per lane — row i of the affine part of the dynamics (generated table
//     M::DYN_AFF) on lane i, the nonlinear remainder in wave-cooperative form (M::dyn_rem_wave).
// Same reference semantics and citations as ilqr_device.hpp.
#pragma once

namespace ilqr {

constexpr int r16(int v) { return (v + 15) & ~15; }
constexpr int r4(int v) { return (v + 3) & ~3; }

template <class M>
struct LargeDims {
    static constexpr int n = M::NX, m = M::NU;
    static constexpr int NP = r16(n), MP = r16(m);          // whole 16x16 tiles; the padding is kept at zero
    static constexpr int ld = NP + 1, ldm = MP + 1;         // odd leading dimensions: conflict-free LDS column walks
    // LDS carve (doubles); must match large_lds_doubles() of ilqr_layout.hpp
    static constexpr int oP = 0, oFx = oP + NP * ld, oT = oFx + NP * ld, oFu = oT + NP * ld, oUh = oFu + MP * ld,
                         oQux = oUh + NP * ldm, oK = oQux + NP * ldm, oUxt = oK + NP * ldm, oQuu = oUxt + NP * ldm,
                         oVec = oQuu + MP * ldm, oLay = oVec + 4 * NP + 4 * MP + 8, total = oLay + LAYOUT_LDS_DOUBLES;
};

typedef double double4_t __attribute__((ext_vector_type(4)));

// D-layout of the tile: element r of lane (li, lk) is (row I0 + lk + 4r, column J0 + li)
template <int LDD>
__device__ __forceinline__ void tile_store(double* D, double4_t acc, int I0, int J0, int li, int lk) {
#pragma unroll
    for (int r = 0; r < 4; ++r) D[(J0 + li) * LDD + I0 + lk + 4 * r] = acc[r];
}
// the same elements of a packed column-major global matrix (R x Cc), zero outside
template <int R, int Cc, class Ptr>
__device__ __forceinline__ double4_t tile_load_global(Ptr G, int I0, int J0, int li, int lk) {
    double4_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = I0 + lk + 4 * r, col = J0 + li;
        if constexpr (R % 16 == 0 && Cc % 16 == 0) v[r] = G[col * R + row];
        else {                                      // clamped address + select: no exec-mask branch
            const bool in = row < R && col < Cc;
            const double x = G[in ? col * R + row : 0];
            v[r] = in ? x : 0.0;
        }
    }
    return v;
}

// Register-blocked product of TR x TC output tiles on v_mfma_f64_16x16x4_f64: acc(a,c) += sum_{k<KD} A_a(i,k) B(k,j)
// with A(i,k) at A[AI*i + AK*k] and B(k,j) at B[BK*k + BJ*j] (any transposition is just a stride pattern; strides
// and KD are compile-time, operands zero-padded: no bounds checks). All operand fragments are read from LDS FIRST (TR*KD/4 + TC*KD/4
// doubles per lane), then the MFMAs issue back to back, k-step outermost so that consecutive instructions hit
// different accumulators. Row tile a takes its A operand from Aop[a] (so two matrices can share one B pass).
template <int TR, int TC, int KD, int AI, int AK, int BK, int BJ>
__device__ __forceinline__ void tiles_mac(double4_t (&acc)[TR * TC], const double* const (&Aop)[TR], const int (&Arow)[TR],
                                          const double* B, int li, int lk, int c0 = 0) {
    constexpr int KS = KD / 4;
    double fa[TR][KS], fb[TC][KS];
#pragma unroll
    for (int a = 0; a < TR; ++a) {
        const double* pa = Aop[a] + AI * (Arow[a] + li) + AK * lk;
#pragma unroll
        for (int s = 0; s < KS; ++s) fa[a][s] = pa[AK * 4 * s];
    }
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        const double* pb = B + BK * lk + BJ * (16 * (c0 + c) + li);
#pragma unroll
        for (int s = 0; s < KS; ++s) fb[c][s] = pb[BK * 4 * s];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int a = 0; a < TR; ++a)
#pragma unroll
            for (int c = 0; c < TC; ++c)
                acc[a * TC + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][s], fb[c][s], acc[a * TC + c], 0, 0, 0);
}

// The phase is a real (noinline) function: the fused solve kernel inlines every other phase (dense 32-state model
// code included), and inside that one register allocation the Riccati loop ended up with hundreds of spills
// whose scratch reloads wait on vmcnt(0), i.e. on the HBM prefetch. As a function it gets its own allocation.
// Pointers cross the call with their address space attached (a plain double* would turn every access into a
// flat_load), LDS is re-derived from the dynamic shared symbol.
typedef __attribute__((address_space(1))) double gdbl;
template <class T> __device__ __forceinline__ gdbl* as_global(T* p) { return (gdbl*)p; }
// LDS traffic of ONE wave is in order: a wave-local exchange between lanes needs no workgroup barrier
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ gdbl* uniform_ptr(gdbl* p) {
    unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gdbl*)(((unsigned long long)hi << 32) | lo);
}
// What the large-path phase functions need of one instance. They are real calls: a struct of twenty pointers would travel
// on the stack (scratch), so only the instance's block pointer crosses the call; the Layout is read back from the tail of
// the dynamic LDS, where the kernel parked it (store_layout_lds), and the pointers are rebuilt wave-uniform in the callee.
struct LargeArgs {
    gdbl *xb, *ub, *x, *u, *fx, *fu, *gx, *gu, *K, *k, *Lx, *Lu, *c, *lam, *rho, *act, *w, *gxx, *guu, *gux, *P, *p, *scal;
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
    return LargeArgs{g + L.xb, g + L.ub, g + L.x, g + L.u, g + L.fx, g + L.fu, g + L.gx, g + L.gu, g + L.K, g + L.k, g + L.Lx, g + L.Lu,
                     g + L.c, g + L.lam, g + L.rho, g + L.act, g + L.w, g + L.gxx, g + L.guu, g + L.gux, g + L.P, g + L.p, g + L.scal,
                     L.T, L.T - 1};
}

// ---------------------------------------------------------------- gradients! (one timestep per lane)
// Jacobian entries that do not depend on (x, u, θ) — 1248 of 1280 for synth32 — come from the generated tables
// M::JAC_CONST_* and are written by coalesced wave-wide stores; only the M::JAC_NVAR state-dependent entries
// are evaluated per timestep (M::dyn_jac_var_mem). Same `.=` semantics as src/dynamics.jl:45-46 every call.
template <class M>
__attribute__((noinline)) __device__ void gradients_large_fn(gdbl* base, int constrained) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT, W = waves_of<M>::value;
    const LargeArgs A = large_args_from_lds<M>(base);
    constrained = __builtin_amdgcn_readfirstlane(constrained);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), T = A.T, N = A.N;
    constexpr bool split = M::JAC_NVAR < n * n + n * m;
    if constexpr (split) {
        // The constant entries are written once per buffer lifetime: 10 KB per timestep and instance that would
        // otherwise be re-sent to HBM every iteration (530 MB per call for 512 synth32 instances — that alone
        // was half of this phase). ilqr_reset and ilqr_set_buffer on the Jacobians clear the flag.
        const bool fresh = A.scal[S_JAC_CONST] == 0.0;   // read by both waves BEFORE anyone may set it ...
        __syncthreads();                                  // ... so that both take the same branch (barriers inside)
        if (fresh) {
            constexpr int EFX = (n * n + 63) / 64, EFU = (n * m + 63) / 64;
            double cfx[EFX], cfu[EFU];
#pragma unroll
            for (int q = 0; q < EFX; ++q) { const int e = lane + 64 * q; cfx[q] = e < n * n ? M::JAC_CONST_FX[0][e] : 0.0; }
#pragma unroll
            for (int q = 0; q < EFU; ++q) { const int e = lane + 64 * q; cfu[q] = e < n * m ? M::JAC_CONST_FU[0][e] : 0.0; }
            for (int t = wave; t < N; t += W) {
#pragma unroll
                for (int q = 0; q < EFX; ++q) { const int e = lane + 64 * q; if ((n * n) % 64 == 0 || e < n * n) A.fx[(size_t)t * n * n + e] = cfx[q]; }
#pragma unroll
                for (int q = 0; q < EFU; ++q) { const int e = lane + 64 * q; if ((n * m) % 64 == 0 || e < n * m) A.fu[(size_t)t * n * m + e] = cfu[q]; }
            }
            __syncthreads();             // the state-dependent entries below overwrite some of these addresses
            if (lane == 0 && wave == 0) A.scal[S_JAC_CONST] = 1.0;
        }
    }
    for (int t = lane + 64 * wave; t < T; t += 64 * W) {                 // Hessians accumulate: each timestep exactly once
        double w[cdim<M::NW>::v];
        load_w<M::NW>((const double*)A.w, t, w);
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = A.xb[t * n + i];
        double* gxx = (double*)(A.gxx + (size_t)t * n * n);
        if (t < N) {
            double ut[m];
#pragma unroll
            for (int i = 0; i < m; ++i) ut[i] = A.ub[t * m + i];
            double* fx = (double*)(A.fx + (size_t)t * n * n);
            double* fu = (double*)(A.fu + (size_t)t * n * m);
            double* guu = (double*)(A.guu + (size_t)t * m * m);
            double* gux = (double*)(A.gux + (size_t)t * m * n);
            if constexpr (split) M::dyn_jac_var_mem(xt, ut, w, fx, fu);                       // `.=`  (src/dynamics.jl:45-46)
            else M::dyn_jac_mem(xt, ut, w, fx, fu);
            double gx[n], gu[m];
            M::cost_s_grad(xt, ut, w, gx, gu);                                                // `.=`  (src/costs.jl:61,65)
            M::cost_s_hess_acc(xt, ut, w, gxx, guu, gux);                                     // `.+=` (src/costs.jl:74-80)
            if constexpr (ncs > 0) {
                if (constrained) {                                                            // src/gradients.jl:54-80
                    double ct[ncs], ir[ncs];
                    const int off = t * ncs;
#pragma unroll
                    for (int i = 0; i < ncs; ++i) {
                        ir[i] = A.rho[off + i] * A.act[off + i];
                        ct[i] = A.lam[off + i] + ir[i] * A.c[off + i];
                    }
                    M::al_s(xt, ut, w, ct, ir, gx, gu, gxx, guu, gux);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) A.gx[t * n + i] = gx[i];
#pragma unroll
            for (int i = 0; i < m; ++i) A.gu[t * m + i] = gu[i];
        } else {
            double gx[n];
            M::cost_t_grad(xt, w, gx);
            M::cost_t_hess_acc(xt, w, gxx);
            if constexpr (nct > 0) {
                if (constrained) {
                    double ct[nct], ir[nct];
                    const int off = N * ncs;
#pragma unroll
                    for (int i = 0; i < nct; ++i) {
                        ir[i] = A.rho[off + i] * A.act[off + i];
                        ct[i] = A.lam[off + i] + ir[i] * A.c[off + i];
                    }
                    M::al_t(xt, w, ct, ir, gx, gxx);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) A.gx[t * n + i] = gx[i];
        }
    }
    __syncthreads();
}
template <class M>
__device__ __forceinline__ void gradients_large(Inst<M>& I, bool constrained) {
    ILQR_PROF_BEGIN();
    gradients_large_fn<M>(as_global(I.gbase), constrained ? 1 : 0);
    ILQR_PROF_END(I, PROF_GRAD);
}

// ---------------------------------------------------------------- backward_pass! (MFMA 16x16x4 tiles)
// One Riccati step (src/backward_pass.jl:42-90) for n = 32, m = 8 issues 132 tile MFMAs (64 cycles each on
// gfx950 ⇒ 8.4 k cycles of one matrix pipe) around a serial Cholesky/solve chain of ~4 k cycles. The instance's
// TWO waves (two SIMDs) split it:
//     both    stage fx, fu (HBM → registers a step ahead → LDS);   [T; ûx] = [fx fu]ᵀ P′, one column tile each
//     wave 0  Qx, Qux, Quu → potrf → potrs (K, k) → ûxt = Quu K → p, ∇L        (the serial chain)
//     wave 1  Qu, Qxx = T fx + gxx (overlaps wave 0's chain) → ûxt → P = Kᵀûxt + KᵀQux + QuxᵀK + Qxx
// Three workgroup barriers per timestep. Operand fragments of a product are read from zero-padded LDS matrices first
// (odd leading dimensions, transposition = stride pattern, no bounds checks), then the MFMAs issue back to back;
// Qxx never leaves wave 1's accumulators; the cost Hessians are prefetched in the D-layout of the tiles they are
// added to, by the wave that owns those tiles.
struct RiccatiOut {
    double gradient_norm; int potrf_info;
#if defined(ILQR_PROFILE) && defined(ILQR_PROFILE_SUB)
    double prof[6];
#endif
};

template <class M, bool STORE_VALUE>
__attribute__((noinline)) __device__ RiccatiOut backward_pass_large_fn(gdbl* base, gdbl* Qbase) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, NP = LD::NP, MP = LD::MP, ld = LD::ld, ldm = LD::ldm;
    constexpr int n4 = r4(n), m4 = r4(m), TN = NP / 16, TM = MP / 16, NT = 128;
    constexpr int EFX = (n * n + NT - 1) / NT, EFU = (n * m + NT - 1) / NT;
    static_assert(n <= 64 && m <= 16, "large path: nx <= 64 (one state component per lane), nu <= 16");
    static_assert(LD::total == large_lds_doubles(n, m), "LDS carve and host-side size disagree");
    static_assert(waves_of<M>::value == 2, "the Riccati step is written for two waves per instance");
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    struct { gdbl *fx, *fu, *gx, *gu, *gxx, *guu, *gux, *K, *k, *Lx, *Lu, *P, *p; int potrf_info; double prof[6]; } I;
    const LargeArgs A = large_args_from_lds<M>(base);
    I.fx = A.fx; I.fu = A.fu; I.gx = A.gx; I.gu = A.gu; I.gxx = A.gxx; I.guu = A.guu; I.gux = A.gux; I.K = A.K;
    I.k = A.k; I.Lx = A.Lx; I.Lu = A.Lu; I.P = A.P; I.p = A.p;
    I.potrf_info = 0;
    for (int q = 0; q < 6; ++q) I.prof[q] = 0.0;
    gdbl* const Qv = (STORE_VALUE && Qbase != nullptr) ? uniform_ptr(Qbase) : nullptr;   // optional action-value buffers (stage kernel only)
    const QLayout QL = make_qlayout(n, m, A.T);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = A.N, li = lane & 15, lk = lane >> 4;
    const bool w0 = wave == 0;
    double* S = lds_dyn;
    double *sP = S + LD::oP, *sFx = S + LD::oFx, *sT = S + LD::oT, *sFu = S + LD::oFu, *sUh = S + LD::oUh,
           *sQux = S + LD::oQux, *sK = S + LD::oK, *sUxt = S + LD::oUxt, *sQuu = S + LD::oQuu;
    double* sUxt0 = sT;                                                   // wave 0's copy of ûxt (T is dead by then)
    double *sp = S + LD::oVec, *sQx = sp + NP, *sQu = sQx + NP, *sk = sQu + MP, *sOut = sk + MP;
    for (int e = tid; e < LD::oLay; e += NT) S[e] = 0.0;                  // the tile padding must read as zero (the Layout copy behind it stays)
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) {                               // P[H] .= gxx[H]  (:39)
        const double v = I.gxx[(size_t)N * n * n + e];
        sP[(e / n) * ld + e % n] = v;
        if (STORE_VALUE) I.P[(size_t)N * n * n + e] = v;
    }
    for (int i = tid; i < n; i += NT) {                                   // p[H] .= gx[H]   (:40)
        const double v = I.gx[N * n + i];
        sp[i] = v;
        if (STORE_VALUE) I.p[N * n + i] = v;
    }
    // register prefetch of step t's operands (each wave fetches what it will consume)
    double rfx[EFX], rfu[EFU], rgv = 0.0;
    double4_t rgxx[TN * TN], rguu[TM * TM], rgux[TM * TN];
    auto fetch_early = [&](int t) {
#pragma unroll
        for (int q = 0; q < EFX; ++q) {
            const int e = tid + NT * q;
            if constexpr ((n * n) % NT == 0) rfx[q] = I.fx[(size_t)t * n * n + e];
            else rfx[q] = e < n * n ? I.fx[(size_t)t * n * n + e] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < EFU; ++q) {
            const int e = tid + NT * q;
            if constexpr ((n * m) % NT == 0) rfu[q] = I.fu[(size_t)t * n * m + e];
            else rfu[q] = e < n * m ? I.fu[(size_t)t * n * m + e] : 0.0;
        }
        rgv = w0 ? I.gx[t * n + (lane < n ? lane : 0)] : I.gu[t * m + (lane < m ? lane : 0)];
    };
    auto fetch_late = [&](int t) {
        if (w0) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int c = 0; c < TM; ++c) rguu[a * TM + c] = tile_load_global<m, m>(I.guu + (size_t)t * m * m, 16 * a, 16 * c, li, lk);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int c = 0; c < TN; ++c) rgux[a * TN + c] = tile_load_global<m, n>(I.gux + (size_t)t * m * n, 16 * a, 16 * c, li, lk);
        } else {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int c = 0; c < TN; ++c) rgxx[a * TN + c] = tile_load_global<n, n>(I.gxx + (size_t)t * n * n, 16 * a, 16 * c, li, lk);
        }
    };
#pragma unroll
    for (int q = 0; q < TN * TN; ++q) rgxx[q] = double4_t{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < TM * TM; ++q) rguu[q] = double4_t{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < TM * TN; ++q) rgux[q] = double4_t{0, 0, 0, 0};
    if (N > 0) { fetch_early(N - 1); fetch_late(N - 1); }
    double gmax = 0.0;
    __syncthreads();
    for (int t = N - 1; t >= 0; --t) {                                    // (:42)
        ILQR_SUB_BEGIN();
#pragma unroll
        for (int q = 0; q < EFX; ++q) {
            const int e = tid + NT * q;
            if ((n * n) % NT == 0 || e < n * n) sFx[(e / n) * ld + e % n] = rfx[q];
        }
#pragma unroll
        for (int q = 0; q < EFU; ++q) {
            const int e = tid + NT * q;
            if ((n * m) % NT == 0 || e < n * m) sFu[(e / n) * ld + e % n] = rfu[q];
        }
        const double gv = rgv;
        __syncthreads();                                                  // (1) fx, fu staged; P′, p′ of the previous step visible
        fetch_early(t > 0 ? t - 1 : 0);                                   // (t = 0: a harmless re-read instead of a branch)
        // Qx = fx^T p' + gx (wave 0), Qu = fu^T p' + gu (wave 1)   (:44-49): one output per lane
        {
            const int col = w0 ? (lane < n ? lane : 0) : (lane < m ? lane : 0);
            const double* colp = w0 ? sFx + col * ld : sFu + col * ld;
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < n; ++l) acc += colp[l] * sp[l];
            if (w0 && lane < n) sQx[lane] = acc + gv;
            if (!w0 && lane < m) sQu[lane] = acc + gv;
        }
        ILQR_SUB_MARK(I, 0);
        // [T; ux_hat] = [fx fu]^T P'  (:52, :57, :62): each wave one column tile (wave 0 all of them when there is one)
        {
            constexpr int TR = TN + TM, TC = TN == 2 ? 1 : TN;
            if (TN == 2 || w0) {
                const int c0 = TN == 2 ? wave : 0;
                double4_t acc[TR * TC];
#pragma unroll
                for (int q = 0; q < TR * TC; ++q) acc[q] = double4_t{0, 0, 0, 0};
                const double* Aop[TR]; int Arow[TR];
#pragma unroll
                for (int a = 0; a < TR; ++a) { Aop[a] = a < TN ? sFx : sFu; Arow[a] = a < TN ? 16 * a : 16 * (a - TN); }
                tiles_mac<TR, TC, n4, ld, 1, 1, ld>(acc, Aop, Arow, sP, li, lk, c0);
#pragma unroll
                for (int a = 0; a < TR; ++a)
#pragma unroll
                    for (int c = 0; c < TC; ++c) {
                        if (a < TN) tile_store<ld>(sT, acc[a * TC + c], 16 * a, 16 * (c0 + c), li, lk);
                        else tile_store<ldm>(sUh, acc[a * TC + c], 16 * (a - TN), 16 * (c0 + c), li, lk);
                    }
            }
        }
        __syncthreads();                                                  // (2) T, ux_hat, Qx, Qu complete
        ILQR_SUB_MARK(I, 1);
        double4_t qxx[TN * TN];
#pragma unroll
        for (int q = 0; q < TN * TN; ++q) qxx[q] = double4_t{0, 0, 0, 0};
        if (w0) {
            // Qux = ux_hat fx + gux (:63-64), Quu = ux_hat fu + guu (:58-59): one pass over the ux_hat fragments
            {
                constexpr int KS = n4 / 4, TC = TN + TM;
                double4_t acc[TM * TC];
#pragma unroll
                for (int q = 0; q < TM * TC; ++q) acc[q] = double4_t{0, 0, 0, 0};
                double fa[TM][KS], fb[TC][KS];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int sx = 0; sx < KS; ++sx) fa[a][sx] = sUh[(4 * sx + lk) * ldm + 16 * a + li];
#pragma unroll
                for (int c = 0; c < TC; ++c)
#pragma unroll
                    for (int sx = 0; sx < KS; ++sx)
                        fb[c][sx] = c < TN ? sFx[(16 * c + li) * ld + 4 * sx + lk] : sFu[(16 * (c - TN) + li) * ld + 4 * sx + lk];
#pragma unroll
                for (int sx = 0; sx < KS; ++sx)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int c = 0; c < TC; ++c)
                            acc[a * TC + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a][sx], fb[c][sx], acc[a * TC + c], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int c = 0; c < TC; ++c) {
                        if (c < TN) tile_store<ldm>(sQux, acc[a * TC + c] + rgux[a * TN + c], 16 * a, 16 * c, li, lk);
                        else tile_store<ldm>(sQuu, acc[a * TC + c] + rguu[a * TM + (c - TN)], 16 * a, 16 * (c - TN), li, lk);
                    }
            }
            wave_lds_fence();                                             // this wave's own Quu, Qux writes
            if (STORE_VALUE && Qv != nullptr) {                           // policy.action_value.* (src/data/policy.jl:58-64)
                for (int e = lane; e < m * n; e += 64) Qv[QL.Qux + (size_t)t * m * n + e] = sQux[(e / m) * ldm + e % m];
                for (int e = lane; e < m * m; e += 64) Qv[QL.Quu + (size_t)t * m * m + e] = sQuu[(e / m) * ldm + e % m];
                if (lane < n) Qv[QL.Qx + t * n + lane] = sQx[lane];
            }
            ILQR_SUB_MARK(I, 2);
            // potrf('U') on wave-uniform registers (info ignored, :68-69)
            double Uc[m * m], Ur[m];
#pragma unroll
            for (int j = 0; j < m; ++j)
#pragma unroll
                for (int i = 0; i < m; ++i) Uc[j * m + i] = (i <= j) ? sQuu[j * ldm + i] : 0.0;
            const int info = potrf_U<m>(Uc, Ur);
            if (info != 0 && I.potrf_info == 0) I.potrf_info = info;
            fetch_late(t > 0 ? t - 1 : 0);
            ILQR_SUB_MARK(I, 3);
            // potrs('U'): column j of Qux per lane, Qu on lane n   (:70-75)
            // (nx = 64 leaves no lane for k: a second pass on lane 0)
#pragma unroll
            for (int pass = 0; pass < (n < 64 ? 1 : 2); ++pass) {
                const bool mine = n < 64 ? lane <= n : (pass == 0 || lane == 0);
                if (mine) {
                    const int j = (n < 64 || pass == 0) ? lane : n;
                    double b[m];
#pragma unroll
                    for (int i = 0; i < m; ++i) b[i] = j < n ? sQux[j * ldm + i] : sQu[i];
                    potrs_U_rdiag<m, 1>(Uc, Ur, b);       // inverted diagonal from the factorisation: no divisions here
#pragma unroll
                    for (int i = 0; i < m; ++i) {
                        const double v = b[i] * -1.0;
                        if (j < n) { sK[j * ldm + i] = v; I.K[(size_t)t * m * n + j * m + i] = v; }
                        else { sk[i] = v; I.k[t * m + i] = v; }
                    }
                }
            }
        } else {
            // Qxx = T fx + gxx (:53-54): stays in this wave's accumulators, overlaps wave 0's Cholesky chain
            const double* Aop[TN]; int Arow[TN];
#pragma unroll
            for (int a = 0; a < TN; ++a) { Aop[a] = sT; Arow[a] = 16 * a; }
            tiles_mac<TN, TN, n4, 1, ld, 1, ld>(qxx, Aop, Arow, sFx, li, lk);
#pragma unroll
            for (int q = 0; q < TN * TN; ++q) qxx[q] = qxx[q] + rgxx[q];
            if (STORE_VALUE && Qv != nullptr) {
                if (lane < m) Qv[QL.Qu + t * m + lane] = sQu[lane];
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int c = 0; c < TN; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * a + lk + 4 * r, col = 16 * c + li;
                            if (row < n && col < n) Qv[QL.Qxx + (size_t)t * n * n + col * n + row] = qxx[a * TN + c][r];
                        }
            }
            fetch_late(t > 0 ? t - 1 : 0);
        }
        __syncthreads();                                                  // (3) K, k in LDS; T no longer needed
        ILQR_SUB_MARK(I, 4);
        // ux_tmp = Quu K   (:79): both waves, each into its own buffer (saves a barrier)
        {
            double4_t acc[TM * TN];
#pragma unroll
            for (int q = 0; q < TM * TN; ++q) acc[q] = double4_t{0, 0, 0, 0};
            const double* Aop[TM]; int Arow[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) { Aop[a] = sQuu; Arow[a] = 16 * a; }
            tiles_mac<TM, TN, m4, 1, ldm, 1, ldm>(acc, Aop, Arow, sK, li, lk);
            double* dst = w0 ? sUxt0 : sUxt;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int c = 0; c < TN; ++c) tile_store<ldm>(dst, acc[a * TN + c], 16 * a, 16 * c, li, lk);
            wave_lds_fence();
        }
        if (!w0) {
            // P = K^T ux_tmp + K^T Qux + Qux^T K + Qxx   (:81-84), accumulated in this order
            double4_t acc[TN * TN];
#pragma unroll
            for (int q = 0; q < TN * TN; ++q) acc[q] = double4_t{0, 0, 0, 0};
            const double* Ak[TN]; const double* Aq[TN]; int Arow[TN];
#pragma unroll
            for (int a = 0; a < TN; ++a) { Ak[a] = sK; Aq[a] = sQux; Arow[a] = 16 * a; }
            tiles_mac<TN, TN, m4, ldm, 1, 1, ldm>(acc, Ak, Arow, sUxt, li, lk);
            tiles_mac<TN, TN, m4, ldm, 1, 1, ldm>(acc, Ak, Arow, sQux, li, lk);
            tiles_mac<TN, TN, m4, ldm, 1, 1, ldm>(acc, Aq, Arow, sK, li, lk);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int c = 0; c < TN; ++c) {
                    const double4_t v = acc[a * TN + c] + qxx[a * TN + c];
                    tile_store<ld>(sP, v, 16 * a, 16 * c, li, lk);
                    if (STORE_VALUE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * a + lk + 4 * r, col = 16 * c + li;
                            if (row < n && col < n) I.P[(size_t)t * n * n + col * n + row] = v[r];
                        }
                    }
                }
        } else {
            // p = ux_tmp^T k + K^T Qu + Qux^T k + Qx   (:86-89); Lagrangian gradient (src/solve.jl:73-81)
            if (lane < n) {
                double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) {
                    a1 += sUxt0[lane * ldm + l] * sk[l];
                    a2 += sK[lane * ldm + l] * sQu[l];
                    a3 += sQux[lane * ldm + l] * sk[l];
                }
                const double pn = ((a1 + a2) + a3) + sQx[lane];
                const double Lx = sQx[lane] - pn;
                gmax = nanmax(gmax, fabs(Lx));
                I.Lx[t * n + lane] = Lx;
                if (STORE_VALUE) I.p[t * n + lane] = pn;
                sp[lane] = pn;                                            // p' of the next step (read after barrier (1))
            }
            if (lane < m) {
                gmax = nanmax(gmax, fabs(sQu[lane]));
                I.Lu[t * m + lane] = sQu[lane];
            }
        }
        ILQR_SUB_MARK(I, 5);
    }
    __syncthreads();
    // the serial chain lived on wave 0: hand its scalars to both waves (identical control flow afterwards)
    const double gn = wave_max(gmax);
    if (tid == 0) { sOut[0] = gn; sOut[1] = (double)I.potrf_info; }
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
// The closed-loop rollout (src/rollout.jl:1-31) and, on the first line-search trial, the sensitivity recursion with its
// dot product (src/data/methods.jl:42-54, src/forward_pass.jl:20) are independent forward sweeps over t: wave 0 runs
// the rollout while wave 1 runs the sensitivities. Each works in its own LDS staging (wave-local fences, no workgroup
// barrier inside the sweep); one barrier at the end.
//
// rollout: u = αk + ū + Kx − Kx̄ (:24-28) with K x on lanes 0..m-1 and K x̄ on lanes 32..32+m-1 at the same time (K_t
// staged in LDS, fetched from HBM a step ahead); dynamics: lane i evaluates row i of the affine part
// y = DYN_AFF [x; u; 1] with its coefficient row held in registers for the whole rollout, plus the generated
// remainder (wave-cooperative trig). x and u travel between lanes through LDS.
template <class M>
__device__ __forceinline__ void rollout_large_body(const LargeArgs& A, double alpha, int lane) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, EK = (m * n + 63) / 64;
    static_assert(n <= 64 && m <= 16, "large path: nx <= 64 (one state component per lane), nu <= 16");
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const int N = A.N;
    double* sK = lds_dyn + LD::oK;                     // packed m x n
    double* sx = lds_dyn + LD::oVec;                   // x (n), then x̄ (n), then u (m)
    double* sxb = sx + LD::NP;
    double* su = sxb + LD::NP;
    const int row = lane < n ? lane : n - 1;
    double aff[n + m + 1];
#pragma unroll
    for (int j = 0; j < n + m + 1; ++j) aff[j] = M::DYN_AFF[row][j];
    double xl = lane < n ? A.xb[lane] : 0.0;                              // x[1] = x̄[1]  (:19)
    if (lane < n) A.x[lane] = xl;
    double rK[EK];
    auto fetchK = [&](int t) {
#pragma unroll
        for (int q = 0; q < EK; ++q) { const int e = lane + 64 * q; rK[q] = ((m * n) % 64 == 0 || e < m * n) ? A.K[(size_t)t * m * n + e] : 0.0; }
    };
    if (N > 0) fetchK(0);
    double xbl = lane < n ? A.xb[lane] : 0.0;                             // x̄_t component of this lane
    const bool hi = lane >= 32;                                           // lanes 32.. work on K x̄
    const int ui = (lane & 31) < m ? (lane & 31) : m - 1;
    for (int t = 0; t < N; ++t) {
#pragma unroll
        for (int q = 0; q < EK; ++q) { const int e = lane + 64 * q; if ((m * n) % 64 == 0 || e < m * n) sK[e] = rK[q]; }
        if (lane < n) { sx[lane] = xl; sxb[lane] = xbl; }
        wave_lds_fence();
        if (t + 1 < N) fetchK(t + 1);
        const double xb_next = lane < n ? A.xb[(t + 1) * n + lane] : 0.0;
        const double kv = A.k[t * m + ui], ubv = A.ub[t * m + ui];
        double xa[n];
#pragma unroll
        for (int j = 0; j < n; ++j) xa[j] = sx[j];
        const double* src = hi ? sxb : sx;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < n; ++j) acc += sK[j * m + ui] * src[j];       // K x (lanes < 32) | K x̄ (lanes >= 32)
        const double a2 = __shfl(acc, lane + 32);
        double v = kv * alpha;                                            // (:24-25)
        v += ubv;                                                         // (:26)
        v += acc;                                                         // (:27)
        v += -1.0 * a2;                                                   // (:28)
        if (lane < m) { su[lane] = v; A.u[t * m + lane] = v; }
        wave_lds_fence();
        double ua[m];
#pragma unroll
        for (int j = 0; j < m; ++j) ua[j] = su[j];
        double y = aff[n + m];                                            // (:29) row `lane` of the dynamics
#pragma unroll
        for (int j = 0; j < n; ++j) y += aff[j] * xa[j];
#pragma unroll
        for (int j = 0; j < m; ++j) y += aff[n + j] * ua[j];
        if constexpr (M::DYN_HAS_REM) {
            double w[cdim<M::NW>::v], r[n];
            load_w<M::NW>((const double*)A.w, t, w);
            M::dyn_rem_wave(lane, xa, ua, w, r);
            double rl = r[0];
#pragma unroll
            for (int i = 1; i < n; ++i) rl = (lane == i) ? r[i] : rl;
            y += rl;
        }
        xl = y;
        xbl = xb_next;
        if (lane < n) A.x[(t + 1) * n + lane] = y;
        wave_lds_fence();
    }
}

// Δu = k + KΔx on lanes 0..m-1, Δx⁺ = fuΔu + fxΔx one row per lane; K, fx, fu staged in LDS (HBM fetch a step ahead).
// Staging buffers disjoint from the rollout's.
template <class M>
__device__ __forceinline__ double delta_large_body(const LargeArgs& A, int lane) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, ld = LD::ld;
    constexpr int EK = (m * n + 63) / 64, EFX = (n * n + 63) / 64, EFU = (n * m + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const int N = A.N;
    double *sK = lds_dyn + LD::oQux, *sFx = lds_dyn + LD::oFx, *sFu = lds_dyn + LD::oFu;
    double* zx = lds_dyn + LD::oVec + 2 * LD::NP + LD::MP;        // n   (behind the rollout's x, x̄, u)
    double* zu = zx + LD::NP;                                     // m
    double rK[EK], rfx[EFX], rfu[EFU];
    auto fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < EK; ++q) { const int e = lane + 64 * q; rK[q] = ((m * n) % 64 == 0 || e < m * n) ? A.K[(size_t)t * m * n + e] : 0.0; }
#pragma unroll
        for (int q = 0; q < EFX; ++q) { const int e = lane + 64 * q; rfx[q] = ((n * n) % 64 == 0 || e < n * n) ? A.fx[(size_t)t * n * n + e] : 0.0; }
#pragma unroll
        for (int q = 0; q < EFU; ++q) { const int e = lane + 64 * q; rfu[q] = ((n * m) % 64 == 0 || e < n * m) ? A.fu[(size_t)t * n * m + e] : 0.0; }
    };
    if (N > 0) fetch(0);
    if (lane < n) zx[lane] = 0.0;
    double dpart = 0.0;
    const int ui = lane < m ? lane : m - 1, xi = lane < n ? lane : n - 1;
    for (int t = 0; t < N; ++t) {
#pragma unroll
        for (int q = 0; q < EK; ++q) { const int e = lane + 64 * q; if ((m * n) % 64 == 0 || e < m * n) sK[e] = rK[q]; }
#pragma unroll
        for (int q = 0; q < EFX; ++q) { const int e = lane + 64 * q; if ((n * n) % 64 == 0 || e < n * n) sFx[(e / n) * ld + e % n] = rfx[q]; }
#pragma unroll
        for (int q = 0; q < EFU; ++q) { const int e = lane + 64 * q; if ((n * m) % 64 == 0 || e < n * m) sFu[(e / n) * ld + e % n] = rfu[q]; }
        wave_lds_fence();
        if (t + 1 < N) fetch(t + 1);
        const double kv = A.k[t * m + ui], Luv = A.Lu[t * m + ui], Lxv = A.Lx[t * n + xi];
        double za[n];
#pragma unroll
        for (int j = 0; j < n; ++j) za[j] = zx[j];
        double acc = 0.0;                                                 // Δu = k + K Δx
#pragma unroll
        for (int j = 0; j < n; ++j) acc += sK[j * m + ui] * za[j];
        const double du = kv + acc;
        if (lane < m) { zu[lane] = du; dpart += Luv * du; }
        if (lane < n) dpart += Lxv * zx[xi];
        wave_lds_fence();
        double a1 = 0.0, a2 = 0.0;                                        // Δx⁺ = fu Δu + fx Δx
#pragma unroll
        for (int j = 0; j < m; ++j) a1 += sFu[j * ld + xi] * zu[j];
#pragma unroll
        for (int j = 0; j < n; ++j) a2 += sFx[j * ld + xi] * za[j];
        wave_lds_fence();
        if (lane < n) zx[lane] = a1 + a2;
        wave_lds_fence();
    }
    return wave_sum(dpart);
}

template <class M>
__attribute__((noinline)) __device__ double forward_sweep_large_fn(gdbl* base, double alpha, int want_delta) {
    typedef LargeDims<M> LD;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    const LargeArgs A = large_args_from_lds<M>(base);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    want_delta = __builtin_amdgcn_readfirstlane(want_delta);
    double* sOut = lds_dyn + LD::oVec + 4 * LD::NP + 4 * LD::MP;
    double d = 0.0;
    if (wave == 0) rollout_large_body<M>(A, alpha, lane);
    else if (want_delta) d = delta_large_body<M>(A, lane);
    __syncthreads();
    if (want_delta) {                     // hand wave 1's scalar to both waves (identical control flow afterwards)
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

}  // namespace ilqr
