// Batched iLQR / augmented-Lagrangian solve for gfx950 (MI355X), hand-written HIP.
//
// Mapping: ONE WAVEFRONT (64 lanes, one workgroup) PER PROBLEM INSTANCE.
//   * the per-instance working set (trajectories, Jacobians, gradients, gains)
//     lives in LDS for the whole solve; HBM sees one load and one store of it;
//   * time-parallel stages (cost, linearisation) put one timestep on each lane;
//   * time-sequential stages (Riccati recursion, closed-loop rollout) run the
//     recursion wave-uniformly out of registers, reading LDS by broadcast;
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
};

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

template <int N_> struct cdim { static constexpr int v = N_ > 0 ? N_ : 1; };

// Per-instance context. LDS pointers first, then HBM pointers, then the
// wave-uniform SolverData scalars (src/data/solver.jl:4-18) kept in registers.
template <class M>
struct Inst {
    static constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    double *xb, *ub, *x, *u, *fx, *fu, *gx, *gu, *K, *k, *Lx, *Lu, *c, *lam, *rho, *act;
    double *gxx, *guu, *gux, *P, *p, *scal;
    int T, N, C, lane;
    double objective, max_violation, step_size, gradient_norm;
    int status, iterations, outer_iterations, potrf_info, rollouts, states_eq_nominal;
};

// ------------------------------------------------------------------ cost!
// One timestep per lane. upd_J: evaluate J and the active set at (X,U)
// (src/augmented_lagrangian.jl:39-85); upd_viol: overwrite the violations
// buffer and compute max_violation at (X,U) (src/data/constraints.jl:23-46).
template <class M>
__device__ void cost_pass(Inst<M>& I, const double* X, const double* U, bool upd_J, bool upd_viol,
                          bool constrained, double& J_out, double& viol_out) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    double Jp = 0.0, vp = 0.0;
    const double w[cdim<M::NW>::v] = {0.0};
    for (int t = I.lane; t < I.T; t += 64) {
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = X[t * n + i];
        if (t < I.N) {
            double ut[m];
#pragma unroll
            for (int i = 0; i < m; ++i) ut[i] = U[t * m + i];
            if (upd_J) Jp += M::cost_s(xt, ut, w);
            if constexpr (ncs > 0) {
                if (constrained) {
                    double cv[ncs];
                    M::con_s(xt, ut, w, cv);
                    const int off = t * ncs;
                    if (upd_J) {
                        double dot = 0.0, pen = 0.0;
#pragma unroll
                        for (int i = 0; i < ncs; ++i) {
                            const double lam = I.lam[off + i];
                            const bool ineq = (M::INEQ_S >> i) & 1ull;
                            const bool inactive = ineq && cv[i] < 0.0 && lam == 0.0;
                            I.act[off + i] = inactive ? 0.0 : 1.0;
                            dot += lam * cv[i];
                            if (!inactive) pen += 0.5 * I.rho[off + i] * (cv[i] * cv[i]);
                        }
                        Jp += dot;
                        Jp += pen;
                    }
                    if (upd_viol) {
#pragma unroll
                        for (int i = 0; i < ncs; ++i) {
                            I.c[off + i] = cv[i];
                            const bool ineq = (M::INEQ_S >> i) & 1ull;
                            vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                        }
                    }
                }
            }
        } else {
            if (upd_J) Jp += M::cost_t(xt, w);
            if constexpr (nct > 0) {
                if (constrained) {
                    double cv[nct];
                    M::con_t(xt, w, cv);
                    const int off = I.N * ncs;
                    if (upd_J) {
                        double dot = 0.0, pen = 0.0;
#pragma unroll
                        for (int i = 0; i < nct; ++i) {
                            const double lam = I.lam[off + i];
                            const bool ineq = (M::INEQ_T >> i) & 1ull;
                            const bool inactive = ineq && cv[i] < 0.0 && lam == 0.0;
                            I.act[off + i] = inactive ? 0.0 : 1.0;
                            dot += lam * cv[i];
                            if (!inactive) pen += 0.5 * I.rho[off + i] * (cv[i] * cv[i]);
                        }
                        Jp += dot;
                        Jp += pen;
                    }
                    if (upd_viol) {
#pragma unroll
                        for (int i = 0; i < nct; ++i) {
                            I.c[off + i] = cv[i];
                            const bool ineq = (M::INEQ_T >> i) & 1ull;
                            vp = nanmax(vp, ineq ? nanmax(0.0, cv[i]) : fabs(cv[i]));
                        }
                    }
                }
            }
        }
    }
    J_out = wave_sum(Jp);
    viol_out = wave_max(vp);
    __syncthreads();
}

// cost!(data, problem, mode) — src/data/methods.jl:13-30.
// The violations buffer / max_violation are ALWAYS taken at problem.states
// (SURVEY Appendix A, Q2); when states == nominal bitwise one pass suffices.
template <class M>
__device__ void cost_bang(Inst<M>& I, bool mode_current, bool constrained) {
    double J, v;
    if (mode_current || I.states_eq_nominal) {
        cost_pass<M>(I, mode_current ? I.x : I.xb, mode_current ? I.u : I.ub, true, true, constrained, J, v);
        I.objective = J;
        if (constrained) I.max_violation = v;
    } else {
        cost_pass<M>(I, I.xb, I.ub, true, false, constrained, J, v);
        I.objective = J;
        if (constrained) {
            double J2;
            cost_pass<M>(I, I.x, I.u, false, true, constrained, J2, v);
            I.max_violation = v;
        }
    }
}

// -------------------------------------------------------------- gradients!
// One timestep per lane: dynamics Jacobians (`.=`), cost gradients (`.=`),
// cost Hessians (`.+=` — accumulate, Appendix A Q1) and the Gauss-Newton AL
// terms of src/gradients.jl:54-80 using the violations BUFFER (Q2).
template <class M>
__device__ void gradients(Inst<M>& I, bool constrained) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    const double w[cdim<M::NW>::v] = {0.0};
    for (int t = I.lane; t < I.T; t += 64) {
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
            double gx[n], gu[m], hxx[n * n], huu[m * m], hux[m * n];
            M::cost_s_grad(xt, ut, w, gx, gu);
            M::cost_s_hess(xt, ut, w, hxx, huu, hux);
            double axx[n * n], auu[m * m], aux[m * n];
#pragma unroll
            for (int i = 0; i < n * n; ++i) axx[i] = I.gxx[t * n * n + i] + hxx[i];
#pragma unroll
            for (int i = 0; i < m * m; ++i) auu[i] = I.guu[t * m * m + i] + huu[i];
#pragma unroll
            for (int i = 0; i < m * n; ++i) aux[i] = I.gux[t * m * n + i] + hux[i];
            if constexpr (ncs > 0) {
                if (constrained) {
                    double cx[ncs * n], cu[ncs * m], ct[ncs], ir[ncs];
                    M::con_s_jac(xt, ut, w, cx, cu);
                    const int off = t * ncs;
#pragma unroll
                    for (int i = 0; i < ncs; ++i) {
                        ir[i] = I.rho[off + i] * I.act[off + i];
                        ct[i] = I.lam[off + i] + ir[i] * I.c[off + i];
                    }
#pragma unroll
                    for (int j = 0; j < n; ++j) {
                        double acc = 0.0;
#pragma unroll
                        for (int i = 0; i < ncs; ++i) acc += cx[j * ncs + i] * ct[i];
                        gx[j] += acc;
                    }
#pragma unroll
                    for (int j = 0; j < n; ++j)
#pragma unroll
                        for (int i2 = 0; i2 < n; ++i2) {
                            double acc = 0.0;
#pragma unroll
                            for (int i = 0; i < ncs; ++i) acc += cx[i2 * ncs + i] * (ir[i] * cx[j * ncs + i]);
                            axx[j * n + i2] += acc;
                        }
#pragma unroll
                    for (int j = 0; j < m; ++j) {
                        double acc = 0.0;
#pragma unroll
                        for (int i = 0; i < ncs; ++i) acc += cu[j * ncs + i] * ct[i];
                        gu[j] += acc;
                    }
#pragma unroll
                    for (int j = 0; j < m; ++j)
#pragma unroll
                        for (int i2 = 0; i2 < m; ++i2) {
                            double acc = 0.0;
#pragma unroll
                            for (int i = 0; i < ncs; ++i) acc += cu[i2 * ncs + i] * (ir[i] * cu[j * ncs + i]);
                            auu[j * m + i2] += acc;
                        }
#pragma unroll
                    for (int j = 0; j < n; ++j)
#pragma unroll
                        for (int i2 = 0; i2 < m; ++i2) {
                            double acc = 0.0;
#pragma unroll
                            for (int i = 0; i < ncs; ++i) acc += cu[i2 * ncs + i] * (ir[i] * cx[j * ncs + i]);
                            aux[j * m + i2] += acc;
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) I.gx[t * n + i] = gx[i];
#pragma unroll
            for (int i = 0; i < m; ++i) I.gu[t * m + i] = gu[i];
#pragma unroll
            for (int i = 0; i < n * n; ++i) I.gxx[t * n * n + i] = axx[i];
#pragma unroll
            for (int i = 0; i < m * m; ++i) I.guu[t * m * m + i] = auu[i];
#pragma unroll
            for (int i = 0; i < m * n; ++i) I.gux[t * m * n + i] = aux[i];
        } else {
            double gx[n], hxx[n * n], axx[n * n];
            M::cost_t_grad(xt, w, gx);
            M::cost_t_hess(xt, w, hxx);
#pragma unroll
            for (int i = 0; i < n * n; ++i) axx[i] = I.gxx[t * n * n + i] + hxx[i];
            if constexpr (nct > 0) {
                if (constrained) {
                    double cx[nct * n], ct[nct], ir[nct];
                    M::con_t_jac(xt, w, cx);
                    const int off = I.N * ncs;
#pragma unroll
                    for (int i = 0; i < nct; ++i) {
                        ir[i] = I.rho[off + i] * I.act[off + i];
                        ct[i] = I.lam[off + i] + ir[i] * I.c[off + i];
                    }
#pragma unroll
                    for (int j = 0; j < n; ++j) {
                        double acc = 0.0;
#pragma unroll
                        for (int i = 0; i < nct; ++i) acc += cx[j * nct + i] * ct[i];
                        gx[j] += acc;
                    }
#pragma unroll
                    for (int j = 0; j < n; ++j)
#pragma unroll
                        for (int i2 = 0; i2 < n; ++i2) {
                            double acc = 0.0;
#pragma unroll
                            for (int i = 0; i < nct; ++i) acc += cx[i2 * nct + i] * (ir[i] * cx[j * nct + i]);
                            axx[j * n + i2] += acc;
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) I.gx[t * n + i] = gx[i];
#pragma unroll
            for (int i = 0; i < n * n; ++i) I.gxx[t * n * n + i] = axx[i];
        }
    }
    __syncthreads();
}

// ------------------------------------------------ LAPACK potrf('U') / potrs('U')
// Unblocked right-looking order of dpotf2; returns LAPACK info (ignored by the
// reference, src/backward_pass.jl:69 — we only record it).
template <int m>
__device__ __forceinline__ int potrf_U(double (&A)[m * m]) {
#pragma unroll
    for (int j = 0; j < m; ++j) {
        double ajj = A[j * m + j];
#pragma unroll
        for (int l = 0; l < j; ++l) ajj -= A[j * m + l] * A[j * m + l];
        if (!(ajj > 0.0)) { A[j * m + j] = ajj; return j + 1; }
        ajj = sqrt(ajj);
        A[j * m + j] = ajj;
#pragma unroll
        for (int c = j + 1; c < m; ++c) {
            double v = A[c * m + j];
#pragma unroll
            for (int l = 0; l < j; ++l) v -= A[j * m + l] * A[c * m + l];
            A[c * m + j] = v / ajj;
        }
    }
    return 0;
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

// ---------------------------------------------------------- backward_pass!
// Sequential Riccati recursion, wave-uniform, value function in registers.
// Also produces the Lagrangian gradient (src/solve.jl:67-83) and its ∞-norm.
template <class M, bool STORE_VALUE>
__device__ void backward_pass(Inst<M>& I) {
    constexpr int n = M::NX, m = M::NU;
    const int N = I.N;
    double P[n * n], p[n];
#pragma unroll
    for (int i = 0; i < n * n; ++i) P[i] = I.gxx[N * n * n + i];       // P[H] .= gxx[H]  (:39)
#pragma unroll
    for (int i = 0; i < n; ++i) p[i] = I.gx[N * n + i];                // p[H] .= gx[H]   (:40)
    if (STORE_VALUE && I.lane == 0) {
#pragma unroll
        for (int i = 0; i < n * n; ++i) I.P[N * n * n + i] = P[i];
#pragma unroll
        for (int i = 0; i < n; ++i) I.p[N * n + i] = p[i];
    }
    double gnorm = 0.0;
    // software prefetch of the HBM-resident accumulated Hessians, one step ahead
    double nxx[n * n], nuu[m * m], nux[m * n];
    if (N > 0) {
#pragma unroll
        for (int i = 0; i < n * n; ++i) nxx[i] = I.gxx[(N - 1) * n * n + i];
#pragma unroll
        for (int i = 0; i < m * m; ++i) nuu[i] = I.guu[(N - 1) * m * m + i];
#pragma unroll
        for (int i = 0; i < m * n; ++i) nux[i] = I.gux[(N - 1) * m * n + i];
    }
    for (int t = N - 1; t >= 0; --t) {                                  // (:42)
        double gxx[n * n], guu[m * m], gux[m * n];
#pragma unroll
        for (int i = 0; i < n * n; ++i) gxx[i] = nxx[i];
#pragma unroll
        for (int i = 0; i < m * m; ++i) guu[i] = nuu[i];
#pragma unroll
        for (int i = 0; i < m * n; ++i) gux[i] = nux[i];
        if (t > 0) {
#pragma unroll
            for (int i = 0; i < n * n; ++i) nxx[i] = I.gxx[(t - 1) * n * n + i];
#pragma unroll
            for (int i = 0; i < m * m; ++i) nuu[i] = I.guu[(t - 1) * m * m + i];
#pragma unroll
            for (int i = 0; i < m * n; ++i) nux[i] = I.gux[(t - 1) * m * n + i];
        }
        double fx[n * n], fu[n * m], gx[n], gu[m];
#pragma unroll
        for (int i = 0; i < n * n; ++i) fx[i] = I.fx[t * n * n + i];
#pragma unroll
        for (int i = 0; i < n * m; ++i) fu[i] = I.fu[t * n * m + i];
#pragma unroll
        for (int i = 0; i < n; ++i) gx[i] = I.gx[t * n + i];
#pragma unroll
        for (int i = 0; i < m; ++i) gu[i] = I.gu[t * m + i];

        double Qx[n], Qu[m], Qxx[n * n], Quu[m * m], Qux[m * n];
        // Qx = fxᵀp' + gx   (:44-45) ; Qu = fuᵀp' + gu   (:48-49)
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < n; ++l) acc += fx[i * n + l] * p[l];
            Qx[i] = acc + gx[i];
        }
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < n; ++l) acc += fu[i * n + l] * p[l];
            Qu[i] = acc + gu[i];
        }
        // Qxx = (fxᵀP')fx + gxx   (:52-54)
        {
            double tmp[n * n];
#pragma unroll
            for (int j = 0; j < n; ++j)
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < n; ++l) acc += fx[i * n + l] * P[j * n + l];
                    tmp[j * n + i] = acc;
                }
#pragma unroll
            for (int j = 0; j < n; ++j)
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < n; ++l) acc += tmp[l * n + i] * fx[j * n + l];
                    Qxx[j * n + i] = acc + gxx[j * n + i];
                }
        }
        // Quu = (fuᵀP')fu + guu (:57-59) ; Qux = (fuᵀP')fx + gux (:62-64)
        {
            double uh[m * n];
#pragma unroll
            for (int j = 0; j < n; ++j)
#pragma unroll
                for (int i = 0; i < m; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < n; ++l) acc += fu[i * n + l] * P[j * n + l];
                    uh[j * m + i] = acc;
                }
#pragma unroll
            for (int j = 0; j < m; ++j)
#pragma unroll
                for (int i = 0; i < m; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < n; ++l) acc += uh[l * m + i] * fu[j * n + l];
                    Quu[j * m + i] = acc + guu[j * m + i];
                }
#pragma unroll
            for (int j = 0; j < n; ++j)
#pragma unroll
                for (int i = 0; i < m; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int l = 0; l < n; ++l) acc += uh[l * m + i] * fx[j * n + l];
                    Qux[j * m + i] = acc + gux[j * m + i];
                }
        }
        // K = −Quu⁻¹Qux, k = −Quu⁻¹Qu via potrf/potrs ('U'), info ignored   (:68-75)
        double K[m * n], k[m];
        {
            double Uc[m * m];
#pragma unroll
            for (int i = 0; i < m * m; ++i) Uc[i] = Quu[i];
            const int info = potrf_U<m>(Uc);
            if (info != 0 && I.potrf_info == 0) I.potrf_info = info;
#pragma unroll
            for (int i = 0; i < m * n; ++i) K[i] = Qux[i];
#pragma unroll
            for (int i = 0; i < m; ++i) k[i] = Qu[i];
            potrs_U<m, n>(Uc, K);
            potrs_U<m, 1>(Uc, k);
#pragma unroll
            for (int i = 0; i < m * n; ++i) K[i] *= -1.0;
#pragma unroll
            for (int i = 0; i < m; ++i) k[i] *= -1.0;
        }
        // ux_tmp = Quu K   (:79)
        double uxt[m * n];
#pragma unroll
        for (int j = 0; j < n; ++j)
#pragma unroll
            for (int i = 0; i < m; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) acc += Quu[l * m + i] * K[j * m + l];
                uxt[j * m + i] = acc;
            }
        // P = Kᵀ ux_tmp + Kᵀ Qux + Quxᵀ K + Qxx   (:81-84)
#pragma unroll
        for (int j = 0; j < n; ++j)
#pragma unroll
            for (int i = 0; i < n; ++i) {
                double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int l = 0; l < m; ++l) {
                    a1 += K[i * m + l] * uxt[j * m + l];
                    a2 += K[i * m + l] * Qux[j * m + l];
                    a3 += Qux[i * m + l] * K[j * m + l];
                }
                P[j * n + i] = ((a1 + a2) + a3) + Qxx[j * n + i];
            }
        // p = ux_tmpᵀ k + Kᵀ Qu + Quxᵀ k + Qx   (:86-89)
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) {
                a1 += uxt[i * m + l] * k[l];
                a2 += K[i * m + l] * Qu[l];
                a3 += Qux[i * m + l] * k[l];
            }
            p[i] = ((a1 + a2) + a3) + Qx[i];
        }
        // lagrangian_gradient!: Lx = Qx − p[t], Lu = Qu   (src/solve.jl:73-81)
        double Lx[n];
#pragma unroll
        for (int i = 0; i < n; ++i) { Lx[i] = Qx[i] - p[i]; gnorm = nanmax(gnorm, fabs(Lx[i])); }
#pragma unroll
        for (int i = 0; i < m; ++i) gnorm = nanmax(gnorm, fabs(Qu[i]));
        if (I.lane == 0) {
#pragma unroll
            for (int i = 0; i < m * n; ++i) I.K[t * m * n + i] = K[i];
#pragma unroll
            for (int i = 0; i < m; ++i) I.k[t * m + i] = k[i];
#pragma unroll
            for (int i = 0; i < n; ++i) I.Lx[t * n + i] = Lx[i];
#pragma unroll
            for (int i = 0; i < m; ++i) I.Lu[t * m + i] = Qu[i];
            if (STORE_VALUE) {
#pragma unroll
                for (int i = 0; i < n * n; ++i) I.P[t * n * n + i] = P[i];
#pragma unroll
                for (int i = 0; i < n; ++i) I.p[t * n + i] = p[i];
            }
        }
    }
    I.gradient_norm = gnorm;
    __syncthreads();
}

// ------------------------------------------------------------- rollout!
// Closed-loop rollout u = αk + ū + Kx − Kx̄ in the reference's operation order
// (src/rollout.jl:24-28), wave-uniform; lane 0 writes the trial trajectory.
template <class M>
__device__ void rollout_bang(Inst<M>& I, double alpha) {
    constexpr int n = M::NX, m = M::NU;
    const double w[cdim<M::NW>::v] = {0.0};
    double xt[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xt[i] = I.xb[i];                      // (:19)
    if (I.lane == 0) {
#pragma unroll
        for (int i = 0; i < n; ++i) I.x[i] = xt[i];
    }
    // policy operands of step t are fetched from LDS one step ahead so that their
    // latency hides under the previous step's dynamics chain
    double Kn[m * n], kn[m], ubn[m], xbn[n];
#pragma unroll
    for (int i = 0; i < m * n; ++i) Kn[i] = I.K[i];
#pragma unroll
    for (int i = 0; i < m; ++i) { kn[i] = I.k[i]; ubn[i] = I.ub[i]; }
#pragma unroll
    for (int i = 0; i < n; ++i) xbn[i] = xt[i];
    for (int t = 0; t < I.N; ++t) {
        double Kt[m * n], kt[m], ubt[m], xbt[n];
#pragma unroll
        for (int i = 0; i < m * n; ++i) Kt[i] = Kn[i];
#pragma unroll
        for (int i = 0; i < m; ++i) { kt[i] = kn[i]; ubt[i] = ubn[i]; }
#pragma unroll
        for (int i = 0; i < n; ++i) xbt[i] = xbn[i];
        if (t + 1 < I.N) {
#pragma unroll
            for (int i = 0; i < m * n; ++i) Kn[i] = I.K[(t + 1) * m * n + i];
#pragma unroll
            for (int i = 0; i < m; ++i) { kn[i] = I.k[(t + 1) * m + i]; ubn[i] = I.ub[(t + 1) * m + i]; }
#pragma unroll
            for (int i = 0; i < n; ++i) xbn[i] = I.xb[(t + 1) * n + i];
        }
        double ut[m];
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double v = kt[i] * alpha;                                 // (:24-25)
            v += ubt[i];                                              // (:26)
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                a1 += Kt[j * m + i] * xt[j];
                a2 += Kt[j * m + i] * xbt[j];
            }
            v += a1;                                                  // (:27)
            v += -1.0 * a2;                                           // (:28)
            ut[i] = v;
        }
        double y[n];
        M::dyn_wave(I.lane, xt, ut, w, y);                            // (:29)
        if (I.lane == 0) {
#pragma unroll
            for (int i = 0; i < m; ++i) I.u[t * m + i] = ut[i];
#pragma unroll
            for (int i = 0; i < n; ++i) I.x[(t + 1) * n + i] = y[i];
        }
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = y[i];
    }
    I.rollouts += 1;
    I.states_eq_nominal = 0;
    __syncthreads();
}

// --------------------------------------------------------- forward_pass!
template <class M>
__device__ void forward_pass(Inst<M>& I, const ilqr_options& opt, bool constrained) {
    constexpr int n = M::NX, m = M::NU;
    const double c1 = 1.0e-4;
    const int max_iterations = 25;
    I.status = 0;                                                     // (:10)
    const double J_prev = I.objective;                                // (:13)
    // lagrangian_gradient! (:16) was produced by the backward pass (Lx, Lu in LDS).
    // trajectory_sensitivities (src/data/methods.jl:42-54) fused with the
    // product gradientᵀ·Δz (:20).
    double delta = 0.0;
    if (opt.line_search == 1) {
        double zx[n];
#pragma unroll
        for (int i = 0; i < n; ++i) zx[i] = 0.0;
        for (int t = 0; t < I.N; ++t) {
            double zu[m], zy[n];
#pragma unroll
            for (int i = 0; i < m; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < n; ++j) acc += I.K[t * m * n + j * m + i] * zx[j];
                zu[i] = I.k[t * m + i] + acc;
            }
#pragma unroll
            for (int i = 0; i < n; ++i) delta += I.Lx[t * n + i] * zx[i];
#pragma unroll
            for (int i = 0; i < m; ++i) delta += I.Lu[t * m + i] * zu[i];
#pragma unroll
            for (int i = 0; i < n; ++i) {
                double a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int j = 0; j < m; ++j) a1 += I.fu[t * n * m + j * n + i] * zu[j];
#pragma unroll
                for (int j = 0; j < n; ++j) a2 += I.fx[t * n * n + j * n + i] * zx[j];
                zy[i] = a1 + a2;
            }
#pragma unroll
            for (int i = 0; i < n; ++i) zx[i] = zy[i];
        }
    }
    I.step_size = 1.0;                                                // (:26)
    int iteration = 1;
    while (I.step_size >= opt.min_step_size) {                        // (:28)
        if (iteration > max_iterations) break;                        // (:29)
        rollout_bang<M>(I, I.step_size);                              // (:34)
        cost_bang<M>(I, true, constrained);                           // (:36)
        const double J = I.objective;
        if (J <= J_prev + c1 * I.step_size * delta) {                 // (:44) NaN ⇒ reject
            // update_nominal_trajectory! (src/data/methods.jl:32-39)
            for (int i = I.lane; i < I.T * n; i += 64) I.xb[i] = I.x[i];
            for (int i = I.lane; i < I.N * m; i += 64) I.ub[i] = I.u[i];
            I.states_eq_nominal = 1;
            I.status = 1;
            __syncthreads();
            break;
        } else {
            I.step_size *= 0.5;                                       // (:51)
            iteration += 1;
        }
    }
}

// reset!(problem.model); reset!(problem.objective) — src/solve.jl:9-10
template <class M>
__device__ void reset_model_objective(Inst<M>& I) {
    constexpr int n = M::NX, m = M::NU;
    for (int i = I.lane; i < I.N * n * n; i += 64) I.fx[i] = 0.0;
    for (int i = I.lane; i < I.N * n * m; i += 64) I.fu[i] = 0.0;
    for (int i = I.lane; i < I.T * n; i += 64) I.gx[i] = 0.0;
    for (int i = I.lane; i < I.N * m; i += 64) I.gu[i] = 0.0;
    for (int i = I.lane; i < I.T * n * n; i += 64) I.gxx[i] = 0.0;
    for (int i = I.lane; i < I.N * m * m; i += 64) I.guu[i] = 0.0;
    for (int i = I.lane; i < I.N * m * n; i += 64) I.gux[i] = 0.0;
    __syncthreads();
}

// ilqr_solve! — src/solve.jl:1-54
template <class M, bool STORE_VALUE>
__device__ void ilqr_solve(Inst<M>& I, const ilqr_options& opt, bool constrained) {
    reset_model_objective<M>(I);                                      // (:9-10)
    if (opt.reset_cache) {                                            // (:12) reset!(data)
        I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0;
    }
    cost_bang<M>(I, false, constrained);                              // (:14)
    gradients<M>(I, constrained);                                     // (:16)
    backward_pass<M, STORE_VALUE>(I);                                 // (:18)
    double obj_prev = I.objective;                                    // (:21)
    for (int i = 1; i <= opt.max_iterations; ++i) {                   // (:22)
        forward_pass<M>(I, opt, constrained);                         // (:23)
        if (opt.line_search != 0) {                                   // (:27-33)
            gradients<M>(I, constrained);
            backward_pass<M, STORE_VALUE>(I);
        }
        I.iterations += 1;                                            // (:39)
        if (I.gradient_norm < opt.lagrangian_gradient_tolerance) break;          // (:48)
        if (fabs(I.objective - obj_prev) < opt.objective_tolerance) break;       // (:49)
        obj_prev = I.objective;
        if (!I.status) break;                                         // (:50)
    }
}

// augmented_lagrangian_update! — src/augmented_lagrangian.jl:87-110
template <class M>
__device__ void al_update(Inst<M>& I, const ilqr_options& opt) {
    constexpr int ncs = M::NCS;
    for (int i = I.lane; i < I.C; i += 64) {
        const int ns = I.N * ncs;
        bool ineq;
        if (i < ns) ineq = ncs > 0 ? ((M::INEQ_S >> (i % (ncs > 0 ? ncs : 1))) & 1ull) : false;
        else ineq = (M::INEQ_T >> (i - ns)) & 1ull;
        double lam = I.lam[i] + I.rho[i] * I.c[i];
        if (ineq) lam = nanmax(0.0, lam);
        I.lam[i] = lam;
        const double r = opt.scaling_penalty * I.rho[i];
        I.rho[i] = r < opt.max_penalty ? r : opt.max_penalty;
    }
    __syncthreads();
}

// constrained_ilqr_solve! — src/solve.jl:88-129
template <class M>
__device__ void constrained_ilqr_solve(Inst<M>& I, const ilqr_options& opt) {
    // reset!(solver.data) (:93, src/data/solver.jl:49-59)
    I.objective = 0.0; I.max_violation = 0.0; I.status = 0; I.iterations = 0; I.gradient_norm = 0.0;
    for (int i = I.lane; i < I.N * M::NX; i += 64) I.Lx[i] = 0.0;
    for (int i = I.lane; i < I.N * M::NU; i += 64) I.Lu[i] = 0.0;
    for (int i = I.lane; i < I.C; i += 64) {                          // (:96-103)
        I.lam[i] = 0.0;
        I.rho[i] = opt.initial_constraint_penalty;
    }
    __syncthreads();
    I.outer_iterations = 0;
    for (int i = 1; i <= opt.max_dual_updates; ++i) {                 // (:105)
        I.outer_iterations = i;
        ilqr_solve<M, false>(I, opt, true);                           // (:109)
        cost_bang<M>(I, false, true);                                 // (:113)
        if (I.max_violation <= opt.constraint_tolerance) break;       // (:117)
        al_update<M>(I, opt);                                         // (:120-122)
    }
}

// ------------------------------------------------------------ kernel glue
template <class M>
__device__ __forceinline__ void inst_setup(Inst<M>& I, const KArgs& a, double* smem, int b) {
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    I.xb = smem + L.xb; I.ub = smem + L.ub; I.x = smem + L.x; I.u = smem + L.u;
    I.fx = smem + L.fx; I.fu = smem + L.fu; I.gx = smem + L.gx; I.gu = smem + L.gu;
    I.K = smem + L.K; I.k = smem + L.k; I.Lx = smem + L.Lx; I.Lu = smem + L.Lu;
    I.c = smem + L.c; I.lam = smem + L.lam; I.rho = smem + L.rho; I.act = smem + L.act;
    I.gxx = g + L.gxx; I.guu = g + L.guu; I.gux = g + L.gux; I.P = g + L.P; I.p = g + L.p; I.scal = g + L.scal;
    I.T = L.T; I.N = L.T - 1; I.C = L.C; I.lane = threadIdx.x;
    // LDS-resident set: one coalesced 16-B-per-lane stream from HBM
    const double2* src = reinterpret_cast<const double2*>(g);
    double2* dst = reinterpret_cast<double2*>(smem);
    for (int i = I.lane; i < L.lds_doubles / 2; i += 64) dst[i] = src[i];
    I.objective = I.scal[S_OBJECTIVE]; I.max_violation = I.scal[S_MAX_VIOLATION];
    I.step_size = I.scal[S_STEP_SIZE]; I.gradient_norm = I.scal[S_GRADIENT_NORM];
    I.status = (int)I.scal[S_STATUS]; I.iterations = (int)I.scal[S_ITERATIONS];
    I.outer_iterations = (int)I.scal[S_OUTER_ITERATIONS]; I.potrf_info = (int)I.scal[S_POTRF_INFO];
    I.rollouts = (int)I.scal[S_ROLLOUTS]; I.states_eq_nominal = (int)I.scal[S_STATES_EQ_NOMINAL];
    __syncthreads();
}

template <class M>
__device__ __forceinline__ void inst_writeback(Inst<M>& I, const KArgs& a, double* smem, int b) {
    __syncthreads();
    const Layout& L = a.L;
    double* g = a.ws + (size_t)b * (size_t)L.stride;
    double2* dst = reinterpret_cast<double2*>(g);
    const double2* src = reinterpret_cast<const double2*>(smem);
    for (int i = I.lane; i < L.lds_doubles / 2; i += 64) dst[i] = src[i];
    if (I.lane == 0) {
        I.scal[S_OBJECTIVE] = I.objective; I.scal[S_MAX_VIOLATION] = I.max_violation;
        I.scal[S_STEP_SIZE] = I.step_size; I.scal[S_GRADIENT_NORM] = I.gradient_norm;
        I.scal[S_STATUS] = (double)I.status; I.scal[S_ITERATIONS] = (double)I.iterations;
        I.scal[S_OUTER_ITERATIONS] = (double)I.outer_iterations; I.scal[S_POTRF_INFO] = (double)I.potrf_info;
        I.scal[S_ROLLOUTS] = (double)I.rollouts; I.scal[S_STATES_EQ_NOMINAL] = (double)I.states_eq_nominal;
    }
}

// solve!(solver) for every instance — src/solve.jl:137-143
template <class M>
__global__ __launch_bounds__(64) void solve_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    Inst<M> I;
    inst_setup<M>(I, a, smem, b);
    I.potrf_info = 0; I.rollouts = 0;
    if (a.constrained) constrained_ilqr_solve<M>(I, a.opt);
    else { I.outer_iterations = 0; ilqr_solve<M, false>(I, a.opt, false); }
    inst_writeback<M>(I, a, smem, b);
}

// single stages for parity tests (STORE_VALUE: P, p are written to HBM)
template <class M>
__global__ __launch_bounds__(64) void stage_kernel(KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    Inst<M> I;
    inst_setup<M>(I, a, smem, b);
    const bool con = a.constrained != 0;
    switch (a.stage) {
        case ILQR_STAGE_COST_NOMINAL: cost_bang<M>(I, false, con); break;
        case ILQR_STAGE_GRADIENTS: gradients<M>(I, con); break;
        case ILQR_STAGE_BACKWARD_PASS: backward_pass<M, true>(I); break;
        case ILQR_STAGE_FORWARD_PASS: forward_pass<M>(I, a.opt, con); break;
        case ILQR_STAGE_RESET_MODEL_OBJECTIVE: reset_model_objective<M>(I); break;
        case ILQR_STAGE_ILQR_SOLVE: ilqr_solve<M, true>(I, a.opt, con); break;
        case ILQR_STAGE_AL_UPDATE: al_update<M>(I, a.opt); break;
        default: break;
    }
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
    const double w[cdim<M::NW>::v] = {0.0};
    double xt[n];
#pragma unroll
    for (int i = 0; i < n; ++i) { xt[i] = a.x1[(size_t)b * n + i]; g[L.xb + i] = xt[i]; }
    for (int t = 0; t < N; ++t) {
        double ut[m], y[n];
#pragma unroll
        for (int i = 0; i < m; ++i) { ut[i] = a.u_in[((size_t)b * N + t) * m + i]; g[L.ub + t * m + i] = ut[i]; }
        M::dyn(xt, ut, w, y);
#pragma unroll
        for (int i = 0; i < n; ++i) { xt[i] = y[i]; g[L.xb + (t + 1) * n + i] = y[i]; }
    }
    g[L.scal + S_STATES_EQ_NOMINAL] = 0.0;
}

}  // namespace ilqr

// Model module interface: what a compiled model (built-in or generated by
// iterativelqr.jl_amd/codegen.py) registers with the library.
extern "C" struct ilqr_model_vtable {
    const char* name;
    int nx, nu, nw, ncs, nct;
    unsigned long long ineq_s, ineq_t;
    int (*launch_solve)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);
    int (*launch_stage)(const ilqr::KArgs* a, size_t lds_bytes, void* stream);
    int (*launch_init)(const ilqr::KArgs* a, void* stream);
};

namespace ilqr {
template <class M>
struct ModelModule {
    static int launch_solve(const KArgs* a, size_t lds, void* stream) {
        static size_t configured = 0;
        if (lds > configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<M>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            configured = lds;
        }
        hipLaunchKernelGGL(solve_kernel<M>, dim3(a->B), dim3(64), lds, (hipStream_t)stream, *a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static int launch_stage(const KArgs* a, size_t lds, void* stream) {
        static size_t configured = 0;
        if (lds > configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<M>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            configured = lds;
        }
        hipLaunchKernelGGL(stage_kernel<M>, dim3(a->B), dim3(64), lds, (hipStream_t)stream, *a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static int launch_init(const KArgs* a, void* stream) {
        hipLaunchKernelGGL(init_rollout_kernel<M>, dim3((a->B + 63) / 64), dim3(64), 0, (hipStream_t)stream, *a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    static const ilqr_model_vtable* vtable() {
        static const ilqr_model_vtable vt = {M::NAME, M::NX, M::NU, M::NW, M::NCS, M::NCT, M::INEQ_S, M::INEQ_T,
                                             &launch_solve, &launch_stage, &launch_init};
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
