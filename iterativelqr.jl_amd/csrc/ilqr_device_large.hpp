// Large-model path (nx > 4 or nu > 4, up to nx = 32, nu = 16), e.g. BASELINE config "synth32".
//
// Differences from the LDS-resident small-model path of ilqr_device.hpp:
//   * the per-instance workspace (2.4 MB for nx=32, nu=8, T=101) stays in HBM; the wave streams the
//     per-timestep matrices through LDS staging buffers;
//   * linearisation writes Jacobians straight to memory (M::dyn_jac_mem), accumulates only the
//     structurally non-zero Hessian entries (M::cost_*_hess_acc) and uses the symbolically derived
//     Gauss-Newton AL terms (M::al_s / M::al_t);
//   * the Riccati step's contractions fx^T P' fx, fu^T P' fx, ... are real matrix products here and
//     run as 16x16x4 fp64 MFMA tiles (v_mfma_f64_16x16x4_f64) out of LDS operands
//     (A[i][k]: i = lane&15, k = lane>>4; B[k][j]: k = lane>>4, j = lane&15;
//      C/D[i][j]: j = lane&15, i = (lane>>4) + 4*reg  — cdna_hip_programming.md §3);
//   * Cholesky of Quu runs on wave-uniform registers, the triangular solves one column per lane.
// Same reference semantics and citations as ilqr_device.hpp.
#pragma once

namespace ilqr {

template <class M>
struct LargeDims {
    static constexpr int n = M::NX, m = M::NU;
    static constexpr int ld = n | 1;      // odd leading dimensions: conflict-free LDS column walks
    static constexpr int ldm = m | 1;
    // LDS carve (doubles)
    static constexpr int oP = 0, oFx = oP + n * ld, oT = oFx + n * ld, oQxx = oT + n * ld,
                         oFu = oQxx + n * ld, oUh = oFu + m * ld, oQux = oUh + n * ldm, oK = oQux + n * ldm,
                         oUxt = oK + n * ldm, oQuu = oUxt + n * ldm, oVec = oQuu + m * ldm,
                         total = oVec + 4 * n + 4 * m + 8;
};

typedef double double4_t __attribute__((ext_vector_type(4)));

// D (R x Cc, column-major, leading dimension ldd, in LDS) = [D +] op(A) * op(B), inner dimension Kd.
// A(i,k) lives at A[a_i * i + a_k * k], B(k,j) at B[b_k * k + b_j * j] (any transposition for free).
// Everything outside the real extents reads as zero, so dimensions need not be multiples of the tile.
__device__ __forceinline__ void tile_gemm(double* D, int ldd, int R, int Cc, const double* A, int a_i, int a_k,
                                          const double* B, int b_k, int b_j, int Kd, bool accumulate, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    for (int I0 = 0; I0 < R; I0 += 16)
        for (int J0 = 0; J0 < Cc; J0 += 16) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
            const int col = J0 + li;
            if (accumulate) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = I0 + lk + 4 * r;
                    acc[r] = (row < R && col < Cc) ? D[col * ldd + row] : 0.0;
                }
            }
            for (int s = 0; s < Kd; s += 4) {
                const int k = s + lk;
                const double a = (I0 + li < R && k < Kd) ? A[a_i * (I0 + li) + a_k * k] : 0.0;
                const double b = (col < Cc && k < Kd) ? B[b_k * k + b_j * col] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I0 + lk + 4 * r;
                if (row < R && col < Cc) D[col * ldd + row] = acc[r];
            }
        }
}

// LDS matrix (R x Cc, ldd) += / = global matrix (packed column-major R x Cc)
__device__ __forceinline__ void lds_add_global(double* D, int ldd, int R, int Cc, const double* G, int lane) {
    for (int e = lane; e < R * Cc; e += 64) { const int i = e % R, j = e / R; D[j * ldd + i] += G[e]; }
}
__device__ __forceinline__ void lds_load_global(double* D, int ldd, int R, int Cc, const double* G, int lane) {
    for (int e = lane; e < R * Cc; e += 64) { const int i = e % R, j = e / R; D[j * ldd + i] = G[e]; }
}

// ---------------------------------------------------------------- gradients! (one timestep per lane)
template <class M>
__device__ void gradients_large(Inst<M>& I, bool constrained) {
    constexpr int n = M::NX, m = M::NU, ncs = M::NCS, nct = M::NCT;
    ILQR_PROF_BEGIN();
    for (int t = I.lane; t < I.T; t += 64) {
        double w[cdim<M::NW>::v];
        load_w<M::NW>(I.w, t, w);
        double xt[n];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = I.xb[t * n + i];
        if (t < I.N) {
            double ut[m];
#pragma unroll
            for (int i = 0; i < m; ++i) ut[i] = I.ub[t * m + i];
            M::dyn_jac_mem(xt, ut, w, I.fx + (size_t)t * n * n, I.fu + (size_t)t * n * m);     // `.=`  (src/dynamics.jl:45-46)
            double gx[n], gu[m];
            M::cost_s_grad(xt, ut, w, gx, gu);                                                // `.=`  (src/costs.jl:61,65)
            M::cost_s_hess_acc(xt, ut, w, I.gxx + (size_t)t * n * n, I.guu + (size_t)t * m * m,
                               I.gux + (size_t)t * m * n);                                    // `.+=` (src/costs.jl:74-80)
            if constexpr (ncs > 0) {
                if (constrained) {                                                            // src/gradients.jl:54-80
                    double ct[ncs], ir[ncs];
                    const int off = t * ncs;
#pragma unroll
                    for (int i = 0; i < ncs; ++i) {
                        ir[i] = I.rho[off + i] * I.act[off + i];
                        ct[i] = I.lam[off + i] + ir[i] * I.c[off + i];
                    }
                    M::al_s(xt, ut, w, ct, ir, gx, gu, I.gxx + (size_t)t * n * n, I.guu + (size_t)t * m * m,
                            I.gux + (size_t)t * m * n);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) I.gx[t * n + i] = gx[i];
#pragma unroll
            for (int i = 0; i < m; ++i) I.gu[t * m + i] = gu[i];
        } else {
            double gx[n];
            M::cost_t_grad(xt, w, gx);
            M::cost_t_hess_acc(xt, w, I.gxx + (size_t)t * n * n);
            if constexpr (nct > 0) {
                if (constrained) {
                    double ct[nct], ir[nct];
                    const int off = I.N * ncs;
#pragma unroll
                    for (int i = 0; i < nct; ++i) {
                        ir[i] = I.rho[off + i] * I.act[off + i];
                        ct[i] = I.lam[off + i] + ir[i] * I.c[off + i];
                    }
                    M::al_t(xt, w, ct, ir, gx, I.gxx + (size_t)t * n * n);
                }
            }
#pragma unroll
            for (int i = 0; i < n; ++i) I.gx[t * n + i] = gx[i];
        }
    }
    __syncthreads();
    ILQR_PROF_END(I, PROF_GRAD);
}

// ---------------------------------------------------------------- backward_pass! (MFMA 16x16x4 tiles)
template <class M, bool STORE_VALUE>
__device__ void backward_pass_large(Inst<M>& I) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU, ld = LD::ld, ldm = LD::ldm;
    static_assert(n <= 32 && m <= 16, "large path: nx <= 32, nu <= 16");
    const int lane = I.lane, N = I.N;
    double* S = I.lds;
    double *sP = S + LD::oP, *sFx = S + LD::oFx, *sT = S + LD::oT, *sQxx = S + LD::oQxx, *sFu = S + LD::oFu,
           *sUh = S + LD::oUh, *sQux = S + LD::oQux, *sK = S + LD::oK, *sUxt = S + LD::oUxt, *sQuu = S + LD::oQuu;
    double *sp = S + LD::oVec, *sQx = sp + n, *sQu = sQx + n, *sk = sQu + m;
    lds_load_global(sP, ld, n, n, I.gxx + (size_t)N * n * n, lane);       // P[H] .= gxx[H]  (:39)
    for (int i = lane; i < n; i += 64) sp[i] = I.gx[N * n + i];           // p[H] .= gx[H]   (:40)
    if (STORE_VALUE) {
        for (int e = lane; e < n * n; e += 64) I.P[(size_t)N * n * n + e] = I.gxx[(size_t)N * n * n + e];
        for (int i = lane; i < n; i += 64) I.p[N * n + i] = I.gx[N * n + i];
    }
    double gmax = 0.0;
    __syncthreads();
    for (int t = N - 1; t >= 0; --t) {                                    // (:42)
        lds_load_global(sFx, ld, n, n, I.fx + (size_t)t * n * n, lane);
        lds_load_global(sFu, ld, n, m, I.fu + (size_t)t * n * m, lane);
        __syncthreads();
        // Qx = fx^T p' + gx, Qu = fu^T p' + gu   (:44-49): one output per lane
        for (int i = lane; i < n + m; i += 64) {
            const double* colp = i < n ? sFx + i * ld : sFu + (i - n) * ld;
            double acc = 0.0;
            for (int l = 0; l < n; ++l) acc += colp[l] * sp[l];
            if (i < n) sQx[i] = acc + I.gx[t * n + i];
            else sQu[i - n] = acc + I.gu[t * m + (i - n)];
        }
        // Qxx = (fx^T P') fx + gxx   (:52-54)
        tile_gemm(sT, ld, n, n, sFx, ld, 1, sP, 1, ld, n, false, lane);
        // ux_hat = fu^T P'   (:57, :62)
        tile_gemm(sUh, ldm, m, n, sFu, ld, 1, sP, 1, ld, n, false, lane);
        __syncthreads();
        tile_gemm(sQxx, ld, n, n, sT, 1, ld, sFx, 1, ld, n, false, lane);
        tile_gemm(sQuu, ldm, m, m, sUh, 1, ldm, sFu, 1, ld, n, false, lane);   // Quu = ux_hat fu + guu (:58-59)
        tile_gemm(sQux, ldm, m, n, sUh, 1, ldm, sFx, 1, ld, n, false, lane);   // Qux = ux_hat fx + gux (:63-64)
        __syncthreads();
        lds_add_global(sQxx, ld, n, n, I.gxx + (size_t)t * n * n, lane);
        lds_add_global(sQuu, ldm, m, m, I.guu + (size_t)t * m * m, lane);
        lds_add_global(sQux, ldm, m, n, I.gux + (size_t)t * m * n, lane);
        __syncthreads();
        // potrf('U') on wave-uniform registers (info ignored, :68-69)
        double Uc[m * m];
#pragma unroll
        for (int j = 0; j < m; ++j)
#pragma unroll
            for (int i = 0; i < m; ++i) Uc[j * m + i] = (i <= j) ? sQuu[j * ldm + i] : 0.0;
        const int info = potrf_U<m>(Uc);
        if (info != 0 && I.potrf_info == 0) I.potrf_info = info;
        // potrs('U'): column j of Qux per lane, Qu on lane n   (:70-75)
        for (int j = lane; j <= n; j += 64) {
            double b[m];
#pragma unroll
            for (int i = 0; i < m; ++i) b[i] = j < n ? sQux[j * ldm + i] : sQu[i];
            potrs_U<m, 1>(Uc, b);
#pragma unroll
            for (int i = 0; i < m; ++i) {
                const double v = b[i] * -1.0;
                if (j < n) { sK[j * ldm + i] = v; I.K[(size_t)t * m * n + j * m + i] = v; }
                else { sk[i] = v; I.k[t * m + i] = v; }
            }
        }
        __syncthreads();
        // ux_tmp = Quu K   (:79)
        tile_gemm(sUxt, ldm, m, n, sQuu, 1, ldm, sK, 1, ldm, m, false, lane);
        __syncthreads();
        // P = K^T ux_tmp + K^T Qux + Qux^T K + Qxx   (:81-84), accumulated in this order into sT
        tile_gemm(sT, ld, n, n, sK, ldm, 1, sUxt, 1, ldm, m, false, lane);
        __syncthreads();
        tile_gemm(sT, ld, n, n, sK, ldm, 1, sQux, 1, ldm, m, true, lane);
        __syncthreads();
        tile_gemm(sT, ld, n, n, sQux, ldm, 1, sK, 1, ldm, m, true, lane);
        __syncthreads();
        for (int e = lane; e < n * n; e += 64) {
            const int i = e % n, j = e / n;
            const double v = sT[j * ld + i] + sQxx[j * ld + i];
            sP[j * ld + i] = v;
            if (STORE_VALUE) I.P[(size_t)t * n * n + e] = v;
        }
        // p = ux_tmp^T k + K^T Qu + Qux^T k + Qx   (:86-89); Lagrangian gradient (src/solve.jl:73-81)
        for (int i = lane; i < n; i += 64) {
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int l = 0; l < m; ++l) {
                a1 += sUxt[i * ldm + l] * sk[l];
                a2 += sK[i * ldm + l] * sQu[l];
                a3 += sQux[i * ldm + l] * sk[l];
            }
            const double pn = ((a1 + a2) + a3) + sQx[i];
            const double Lx = sQx[i] - pn;
            gmax = nanmax(gmax, fabs(Lx));
            I.Lx[t * n + i] = Lx;
            if (STORE_VALUE) I.p[t * n + i] = pn;
            sQx[i] = pn;                      // becomes p' of the next step (copied below)
        }
        for (int i = lane; i < m; i += 64) {
            gmax = nanmax(gmax, fabs(sQu[i]));
            I.Lu[t * m + i] = sQu[i];
        }
        __syncthreads();
        for (int i = lane; i < n; i += 64) sp[i] = sQx[i];
        __syncthreads();
    }
    I.gradient_norm = wave_max(gmax);
    __syncthreads();
}

// ---------------------------------------------------------------- rollout! (wave-uniform state, lane-parallel policy)
template <class M>
__device__ void rollout_large(Inst<M>& I, double alpha) {
    constexpr int n = M::NX, m = M::NU;
    ILQR_PROF_BEGIN();
    double xt[n];
#pragma unroll
    for (int i = 0; i < n; ++i) xt[i] = I.xb[i];                          // (:19)
    if (I.lane == 0) {
#pragma unroll
        for (int i = 0; i < n; ++i) I.x[i] = xt[i];
    }
    const int li = I.lane < m ? I.lane : m - 1;                           // lane i evaluates control component i
    for (int t = 0; t < I.N; ++t) {
        double v = I.k[t * m + li] * alpha;                               // (:24-25)
        v += I.ub[t * m + li];                                            // (:26)
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int j = 0; j < n; ++j) {
            const double Kij = I.K[(size_t)t * m * n + j * m + li];
            a1 += Kij * xt[j];
            a2 += Kij * I.xb[t * n + j];
        }
        v += a1;                                                          // (:27)
        v += -1.0 * a2;                                                   // (:28)
        double ut[m];
        bcast_array<m>(v, ut);
        double w[cdim<M::NW>::v], y[n];
        load_w<M::NW>(I.w, t, w);
        M::dyn_wave(I.lane, xt, ut, w, y);                                // (:29)
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
    ILQR_PROF_END(I, PROF_ROLLOUT);
}

// ---------------------------------------------------------------- trajectory_sensitivities + gradient^T dz
template <class M>
__device__ double delta_large(Inst<M>& I) {
    typedef LargeDims<M> LD;
    constexpr int n = M::NX, m = M::NU;
    ILQR_PROF_BEGIN();
    double* zx = I.lds + LD::oVec;          // n
    double* zu = zx + n;                    // m
    const int lane = I.lane;
    for (int i = lane; i < n; i += 64) zx[i] = 0.0;
    __syncthreads();
    double dpart = 0.0;
    for (int t = 0; t < I.N; ++t) {
        for (int i = lane; i < m; i += 64) {                              // Δu = k + K Δx
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += I.K[(size_t)t * m * n + j * m + i] * zx[j];
            const double v = I.k[t * m + i] + acc;
            zu[i] = v;
            dpart += I.Lu[t * m + i] * v;
        }
        for (int i = lane; i < n; i += 64) dpart += I.Lx[t * n + i] * zx[i];
        __syncthreads();
        double zy = 0.0;
        if (lane < n) {                                                   // Δx⁺ = fu Δu + fx Δx
            double a1 = 0.0, a2 = 0.0;
            for (int j = 0; j < m; ++j) a1 += I.fu[(size_t)t * n * m + j * n + lane] * zu[j];
            for (int j = 0; j < n; ++j) a2 += I.fx[(size_t)t * n * n + j * n + lane] * zx[j];
            zy = a1 + a2;
        }
        __syncthreads();
        if (lane < n) zx[lane] = zy;
        __syncthreads();
    }
    const double d = wave_sum(dpart);
    ILQR_PROF_END(I, PROF_DELTA);
    return d;
}

}  // namespace ilqr
