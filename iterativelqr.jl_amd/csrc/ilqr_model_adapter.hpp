// Adapter from the reference's CALLABLE CONTRACT to the model interface of the kernels — the seam behind
// ilqr_compile_model (include/ilqr_hip.h) for callers that have no Python / sympy (a Julia or C host).
//
// The reference takes user functions in-place, `(out, x, u, w) -> nothing`, `out` column-major and pre-zeroed by the caller
// (src/dynamics.jl:55-60, src/constraints.jl:54-64, src/costs.jl:1-15). ilqr_compile_model wraps C source that defines
//     dynamics, dynamics_jacobian_state, dynamics_jacobian_action,
//     cost_stage, cost_stage_gradient_state, cost_stage_gradient_action,
//     cost_stage_hessian_state_state, cost_stage_hessian_action_action, cost_stage_hessian_action_state,
//     cost_terminal, cost_terminal_gradient_state, cost_terminal_hessian_state_state,
//     constraint_stage, constraint_stage_jacobian_state, constraint_stage_jacobian_action      (when nc_stage > 0)
//     constraint_terminal, constraint_terminal_jacobian_state                                   (when nc_term > 0)
// — each `ILQR_MODEL_FN void name(double* out, const double* x, const double* u, const double* w)` — into a struct F of
// static forwarders and instantiates AdaptedModel<F, ...>, which the solve / stage / packed kernels take like a generated
// model. No wave-cooperative rollout form exists for such code: every lane evaluates the dynamics (correct, not the fastest),
// the Gauss-Newton AL terms are the dense products of src/gradients.jl:54-80.
// Models with nx > 4 or nu > 4 (up to nx = 64, nu = 16) get AdaptedLargeModel: the compact forms the large path streams
// (ilqr_device_large.hpp), with every Jacobian entry treated as state-dependent and every Hessian entry as structurally
// non-zero — black-box callables carry no structure; the symbolic generator (codegen.py) is the way to constant / sparse tables.
#pragma once

#include "ilqr_device.hpp"

#define ILQR_MODEL_FN __device__ __forceinline__ static

namespace ilqr {

template <class F, int NX_, int NU_, int NW_, int NCS_, int NCT_, unsigned long long INEQ_S_, unsigned long long INEQ_T_>
struct AdaptedModel {
    static constexpr int NX = NX_, NU = NU_, NW = NW_, NCS = NCS_, NCT = NCT_;
    static constexpr unsigned long long INEQ_S = INEQ_S_, INEQ_T = INEQ_T_;
    static_assert(NX >= 1 && NX <= 64 && NU >= 1 && NU <= 16, "ilqr_compile_model: nx <= 64, nu <= 16");
    static_assert(NCS <= ILQR_MAX_CONSTRAINT_ROWS && NCT <= ILQR_MAX_CONSTRAINT_ROWS, "at most 256 constraint rows per stage");
    static constexpr int W = cdim<NW>::v, CS = cdim<NCS>::v, CT = cdim<NCT>::v;

    template <int N> __device__ __forceinline__ static void zero(double (&a)[N]) {
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] = 0.0;
    }

    // ---- Dynamics (src/dynamics.jl:36-50)
    __device__ __forceinline__ static void dyn(const double (&x)[NX], const double (&u)[NU], const double (&w)[W], double (&y)[NX]) {
        zero(y);
        F::dynamics(y, x, u, w);
    }
    struct WaveCtx {};
    template <bool PIN_CONSTANTS = false> __device__ __forceinline__ static WaveCtx wave_ctx(const int) { return WaveCtx{}; }
    template <class BC = RowBC>
    __device__ __forceinline__ static void dyn_wave(const WaveCtx&, const int, const double (&x)[NX], const double (&u)[NU], const double (&w)[W], double (&y)[NX]) {
        dyn(x, u, w, y);
    }
    __device__ __forceinline__ static void dyn_jac(const double (&x)[NX], const double (&u)[NU], const double (&w)[W],
                                                   double (&fx)[NX * NX], double (&fu)[NX * NU]) {
        zero(fx); zero(fu);
        F::dynamics_jacobian_state(fx, x, u, w);
        F::dynamics_jacobian_action(fu, x, u, w);
    }
    __device__ __forceinline__ static void dyn_jac_mem(const double (&x)[NX], const double (&u)[NU], const double (&w)[W],
                                                       double* __restrict__ fx, double* __restrict__ fu) {
        double a[NX * NX], b[NX * NU];
        dyn_jac(x, u, w, a, b);
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) fx[i] = a[i];
#pragma unroll
        for (int i = 0; i < NX * NU; ++i) fu[i] = b[i];
    }
    // ---- Cost (src/costs.jl:48-84)
    __device__ __forceinline__ static double cost_s(const double (&x)[NX], const double (&u)[NU], const double (&w)[W]) {
        double l[1] = {0.0};
        F::cost_stage(l, x, u, w);
        return l[0];
    }
    __device__ __forceinline__ static void cost_s_grad(const double (&x)[NX], const double (&u)[NU], const double (&w)[W],
                                                       double (&gx)[NX], double (&gu)[NU]) {
        zero(gx); zero(gu);
        F::cost_stage_gradient_state(gx, x, u, w);
        F::cost_stage_gradient_action(gu, x, u, w);
    }
    __device__ __forceinline__ static void cost_s_hess(const double (&x)[NX], const double (&u)[NU], const double (&w)[W],
                                                       double (&hxx)[NX * NX], double (&huu)[NU * NU], double (&hux)[NU * NX]) {
        zero(hxx); zero(huu); zero(hux);
        F::cost_stage_hessian_state_state(hxx, x, u, w);
        F::cost_stage_hessian_action_action(huu, x, u, w);
        F::cost_stage_hessian_action_state(hux, x, u, w);
    }
    __device__ __forceinline__ static void cost_s_hess_acc(const double (&x)[NX], const double (&u)[NU], const double (&w)[W],
                                                           double* __restrict__ gxx, double* __restrict__ guu, double* __restrict__ gux) {
        double hxx[NX * NX], huu[NU * NU], hux[NU * NX];
        cost_s_hess(x, u, w, hxx, huu, hux);
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) gxx[i] += hxx[i];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) guu[i] += huu[i];
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) gux[i] += hux[i];
    }
    __device__ __forceinline__ static double cost_t(const double (&x)[NX], const double (&w)[W]) {
        double l[1] = {0.0};
        const double u0[1] = {0.0};
        F::cost_terminal(l, x, u0, w);
        return l[0];
    }
    __device__ __forceinline__ static void cost_t_grad(const double (&x)[NX], const double (&w)[W], double (&gx)[NX]) {
        const double u0[1] = {0.0};
        zero(gx);
        F::cost_terminal_gradient_state(gx, x, u0, w);
    }
    __device__ __forceinline__ static void cost_t_hess(const double (&x)[NX], const double (&w)[W], double (&hxx)[NX * NX]) {
        const double u0[1] = {0.0};
        zero(hxx);
        F::cost_terminal_hessian_state_state(hxx, x, u0, w);
    }
    __device__ __forceinline__ static void cost_t_hess_acc(const double (&x)[NX], const double (&w)[W], double* __restrict__ gxx) {
        double hxx[NX * NX];
        cost_t_hess(x, w, hxx);
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) gxx[i] += hxx[i];
    }
    // ---- Constraint (src/constraints.jl:66-87)
    __device__ __forceinline__ static void con_s(const double (&x)[NX], const double (&u)[NU], const double (&w)[W], double (&c)[CS]) {
        zero(c);
        if constexpr (NCS > 0) F::constraint_stage(c, x, u, w);
    }
    __device__ __forceinline__ static void con_s_jac(const double (&x)[NX], const double (&u)[NU], const double (&w)[W],
                                                     double (&cx)[cdim<NCS * NX>::v], double (&cu)[cdim<NCS * NU>::v]) {
        zero(cx); zero(cu);
        if constexpr (NCS > 0) {
            F::constraint_stage_jacobian_state(cx, x, u, w);
            F::constraint_stage_jacobian_action(cu, x, u, w);
        }
    }
    __device__ __forceinline__ static void con_t(const double (&x)[NX], const double (&w)[W], double (&c)[CT]) {
        const double u0[1] = {0.0};
        zero(c);
        if constexpr (NCT > 0) F::constraint_terminal(c, x, u0, w);
    }
    __device__ __forceinline__ static void con_t_jac(const double (&x)[NX], const double (&w)[W], double (&cx)[cdim<NCT * NX>::v]) {
        const double u0[1] = {0.0};
        zero(cx);
        if constexpr (NCT > 0) F::constraint_terminal_jacobian_state(cx, x, u0, w);
    }
    // ---- Gauss-Newton AL terms, dense (src/gradients.jl:54-80): gx += cxᵀc̃, gxx += cxᵀ Iρ cx, gu += cuᵀc̃, guu += cuᵀ Iρ cu, gux += cuᵀ Iρ cx
    __device__ __forceinline__ static void al_s(const double (&x)[NX], const double (&u)[NU], const double (&w)[W],
                                                const double (&ct)[CS], const double (&ir)[CS], double (&gx)[NX], double (&gu)[NU],
                                                double* __restrict__ gxx, double* __restrict__ guu, double* __restrict__ gux) {
        if constexpr (NCS > 0) {
            double cx[NCS * NX], cu[NCS * NU];
            con_s_jac(x, u, w, cx, cu);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < NCS; ++i) acc += cx[j * NCS + i] * ct[i];
                gx[j] += acc;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int i2 = 0; i2 < NX; ++i2) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NCS; ++i) acc += cx[i2 * NCS + i] * (ir[i] * cx[j * NCS + i]);
                    gxx[j * NX + i2] += acc;
                }
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < NCS; ++i) acc += cu[j * NCS + i] * ct[i];
                gu[j] += acc;
            }
#pragma unroll
            for (int j = 0; j < NU; ++j)
#pragma unroll
                for (int i2 = 0; i2 < NU; ++i2) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NCS; ++i) acc += cu[i2 * NCS + i] * (ir[i] * cu[j * NCS + i]);
                    guu[j * NU + i2] += acc;
                }
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int i2 = 0; i2 < NU; ++i2) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NCS; ++i) acc += cu[i2 * NCS + i] * (ir[i] * cx[j * NCS + i]);
                    gux[j * NU + i2] += acc;
                }
        }
    }
    __device__ __forceinline__ static void al_t(const double (&x)[NX], const double (&w)[W], const double (&ct)[CT], const double (&ir)[CT],
                                                double (&gx)[NX], double* __restrict__ gxx) {
        if constexpr (NCT > 0) {
            double cx[NCT * NX];
            con_t_jac(x, w, cx);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < NCT; ++i) acc += cx[j * NCT + i] * ct[i];
                gx[j] += acc;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int i2 = 0; i2 < NX; ++i2) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NCT; ++i) acc += cx[i2 * NCT + i] * (ir[i] * cx[j * NCT + i]);
                    gxx[j * NX + i2] += acc;
                }
        }
    }
};

// Index tables of the compact forms the large path streams. Jacobian entry q of the compact row is entry jac_idx[q] of [fx | fu]
// (column-major inside each matrix); entries not listed are CONSTANTS taken from fxc / fuc. The Hessian row is
// [gxx entries sorted by the 16x16 tile they fall in | guu | gux], hess_idx giving each one's column-major index inside its
// matrix; entries not listed are structurally zero. DenseTables lists everything (black-box callables without structure);
// ilqr_compile_model emits a table type of the same shape from what it finds by probing the callables on the host
// (constant / zero Jacobian entries, structurally zero Hessian entries: the role Symbolics' sparse expressions play for the
// reference, src/dynamics.jl:16-34, src/costs.jl:17-44).
template <int NX, int NU>
struct DenseTables {
    static constexpr int TN = (NX + 15) / 16, NXX = NX * NX, NUU = NU * NU, NUX = NU * NX, HS = NXX + NUU + NUX, JV = NX * NX + NX * NU;
    struct Tab { int hess_idx[HS]; int tile_start[TN * TN + 1]; int jac_idx[JV]; double fxc[NX * NX]; double fuc[NX * NU]; };
    static constexpr Tab make() {
        Tab t{};
        int q = 0;
        for (int tile = 0; tile < TN * TN; ++tile) {
            t.tile_start[tile] = q;
            for (int idx = 0; idx < NXX; ++idx) {
                const int col = idx / NX, row = idx % NX;
                if ((row / 16) * TN + col / 16 == tile) t.hess_idx[q++] = idx;
            }
        }
        t.tile_start[TN * TN] = q;
        for (int i = 0; i < NUU; ++i) t.hess_idx[q++] = i;
        for (int i = 0; i < NUX; ++i) t.hess_idx[q++] = i;
        for (int i = 0; i < JV; ++i) t.jac_idx[i] = i;
        return t;
    }
    static constexpr Tab tab = make();
};

template <class F, int NX_, int NU_, int NW_, int NCS_, int NCT_, unsigned long long INEQ_S_, unsigned long long INEQ_T_,
          class DT_ = DenseTables<NX_, NU_>>
struct AdaptedLargeModel : AdaptedModel<F, NX_, NU_, NW_, NCS_, NCT_, INEQ_S_, INEQ_T_> {
    typedef AdaptedModel<F, NX_, NU_, NW_, NCS_, NCT_, INEQ_S_, INEQ_T_> Base;
    typedef DT_ DT;
    static constexpr int NX = NX_, NU = NU_, NW = NW_, NCS = NCS_, NCT = NCT_;
    static constexpr int W = cdim<NW>::v, CS = cdim<NCS>::v, CT = cdim<NCT>::v;
    static_assert(NX > 4 || NU > 4, "the compact forms are the large path's");
    static constexpr int JVN = DT::JV > 0 ? DT::JV : 1, HSN = DT::HS > 0 ? DT::HS : 1;
    // ---- Jacobians: the entries the table lists are evaluated per timestep, the others are its constants
    static constexpr int JAC_NVAR = DT::JV;
    template <int N> struct Row1 { double v[1][N]; };
    template <int N> static constexpr Row1<N> row1(const double (&a)[N]) {
        Row1<N> r{};
        for (int i = 0; i < N; ++i) r.v[0][i] = a[i];
        return r;
    }
    static constexpr Row1<NX * NX> FXC = row1(DT::tab.fxc);
    static constexpr Row1<NX * NU> FUC = row1(DT::tab.fuc);
    static constexpr const double (&JAC_CONST_FX)[1][NX * NX] = FXC.v;     // (the access form of the generated models: [0][e])
    static constexpr const double (&JAC_CONST_FU)[1][NX * NU] = FUC.v;
    static constexpr const int (&JAC_VAR_IDX)[JVN] = DT::tab.jac_idx;
    __device__ __forceinline__ static void dyn_jac_var(const double (&x)[NX], const double (&u)[NU], const double (&w)[W], double (&v)[JVN]) {
        if constexpr (DT::JV == NX * NX + NX * NU) {             // dense: the callables write the compact row directly
#pragma unroll
            for (int i = 0; i < JVN; ++i) v[i] = 0.0;
            F::dynamics_jacobian_state(v, x, u, w);
            F::dynamics_jacobian_action(v + NX * NX, x, u, w);
        } else {
            double full[NX * NX + NX * NU];
#pragma unroll
            for (int i = 0; i < NX * NX + NX * NU; ++i) full[i] = 0.0;
            F::dynamics_jacobian_state(full, x, u, w);
            F::dynamics_jacobian_action(full + NX * NX, x, u, w);
#pragma unroll
            for (int q = 0; q < DT::JV; ++q) v[q] = full[DT::tab.jac_idx[q]];
        }
    }
    // ---- dynamics row form: no affine part known, the whole f is the "remainder", evaluated by every lane
    static constexpr double DYN_AFF[NX][NX + NU + 1] = {};
    static constexpr bool DYN_HAS_REM = true, DYN_REM_ELEMENTWISE = false, JAC_VAR_ELEMENTWISE = false;
    template <class BC = WaveBC>
    __device__ __forceinline__ static void dyn_rem_wave(const int, const double (&x)[NX], const double (&u)[NU], const double (&w)[W], double (&r)[NX]) {
        Base::dyn(x, u, w, r);
    }
    __device__ __forceinline__ static double dyn_rem_own(const double, const double (&)[NU], const double (&)[W]) { return 0.0; }
    // ---- Hessians: dense compact row
    static constexpr int HESS_NXX = DT::NXX, HESS_NUU = DT::NUU, HESS_NUX = DT::NUX;
    static constexpr const int (&HESS_IDX)[HSN] = DT::tab.hess_idx;
    static constexpr const int (&HESS_XX_TILE_START)[DT::TN * DT::TN + 1] = DT::tab.tile_start;
    __device__ __forceinline__ static void cost_s_hess_c(const double (&x)[NX], const double (&u)[NU], const double (&w)[W], double* __restrict__ hs) {
        double hxx[NX * NX], huu[NU * NU], hux[NU * NX];
        Base::cost_s_hess(x, u, w, hxx, huu, hux);
        for (int q = 0; q < DT::NXX; ++q) hs[q] += hxx[DT::tab.hess_idx[q]];
        for (int q = 0; q < DT::NUU; ++q) hs[DT::NXX + q] += huu[DT::tab.hess_idx[DT::NXX + q]];
        for (int q = 0; q < DT::NUX; ++q) hs[DT::NXX + DT::NUU + q] += hux[DT::tab.hess_idx[DT::NXX + DT::NUU + q]];
    }
    __device__ __forceinline__ static void cost_t_hess_c(const double (&x)[NX], const double (&w)[W], double* __restrict__ hs) {
        double hxx[NX * NX];
        Base::cost_t_hess(x, w, hxx);
        for (int q = 0; q < DT::NXX; ++q) hs[q] += hxx[DT::tab.hess_idx[q]];
    }
    // Gauss-Newton AL terms (src/gradients.jl:54-80) on the compact row
    __device__ __forceinline__ static void al_s_c(const double (&x)[NX], const double (&u)[NU], const double (&w)[W], const double (&ct)[CS],
                                                  const double (&ir)[CS], double (&gx)[NX], double (&gu)[NU], double* __restrict__ hs) {
        if constexpr (NCS > 0) {
            double cx[NCS * NX], cu[NCS * NU];
            Base::con_s_jac(x, u, w, cx, cu);
            for (int j = 0; j < NX; ++j) {
                double acc = 0.0;
                for (int i = 0; i < NCS; ++i) acc += cx[j * NCS + i] * ct[i];
                gx[j] += acc;
            }
            for (int j = 0; j < NU; ++j) {
                double acc = 0.0;
                for (int i = 0; i < NCS; ++i) acc += cu[j * NCS + i] * ct[i];
                gu[j] += acc;
            }
            for (int q = 0; q < DT::NXX; ++q) {
                const int idx = DT::tab.hess_idx[q], j = idx / NX, i2 = idx % NX;
                double acc = 0.0;
                for (int i = 0; i < NCS; ++i) acc += cx[i2 * NCS + i] * (ir[i] * cx[j * NCS + i]);
                hs[q] += acc;
            }
            for (int q = 0; q < DT::NUU; ++q) {
                const int idx = DT::tab.hess_idx[DT::NXX + q], j = idx / NU, i2 = idx % NU;
                double acc = 0.0;
                for (int i = 0; i < NCS; ++i) acc += cu[i2 * NCS + i] * (ir[i] * cu[j * NCS + i]);
                hs[DT::NXX + q] += acc;
            }
            for (int q = 0; q < DT::NUX; ++q) {
                const int idx = DT::tab.hess_idx[DT::NXX + DT::NUU + q], j = idx / NU, i2 = idx % NU;
                double acc = 0.0;
                for (int i = 0; i < NCS; ++i) acc += cu[i2 * NCS + i] * (ir[i] * cx[j * NCS + i]);
                hs[DT::NXX + DT::NUU + q] += acc;
            }
        }
    }
    __device__ __forceinline__ static void al_t_c(const double (&x)[NX], const double (&w)[W], const double (&ct)[CT], const double (&ir)[CT],
                                                  double (&gx)[NX], double* __restrict__ hs) {
        if constexpr (NCT > 0) {
            double cx[NCT * NX];
            Base::con_t_jac(x, w, cx);
            for (int j = 0; j < NX; ++j) {
                double acc = 0.0;
                for (int i = 0; i < NCT; ++i) acc += cx[j * NCT + i] * ct[i];
                gx[j] += acc;
            }
            for (int q = 0; q < DT::NXX; ++q) {
                const int idx = DT::tab.hess_idx[q], j = idx / NX, i2 = idx % NX;
                double acc = 0.0;
                for (int i = 0; i < NCT; ++i) acc += cx[i2 * NCT + i] * (ir[i] * cx[j * NCT + i]);
                hs[q] += acc;
            }
        }
    }
};

}  // namespace ilqr
