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
// the Gauss-Newton AL terms are the dense products of src/gradients.jl:54-80. Small models only (nx, nu <= 4).
#pragma once

#include "ilqr_device.hpp"

#define ILQR_MODEL_FN __device__ __forceinline__ static

namespace ilqr {

template <class F, int NX_, int NU_, int NW_, int NCS_, int NCT_, unsigned long long INEQ_S_, unsigned long long INEQ_T_>
struct AdaptedModel {
    static constexpr int NX = NX_, NU = NU_, NW = NW_, NCS = NCS_, NCT = NCT_;
    static constexpr unsigned long long INEQ_S = INEQ_S_, INEQ_T = INEQ_T_;
    static_assert(NX >= 1 && NX <= 4 && NU >= 1 && NU <= 4, "ilqr_compile_model: small models only (nx, nu <= 4)");
    static_assert(NCS <= 64 && NCT <= 64, "at most 64 constraint rows per stage");
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

}  // namespace ilqr
