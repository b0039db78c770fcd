// fp64 sin/cos for the model code: one fused evaluation per distinct argument.
//
// The ROCm device libm spends ~150-190 instructions per sin() or cos() call
// (double-double reduction + Payne-Hanek path); the closed-loop rollout of the
// acrobot needs 8 of them per timestep on its serial critical path. Here:
// 3-constant Cody-Waite reduction by FMA (exact product inside the FMA) for
// |x| < 2^30, fdlibm minimax kernels on [-pi/4, pi/4], branch-free quadrant
// fix-up: ~35 VALU instructions for BOTH sin and cos. Max observed error vs
// a 200-bit reference: 1.4 ulp, mean 0.29 ulp (tests/test_device_math.py).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ILQR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define ILQR_HD inline
#endif

namespace ilqr {

// d = a * b + c as ONE three-address v_fma_f64. hipcc selects the two-address v_fmac_f64 for Horner steps and then copies the
// constant addend into the destination first (a v_mov_b64 per step: ten extra issue slots per sincos on the serial rollout chain).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// value made opaque to the optimiser at this point (keeps a cheap expression out of an exec-masked branch)
#define ILQR_OPAQUE(v) asm volatile("" : "+v"(v))
// 1 / x by v_rcp_f64 and two Newton steps (5 instructions, <= 1 ulp for normal x) instead of the IEEE division sequence
// (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup: 11 instructions). Used on the serial rollout chain only.
__device__ __forceinline__ double recip_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}
// d = sqrt(a) and r = 1 / sqrt(a) together, for the pivots of the Cholesky factorisation (dpotf2: AJJ = SQRT(AJJ);
// DSCAL by ONE / AJJ): v_rsq_f64 + two coupled Newton (Goldschmidt) steps + one residual correction of d — 12 instructions
// instead of the expanded IEEE sqrt (~17) followed by the IEEE division sequence (11) on the serial chain of every column.
// d within 1 ulp, r within 2 ulp for normal a > 0 (the caller has already decided a > 0; a <= 0 never gets here).
__device__ __forceinline__ void sqrt_rsqrt_fast(double a, double& d, double& r) {
    const double r0 = __builtin_amdgcn_rsq(a);
    double g = a * r0, h = 0.5 * r0;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    d = fma(fma(-g, g, a), h, g);
    r = h + h;
}
#else
ILQR_HD void sqrt_rsqrt_fast(double a, double& d, double& r) { d = sqrt(a); r = 1.0 / d; }
ILQR_HD double fma3(double a, double b, double c) { return fma(a, b, c); }
#define ILQR_OPAQUE(v) do {} while (0)
ILQR_HD double recip_fast(double x) { return 1.0 / x; }
#endif

ILQR_HD void sincos_reduced(double r, double& s, double& c) {
    // fdlibm __kernel_sin / __kernel_cos coefficients on |r| <= pi/4
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = r * r;
    double ps = fma3(z, S6, S5);
    ps = fma3(z, ps, S4);
    ps = fma3(z, ps, S3);
    ps = fma3(z, ps, S2);
    ps = fma3(z, ps, S1);
    s = fma(r * z, ps, r);
    double pc = fma3(z, C6, C5);
    pc = fma3(z, pc, C4);
    pc = fma3(z, pc, C3);
    pc = fma3(z, pc, C2);
    pc = fma3(z, pc, C1);
    c = fma(z, fma(z, pc, -0.5), 1.0);       // 1 + z(-1/2 + z*pc): 2 ops instead of fdlibm's 6 (+0.3 ulp)
}

// |x| < 2^30: exact-product FMA reduction keeps the ABSOLUTE error of r below ~2e-16, so
// sin/cos are good to ~1.5 ulp except relative to a tiny result at huge |x|. For
// |x| >= 2^30 (1e9 rad: only diverged line-search trials get there, and those are
// rejected) the argument is first folded coarsely by multiples of 2*pi; that path
// is finite and deterministic but loses ~|x|*2^-52 rad of accuracy, unlike libm.
ILQR_HD void sincos_fast(double x, double& s, double& c) {
    if (!(fabs(x) < 1073741824.0)) {
#pragma clang loop unroll(disable)
        for (int it = 0; it < 24 && !(fabs(x) < 1073741824.0) && x == x; ++it) {
            const double k = rint(x * 1.5915494309189535e-01);
            x = fma(-k, 6.283185307179586, x);
        }
    }
    const double fn = rint(x * 6.36619772367581382433e-01);   // x * 2/pi
    double r = fma(-fn, 1.5707963267948966e+00, x);           // pi/2 = HI + MID + LO
    r = fma(-fn, 6.123233995736766e-17, r);
    r = fma(-fn, -1.4973849048591698e-33, r);                 // keeps RELATIVE accuracy next to multiples of pi/2
    double sr, cr;
    sincos_reduced(r, sr, cr);
    const int q = (int)fn;
    const bool swap = q & 1;
    const double ss = swap ? cr : sr;
    const double cc = swap ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

ILQR_HD double sin_fast(double x) { double s, c; sincos_fast(x, s, c); return s; }
ILQR_HD double cos_fast(double x) { double s, c; sincos_fast(x, s, c); return c; }

#if defined(__HIPCC__)
// broadcast lane I of every 4-lane quad to the whole quad (DPP quad_perm, no LDS)
template <int I>
__device__ __forceinline__ double quad_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, I * 0x55, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, I * 0x55, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// wave-uniform broadcast of lane I (result lives in SGPRs: any number of sources per batch)
template <int I>
__device__ __forceinline__ double wave_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, I);
    hi = __builtin_amdgcn_readlane(hi, I);
    return __hiloint2double(hi, lo);
}
// How the wave-cooperative model code (M::dyn_wave) hands a value from lane I of the cooperating group to every lane of it.
struct WaveBC {      // group = the whole wave (one instance per wave); result is wave-uniform (SGPRs)
    template <int I> static __device__ __forceinline__ double bcast(double v) { return wave_bcast<I>(v); }
};
struct Row16BC {     // group = a row of 16 lanes (four instances per wave): ds_swizzle bit mode, lane' = (lane & 0x10) | I
    template <int I> static __device__ __forceinline__ double bcast(double v) {
        static_assert(I < 16, "a 16-lane row cooperates on at most 16 values");
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_ds_swizzle(lo, 0x10 | (I << 5));
        hi = __builtin_amdgcn_ds_swizzle(hi, 0x10 | (I << 5));
        return __hiloint2double(hi, lo);
    }
};
#endif

}  // namespace ilqr
