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
// a wave-uniform value the compiler must keep in a scalar register pair (it otherwise splits known 64-bit constants into halves and
// re-assembles aligned pairs with s_mov_b32 at every use: 8 of the 148.5 issue slots of an acrobot rollout step)
#define ILQR_OPAQUE_UNIFORM(v) asm volatile("" : "+s"(v))
// 1 / x by v_rcp_f64 and two Newton steps (5 instructions; measured 0.5 ulp, correctly rounded for 99.98 % of arguments — one step
// would leave 11 ulp, tools/probes/probe_rcp.hip) instead of the IEEE division sequence
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
// r alone (bitwise the r of sqrt_rsqrt_fast: 2 fma(h, e, h) = fma(2h, e, 2h)), one instruction shorter on the dependent chain and
// without the four that only serve d — for the factorisations that never read the diagonal of the factor
__device__ __forceinline__ double rsqrt_fast(double a) {
    const double r0 = __builtin_amdgcn_rsq(a);
    double g = a * r0, h = 0.5 * r0;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    const double h2 = h + h;
    return fma(h2, e, h2);
}
#else
ILQR_HD double rsqrt_fast(double a) { return 1.0 / sqrt(a); }
ILQR_HD void sqrt_rsqrt_fast(double a, double& d, double& r) { d = sqrt(a); r = 1.0 / d; }
ILQR_HD double fma3(double a, double b, double c) { return fma(a, b, c); }
#define ILQR_OPAQUE(v) do {} while (0)
#define ILQR_OPAQUE_UNIFORM(v) do {} while (0)
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
// |x| >= 2^30 on SOME lane of the wave: a wave-uniform branch (v_cmp + s_cbranch_vccnz) around the rare coarse folding instead
// of an exec-masked region (v_cmp, s_and_saveexec, s_cbranch_execz, s_or exec) on the serial rollout chain; the folding loop
// itself stays per lane
#if defined(__HIP_DEVICE_COMPILE__)
#define ILQR_ANY_HUGE(x) __builtin_expect(__builtin_amdgcn_ballot_w64(!(fabs(x) < 1073741824.0)) != 0, 0)
#else
#define ILQR_ANY_HUGE(x) (!(fabs(x) < 1073741824.0))
#endif
ILQR_HD void sincos_fast(double x, double& s, double& c) {
    if (ILQR_ANY_HUGE(x)) {
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

// ---- sine and cosine of one angle on a PAIR of lanes (even lane: sine, odd lane: cosine).
// The critical wave of the solve kernels is bound by its instruction COUNT (one wave issues one instruction per ~5-6 clk
// whatever its class, tools/probes/probe_issue.hip), and sincos_fast spends 16 fp64 instructions on two Horner chains that a
// pair of lanes can run as ONE: per-lane constant coefficients (c[0..6]: {0, S6..S1} on sine lanes, {C6..C1, -1/2} on cosine
// lanes; g = r on sine lanes, 1 on cosine lanes), own = g + (g z) q(z) — operation for operation the value sincos_reduced
// produces, so the results are bitwise those of sincos_fast (tests/test_device_math.py) — then the quadrant fix-up takes
// the partner lane's kernel value where sincos_fast swaps, and flips the sign bit.
struct TrigPair {
    double c[7], m, o;      // Horner coefficients, g = fma(r, m, o)
    double red[4];          // 2/pi and the three parts of pi/2 of the argument reduction (wave-uniform, but kept in VGPRs like the rest)
    double huge;            // 2^30: arguments beyond it take the slow reduction (a pinned scalar pair: hipcc rebuilt it with two s_mov per step)
    int odd;                // 1 on cosine lanes
};
ILQR_HD TrigPair trig_pair_constants(bool odd) {
    TrigPair t;
    t.c[0] = odd ? -1.13596475577881948265e-11 : 0.0;
    t.c[1] = odd ? 2.08757232129817482790e-09 : 1.58969099521155010221e-10;
    t.c[2] = odd ? -2.75573143513906633035e-07 : -2.50507602534068634195e-08;
    t.c[3] = odd ? 2.48015872894767294178e-05 : 2.75573137070700676789e-06;
    t.c[4] = odd ? -1.38888888888741095749e-03 : -1.98412698298579493134e-04;
    t.c[5] = odd ? 4.16666666666666019037e-02 : 8.33333333332248946124e-03;
    t.c[6] = odd ? -0.5 : -1.66666666666666324348e-01;
    t.m = odd ? 0.0 : 1.0;
    t.o = odd ? 1.0 : 0.0;
    t.red[0] = 6.36619772367581382433e-01; t.red[1] = 1.5707963267948966e+00;
    t.red[2] = 6.123233995736766e-17; t.red[3] = -1.4973849048591698e-33;
    t.huge = 1073741824.0;
    t.odd = odd ? 1 : 0;
    return t;
}
// kernel value of this lane (sine kernel on even, cosine kernel on odd lanes) and the quadrant of the argument
ILQR_HD double trig_pair_own(double x, const TrigPair& t, int& quadrant) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(fabs(x) < t.huge)) != 0, 0)) {
#else
    if (ILQR_ANY_HUGE(x)) {
#endif
#pragma clang loop unroll(disable)
        for (int it = 0; it < 24 && !(fabs(x) < 1073741824.0) && x == x; ++it) {
            const double k = rint(x * 1.5915494309189535e-01);
            x = fma(-k, 6.283185307179586, x);
        }
    }
    const double fn = rint(x * t.red[0]);
    double r = fma(-fn, t.red[1], x);
    r = fma(-fn, t.red[2], r);
    r = fma(-fn, t.red[3], r);
    const double z = r * r;
    double q = fma3(z, t.c[0], t.c[1]);
    q = fma3(z, q, t.c[2]);
    q = fma3(z, q, t.c[3]);
    q = fma3(z, q, t.c[4]);
    q = fma3(z, q, t.c[5]);
    q = fma3(z, q, t.c[6]);
    const double g = fma(r, t.m, t.o);
    quadrant = (int)fn;
    return fma(g * z, q, g);
}
// own / partner: kernel values of this lane and of the other lane of the pair
ILQR_HD double trig_pair_fix(double own, double partner, int quadrant, const TrigPair& t) {
    const double v = (quadrant & 1) ? partner : own;
#if defined(__HIP_DEVICE_COMPILE__)
    // sign flip as three integer instructions on the high word (shift-add, and, xor) instead of negate / compare / select
    const unsigned flip = (((unsigned)quadrant << 30) + ((unsigned)t.odd << 30)) & 0x80000000u;
    return __hiloint2double(__double2hiint(v) ^ (int)flip, __double2loint(v));
#else
    return ((quadrant + t.odd) & 2) ? -v : v;
#endif
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
// value of the other lane of an (even, odd) lane pair: DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ double pair_swap(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <bool PIN_UNIFORM = false>
__device__ __forceinline__ TrigPair make_trig_pair(int lane) {
    TrigPair t = trig_pair_constants(lane & 1);
    // pinned in registers: the time loops must not rebuild per-lane constants with selects every step
#pragma unroll
    for (int i = 0; i < 7; ++i) ILQR_OPAQUE(t.c[i]);
    ILQR_OPAQUE(t.m); ILQR_OPAQUE(t.o);
    if constexpr (PIN_UNIFORM) {        // wave-uniform constants: opaque scalar-register pairs (pinned in VGPRs they cost the forward pass 32 registers and spills)
#pragma unroll
        for (int i = 0; i < 4; ++i) ILQR_OPAQUE_UNIFORM(t.red[i]);
        ILQR_OPAQUE_UNIFORM(t.huge);
    }
    return t;
}
// sin(x) on even lanes, cos(x) on odd lanes; both lanes of a pair must hold the same x
__device__ __forceinline__ double sincos_pair(double x, const TrigPair& t) {
    int quadrant;
    const double own = trig_pair_own(x, t, quadrant);
    return trig_pair_fix(own, pair_swap(own), quadrant, t);
}
// lane I of every 16-lane row to all lanes of the row: ONE v_mov_b64_dpp row_newbcast (gfx90a+ DPP on 64-bit operands)
template <int I>
__device__ __forceinline__ double row_bcast(double v) {
    static_assert(I >= 0 && I < 16, "row_newbcast takes a lane of the 16-lane row");
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + I, 0xF, 0xF, true);
}
// How the wave-cooperative model code (M::dyn_wave) hands a value from lane I of the cooperating group to every lane of it.
struct WaveBC {      // group = the whole wave (one instance per wave); result is wave-uniform (SGPRs)
    template <int I> static __device__ __forceinline__ double bcast(double v) { return wave_bcast<I>(v); }
};
// group = a row of 16 lanes. Small models cooperate within rows in EVERY kernel: four instances per wave in the packed kernel,
// four identical copies of the one instance in the latency / throughput kernels (so the result stays in VGPRs, one instruction
// per value, instead of two v_readlane into SGPRs that the register allocator then spills).
struct RowBC {
    template <int I> static __device__ __forceinline__ double bcast(double v) { return row_bcast<I>(v); }
};
using Row16BC = RowBC;
#endif

}  // namespace ilqr
