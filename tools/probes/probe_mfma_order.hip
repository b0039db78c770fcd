// Probe: in which order does v_mfma_f64_4x4x4 (gfx950) accumulate the four products of one output element, and with how many
// roundings? Decides whether three rank-1 updates C + a0 b0 + a1 b1 + a2 b2 issued as THREE dependent MFMAs (each with one
// non-zero k slice) can be issued as ONE MFMA with the slices stacked along k without changing a single bit.
//   candidates tested against the hardware, element by element, on random operands:
//     seq    : fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))      (k ascending, one rounding per product)
//     rev    : fma(a0,b0, fma(a1,b1, fma(a2,b2, fma(a3,b3, c))))
//     pair   : (fma(a0,b0,c) + a1 b1) ... pairwise tree variants
//     exact  : one rounding of the exact sum (long double / 2-sum approximation)
// Layout (tools/probes/probe_mfma.hip): A[i][k] at lane i + 4*beta + 16*k, B[k][j] at lane j + 4*beta + 16*k,
// D[i][j] at lane j + 4*beta + 16*i.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

__global__ void one_mfma(const double* a, const double* b, const double* c, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
// three dependent MFMAs, slice k of (a, b) alone in each, against one MFMA with all slices
__global__ void chain_vs_stacked(const double* a, const double* b, const double* c, double* d) {
    const int l = threadIdx.x, k = l >> 4;
    const double al = a[l], bl = b[l];
    double acc = c[l];
    for (int s = 0; s < 4; ++s) {
        const double as = (k == s) ? al : 0.0, bs = (k == s) ? bl : 0.0;
        // slice s moved to k = 0 (what the Riccati step does: every product sits in row 0 of its own operand)
        const double a0 = __shfl(as, (l & 15) + 16 * s), b0 = __shfl(bs, (l & 15) + 16 * s);
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(k == 0 ? a0 : 0.0, k == 0 ? b0 : 0.0, acc, 0, 0, 0);
    }
    d[l] = acc;
    d[64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(al, bl, c[l], 0, 0, 0);
}

static bool same(double x, double y) { return std::memcmp(&x, &y, 8) == 0; }

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-2.0, 2.0);
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dc, 64 * 8); hipMalloc(&dd, 128 * 8);
    long n_seq = 0, n_rev = 0, n_exact = 0, n_tot = 0, n_stack = 0, n_stack_tot = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        std::vector<double> a(64), b(64), c(64), d(128);
        for (int l = 0; l < 64; ++l) { a[l] = U(rng) * std::ldexp(1.0, (int)(rng() % 9) - 4); b[l] = U(rng); c[l] = U(rng) * std::ldexp(1.0, (int)(rng() % 5) - 2); }
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
        one_mfma<<<1, 64>>>(da, db, dc, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        for (int beta = 0; beta < 4; ++beta)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    double av[4], bv[4];
                    for (int k = 0; k < 4; ++k) { av[k] = a[i + 4 * beta + 16 * k]; bv[k] = b[j + 4 * beta + 16 * k]; }
                    const double cc = c[j + 4 * beta + 16 * i], hw = d[j + 4 * beta + 16 * i];
                    const double seq = std::fma(av[3], bv[3], std::fma(av[2], bv[2], std::fma(av[1], bv[1], std::fma(av[0], bv[0], cc))));
                    const double rev = std::fma(av[0], bv[0], std::fma(av[1], bv[1], std::fma(av[2], bv[2], std::fma(av[3], bv[3], cc))));
                    long double ex = (long double)cc;
                    for (int k = 0; k < 4; ++k) ex += (long double)av[k] * (long double)bv[k];
                    n_tot++; n_seq += same(hw, seq); n_rev += same(hw, rev); n_exact += same(hw, (double)ex);
                }
        chain_vs_stacked<<<1, 64>>>(da, db, dc, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) { n_stack_tot++; n_stack += same(d[l], d[64 + l]); }
    }
    printf("v_mfma_f64_4x4x4 vs host models over %ld output elements:\n", n_tot);
    printf("  k ascending fma chain : %ld identical (%.4f)\n", n_seq, (double)n_seq / n_tot);
    printf("  k descending fma chain: %ld identical (%.4f)\n", n_rev, (double)n_rev / n_tot);
    printf("  exact sum, one rounding (long double): %ld identical (%.4f)\n", n_exact, (double)n_exact / n_tot);
    printf("four dependent MFMAs (one k slice each, moved to k = 0) vs ONE stacked MFMA: %ld of %ld identical\n", n_stack, n_stack_tot);
    return 0;
}
