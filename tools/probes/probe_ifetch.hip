// What do two waves of a SIMD compete for? One fp64 VALU instruction occupies the SIMD's pipe for 4 clk, a lone wave issues one per
// ~5 clk — and a 64-bit-encoded instruction is 8 bytes of the instruction fetch path (one 64 KB instruction cache per two CUs).
// Streams of INDEPENDENT v_fma_f64 (VOP3, 8 bytes) against v_fmac_f64_e32 (VOP2, 4 bytes), as a short loop (512 instructions, fits any
// buffer) and as a long straight-line body (8192 instructions: 64 / 32 KB), at one, two and four waves per SIMD: clk per instruction
// of a wave. Equal rates for both encodings = the VALU pipe is what is shared; the 8-byte stream falling behind = instruction fetch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R2(x) x x
#define R4(x) R2(R2(x))
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R512(x) R64(R2(R4(x)))
#define FMA8 "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n" \
             "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
#define FMAC8 "v_fmac_f64_e32 %0, %8, %9\n v_fmac_f64_e32 %1, %8, %9\n v_fmac_f64_e32 %2, %8, %9\n v_fmac_f64_e32 %3, %8, %9\n" \
              "v_fmac_f64_e32 %4, %8, %9\n v_fmac_f64_e32 %5, %8, %9\n v_fmac_f64_e32 %6, %8, %9\n v_fmac_f64_e32 %7, %8, %9\n"
#define BODY(S) asm volatile(S : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y))
template <int ENC, int LONG>
__global__ __launch_bounds__(64, 4) void stream(long long* out, int trips, double x, double y) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const long long t0 = clock64();
    for (int k = 0; k < trips; ++k) {
        if (LONG) { if (ENC) { BODY(R512(FMA8) R512(FMA8)); } else { BODY(R512(FMAC8) R512(FMAC8)); } }
        else { if (ENC) { BODY(R64(FMA8)); } else { BODY(R64(FMAC8)); } }
    }
    const long long t1 = clock64();
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[4096] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int ENC, int LONG>
double run(long long* d, int waves, int instr_total) {
    const int per_trip = LONG ? 8192 : 512, trips = instr_total / per_trip;
    hipLaunchKernelGGL((stream<ENC, LONG>), dim3(waves), dim3(64), 0, 0, d, trips, 1.0000001, 0.9999999);
    std::vector<long long> h(waves);
    hipMemcpy(h.data(), d, waves * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (long long v : h) s += v;
    return s / waves / (double)(trips * per_trip);
}
int main() {
    long long* d; hipMalloc(&d, 8192 * sizeof(long long));
    const int N = 1 << 20;
    printf("clock64 ticks per instruction of a wave (independent fp64 FMAs, eight accumulators)\n");
    for (int waves : {256, 1024, 2048, 4096}) {
        run<1, 0>(d, waves, N);
        printf("%4d waves (%.2f per SIMD): short loop  v_fma_f64 (8 B) %.2f  v_fmac_f64_e32 (4 B) %.2f | straight-line 64 / 32 KB  v_fma_f64 %.2f  v_fmac_f64_e32 %.2f\n",
               waves, waves / 1024.0, run<1, 0>(d, waves, N), run<0, 0>(d, waves, N), run<1, 1>(d, waves, N), run<0, 1>(d, waves, N));
    }
    return 0;
}
