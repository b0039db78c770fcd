// v_mfma_f64_16x16x4_f64 on gfx950: clocks per instruction for a chain on ONE accumulator (the tile products of the large
// Riccati step), for two and four interleaved accumulators, with one and with two waves per SIMD; and a whole tile
// (16 ds_read_b64 fragments -> 8 MFMAs -> 4 ds_write_b64) the way ilqr::tile_mm runs it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void chain(double* out, long long* ticks, int iters) {
    double4_t acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0; for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void tile(double* out, long long* ticks, int iters) {
    __shared__ double sA[32 * 33], sB[32 * 33], sD[32 * 33];
    for (int e = threadIdx.x; e < 32 * 33; e += blockDim.x) { sA[e] = e * 1e-4; sB[e] = 1.0 - e * 1e-5; sD[e] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        double fa[8], fb[8];
        const double* pa = sA + 33 * li + lk; const double* pb = sB + lk + 33 * li;
#pragma unroll
        for (int s = 0; s < 8; ++s) { fa[s] = pa[4 * s]; fb[s] = pb[4 * s]; }
        double4_t acc = double4_t{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[s], acc, 0, 0, 0);
        double* p = sD + li * 33 + lk;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * r] = acc[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sD[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
// the same tile between two workgroup barriers, each wave on its own column block of the LDS matrices and reading what ANOTHER wave
// stored in the previous window: one window of the Riccati step without anything else in it
__global__ __launch_bounds__(256) void tile_window(double* out, long long* ticks, int iters) {
    __shared__ double sA[32 * 33], sB[4][16 * 33], sD[4][16 * 33];
    for (int e = threadIdx.x; e < 32 * 33; e += blockDim.x) sA[e] = e * 1e-4;
    for (int e = threadIdx.x; e < 4 * 16 * 33; e += blockDim.x) { (&sB[0][0])[e] = 1.0 - e * 1e-5; (&sD[0][0])[e] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, wave = threadIdx.x >> 6;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        double fa[8], fb[8];
        const double* src = (i & 1) ? &sD[(wave + 1) & 3][0] : &sB[(wave + 1) & 3][0];
        double* dst = (i & 1) ? &sB[wave][0] : &sD[wave][0];
        const double* pa = sA + 33 * li + lk; const double* pb = src + lk + 33 * li;
#pragma unroll
        for (int s = 0; s < 8; ++s) { fa[s] = pa[4 * s]; fb[s] = pb[4 * s]; }
        double4_t acc = double4_t{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], fb[s], acc, 0, 0, 0);
        double* p = dst + li * 33 + lk;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[4 * r] = acc[r] * 1e-3;
        __syncthreads();
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sD[wave][lane] + sB[wave][lane];
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    double* d; long long* t; hipMalloc(&d, 1 << 22); hipMalloc(&t, 64);
    long long h;
    const int iters = 2000;
    auto run = [&](const char* name, auto kern, int threads, int blocks, double per) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, t, iters);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, t, iters);
        hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("%-64s %7.1f clk per %s\n", name, (double)h / iters / per, per > 1 ? "MFMA" : "tile");
    };
    run("one accumulator, 1 wave on the SIMD (64 threads, 1 block)", chain<1>, 64, 1, 16);
    run("two accumulators, 1 wave", chain<2>, 64, 1, 16);
    run("four accumulators, 1 wave", chain<4>, 64, 1, 16);
    run("one accumulator, 4 waves of a block (one per SIMD)", chain<1>, 256, 1, 16);
    run("one accumulator, 2 waves per SIMD (2 blocks x 256 on a CU: 512 blocks)", chain<1>, 256, 512, 16);
    run("four accumulators, 2 waves per SIMD (512 blocks)", chain<4>, 256, 512, 16);
    run("whole tile (16 ds_read + 8 MFMA + 4 ds_write), 1 wave", tile, 64, 1, 1);
    run("whole tile, 4 waves of a block", tile, 256, 1, 1);
    run("whole tile, 2 blocks per CU (512 blocks x 256)", tile, 256, 512, 1);
    run("tile + workgroup barrier (one window), 4 waves of a block", tile_window, 256, 1, 1);
    run("tile + workgroup barrier, 2 blocks per CU (512 blocks x 256)", tile_window, 256, 512, 1);
    return 0;
}
