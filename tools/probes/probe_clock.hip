// Does the chip hold its clock when every SIMD runs one fp64-dependent wave? (DVFS probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain_fma(double* out, int n, double x, int active_lanes) {
    double a = x + threadIdx.x, b = 1.0000001, c = 1e-9;
    if ((int)threadIdx.x < active_lanes) {
        for (int i = 0; i < n; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
int main() {
    double* d; hipMalloc(&d, 4096 * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000000;
    for (int lanes : {64, 16, 4, 1}) {
        for (int blocks : {1, 256, 1024, 2048, 4096}) {
            chain_fma<<<blocks, 64>>>(d, 1000, 1.0, lanes);
            hipDeviceSynchronize();
            hipEventRecord(e0); chain_fma<<<blocks, 64>>>(d, n, 1.0, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("active lanes %2d, blocks %4d: %.3f ms -> %.2f ns per dependent DP FMA per wave; chip rate %.2f Gfma-instr/s\n",
                   lanes, blocks, ms, ms * 1e6 / (4.0 * n), blocks * 4.0 * n / (ms * 1e6));
        }
    }
    return 0;
}
