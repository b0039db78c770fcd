// Probe: lane layout of v_mfma_f64_4x4x4 (4 blocks) and latency of dependent fp64 chains on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe_layout(double* out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < 16; ++t) {
        double a = (double)(lane + 1);
        double b = ((lane & 15) == t) ? 1.0 : 0.0;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[t * 64 + lane] = d;
    }
}
__global__ void chain_fma(double* out, int n, double x) {
    double a = x + threadIdx.x, b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (4.0 * n);
}
__global__ void indep_fma(double* out, int n, double x) {
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); }
    long long t1 = clock64();
    out[threadIdx.x] = a0 + a1 + a2 + a3; if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (4.0 * n);
}
__global__ void chain_mfma(double* out, int n, double x) {
    double a = 1.0 + 1e-9 * threadIdx.x, b = ((threadIdx.x & 3) == ((threadIdx.x >> 2) & 3)) ? 1.0 : 0.0, d = x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = d; if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (4.0 * n);
}
// D of one MFMA feeds the B operand of the next (the Riccati chain pattern)
__global__ void chain_mfma_b(double* out, int n, double x) {
    double a = ((threadIdx.x & 3) == ((threadIdx.x >> 2) & 3)) ? 1.0 : 0.0, d = x + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.0, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.0, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.0, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, d, 0.0, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = d; if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (4.0 * n);
}
__global__ void chain_div(double* out, int n, double x) {
    double a = x + threadIdx.x, b = 1.0000001;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { a = b / a; a = b / a; }
    long long t1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (2.0 * n);
}
__global__ void chain_lds(double* out, int n) {
    __shared__ double s[256];
    s[threadIdx.x] = (double)((threadIdx.x * 7 + 3) & 63);
    __syncthreads();
    int idx = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { idx = (int)s[idx]; idx = (int)s[idx]; }
    long long t1 = clock64();
    out[threadIdx.x] = idx; if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (2.0 * n);
}
int main() {
    double* d; hipMalloc(&d, 16 * 64 * 8 + 1024);
    std::vector<double> h(16 * 64 + 128);
    probe_layout<<<1, 64>>>(d);
    hipMemcpy(h.data(), d, 16 * 64 * 8, hipMemcpyDeviceToHost);
    printf("LAYOUT probe: row t = B is 1 at lane16==t; entries = D per lane (A[l]=l+1)\n");
    for (int t = 0; t < 16; ++t) { printf("t=%2d:", t); for (int l = 0; l < 64; ++l) printf(" %2.0f", h[t * 64 + l]); printf("\n"); }
    const int n = 20000;
    chain_fma<<<1, 64>>>(d, n, 1.0); hipMemcpy(h.data(), d, 65 * 8, hipMemcpyDeviceToHost); printf("dependent v_fma_f64: %.2f clk/op\n", h[64]);
    indep_fma<<<1, 64>>>(d, n, 1.0); hipMemcpy(h.data(), d, 65 * 8, hipMemcpyDeviceToHost); printf("independent(4) v_fma_f64: %.2f clk/op\n", h[64]);
    chain_mfma<<<1, 64>>>(d, n, 1.0); hipMemcpy(h.data(), d, 65 * 8, hipMemcpyDeviceToHost); printf("dependent mfma_f64_4x4x4 (C chain): %.2f clk/op\n", h[64]);
    chain_mfma_b<<<1, 64>>>(d, n, 1.0); hipMemcpy(h.data(), d, 65 * 8, hipMemcpyDeviceToHost); printf("dependent mfma_f64_4x4x4 (D->B chain): %.2f clk/op\n", h[64]);
    chain_div<<<1, 64>>>(d, n, 3.0); hipMemcpy(h.data(), d, 65 * 8, hipMemcpyDeviceToHost); printf("dependent f64 division: %.2f clk/op\n", h[64]);
    chain_lds<<<1, 64>>>(d, n); hipMemcpy(h.data(), d, 65 * 8, hipMemcpyDeviceToHost); printf("dependent ds_read_b64 + cvt: %.2f clk/op\n", h[64]);
    // clock64 ticks vs real time
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); chain_fma<<<1, 64>>>(d, 2000000, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h.data(), d, 65 * 8, hipMemcpyDeviceToHost);
    printf("8M dependent fma: %.3f ms -> %.2f ns/op; clock64 says %.2f ticks/op\n", ms, ms * 1e6 / 8e6, h[64]);
    return 0;
}
