// Probe: cost of ds_write_b64 / ds_read_b64 seen by one wave for different lane -> address patterns (gfx950).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/probe_lds.hip -o tools/probes/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#define X32(s) s s s s s s s s s s s s s s s s s s s s s s s s s s s s s s s s
// mode: 0 all lanes one address, 1 lane*8 (conflict-free), 2 four groups of 16 lanes writing 16 addresses (the MFMA tile stores),
//       3 only lane 0 active (exec mask), 4 all lanes one address but b128
__global__ void k_write(double* out, int n, int mode) {
    __shared__ double buf[1024];
    buf[threadIdx.x] = 0.0;
    __syncthreads();
    unsigned addr = mode == 1 ? threadIdx.x * 8u : mode == 2 ? (threadIdx.x & 15) * 8u : 0u;
    addr += (unsigned)(size_t)(__attribute__((address_space(3))) double*)buf;
    double v = 1.0 + threadIdx.x;
    long long t0 = clock64();
    if (mode == 3) {
        if (threadIdx.x == 0)
            for (int i = 0; i < n; ++i) asm volatile(X32("ds_write_b64 %0, %1\n") :: "v"(addr), "v"(v) : "memory");
    } else if (mode == 4) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 vv = {v, v};
        for (int i = 0; i < n; ++i) asm volatile(X32("ds_write_b128 %0, %1\n") :: "v"(addr), "v"(vv) : "memory");
    } else {
        for (int i = 0; i < n; ++i) asm volatile(X32("ds_write_b64 %0, %1\n") :: "v"(addr), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (32.0 * n);
}
// writes interleaved with fp64 work (3 fma per write): does the write hide behind them?
__global__ void k_write_mix(double* out, int n, int mode) {
    __shared__ double buf[1024];
    buf[threadIdx.x] = 0.0;
    __syncthreads();
    unsigned addr = mode == 1 ? threadIdx.x * 8u : mode == 2 ? (threadIdx.x & 15) * 8u : 0u;
    addr += (unsigned)(size_t)(__attribute__((address_space(3))) double*)buf;
    double v = 1.0 + threadIdx.x, a = v, b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i)
        asm volatile(X32("ds_write_b64 %1, %0\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n") : "+v"(a) : "v"(addr), "v"(b), "v"(c) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (double)(t1 - t0) / (32.0 * n); out[1] = a; }
}
int main() {
    double* d; hipMalloc(&d, 64);
    const char* names[] = {"all 64 lanes one address", "lane * 8 (conflict-free)", "4 x 16 lanes on 16 addresses", "lane 0 only (exec)", "b128, one address"};
    printf("clk per ds_write seen by the issuing wave; columns: 1 wave on the chip | 1 per SIMD | 2 per SIMD\n");
    for (int mode = 0; mode < 5; ++mode) {
        printf("ds_write_b64 %-32s", names[mode]);
        for (int blocks : {1, 1024, 2048}) { k_write<<<blocks, 64>>>(d, 2000, mode); double h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf(" %7.2f", h); }
        printf("\n"); fflush(stdout);
    }
    for (int mode = 0; mode < 3; ++mode) {
        printf("write + 3 fma64: %-28s", names[mode]);
        for (int blocks : {1, 1024, 2048}) { k_write_mix<<<blocks, 64>>>(d, 2000, mode); double h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf(" %7.2f", h); }
        printf("   (per group of 4 instructions)\n"); fflush(stdout);
    }
    return 0;
}
