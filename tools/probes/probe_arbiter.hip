// Which of two waves that share a SIMD does the arbiter serve? 1024 workgroups of two waves, 39 KB of LDS each:
// four per CU, two waves per SIMD — the latency kernel's geometry. Every wave runs the same dependent fp64 FMA chain and reports
// its shader-clock time, its wave slot on the SIMD (HW_REG_HW_ID bits 3:0: slot 0 was dispatched first) and its SIMD.
//   pass 0: nobody sets a priority             -> the slot-0 (older) wave of a SIMD is the faster one
//   pass L = 1, 2, 3: the slot-1 waves run at s_setprio L  -> L = 1 changes nothing, L = 2 and 3 turn the order round
//   hipcc -O2 --offload-arch=gfx950 probe_arbiter.hip -o probe_arbiter && ./probe_arbiter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(128, 2) void probe(unsigned long long* out, int iters, int level, int mode) {
    extern __shared__ double smem[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int slot = hwid & 15;
    if (slot != 0) {
        if (level == 1) __builtin_amdgcn_s_setprio(1);
        else if (level == 2) __builtin_amdgcn_s_setprio(2);
        else if (level == 3) __builtin_amdgcn_s_setprio(3);
    }
    double a = 1.0 + 1e-9 * threadIdx.x, b = 0.999999, c = 1e-7;
    __syncthreads();
    const long long t0 = clock64();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) a = __builtin_fma(a, b, c);
        }
    } else {
        // the solver's mix: dependent FMAs, LDS reads whose address depends on nothing, a 4x4x4 f64 MFMA on the chain, a scalar op
        double m = 0.0;
        smem[threadIdx.x] = 1e-9; smem[threadIdx.x + 128] = 2e-9;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int j = 0; j < 6; ++j) a = __builtin_fma(a, b, c);
                const double l0 = ((volatile double*)smem)[threadIdx.x], l1 = ((volatile double*)smem)[threadIdx.x + 128];
                m = __builtin_amdgcn_mfma_f64_4x4x4f64(a, l0, m, 0, 0, 0);
                a = __builtin_fma(m, 1e-30, a) + l1;
            }
        }
    }
    const long long t1 = clock64();
    smem[threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 2 + (threadIdx.x >> 6);
        out[3 * w] = (unsigned long long)(t1 - t0);
        out[3 * w + 1] = hwid | ((unsigned long long)(xcc & 15) << 32);
        out[3 * w + 2] = (unsigned long long)(smem[threadIdx.x] > 0.0);
    }
}
int main() {
    const int B = 1024, iters = 20000;
    unsigned long long* d;
    hipMalloc(&d, B * 2 * 3 * sizeof(unsigned long long));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 39424);
    std::vector<unsigned long long> h(B * 2 * 3);
    for (int mode = 0; mode < 2; ++mode) {
        std::printf(mode == 0 ? "== dependent fp64 FMAs only\n" : "== FMAs + LDS reads + a 4x4x4 f64 MFMA on the chain (per 32 instructions: 28 FMA / add, 8 ds_read, 4 MFMA)\n");
        for (int level = 0; level <= 3; ++level) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(probe, dim3(B), dim3(128), 39424, 0, d, iters, level, mode);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double sum[2] = {0, 0}; int cnt[2] = {0, 0};
            for (int w = 0; w < 2 * B; ++w) {
                const int slot = (int)(h[3 * w + 1] & 15) != 0;
                sum[slot] += (double)h[3 * w]; cnt[slot]++;
            }
            const double per = 32.0 * iters;
            std::printf("slot-1 waves at s_setprio %d: %4d waves in slot 0: %.2f clk per instruction | %4d waves in slot 1: %.2f clk per instruction\n",
                        level, cnt[0], cnt[0] ? sum[0] / cnt[0] / per : 0.0, cnt[1], cnt[1] ? sum[1] / cnt[1] / per : 0.0);
        }
    }
    // one wave per SIMD for reference
    hipLaunchKernelGGL(probe, dim3(512), dim3(128), 39424, 0, d, iters, 0, 0);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 512 * 2 * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0; for (int w = 0; w < 1024; ++w) s += (double)h[3 * w];
    std::printf("== 512 workgroups (one wave per SIMD), FMAs only: %.2f clk per instruction\n", s / 1024 / (32.0 * iters));
    return 0;
}
