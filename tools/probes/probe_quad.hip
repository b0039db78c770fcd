// Would four instances per workgroup put an instance's two waves on ONE SIMD? (an eight-wave workgroup, 157 KB of dynamic LDS: one
// per CU). Prints which SIMD hosts wave w of the workgroup, and the cost of a meeting of two waves through LDS counters (what
// would replace the workgroup barrier, which such a kernel cannot use: its four instances have their own control flow) against
// s_barrier in a two-wave workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 2) void placement(unsigned* out, int spin) {
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    double acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 0.5;
    smem[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = hwid;
}
typedef __attribute__((address_space(3))) int ldsi;
// pair meeting: monotonic counters, one per wave of the pair
__device__ __forceinline__ void pair_meet(ldsi* cnt, int role, int& epoch) {
    ++epoch;
    if ((threadIdx.x & 63) == 0) cnt[role] = epoch;
    while (__builtin_amdgcn_readfirstlane(cnt[role ^ 1]) < epoch) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// MODE 0: two-wave workgroup, s_barrier. MODE 1: eight-wave workgroup, waves j and j + 4 meet through LDS counters.
// `work` dependent fmas between meetings on role 0, work / 2 on role 1 (so that role 0 arrives last, like a critical wave)
template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 128 : 512, 2) void meet(long long* out, int meetings, int work) {
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6;
    const int inst = MODE == 0 ? 0 : (wave & 3), role = MODE == 0 ? wave : (wave >> 2);
    ldsi* cnt = (ldsi*)smem + 2 * inst;
    if (threadIdx.x < 16) ((ldsi*)smem)[threadIdx.x] = 0;
    __syncthreads();
    int epoch = 0;
    double acc = threadIdx.x;
    const int mywork = role == 0 ? work : work / 2;
    const long long t0 = clock64();
    for (int k = 0; k < meetings; ++k) {
        for (int i = 0; i < mywork; ++i) acc = acc * 1.0000001 + 0.5;
        if (MODE == 0) __syncthreads(); else pair_meet(cnt, role, epoch);
    }
    const long long t1 = clock64();
    if (acc == 12345.678) out[1000] = 1;
    if ((threadIdx.x & 63) == 0 && role == 0 && inst == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    const int B = 256;
    unsigned* d; hipMalloc(&d, B * 8 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)placement, hipFuncAttributeMaxDynamicSharedMemorySize, 157440);
    hipLaunchKernelGGL(placement, dim3(B), dim3(512), 157440, 0, d, 100000);
    std::vector<unsigned> h(B * 8);
    hipMemcpy(h.data(), d, B * 8 * sizeof(unsigned), hipMemcpyDeviceToHost);
    int pairs_same = 0, cus = 0;
    for (int b = 0; b < B; ++b) {
        for (int j = 0; j < 4; ++j) pairs_same += ((h[b * 8 + j] >> 4) & 3) == ((h[b * 8 + j + 4] >> 4) & 3);
        bool one_cu = true;
        for (int w = 1; w < 8; ++w) one_cu &= ((h[b * 8 + w] >> 8) & 0xff) == ((h[b * 8] >> 8) & 0xff);
        cus += one_cu;
    }
    printf("eight-wave workgroups: waves j and j + 4 on the same SIMD in %d of %d pairs\n", pairs_same, 4 * B);
    for (int b = 0; b < 4; ++b) { printf("wg %d: simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
    long long* t; hipMalloc(&t, 2048 * sizeof(long long));
    std::vector<long long> ht(2048);
    for (int work : {0, 20, 100}) {
        for (int mode = 0; mode < 2; ++mode) {
            const int meetings = 2000;
            if (mode == 0) hipLaunchKernelGGL(meet<0>, dim3(1024), dim3(128), 39360, 0, t, meetings, work);
            else { hipFuncSetAttribute((const void*)meet<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 157440); hipLaunchKernelGGL(meet<1>, dim3(256), dim3(512), 157440, 0, t, meetings, work); }
            hipMemcpy(ht.data(), t, 256 * sizeof(long long), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += ht[i];
            printf("work %3d fmas between meetings, %s: %.1f ticks (100 MHz) per meeting+work\n", work, mode == 0 ? "two-wave workgroup + s_barrier (4 per CU)" : "eight-wave workgroup + LDS pair counters", s / 256 / meetings);
        }
    }
    return 0;
}
