// Where do the two waves of a 128-thread workgroup land? (latency kernel geometry: 1024 workgroups, 39 KB dynamic LDS each,
// 4 per CU). Prints, per CU, which SIMD hosts wave 0 / wave 1 of each resident workgroup: the solve's critical chain runs on
// wave 0, so a SIMD that hosts two wave-0s is twice as loaded as one that hosts two wave-1s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(128, 2) void probe(unsigned* out, int spin) {
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 0.5;      // keep every workgroup resident for a while
    smem[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 2 + wave) * 2] = hwid; out[(blockIdx.x * 2 + wave) * 2 + 1] = xcc; }
}
int main() {
    const int B = 1024;
    unsigned* d; hipMalloc(&d, B * 4 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 39360);
    hipLaunchKernelGGL(probe, dim3(B), dim3(128), 39360, 0, d, 200000);
    std::vector<unsigned> h(B * 4);
    hipMemcpy(h.data(), d, B * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned long, std::vector<int>> simd_roles;   // (xcc, se, sh, cu, simd) -> roles
    int same = 0;
    for (int b = 0; b < B; ++b) {
        int s[2];
        for (int w = 0; w < 2; ++w) {
            const unsigned id = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 0xf;
            const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            s[w] = simd;
            simd_roles[((unsigned long)xcc << 24) | (se << 16) | (sh << 12) | (cu << 4) | simd].push_back(w);
        }
        same += s[0] == s[1];
    }
    int hist[3][3] = {};
    for (auto& kv : simd_roles) { int n0 = 0, n1 = 0; for (int r : kv.second) (r ? n1 : n0)++; if (n0 < 3 && n1 < 3) hist[n0][n1]++; }
    printf("workgroups whose two waves share a SIMD: %d of %d; SIMDs used: %zu\n", same, B, simd_roles.size());
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) if (hist[a][c]) printf("SIMDs hosting %d wave-0 and %d wave-1: %d\n", a, c, hist[a][c]);
    for (int b = 0; b < 8; ++b) printf("wg %d: wave0 simd %u cu %u se %u xcc %u | wave1 simd %u cu %u\n", b, (h[b*4] >> 4) & 3, (h[b*4] >> 8) & 15, (h[b*4] >> 13) & 7, h[b*4+1] & 0xf, (h[b*4+2] >> 4) & 3, (h[b*4+2] >> 8) & 15);
    return 0;
}
