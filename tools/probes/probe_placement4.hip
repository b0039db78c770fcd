// Large-path geometry: 512 workgroups of FOUR waves, ~50 KB dynamic LDS each, two per CU. Where do the waves land, and does
// HW_REG_LDS_ALLOC.LDS_BASE tell the two co-resident workgroups of a CU apart? The Riccati chain (role 0) is VALU-bound: two of
// them on one SIMD slow each other down. Prints how many SIMDs host 0 / 1 / 2 chain waves with role = wave index and with
// role = (simd + 2 * (lds_base != 0)) & 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6;
    unsigned hwid, xcc, lds;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds));
    double acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 0.5;      // keep every workgroup resident for a while
    smem[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { unsigned* o = out + (blockIdx.x * 4 + wave) * 4; o[0] = hwid; o[1] = xcc; o[2] = lds; o[3] = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)); }
}
int main() {
    const int B = 512, LDS = 51200;
    unsigned* d; hipMalloc(&d, B * 16 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(probe, dim3(B), dim3(256), LDS, 0, d, 200000);
    std::vector<unsigned> h(B * 16);
    hipMemcpy(h.data(), d, B * 16 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned long, int> chains_plain, chains_mapped, wgs_per_cu;
    int perm = 0, builtin_ok = 0;
    for (int b = 0; b < B; ++b) {
        unsigned mask = 0, cukey = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned* o = &h[(b * 4 + w) * 4];
            const unsigned id = o[0], xcc = o[1] & 0xf, simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            const unsigned base = o[2] & 0xff;
            builtin_ok += o[3] == simd;
            mask |= 1u << simd;
            cukey = (xcc << 24) | (se << 16) | (sh << 12) | (cu << 4);
            const int role_plain = w, role_mapped = (simd + 2 * (base != 0)) & 3;
            if (role_plain == 0) chains_plain[cukey | simd]++;
            if (role_mapped == 0) chains_mapped[cukey | simd]++;
        }
        perm += mask == 0xf;
        wgs_per_cu[cukey]++;
    }
    printf("workgroups whose four waves sit on four different SIMDs: %d of %d; s_getreg builtin agrees with HW_ID: %d of %d\n", perm, B, builtin_ok, 4 * B);
    int hp[4] = {}, hm[4] = {}, hc[8] = {};
    for (auto& kv : chains_plain) hp[kv.second < 3 ? kv.second : 3]++;
    for (auto& kv : chains_mapped) hm[kv.second < 3 ? kv.second : 3]++;
    for (auto& kv : wgs_per_cu) hc[kv.second < 7 ? kv.second : 7]++;
    printf("CUs by resident workgroups: 1: %d, 2: %d, 3+: %d\n", hc[1], hc[2], hc[3] + hc[4] + hc[5] + hc[6] + hc[7]);
    printf("role = wave index:          SIMDs hosting 1 chain wave: %d, 2: %d, 3+: %d\n", hp[1], hp[2], hp[3]);
    printf("role = (simd + 2 lds) & 3:  SIMDs hosting 1 chain wave: %d, 2: %d, 3+: %d\n", hm[1], hm[2], hm[3]);
    for (int b = 0; b < 6; ++b) {
        printf("wg %3d:", b);
        for (int w = 0; w < 4; ++w) { const unsigned* o = &h[(b * 4 + w) * 4]; printf("  w%d simd %u cu %u se %u xcc %u lds_base %u size %u |", w, (o[0] >> 4) & 3, (o[0] >> 8) & 15, (o[0] >> 13) & 7, o[1] & 0xf, o[2] & 0xff, (o[2] >> 12) & 0x1ff); }
        printf("\n");
    }
    return 0;
}
