// Can a straggler of a FULL chip be given room? 2048 one-wave workgroups of 256 VGPRs and 14.5 KB of LDS (two per SIMD, eight per CU:
// the packed kernel's one-wave form at 8192 instances) spin for `ms` milliseconds; after `at` ms the wave of workgroup `who` posts an
// eviction request for its CU (HW_ID / XCC_ID) and leaves, the first other wave of that CU to see the request leaves too. A second
// kernel (two-wave workgroups of 256 VGPRs, 39 KB of LDS: the latency kernel's shape), launched behind the first on another stream,
// can only start where two wave slots of one CU are free. Prints: waves per (xcc, se, sh, cu) id, when and where the pool's workgroups
// started.     hipcc -O2 --offload-arch=gfx950 probe_evict.hip -o probe_evict && ./probe_evict
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ unsigned cu_of() {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    return ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
}
__device__ __forceinline__ unsigned simd_of() { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); return (hw >> 4) & 3u; }
struct Rec { unsigned cu, simd; long long t0, t1; int left; };
__global__ __launch_bounds__(64, 2) void bulk(Rec* rec, int* evict, long long* tstart, int who, long long at_ticks, long long ms_ticks) {
    extern __shared__ double smem[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const unsigned cu = cu_of();
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) *tstart = t0;
    int left = 0;
    for (;;) {
        const long long t = wall_clock64() - t0;
        if (t > ms_ticks) break;
        if ((int)blockIdx.x == who && t > at_ticks) { if (threadIdx.x == 0) atomicExch(&evict[cu], 1); left = 1; break; }
        int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&evict[cu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v == 1) {
            int got = 0;
            if (threadIdx.x == 0) got = atomicCAS(&evict[cu], 1, 2) == 1;
            if (__builtin_amdgcn_readfirstlane(got)) { left = 2; break; }
        }
        __builtin_amdgcn_s_sleep(64);
    }
    smem[threadIdx.x] = (double)left;
    if (threadIdx.x == 0) { rec[blockIdx.x].cu = cu; rec[blockIdx.x].simd = simd_of(); rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); rec[blockIdx.x].left = left; }
}
__global__ __launch_bounds__(128, 2) void pool(Rec* rec, long long hold_ticks) {
    extern __shared__ double smem[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(64);
    smem[threadIdx.x] = 1.0;
    if (blockDim.x == 64 && threadIdx.x == 0) rec[blockIdx.x * 2 + 1] = Rec{cu_of(), simd_of(), t0, t0, 0};
    if ((threadIdx.x & 63) == 0) { Rec& r = rec[blockIdx.x * 2 + (threadIdx.x >> 6)]; r.cu = cu_of(); r.simd = simd_of(); r.t0 = t0; r.t1 = wall_clock64(); r.left = 0; }
}
int main(int argc, char** argv) {
    const int NP = argc > 1 ? atoi(argv[1]) : 4, NB = argc > 2 ? atoi(argv[2]) : 2048, PT = argc > 3 ? atoi(argv[3]) : 128, PL = argc > 4 ? atoi(argv[4]) : 39360, who = 777;
    printf("-- %d bulk workgroups, pool: %d workgroups of %d threads, %d B of LDS\n", NB, NP, PT, PL);
    const double ms = 20.0, at = 5.0, tick_per_ms = 1.0e5;    // wall_clock64: 100 MHz
    Rec *rb, *rp; int* evict; long long* tstart;
    CHECK(hipMalloc(&rb, NB * sizeof(Rec))); CHECK(hipMalloc(&rp, 2 * NP * sizeof(Rec))); CHECK(hipMalloc(&evict, 4096 * sizeof(int))); CHECK(hipMalloc(&tstart, 8));
    CHECK(hipMemset(evict, 0, 4096 * sizeof(int))); CHECK(hipMemset(rp, 0, 2 * NP * sizeof(Rec)));
    hipStream_t s1, s2; CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pool), hipFuncAttributeMaxDynamicSharedMemorySize, 39360));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(evict, 0, 4096 * sizeof(int)));
        hipLaunchKernelGGL(bulk, dim3(NB), dim3(64), 14480, s1, rb, evict, tstart, who, (long long)(at * tick_per_ms), (long long)(ms * tick_per_ms));
        hipLaunchKernelGGL(pool, dim3(NP), dim3(PT), PL, s2, rp, (long long)(2.0 * tick_per_ms));
        CHECK(hipDeviceSynchronize());
    }
    std::vector<Rec> hb(NB), hp(2 * NP); long long t00;
    CHECK(hipMemcpy(hb.data(), rb, NB * sizeof(Rec), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hp.data(), rp, 2 * NP * sizeof(Rec), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&t00, tstart, 8, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_cu; std::map<int, int> hist; long long tmin = hb[0].t0, tmax = hb[0].t0;
    for (auto& r : hb) { per_cu[r.cu]++; tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t0); }
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("bulk: %zu distinct (xcc, se, sh, cu) ids; waves per id:", per_cu.size());
    for (auto& kv : hist) printf("  %d ids with %d waves", kv.second, kv.first);
    printf("\nbulk start spread %.1f us\n", (tmax - tmin) / 100.0);
    for (int b = 0; b < NB; ++b) if (hb[b].left) printf("bulk workgroup %d left (%s) at %.3f ms: cu id 0x%03x simd %u\n", b, hb[b].left == 1 ? "requester" : "CU-mate", (hb[b].t1 - tmin) / 1.0e5, hb[b].cu, hb[b].simd);
    for (int p = 0; p < NP; ++p) printf("pool workgroup %d: started at %.3f ms on cu id 0x%03x (simds %u, %u)%s\n", p, (hp[2 * p].t0 - tmin) / 1.0e5, hp[2 * p].cu, hp[2 * p].simd, hp[2 * p + 1].simd,
                                        hp[2 * p].cu != hp[2 * p + 1].cu ? "  !! waves on different CUs?" : "");
    return 0;
}
