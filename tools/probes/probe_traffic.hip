// Known-traffic kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against the access patterns of the solve
// kernels (tools/calibrate_counters.py runs this under two --pmc passes and divides). Every kernel touches each byte of its
// buffer exactly once per launch; buffers are 1 GiB (far beyond the 256 MB of L2 + MALL reach per launch) and are written /
// read by a different kernel in between, so hits in the cache hierarchy cannot hide traffic.
//   read16_coalesced   16 B per lane, consecutive lanes (the guide's calibration pattern: FETCH_SIZE counts half)
//   read8_coalesced     8 B per lane, consecutive lanes (large path: compact rows, K ring staging)
//   read8_tile128       a wave reads 128 contiguous bytes per step (16 lanes x 8 B), walking backwards in time through its own
//                       block — the Hessian-tile fetch of the small-model Riccati recursion (ilqr_device.hpp, backward_pass_split)
//   write8_coalesced    8 B per lane, consecutive lanes
//   rmw8_strided        one lane per timestep adds to an 8-byte entry at a 128-byte stride (read-modify-write of the accumulated
//                       Hessians, gradients_small / packed linearise_stage)
//   write8_rows16       16-lane rows each store 128 contiguous bytes per step (packed kernel: K, gains, trajectories)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));
static const size_t BYTES = 1ull << 30;
__global__ void read16_coalesced(const double2_t* p, double* out, size_t n16) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const double2_t v = p[i]; acc += v[0] + v[1]; }
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void read8_coalesced(const double* p, double* out, size_t n8) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345e-300) out[0] = acc;
}
// one wave per block of `steps` x 16 doubles; lanes 0..15 read the tile of step t (the other lanes repeat it), t walking down
__global__ void read8_tile128(const double* p, double* out, int steps) {
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const int l = threadIdx.x & 15;
    const double* base = p + wave * (size_t)steps * 16;
    double acc = 0;
    for (int t = steps - 1; t >= 0; --t) acc += base[(size_t)t * 16 + l];
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void write8_coalesced(double* p, size_t n8) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
// lane = timestep: entry e of the 16-double tile of step t, all 16 entries one after the other (every byte once)
__global__ void rmw8_strided(double* p, int steps) {
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    double* base = p + wave * (size_t)steps * 16;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const int t = t0 + lane;
        if (t < steps)
            for (int e = 0; e < 16; ++e) base[(size_t)t * 16 + e] += 0.2;
    }
}
// ONE 8-byte entry per 128-byte tile (entry 10): the accumulated-Hessian update of a model whose stage cost touches a single
// entry per matrix — what granule does WRITE_SIZE / FETCH_SIZE count for a lone 8-byte read-modify-write?
__global__ void rmw8_sparse(double* p, int steps) {
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    double* base = p + wave * (size_t)steps * 16;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const int t = t0 + lane;
        if (t < steps) base[(size_t)t * 16 + 10] += 0.2;
    }
}
__global__ void write8_rows16(double* p, int steps) {
    const size_t row = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 4;      // four rows per wave, each its own block
    const int l = threadIdx.x & 15;
    double* base = p + row * (size_t)steps * 16;
    for (int t = 0; t < steps; ++t) base[(size_t)t * 16 + l] = (double)t;
}
int main() {
    double *a, *b, *out;
    if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&b, BYTES) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const size_t n8 = BYTES / 8, n16 = BYTES / 16;
    const int steps = 128;                                    // 128 steps x 128 B = 16 KB per wave / row block
    const size_t waves = n8 / ((size_t)steps * 16), rows = waves;
    hipMemset(a, 0, BYTES); hipMemset(b, 0, BYTES);
    auto flush = [&]() { hipMemset(b, 1, BYTES); hipDeviceSynchronize(); };          // 1 GiB of other traffic between the kernels
    for (int rep = 0; rep < 2; ++rep) {
        flush(); hipLaunchKernelGGL(read16_coalesced, dim3(8192), dim3(256), 0, 0, (const double2_t*)a, out, n16);
        flush(); hipLaunchKernelGGL(read8_coalesced, dim3(8192), dim3(256), 0, 0, a, out, n8);
        flush(); hipLaunchKernelGGL(read8_tile128, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, a, out, steps);
        flush(); hipLaunchKernelGGL(write8_coalesced, dim3(8192), dim3(256), 0, 0, a, n8);
        flush(); hipLaunchKernelGGL(rmw8_strided, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, a, steps);
        flush(); hipLaunchKernelGGL(rmw8_sparse, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, a, steps);
        flush(); hipLaunchKernelGGL(write8_rows16, dim3((unsigned)(rows / 16)), dim3(256), 0, 0, a, steps);
        hipDeviceSynchronize();
    }
    // kernel, true bytes read, true bytes written (per launch)
    printf("TRUE read16_coalesced %zu 0\nTRUE read8_coalesced %zu 0\nTRUE read8_tile128 %zu 0\nTRUE write8_coalesced 0 %zu\n"
           "TRUE rmw8_strided %zu %zu\nTRUE write8_rows16 0 %zu\nTRUE rmw8_sparse %zu %zu\n", BYTES, BYTES, BYTES, BYTES, BYTES, BYTES, BYTES,
           BYTES / 16, BYTES / 16);
    return 0;
}
