// Probe: accuracy of v_rcp_f64 and of one / two Newton steps on it (ulp of the correctly rounded 1/x), gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    double e = fma(-a, r, 1.0); r = fma(r, e, r); r1[i] = r;
    e = fma(-a, r, 1.0); r = fma(r, e, r); r2[i] = r;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), r0(n), r1(n), r2(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (s >> 11) * (1.0 / 9007199254740992.0), (int)(s % 41) - 20); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double w0 = 0, w1 = 0, w2 = 0; long ex1 = 0, ex2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        double c = (double)t;                               // correctly rounded (x87 extended: 64-bit mantissa suffices for ulp counts here)
        double u = std::nextafter(fabs(c), INFINITY) - fabs(c);
        w0 = fmax(w0, (double)fabsl((long double)r0[i] - t) / u); w1 = fmax(w1, (double)fabsl((long double)r1[i] - t) / u); w2 = fmax(w2, (double)fabsl((long double)r2[i] - t) / u);
        ex1 += r1[i] == c; ex2 += r2[i] == c;
    }
    printf("v_rcp_f64: max error %.3g ulp; + one Newton step: %.3f ulp (correctly rounded in %.2f %%); + two: %.3f ulp (%.2f %%)\n", w0, w1, 100.0 * ex1 / n, w2, 100.0 * ex2 / n);
    return 0;
}
